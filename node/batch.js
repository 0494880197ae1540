'use strict';
// Batch host: many .wav files through one GPU with the copies overlapped.  `slots` contexts (own
// stream and device staging each) with page-locked input/output buffers; file i goes to slot i % slots:
// its samples are read from disk straight into the slot's page-locked buffer (PCM16 stays int16 all the
// way to the kernel), the transform is enqueued without waiting, and the result of the file that used
// the slot before is handed to the callback first.  What it replaces in the reference: the one-file-at-a-
// time load loop of modules/fileLoader.js:110-157 feeding the worker (main.js:1417).
const fs = require('fs');
const S = require('./index.js');
const wav = require('./wav.js');
const N = S.native;

class FileBatch {
  // options: { fftSize = 1024, overlap = 75 (percent), out = 'rgba' | 'u8', maxSeconds = 20, maxSampleRate = 500000,
  //            slots = 3, colorMap (256 x [r,g,b,a] floats, rgba only), device = 0 }
  constructor(options) {
    const o = Object.assign({ fftSize: 1024, overlap: 75, out: 'rgba', maxSeconds: 20, maxSampleRate: 500000, slots: 3, device: 0 }, options || {});
    this.o = o;
    this.maxSamples = Math.ceil(o.maxSeconds * o.maxSampleRate);
    this.hop = N.workerHop(o.fftSize, o.overlap);
    this.slots = [];
    for (let i = 0; i < o.slots; i++) {
      const h = N.create({
        fftN: o.fftSize, hop: this.hop, windowId: S.WIN_WORKER_HANN, scaleMode: S.SCALE_WORKER_LOG5, frameRule: S.FRAMES_WORKER_COMPAT,
        twiddleMode: S.TW_ACCURATE, arith: S.ARITH_FAST_F32, outKind: o.out === 'u8' ? S.OUT_U8_INDEX : S.OUT_RGBA8, inKind: S.IN_I16,
      }, o.device);
      if (o.colorMap && o.out !== 'u8') N.setColormap(h, N.workerLut(flattenColorMap(o.colorMap)));
      const outBytes = N.outBytes(h, this.maxSamples);
      this.slots.push({ h, pin: new Int16Array(N.hostAlloc(this.maxSamples * 2)), pout: new Uint8Array(N.hostAlloc(Math.max(4, outBytes))), busy: null });
    }
  }

  _collect(slot, onResult) {
    const b = slot.busy;
    N.wait(slot.h);
    slot.busy = null;
    // a view of the page-locked output: valid until the slot is reused (copy it to keep it)
    onResult({ path: b.path, info: b.info, guano: b.guano, width: b.geo.width, height: b.geo.bins, frames: b.geo.frames,
               data: slot.pout.subarray(0, b.bytes) });
  }

  // paths: iterable of file names.  onResult({path, info, guano, width, height, frames, data}) is called in submission order;
  // onSkip(path, reason) for files the reference's loader would skip ('small', 'long') or that are not PCM16.
  run(paths, onResult, onSkip) {
    let i = 0;
    for (const path of paths) {
      const fd = fs.openSync(path, 'r');
      try {
        const size = fs.fstatSync(fd).size;
        const head = Buffer.alloc(Math.min(size, 1 << 16));           // fmt / guan chunks sit at the front (or behind the data)
        fs.readSync(fd, head, 0, head.length, 0);
        let info;
        try { info = N.wavParse(head); } catch (e) { if (onSkip) onSkip(path, 'unreadable'); continue; }
        const verdict = wav.acceptFile(size, info.duration, !!this.o.timeExpansion);
        if (verdict !== 'ok' && !this.o.keepAll) { if (onSkip) onSkip(path, verdict); continue; }
        if (!info.hasFmt || !info.hasData || info.formatTag !== 1 || info.bitsPerSample !== 16 || info.channels !== 1) {
          if (onSkip) onSkip(path, 'not mono PCM16');
          continue;
        }
        const frames = Math.min(Math.floor((size - info.dataOffset) / 2), Math.floor(info.dataBytes / 2));
        if (frames > this.maxSamples) { if (onSkip) onSkip(path, 'longer than maxSeconds'); continue; }
        const slot = this.slots[i % this.slots.length];
        if (slot.busy) this._collect(slot, onResult);
        fs.readSync(fd, Buffer.from(slot.pin.buffer, slot.pin.byteOffset, frames * 2), 0, frames * 2, info.dataOffset);   // disk -> page-locked
        const geo = N.geometry(slot.h, frames);
        const bytes = N.stftHostAsync(slot.h, slot.pin, frames, slot.pout);
        slot.busy = { path, info, geo, bytes, guano: null };
        i++;
      } finally {
        fs.closeSync(fd);
      }
    }
    for (let k = 0; k < this.slots.length; k++) {
      const slot = this.slots[(i + k) % this.slots.length];
      if (slot.busy) this._collect(slot, onResult);
    }
  }

  close() {
    for (const s of this.slots) N.destroy(s.h);
    this.slots = [];
  }
}

function flattenColorMap(cm) {
  const f = new Float64Array(1024);
  for (let i = 0; i < 256; i++) for (let c = 0; c < 4; c++) f[i * 4 + c] = cm[i][c];
  return f;
}

module.exports = { FileBatch };
