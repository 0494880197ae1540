'use strict';
// GPU test of the batch host: synthetic .wav files on disk -> FileBatch (overlapped, page-locked buffers) must give,
// file by file and in order, the bytes a synchronous stft() of the same samples gives; the loader's filters skip the rest.
const fs = require('fs');
const os = require('os');
const path = require('path');
const S = require('./index.js');
const { FileBatch } = require('./batch.js');
const N = S.native;

function wavFile(samples, rate, opts) {
  opts = opts || {};
  const ch = opts.channels || 1, bits = opts.bits || 16;
  const data = Buffer.alloc(samples.length * ch * bits / 8);
  for (let i = 0; i < samples.length; i++) for (let c = 0; c < ch; c++) data.writeInt16LE(samples[i], (i * ch + c) * 2);
  const fmt = Buffer.alloc(24);
  fmt.write('fmt ', 0, 4, 'latin1'); fmt.writeUInt32LE(16, 4); fmt.writeUInt16LE(1, 8); fmt.writeUInt16LE(ch, 10); fmt.writeUInt32LE(rate, 12);
  fmt.writeUInt32LE(rate * ch * bits / 8, 16); fmt.writeUInt16LE(ch * bits / 8, 20); fmt.writeUInt16LE(bits, 22);
  const guan = Buffer.from('GUANO|Version: 1.0\nTimestamp: 2024-06-01 21:00:00+08:00\n', 'utf8');
  const gh = Buffer.alloc(8); gh.write('guan', 0, 4, 'latin1'); gh.writeUInt32LE(guan.length, 4);
  const dh = Buffer.alloc(8); dh.write('data', 0, 4, 'latin1'); dh.writeUInt32LE(data.length, 4);
  const body = Buffer.concat([fmt, gh, guan, guan.length & 1 ? Buffer.alloc(1) : Buffer.alloc(0), dh, data]);
  const head = Buffer.alloc(12); head.write('RIFF', 0, 4, 'latin1'); head.writeUInt32LE(4 + body.length, 4); head.write('WAVE', 8, 4, 'latin1');
  return Buffer.concat([head, body]);
}

const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'sonobatch-'));
let seed = 7;
const rnd = () => (seed = (Math.imul(1103515245, seed) + 12345) & 0x7fffffff) / 0x40000000 - 1;
const files = [], samplesOf = {};
for (let i = 0; i < 9; i++) {
  const n = 120000 + ((i * 37123) % 90000);                         // 240-420 KiB of PCM16: above the 200 KiB filter
  const x = new Int16Array(n);
  for (let k = 0; k < n; k++) x[k] = Math.round(3000 * rnd() + 9000 * Math.sin(k * (0.2 + 0.05 * i)) * (k % 30000 < 2000 ? 1 : 0));
  const p = path.join(dir, 'f' + i + '.wav');
  fs.writeFileSync(p, wavFile(x, 384000));
  files.push(p); samplesOf[p] = x;
}
const small = path.join(dir, 'small.wav'); fs.writeFileSync(small, wavFile(new Int16Array(5000), 384000));
const stereo = path.join(dir, 'stereo.wav'); fs.writeFileSync(stereo, wavFile(new Int16Array(120000), 384000, { channels: 2 }));
const order = files.slice(0, 4).concat([small, stereo], files.slice(4));

let failures = 0, checks = 0;
const ok = (c, m) => { checks++; if (!c) { failures++; console.error('FAIL ' + m); } };
for (const out of ['rgba', 'u8']) {
  const sync = N.create({ fftN: 1024, hop: 256, windowId: S.WIN_WORKER_HANN, scaleMode: S.SCALE_WORKER_LOG5, frameRule: S.FRAMES_WORKER_COMPAT,
                          twiddleMode: S.TW_ACCURATE, arith: S.ARITH_FAST_F32, outKind: out === 'u8' ? S.OUT_U8_INDEX : S.OUT_RGBA8, inKind: S.IN_I16 }, 0);
  const batch = new FileBatch({ out, slots: 3, maxSeconds: 2, maxSampleRate: 384000 });
  const got = [], skipped = [];
  batch.run(order, (r) => {
    const want = N.stft(sync, samplesOf[r.path]);
    let same = want.length === r.data.length;
    for (let i = 0; same && i < want.length; i++) same = want[i] === r.data[i];
    ok(same, out + ' ' + path.basename(r.path) + ' bytes');
    ok(r.height === 512 && r.width === Math.max(1, Math.ceil((samplesOf[r.path].length - 1024) / 256)) && r.info.sampleRate === 384000, out + ' geometry');
    got.push(r.path);
  }, (p, why) => skipped.push(path.basename(p) + ':' + why));
  ok(JSON.stringify(got) === JSON.stringify(files), out + ' order');
  ok(JSON.stringify(skipped) === JSON.stringify(['small.wav:small', 'stereo.wav:not mono PCM16']), out + ' skips ' + JSON.stringify(skipped));
  batch.close();
  N.destroy(sync);
}
for (const f of fs.readdirSync(dir)) fs.unlinkSync(path.join(dir, f));
fs.rmdirSync(dir);
console.log(JSON.stringify({ checks, failures }));
process.exit(failures ? 1 : 0);
