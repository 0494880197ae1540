'use strict';
// Drop-in for the wavesurfer Spectrogram plugin's computing methods, on the GPU.
// Mirrors modules/spectrogram.esm.js (constructor options :207-272, getFrequencies :492-542,
// resample :595-640, drawSpectrogram's LUT/flip/crop :345-388) and the factory of
// modules/wsManager.js:29-54.  Same option names, defaults, error texts and return shapes.
const S = require('./index.js');
const N = S.native;

// Non-linear frequency scales of the plugin (option `scale`), spectrogram.esm.js:389-478: a bank of
// fftSamples/2 two-tap filters over [0, sampleRate/2], evaluated here with this V8's log10 / pow so that
// the weights are the very doubles the reference computes.
const SCALE_IDS = { linear: 0, mel: 1, logarithmic: 2, bark: 3, erb: 4 };
const ERB_A = 1e3 * Math.log(10) / 107.939;                                         // :202
const HZ_TO = {
  mel: (t) => 2595 * Math.log10(1 + t / 700),
  logarithmic: (t) => Math.log10(Math.max(1, t)),
  bark: (t) => { let e = 26.81 * t / (1960 + t) - 0.53; if (e < 2) e += 0.15 * (2 - e); if (e > 20.1) e += 0.22 * (e - 20.1); return e; },
  erb: (t) => ERB_A * Math.log10(1 + 0.00437 * t),
};
const TO_HZ = {
  mel: (t) => 700 * (Math.pow(10, t / 2595) - 1),
  logarithmic: (t) => Math.pow(10, t),
  bark: (t) => { if (t < 2) t = (t - 0.3) / 0.85; if (t > 20.1) t = (t + 4.422) / 1.22; return (t + 0.53) / (26.28 - t) * 1960; },
  erb: (t) => (Math.pow(10, t / ERB_A) - 1) / 0.00437,
};
function filterBank(scale, n, sampleRate) {
  const filters = n / 2, toScale = HZ_TO[scale], toHz = TO_HZ[scale];
  const lo = toScale(0), hi = toScale(sampleRate / 2), h = sampleRate / n;
  const bin = new Int32Array(filters), w0 = new Float64Array(filters), w1 = new Float64Array(filters);
  for (let e = 0; e < filters; e++) {                                               // :401-408
    const s = toHz(lo + e / filters * (hi - lo));
    const o = Math.floor(s / h), l = o * h, c = (s - l) / ((o + 1) * h - l);
    bin[e] = o; w0[e] = 1 - c; w1[e] = c;
  }
  return { bin, w0, w1 };
}

class Spectrogram {
  static create(options) { return new Spectrogram(options || {}); }

  constructor(t) {
    this.options = t;
    if (t.colorMap && typeof t.colorMap !== 'string') {                 // :212-219
      if (t.colorMap.length < 256) throw new Error('Colormap must contain 256 elements');
      for (let e = 0; e < t.colorMap.length; e++)
        if (t.colorMap[e].length !== 4) throw new Error('ColorMap entries must contain 4 values');
      this.colorMap = t.colorMap;
    } else {                                                            // :221-239
      this.colorMap = t.colorMap || 'roseus';
      switch (this.colorMap) {
        case 'gray':
          this.colorMap = [];
          for (let i = 0; i < 256; i++) { const e = (255 - i) / 256; this.colorMap.push([e, e, e, 1]); }
          break;
        case 'igray':
          this.colorMap = [];
          for (let i = 0; i < 256; i++) { const e = i / 256; this.colorMap.push([e, e, e, 1]); }
          break;
        default:
          throw Error("No such colormap '" + this.colorMap + "'");
      }
    }
    this.fftSamples = t.fftSamples || 512;                              // :240-249
    this.height = t.height || 200;
    this.noverlap = t.noverlap || null;
    this.windowFunc = t.windowFunc || 'hann';
    this.alpha = t.alpha;
    this.frequencyMin = t.frequencyMin || 0;
    this.frequencyMax = t.frequencyMax || 0;
    this.gainDB = (t.gainDB !== null && t.gainDB !== undefined) ? t.gainDB : 20;
    this.rangeDB = (t.rangeDB !== null && t.rangeDB !== undefined) ? t.rangeDB : 80;
    this.scale = t.scale || 'mel';
    // engine options without a counterpart in the reference
    this.arith = t.arith || 'fast';          // 'fast' float32 | 'exact' bit-identical double
    this.device = t.device || 0;
    this.canvasWidth = t.canvasWidth || 300; // stands in for canvas.width in the auto-overlap rule (:502-506)
    const cm = new Float64Array(1024);
    for (let i = 0; i < 256; i++) {
      const cc = this.colorMap[i] || [0, 0, 0, 1];
      for (let ch = 0; ch < 4; ch++) cm[4 * i + ch] = cc[ch];
    }
    this._colorMapUint = N.pluginLut(cm);                               // :262-271
    this._engines = new Map();               // GPU contexts, least recently used first
    this._maxEngines = t.maxEngines || 8;    // (automatic overlap makes one per distinct hop, i.e. per zoom step)
    this._evictHooks = new Set();            // called with a context's handle before it is destroyed (tile caches)
    this.buffer = null;
  }

  _engine(hop, sampleRate) {
    if (!(this.scale in SCALE_IDS)) throw new Error("unknown scale '" + this.scale + "'");
    const fbRate = this.scale !== 'linear' && sampleRate ? sampleRate : 0;
    const key = [this.fftSamples, hop, this.windowFunc, this.alpha, this.gainDB, this.rangeDB, this.arith, this.scale, fbRate].join(':');
    let h = this._engines.get(key);
    if (h) {
      this._engines.delete(key);            // most recently used last
      this._engines.set(key, h);
    } else {
      while (this._engines.size >= this._maxEngines) {
        const [oldKey, old] = this._engines.entries().next().value;
        this._engines.delete(oldKey);
        for (const hook of this._evictHooks) hook(old);   // device memory that lives on the context goes first
        N.destroy(old);
      }
      const n = this.fftSamples;
      const p = Math.floor(Math.log(n) / Math.LN2);
      if (Math.pow(2, p) !== n) throw 'Invalid buffer size, must be a power of 2.';   // :168-169
      const exact = this.arith === 'exact';
      h = N.create({
        fftN: n, hop, windowId: S.windowId(this.windowFunc), alpha: this.alpha || 0,
        gainDB: this.gainDB, rangeDB: this.rangeDB, scaleMode: S.SCALE_PLUGIN_DB, frameRule: S.FRAMES_PLUGIN_STRICT,
        twiddleMode: exact ? S.TW_HOST_TABLE : S.TW_PLUGIN_RECURRENCE,
        arith: exact ? S.ARITH_EXACT_F64 : S.ARITH_FAST_F32, outKind: S.OUT_U8_INDEX, inKind: S.IN_F32,
      }, this.device);
      if (exact) {                          // this V8's own recurrence doubles
        const tw = S.pluginTwiddles(n);
        N.setTables(h, null, tw.cos, tw.sin);
      }
      N.setColormap(h, this._colorMapUint);
      if (fbRate) {                         // :509-521, with this V8's own log10 / pow
        const fb = filterBank(this.scale, n, fbRate);
        N.setFilterbank(h, fb.bin, fb.w0, fb.w1);
      }
      this._engines.set(key, h);
    }
    return h;
  }

  _anyEngine() {
    for (const h of this._engines.values()) return h;
    return this._engine(this.fftSamples);
  }

  // :492-542: Array[channels] of Array[frames] of Uint8Array(fftSamples/2)
  getFrequencies(t) {
    if (!t) return;
    const r = this.fftSamples;
    const channels = this.options.splitChannels ? t.numberOfChannels : 1;            // :495
    this.frequencyMax = this.frequencyMax || t.sampleRate / 2;                       // :496
    this.buffer = t;
    const noverlap = N.pluginNoverlap(r, this.noverlap || 0, t.length, this.canvasWidth);   // :502-506
    const hop = r - noverlap;
    if (hop <= 0) throw new Error('noverlap must be smaller than fftSamples');
    const h = this._engine(hop, t.sampleRate);
    const out = [];
    for (let ch = 0; ch < channels; ch++) {
      const flat = N.stft(h, t.getChannelData(ch));
      const bins = r / 2, frames = flat.length / bins, rows = new Array(frames);
      for (let f = 0; f < frames; f++) rows[f] = flat.subarray(f * bins, (f + 1) * bins);
      rows.flat = flat;                     // the same data as one Uint8Array(frames*bins)
      out.push(rows);
    }
    return out;
  }

  // :595-640: Array[F] of Uint8Array(B) -> Array[width] of Uint8Array(B)
  resample(frames, width) {
    const F = frames.length;
    if (!F) return [];
    const B = frames[0].length;
    let flat = frames.flat;
    if (!flat || flat.length !== F * B) {
      flat = new Uint8Array(F * B);
      for (let f = 0; f < F; f++) flat.set(frames[f], f * B);
    }
    const outFlat = N.resample(this._anyEngine(), flat, F, B, width);
    const out = new Array(width);
    for (let a = 0; a < width; a++) out[a] = outFlat.subarray(a * B, (a + 1) * B);
    out.flat = outFlat;
    return out;
  }

  // LUT + flip + crop window of drawSpectrogram (:345-388) for one channel: the full RGBA image
  // (bins rows x width columns) and the source rows the reference hands to createImageBitmap.
  drawSpectrogram(frames, width) {
    const cols = this.resample(frames, width);
    const B = cols.length ? cols[0].length : this.fftSamples / 2;
    const h = this._anyEngine();
    N.setColormap(h, this._colorMapUint);
    const data = N.colorize(h, cols.flat || new Uint8Array(0), width, B, 0, B);
    const crop = N.pluginCropScaled(B, this.buffer.sampleRate, this.frequencyMin, this.frequencyMax, SCALE_IDS[this.scale]);   // :377-380
    return { width, height: B, data, crop };
  }

  // render() (:336-344) for channel 0
  render(buffer, width) {
    const frames = this.getFrequencies(buffer)[0];
    return this.drawSpectrogram(frames, width);
  }

  // The whole of drawSpectrogram (:345-388) for one channel, down to the canvas pixels: ImageData-shaped
  // { width, height, data }.  The browser's drawImage filter is implementation-defined; the engine's is
  // bilinear along y between pixel centres.
  drawCanvas(frames, width) {
    const cols = this.resample(frames, width);
    const B = cols.length ? cols[0].length : this.fftSamples / 2;
    const h = this._anyEngine();
    N.setColormap(h, this._colorMapUint);
    const sr = this.buffer.sampleRate, scale = SCALE_IDS[this.scale];
    const crop = N.pluginCropScaled(B, sr, this.frequencyMin, this.frequencyMax, scale);                 // :377-380
    const rect = N.pluginDrawRect(sr, this.frequencyMin, this.frequencyMax, scale, this.height);         // :381
    const last = this.colorMap[this.colorMap.length - 1];
    const bg = rect.fill ? N.pluginBackground(Float64Array.from(last)) : null;                           // :357-361
    const data = N.canvasDraw(h, cols.flat || new Uint8Array(0), width, B, crop.sy, crop.sh, rect.dy, rect.dh, this.height, bg);
    return { width, height: this.height, data };
  }

  renderCanvas(buffer, width) {
    return this.drawCanvas(this.getFrequencies(buffer)[0], width);
  }

  destroy() {
    for (const h of this._engines.values()) N.destroy(h);
    this._engines.clear();
  }
}

// modules/wsManager.js:29-54
function createSpectrogramPlugin({ colorMap, height = 800, frequencyMin = 10, frequencyMax = 128, fftSamples = 1024,
                                   noverlap = null, windowFunc = 'hann', ...engineOptions }) {
  const baseOptions = Object.assign({
    labels: false, height, fftSamples, frequencyMin: frequencyMin * 1000, frequencyMax: frequencyMax * 1000,
    scale: 'linear', windowFunc, colorMap,
  }, engineOptions);
  if (noverlap !== null) baseOptions.noverlap = noverlap;
  return Spectrogram.create(baseOptions);
}

// modules/wsManager.js:87-89
function noverlapFromPercent(fftSamples, overlapPercent) {
  return overlapPercent !== null ? N.overlapFromPercent(fftSamples, overlapPercent) : null;
}

module.exports = { Spectrogram, createSpectrogramPlugin, noverlapFromPercent };
