#!/usr/bin/env node
// Golden-vector generator (test infrastructure; run once in the build container).
//
// Evaluates the reference's own JavaScript, read by path from /root/reference at
// generation time, inside `vm` contexts with small DOM/canvas stand-ins, drives it with
// seeded synthetic inputs, and writes inputs + reference outputs as raw little-endian
// binaries plus `manifest.json` into this directory.  Only data is written here: no
// reference source text is stored.
//
//   node tests/golden/gen_golden.js [/root/reference]
//
// Covered reference entry points:
//   spectrogramWorker.js            onmessage/init/render, hannWindow, twiddles, fftOptimized
//   modules/spectrogram.esm.js      FFT object (windows, tables, calculateSpectrum),
//                                   getFrequencies, resample, drawSpectrogram (LUT/flip/crop)
//   modules/autoid_HK.js            speciesRules table, autoIdHK verdicts
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');

const REF = process.argv[2] || '/root/reference';
const OUT = __dirname;
const manifest = { generator: 'tests/golden/gen_golden.js', node: process.version, cases: [] };

// ---------------------------------------------------------------- binary helpers
function writeBin(name, typed) {
  fs.writeFileSync(path.join(OUT, name), Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength));
  return name;
}
function flattenFrames(frames) {           // Array[F] of Uint8Array(B) -> Uint8Array(F*B)
  const F = frames.length, B = F ? frames[0].length : 0;
  const out = new Uint8Array(F * B);
  for (let f = 0; f < F; f++) out.set(frames[f], f * B);
  return out;
}

// ---------------------------------------------------------------- seeded inputs
function lcg(seed) {
  let s = seed >>> 0;
  return () => { s = (Math.imul(1103515245, s) + 12345) & 0x7fffffff; return s / 0x80000000; };
}
function sigNoise(L, seed, amp) {
  const r = lcg(seed), x = new Float32Array(L);
  for (let i = 0; i < L; i++) x[i] = amp * (2 * r() - 1);
  return x;
}
function sigTone(L, sr, f, amp, floorAmp, seed) {
  const r = lcg(seed), x = new Float32Array(L);
  for (let i = 0; i < L; i++) x[i] = amp * Math.sin(2 * Math.PI * f * i / sr) + floorAmp * (2 * r() - 1);
  return x;
}
// bat-like call: noise 0.01 + FM down-sweep pulses (Hann envelope) + CF tone in the 2nd half
function sigBat(L, sr, seed) {
  const r = lcg(seed), x = new Float32Array(L);
  const pulseLen = Math.floor(0.005 * sr), period = Math.floor(0.02 * sr);
  for (let i = 0; i < L; i++) {
    let v = 0.01 * (2 * r() - 1);
    const p = i % period;
    if (p < pulseLen) {
      const u = p / pulseLen, t = p / sr;
      const f0 = 80000, f1 = 35000;
      const phase = 2 * Math.PI * (f0 * t + 0.5 * (f1 - f0) * t * t / (pulseLen / sr));
      v += 0.3 * 0.5 * (1 - Math.cos(2 * Math.PI * u)) * Math.sin(phase);
    }
    if (i >= L / 2) v += 0.05 * Math.sin(2 * Math.PI * 110000 * i / sr);
    x[i] = v;
  }
  return x;
}
function sigImpulse(L, at) { const x = new Float32Array(L); x[at] = 1; return x; }
function sigFullScale(L, sr) {
  const x = new Float32Array(L);
  for (let i = 0; i < L; i++) x[i] = Math.sin(2 * Math.PI * 48000 * i / sr) >= 0 ? 1 : -1;
  return x;
}

// ---------------------------------------------------------------- load the reference
function loadWorker() {
  let src = fs.readFileSync(path.join(REF, 'spectrogramWorker.js'), 'utf8');
  src += '\n;self.__x = { fftOptimized, hannWindow, getTwiddleFactorsCached, getBitReverseCached, setColorMapUint,' +
         ' getColorMapUint: () => colorMapUint };\n';
  const posted = [];
  const self = { postMessage: (m) => posted.push(m) };
  const ctx = vm.createContext({ self, Math, Map, Float32Array, Uint32Array, Uint8ClampedArray, Uint8Array, Array, Object, console });
  vm.runInContext(src, ctx, { filename: 'spectrogramWorker.js' });
  let lastImage = null;
  const canvas = {
    width: 0, height: 0,
    getContext: () => ({
      createImageData: (w, h) => ({ width: w, height: h, data: new Uint8ClampedArray(w * h * 4) }),
      putImageData: (img) => { lastImage = img; },
    }),
  };
  return { self, posted, canvas, take: () => { const i = lastImage; lastImage = null; return i; } };
}

function makeDocumentStub() {
  const mkEl = () => ({
    style: {}, children: [],
    appendChild(c) { this.children.push(c); return c; },
    setAttribute() {}, addEventListener() {}, remove() {},
    getContext: () => ({ fillStyle: '', fillRect() {}, drawImage() {} }),
  });
  return { createElement: mkEl, createElementNS: mkEl, querySelector: () => mkEl(), createTextNode: () => ({}) };
}

function loadPlugin() {
  let src = fs.readFileSync(path.join(REF, 'modules', 'spectrogram.esm.js'), 'utf8');
  src = src.replace(/export\s*\{\s*h\s+as\s+default\s*\}\s*;?/, 'globalThis.__plugin = { Spectrogram: h, FFT: a };');
  const captured = { bitmaps: [], fills: [], draws: [] };
  class ImageData { constructor(w, h) { this.width = w; this.height = h; this.data = new Uint8ClampedArray(w * h * 4); } }
  const sandbox = {
    Math, Promise, Set, Object, Array, Float32Array, Uint32Array, Uint8Array, Uint8ClampedArray, Error, isNaN, console,
    document: makeDocumentStub(),
    ImageData,
    createImageBitmap: (img, sx, sy, sw, sh) => {
      captured.bitmaps.push({ img, sx, sy, sw, sh });
      return { then: (cb) => cb({ img, sx, sy, sw, sh }) };
    },
  };
  sandbox.globalThis = sandbox;
  const ctx = vm.createContext(sandbox);
  vm.runInContext(src, ctx, { filename: 'spectrogram.esm.js' });
  return { Spectrogram: sandbox.__plugin.Spectrogram, FFT: sandbox.__plugin.FFT, captured };
}

function loadAutoId() {
  let src = fs.readFileSync(path.join(REF, 'modules', 'autoid_HK.js'), 'utf8');
  src = src.replace(/export\s+function/g, 'function');
  src += '\n;globalThis.__a = { autoIdHK, speciesRules, inRange };\n';
  const sandbox = { Math, Array, Object, isNaN, console };
  sandbox.globalThis = sandbox;
  vm.runInContext(src, vm.createContext(sandbox), { filename: 'autoid_HK.js' });
  return sandbox.__a;
}

function audioBuffer(x, sr) {
  return { sampleRate: sr, length: x.length, numberOfChannels: 1, getChannelData: () => x };
}

// default grey map of the page (formula of modules/brightnessControl.js:45-51 evaluated here
// with the page defaults brightness -0.06, gain 2.1, contrast 1.25)
function defaultColorMap(brightness = -0.06, gain = 2.1, contrast = 1.25) {
  return Array.from({ length: 256 }, (_, i) => {
    const t = Math.pow(i / 255, gain);
    let v = 1 - t + brightness;
    v = (v - 0.5) * contrast + 0.5;
    v = Math.max(0, Math.min(1, v));
    return [v, v, v, 1];
  });
}
function rainbowColorMap() {   // non-grey, non-monotone map so channel order mistakes show
  return Array.from({ length: 256 }, (_, i) => [i / 255, ((i * 7) % 256) / 255, (255 - i) / 255, ((i * 3) % 256) / 255]);
}
function colorMapToF64(cm) { const f = new Float64Array(1024); cm.forEach((c, i) => f.set(c, i * 4)); return f; }

// ================================================================ 1. plugin FFT tables + windows
const P = loadPlugin();
const WINDOWS = ['bartlett', 'bartlettHann', 'blackman', 'cosine', 'gauss', 'hamming', 'hann', 'lanczoz', 'rectangular', 'triangular'];
for (const N of [256, 512, 1024, 2048, 4096]) {
  const all = new Float32Array(N * WINDOWS.length);
  WINDOWS.forEach((w, wi) => { all.set(new P.FFT(N, 384000, w, undefined).windowValues, wi * N); });
  const fft = new P.FFT(N, 384000, 'hann', undefined);
  manifest.cases.push({
    kind: 'plugin_tables', N, windows: WINDOWS,
    window_values: writeBin(`plugin_windows_${N}.f32`, all),
    sin_table: writeBin(`plugin_sin_${N}.f32`, fft.sinTable),
    cos_table: writeBin(`plugin_cos_${N}.f32`, fft.cosTable),
    reverse_table: writeBin(`plugin_rev_${N}.u32`, fft.reverseTable),
  });
}
// alpha variants of the two parametrised windows
{
  const N = 1024, alphas = [0.1, 0.16, 0.25, 0.5];
  const all = new Float32Array(N * alphas.length * 2);
  alphas.forEach((al, i) => {
    all.set(new P.FFT(N, 384000, 'blackman', al).windowValues, (2 * i) * N);
    all.set(new P.FFT(N, 384000, 'gauss', al).windowValues, (2 * i + 1) * N);
  });
  manifest.cases.push({ kind: 'plugin_alpha_windows', N, alphas, order: 'alpha-major [blackman, gauss]',
    window_values: writeBin('plugin_alpha_windows_1024.f32', all) });
}
// unknown window name -> error text
{
  let msg = null;
  try { new P.FFT(512, 384000, 'kaiser', undefined); } catch (e) { msg = String(e.message || e); }
  manifest.cases.push({ kind: 'plugin_window_error', name: 'kaiser', message: msg });
}

// ================================================================ 2. worker tables
const Wk = loadWorker();
for (const N of [256, 512, 1024, 2048, 4096]) {
  const tw = Wk.self.__x.getTwiddleFactorsCached(N);
  // flatten: stage s (half size n1 = 2^s) occupies [n1-1, 2*n1-1) in an (N-1)-entry table
  const c = new Float64Array(N - 1), s = new Float64Array(N - 1);
  for (let st = 0; st < tw.length; st++) {
    const n1 = 1 << st;
    for (let j = 0; j < n1; j++) { c[n1 - 1 + j] = tw[st][j].c; s[n1 - 1 + j] = tw[st][j].s; }
  }
  manifest.cases.push({
    kind: 'worker_tables', N,
    hann: writeBin(`worker_hann_${N}.f32`, Wk.self.__x.hannWindow(N)),
    twiddle_cos: writeBin(`worker_twc_${N}.f64`, c),
    twiddle_sin: writeBin(`worker_tws_${N}.f64`, s),
  });
}

// ================================================================ 3. signals
const SR = 384000;
const signals = {};
function addSignal(name, x, sr) {
  signals[name] = { x, sr: sr || SR, file: writeBin(`sig_${name}.f32`, x) };
}
addSignal('noise', sigNoise(6400, 12345, 0.5));
addSignal('tone', sigTone(6400, SR, 45000, 0.9, 1e-5, 12346));
addSignal('bat', sigBat(11520, SR, 12347));            // (11520-1024)/256 = 41 exactly
addSignal('bat_ragged', sigBat(11433, SR, 12348));     // (11433-1024) % 256 = 169
addSignal('silence', new Float32Array(5120));
addSignal('fullscale', sigFullScale(5120, SR));
addSignal('impulse', sigImpulse(5120, 1500));
addSignal('quiet', sigNoise(5120, 12349, 3e-6));        // below both display floors
addSignal('bat500', sigBat(9000, 500000, 12350), 500000);
{
  const x = sigNoise(5120, 12351, 0.5); x[2000] = NaN;
  addSignal('nan_probe', x);
}
{
  // tone whose plugin magnitude lands (nearly) on 10^(-gain/20): probes the uint8 wrap at r == -gain
  addSignal('edge_gain', sigTone(5120, SR, 48000, 0.4, 0, 1));
}
addSignal('dc_half', new Float32Array(5120).fill(0.5));  // rectangular window: bin 0 magnitude == 1.0 exactly
addSignal('short_eq', sigNoise(1024, 12352, 0.5));      // L == N
addSignal('short_lt', sigNoise(700, 12353, 0.5));       // L <  N
addSignal('short_gt', sigNoise(1025, 12354, 0.5));      // L == N+1
manifest.signals = {};
for (const k of Object.keys(signals)) manifest.signals[k] = { file: signals[k].file, length: signals[k].x.length, sample_rate: signals[k].sr };

// ================================================================ 4. plugin calculateSpectrum magnitudes
function spectrumCase(sigName, N, windowFunc, alpha, nFrames, hop) {
  const { x, sr } = signals[sigName];
  const fft = new P.FFT(N, sr, windowFunc, alpha);
  const mags = new Float32Array(nFrames * (N / 2));
  const peaks = [];
  for (let f = 0; f < nFrames; f++) {
    const m = fft.calculateSpectrum(x.subarray(f * hop, f * hop + N));
    mags.set(m, f * (N / 2));
    peaks.push({ peak: fft.peak, peakBand: fft.peakBand });
  }
  manifest.cases.push({
    kind: 'plugin_spectrum', signal: sigName, N, window: windowFunc, alpha: alpha === undefined ? null : alpha,
    frames: nFrames, hop, mags: writeBin(`pspec_${sigName}_${N}_${windowFunc}.f32`, mags), running_peak: peaks,
  });
}
for (const s of ['noise', 'tone', 'bat', 'silence', 'fullscale', 'impulse']) spectrumCase(s, 1024, 'hann', undefined, 8, 256);
spectrumCase('tone', 512, 'hamming', undefined, 8, 128);
spectrumCase('tone', 2048, 'blackman', undefined, 4, 512);
spectrumCase('tone', 4096, 'hann', undefined, 2, 256);
spectrumCase('bat', 4096, 'gauss', undefined, 4, 256);
spectrumCase('noise', 256, 'rectangular', undefined, 8, 64);

// ================================================================ 5. plugin getFrequencies
function makePlugin(opts, width) {
  const p = P.Spectrogram.create(opts);
  p.wavesurfer = { getWrapper: () => ({ offsetWidth: width }), options: {}, getDecodedData: () => null, on: () => () => {}, un() {} };
  p.canvas.width = width;           // read by the auto-noverlap rule
  return p;
}
let gfId = 0;
function getFreqCase(sigName, o) {
  const { x, sr } = signals[sigName];
  const opts = { fftSamples: o.N, windowFunc: o.window || 'hann', colorMap: 'gray', scale: 'linear' };
  if (o.noverlap != null) opts.noverlap = o.noverlap;
  if (o.gainDB != null) opts.gainDB = o.gainDB;
  if (o.rangeDB != null) opts.rangeDB = o.rangeDB;
  if (o.alpha != null) opts.alpha = o.alpha;
  const p = makePlugin(opts, o.canvasWidth || 300);
  const out = p.getFrequencies(audioBuffer(x, sr));
  const frames = out[0];
  const id = `gf${String(gfId++).padStart(2, '0')}`;
  manifest.cases.push({
    kind: 'plugin_get_frequencies', id, signal: sigName, N: o.N, window: opts.windowFunc, alpha: o.alpha == null ? null : o.alpha,
    noverlap: o.noverlap == null ? null : o.noverlap, canvas_width: o.canvasWidth || 300,
    gainDB: p.gainDB, rangeDB: p.rangeDB, frames: frames.length, bins: o.N / 2,
    data: frames.length ? writeBin(`${id}.u8`, flattenFrames(frames)) : null,
  });
}
for (const s of ['noise', 'tone', 'bat', 'bat_ragged', 'silence', 'fullscale', 'impulse', 'quiet', 'nan_probe', 'edge_gain'])
  getFreqCase(s, { N: 1024, noverlap: 768 });
getFreqCase('bat', { N: 1024, noverlap: 512 });
getFreqCase('bat', { N: 1024, noverlap: 960 });
getFreqCase('bat', { N: 1024 /* auto noverlap */, canvasWidth: 40 });
getFreqCase('bat', { N: 1024 /* auto noverlap, clamps to 0 */, canvasWidth: 5 });
getFreqCase('bat_ragged', { N: 1024, noverlap: 1 });
getFreqCase('bat', { N: 1024, noverlap: 768, gainDB: 0, rangeDB: 100 });
getFreqCase('bat', { N: 1024, noverlap: 768, gainDB: 40, rangeDB: 60 });
getFreqCase('tone', { N: 1024, noverlap: 768, gainDB: 6, rangeDB: 30 });
for (const w of WINDOWS) getFreqCase('bat_ragged', { N: 1024, noverlap: 768, window: w });
getFreqCase('bat', { N: 1024, noverlap: 768, window: 'gauss', alpha: 0.4 });
getFreqCase('bat', { N: 1024, noverlap: 768, window: 'blackman', alpha: 0.3 });
getFreqCase('dc_half', { N: 1024, noverlap: 768, window: 'rectangular', gainDB: 0, rangeDB: 100 });   // r == -gain -> 256 wraps to 0
getFreqCase('dc_half', { N: 1024, noverlap: 768, window: 'rectangular' });
getFreqCase('bat', { N: 256, noverlap: 192 });
getFreqCase('bat', { N: 512, noverlap: 384 });
getFreqCase('bat', { N: 2048, noverlap: 1536 });
getFreqCase('bat', { N: 4096, noverlap: 3840 });
getFreqCase('tone', { N: 4096, noverlap: 3840 });
getFreqCase('bat500', { N: 1024, noverlap: 768 });
getFreqCase('short_eq', { N: 1024, noverlap: 768 });
getFreqCase('short_lt', { N: 1024, noverlap: 768 });
getFreqCase('short_gt', { N: 1024, noverlap: 768 });

// ================================================================ 6. plugin resample
function resampleCase(sigName, N, noverlap, W) {
  const { x, sr } = signals[sigName];
  const p = makePlugin({ fftSamples: N, noverlap, colorMap: 'gray', scale: 'linear' }, W);
  const frames = p.getFrequencies(audioBuffer(x, sr))[0];
  const out = p.resample(frames);
  manifest.cases.push({
    kind: 'plugin_resample', signal: sigName, N, noverlap, in_frames: frames.length, out_width: W, bins: N / 2,
    data: writeBin(`rs_${sigName}_${N}_${W}.u8`, flattenFrames(out)),
  });
}
for (const W of [41, 20, 82, 37, 1, 100]) resampleCase('bat', 1024, 768, W);    // F = 41
resampleCase('bat_ragged', 512, 384, 53);

// ================================================================ 7. plugin render (LUT, flip, crop window)
function renderCase(sigName, o) {
  const { x, sr } = signals[sigName];
  const cm = o.colorMapName === 'rainbow' ? rainbowColorMap() : (o.colorMapName === 'default' ? defaultColorMap() : o.colorMapName);
  const opts = { fftSamples: o.N, noverlap: o.noverlap, colorMap: cm, scale: 'linear', height: o.height || 800,
                 frequencyMin: o.fmin, frequencyMax: o.fmax, windowFunc: o.window || 'hann' };
  const p = makePlugin(opts, o.W);
  p.wavesurfer.getDecodedData = () => audioBuffer(x, sr);
  P.captured.bitmaps.length = 0;
  p.render();
  const b = P.captured.bitmaps[0];
  const id = `pr_${sigName}_${o.N}_${o.W}_${o.tag}`;
  manifest.cases.push({
    kind: 'plugin_render', id, signal: sigName, N: o.N, noverlap: o.noverlap, window: opts.windowFunc, out_width: o.W,
    colormap: typeof o.colorMapName === 'string' ? o.colorMapName : 'array',
    fmin: o.fmin, fmax: o.fmax, image_width: b.img.width, image_height: b.img.height,
    crop: { sx: b.sx, sy: b.sy, sw: b.sw, sh: b.sh },
    lut: writeBin(`${id}_lut.u8`, new Uint8Array(p._colorMapUint)),
    rgba: writeBin(`${id}.rgba`, new Uint8Array(b.img.data)),
  });
}
renderCase('bat', { N: 1024, noverlap: 768, W: 41, fmin: 10000, fmax: 128000, colorMapName: 'default', tag: 'a' });
renderCase('bat', { N: 1024, noverlap: 768, W: 30, fmin: 15000, fmax: 120000, colorMapName: 'rainbow', tag: 'b' });
renderCase('bat', { N: 4096, noverlap: 3840, W: 30, fmin: 15000, fmax: 120000, colorMapName: 'default', tag: 'c' });
renderCase('bat', { N: 1024, noverlap: 768, W: 41, fmin: 0, fmax: 250000, colorMapName: 'gray', tag: 'd' });   // fmax > Nyquist
renderCase('tone', { N: 512, noverlap: 384, W: 50, fmin: 20000, fmax: 90000, colorMapName: 'igray', tag: 'e' });

// ================================================================ 7b. non-linear frequency scales (filter banks)
{
  const SCALES = ['mel', 'logarithmic', 'bark', 'erb'];
  const fbCases = [];
  for (const scale of SCALES)
    for (const [N, sr] of [[1024, 384000], [512, 44100], [2048, 500000]]) {
      const p = makePlugin({ fftSamples: N, noverlap: N / 2, colorMap: 'gray', scale }, 10);
      const fns = { mel: [p.hzToMel, p.melToHz], logarithmic: [p.hzToLog, p.logToHz], bark: [p.hzToBark, p.barkToHz], erb: [p.hzToErb, p.erbToHz] }[scale];
      const bank = p.createFilterBank(N / 2, sr, fns[0], fns[1]);
      // each row has at most two non-zero taps: store (first tap index, weight of the first, weight of the second)
      const bin = new Int32Array(N / 2), w0 = new Float64Array(N / 2), w1 = new Float64Array(N / 2);
      for (let e = 0; e < N / 2; e++) {
        const row = bank[e];
        let o = -1;
        for (let k = 0; k < row.length; k++) if (row[k] !== 0 && row[k] !== undefined) { o = k; break; }
        // rows whose first weight is exactly 0 (c == 1) cannot be told from their second tap alone: recompute o as the reference does
        const h = sr / N, hz = fns[1](fns[0](0) + e / (N / 2) * (fns[0](sr / 2) - fns[0](0)));
        o = Math.floor(hz / h);
        bin[e] = o; w0[e] = row[o] === undefined ? 0 : row[o]; w1[e] = row[o + 1] === undefined ? 0 : row[o + 1];
      }
      const id = `fb_${scale}_${N}_${sr}`;
      fbCases.push({ scale, N, sr, bin: writeBin(id + '_bin.i32', bin), w0: writeBin(id + '_w0.f64', w0), w1: writeBin(id + '_w1.f64', w1) });
    }
  manifest.cases.push({ kind: 'plugin_filterbanks', banks: fbCases });
  const pts = [0, 1, 20, 440, 1000, 1960, 7000, 22050, 96000, 192000, 250000];
  const p0 = makePlugin({ fftSamples: 512, colorMap: 'gray', scale: 'mel' }, 10);
  const conv = {};
  for (const scale of SCALES) {
    p0.scale = scale;
    conv[scale] = pts.map((hz) => ({ hz, to: p0.hzToScale(hz), back: p0.scaleToHz(p0.hzToScale(hz)) }));
  }
  manifest.cases.push({ kind: 'plugin_scale_conversions', conv });
  let sid = 0;
  for (const scale of SCALES)
    for (const [sig, N, nov] of [['bat', 1024, 768], ['tone', 512, 256], ['nan_probe', 1024, 768]]) {
      const { x, sr } = signals[sig];
      const p = makePlugin({ fftSamples: N, noverlap: nov, colorMap: 'gray', scale }, 300);
      const frames = p.getFrequencies(audioBuffer(x, sr))[0];
      const id = `sc${String(sid++).padStart(2, '0')}`;
      manifest.cases.push({ kind: 'plugin_scale_frequencies', id, scale, signal: sig, N, noverlap: nov, frames: frames.length, bins: N / 2,
                            data: writeBin(`${id}.u8`, flattenFrames(frames)) });
    }
  // crop window of drawSpectrogram under each scale
  const rows = [];
  for (const scale of SCALES)
    for (const [N, sr, fmin, fmax] of [[1024, 384000, 10000, 128000], [1024, 384000, 15000, 120000], [512, 44100, 100, 8000], [2048, 500000, 0, 0], [1024, 96000, 1000, 60000]]) {
      const x = new Float32Array(N + 16);
      const p = makePlugin({ fftSamples: N, noverlap: N - 8, colorMap: 'gray', scale, frequencyMin: fmin, frequencyMax: fmax }, 2);
      p.wavesurfer.getDecodedData = () => audioBuffer(x, sr);
      P.captured.bitmaps.length = 0;
      p.render();
      const b = P.captured.bitmaps[0];
      rows.push({ scale, N, sr, fmin, fmax, sy: b.sy, sh: b.sh });
    }
  manifest.cases.push({ kind: 'plugin_scale_crop_grid', rows });
}

// crop-window grid (pure integer rule of drawSpectrogram for scale 'linear')
{
  const rows = [];
  for (const N of [512, 1024, 2048, 4096])
    for (const sr of [96000, 192000, 256000, 384000, 500000])
      for (const [fmin, fmax] of [[10000, 128000], [15000, 120000], [0, 0], [0, 48000], [20000, 250000], [1000, 47999], [12345, 98765]]) {
        const x = new Float32Array(N + 16);
        const p = makePlugin({ fftSamples: N, noverlap: N - 8, colorMap: 'gray', scale: 'linear', frequencyMin: fmin, frequencyMax: fmax }, 2);
        p.wavesurfer.getDecodedData = () => audioBuffer(x, sr);
        P.captured.bitmaps.length = 0;
        p.render();
        const b = P.captured.bitmaps[0];
        rows.push({ N, sr, fmin, fmax, sy: b.sy, sh: b.sh });
      }
  manifest.cases.push({ kind: 'plugin_crop_grid', rows });
}
// constructor validation errors
{
  const errs = [];
  for (const [label, cm] of [['short', Array.from({ length: 10 }, () => [0, 0, 0, 1])],
                             ['bad_entry', Array.from({ length: 256 }, () => [0, 0, 0])],
                             ['default_name', undefined], ['unknown_name', 'viridis']]) {
    let msg = null;
    try { P.Spectrogram.create({ colorMap: cm }); } catch (e) { msg = String(e.message || e); }
    errs.push({ label, message: msg });
  }
  let fftMsg = null;
  try { new P.FFT(1000, 384000, 'hann').calculateSpectrum(new Float32Array(1000)); } catch (e) { fftMsg = String(e.message || e); }
  errs.push({ label: 'non_pow2', message: fftMsg });
  let lenMsg = null;
  try { new P.FFT(1024, 384000, 'hann').calculateSpectrum(new Float32Array(512)); } catch (e) { lenMsg = String(e.message || e); }
  errs.push({ label: 'length_mismatch', message: lenMsg });
  manifest.cases.push({ kind: 'plugin_errors', errors: errs });
}

// ================================================================ 8. worker render
let wkId = 0;
function workerCase(sigName, o) {
  const W = loadWorker();                       // fresh module state per case (colour map is sticky)
  const { x, sr } = signals[sigName];
  const cm = o.colorMapName === 'rainbow' ? rainbowColorMap() : (o.colorMapName === 'default' ? defaultColorMap() : null);
  W.self.onmessage({ data: { type: 'init', canvas: W.canvas, sampleRate: sr, options: cm ? { colorMap: cm } : undefined } });
  const msg = { type: 'render', buffer: x, sampleRate: sr };
  if (o.N != null) msg.fftSize = o.N;
  if (o.overlap != null) msg.overlap = o.overlap;
  if (o.options) msg.options = o.options;
  W.self.onmessage({ data: msg });
  const img = W.take();
  const id = `wk${String(wkId++).padStart(2, '0')}`;
  manifest.cases.push({
    kind: 'worker_render', id, signal: sigName, N: o.N == null ? null : o.N, overlap: o.overlap == null ? null : o.overlap,
    colormap: o.colorMapName || null, options: o.options || null,
    width: img.width, height: img.height, posted: W.posted.slice(),
    lut: cm ? writeBin(`${id}_lut.u8`, new Uint8Array(W.self.__x.getColorMapUint())) : null,
    rgba: writeBin(`${id}.rgba`, new Uint8Array(img.data)),
  });
}
for (const s of ['noise', 'tone', 'bat', 'bat_ragged', 'silence', 'fullscale', 'impulse', 'quiet', 'nan_probe'])
  workerCase(s, { N: 1024, overlap: 75 });
workerCase('bat', { N: 1024, overlap: 75, colorMapName: 'default' });
workerCase('bat_ragged', { N: 1024, overlap: 75, colorMapName: 'rainbow' });
workerCase('bat', { /* defaults: 1024, overlap 0 */ });
workerCase('bat', { N: 1024, overlap: 50 });
workerCase('bat', { N: 1024, overlap: 93.75 });
workerCase('short_gt', { N: 1024, overlap: 99.95 });   // hop clamps to max(1, floor(0.512)) = 1
workerCase('bat', { N: 512, overlap: 75 });
workerCase('bat', { N: 2048, overlap: 75 });
workerCase('bat', { N: 4096, overlap: 93.75, colorMapName: 'default' });
workerCase('bat', { N: 256, overlap: 87.5 });
workerCase('bat500', { N: 1024, overlap: 75 });
workerCase('short_eq', { N: 1024, overlap: 75 });      // L == N: width 1, one frame
workerCase('short_gt', { N: 1024, overlap: 75 });
workerCase('bat', { N: 1024, overlap: 75, options: { windowFunc: 'hamming', gainDB: 0, rangeDB: 100 } });   // accepted, ignored
// message contract: render before init is dropped, setOptions installs a colour map
{
  const W = loadWorker();
  W.self.onmessage({ data: { type: 'render', buffer: signals.noise.x, fftSize: 1024, overlap: 75 } });
  const dropped = W.take() === null && W.posted.length === 0;
  W.self.onmessage({ data: { type: 'init', canvas: W.canvas } });
  W.self.onmessage({ data: { type: 'setOptions', options: { colorMap: rainbowColorMap() } } });
  W.self.onmessage({ data: { type: 'render', buffer: signals.noise.x, fftSize: 1024, overlap: 75 } });
  const img = W.take();
  manifest.cases.push({ kind: 'worker_contract', render_before_init_dropped: dropped, signal: 'noise', N: 1024, overlap: 75,
    width: img.width, height: img.height, posted: W.posted.slice(),
    rgba: writeBin('wk_contract.rgba', new Uint8Array(img.data)) });
}
// raw worker FFT (re/im after fftOptimized on windowed frames)
{
  const N = 1024, { x } = signals.tone;
  const win = Wk.self.__x.hannWindow(N), tw = Wk.self.__x.getTwiddleFactorsCached(N), br = Wk.self.__x.getBitReverseCached(N);
  const nF = 4, out = new Float32Array(nF * N * 2);
  for (let f = 0; f < nF; f++) {
    const re = new Float32Array(N), im = new Float32Array(N);
    for (let j = 0; j < N; j++) re[j] = x[f * 256 + j] * win[j];
    Wk.self.__x.fftOptimized(re, im, tw, br);
    out.set(re, f * 2 * N); out.set(im, f * 2 * N + N);
  }
  manifest.cases.push({ kind: 'worker_fft', signal: 'tone', N, hop: 256, frames: nF, layout: '[frame][re(N), im(N)]',
    data: writeBin('worker_fft_tone_1024.f32', out) });
}

// colour maps used above, as float RGBA (inputs for the other implementations)
manifest.colormaps = {
  default: writeBin('cmap_default.f64', colorMapToF64(defaultColorMap())),
  rainbow: writeBin('cmap_rainbow.f64', colorMapToF64(rainbowColorMap())),
};

// ================================================================ 9. autoIdHK
{
  const A = loadAutoId();
  manifest.species_rules = JSON.parse(JSON.stringify(A.speciesRules));
  const verdicts = [];
  const types = ['CF-FM', 'FM-CF-FM', 'FM', 'FM-QCF', 'QCF'];
  const FIELDS = ['highestFreq', 'lowestFreq', 'kneeFreq', 'heelFreq', 'startFreq', 'endFreq', 'cfStart', 'cfEnd', 'duration',
                  'bandwidth', 'kneeLowTime', 'kneeLowBandwidth', 'heelLowBandwidth', 'kneeHeelBandwidth'];
  const push = (d) => verdicts.push({ data: d, result: A.autoIdHK(d) });
  push({});
  push({ callType: 'FM-QCF', highestFreq: 55, lowestFreq: 48, bandwidth: 7, duration: 5, harmonic: 0 });
  // every rule: mid-point sample, then each bound +-eps, each field missing / NaN
  for (const sp of A.speciesRules) for (const rule of sp.rules) {
    const cts = rule.callType ? rule.callType.split(',').map((s) => s.trim()) : ['FM'];
    const base = { callType: cts[0], harmonic: rule.harmonic ? rule.harmonic[0] : 0 };
    const numeric = FIELDS.filter((f) => rule[f] && typeof rule[f][0] === 'number');
    const relational = FIELDS.filter((f) => rule[f] && typeof rule[f][0] === 'string');
    const nested = FIELDS.filter((f) => rule[f] && Array.isArray(rule[f][0]));
    for (const f of numeric) base[f] = (rule[f][0] + rule[f][1]) / 2;
    for (const f of nested) base[f] = (rule[f][0][0] + rule[f][0][1]) / 2;
    for (const f of relational) {   // satisfy "op refField" with equality where allowed
      const m = rule[f][0].match(/^(=|=>|>=|<|<=|>)\s*(\w+)$/);
      if (m) { if (base[m[2]] == null) base[m[2]] = 50; base[f] = m[1] === '>' ? base[m[2]] + 1 : m[1] === '<' ? base[m[2]] - 1 : base[m[2]]; }
    }
    push(Object.assign({}, base));
    for (const ct of types) push(Object.assign({}, base, { callType: ct }));
    for (const hm of [0, 1, 2, 3, 4]) push(Object.assign({}, base, { harmonic: hm }));
    for (const f of numeric) {
      const [lo, hi] = rule[f];
      for (const v of [lo, hi, lo - 1e-9, hi + 1e-9, lo - 1, hi + 1]) push(Object.assign({}, base, { [f]: v }));
      const miss = Object.assign({}, base); delete miss[f]; push(miss);
      push(Object.assign({}, base, { [f]: NaN }));
      push(Object.assign({}, base, { [f]: null }));
    }
    for (const f of relational) {
      const m = rule[f][0].match(/^(=|=>|>=|<|<=|>)\s*(\w+)$/);
      if (!m) continue;
      for (const dv of [-1, 0, 1]) push(Object.assign({}, base, { [f]: base[m[2]] + dv }));
      const miss = Object.assign({}, base); delete miss[m[2]]; push(miss);
    }
    for (const f of nested) for (const r of rule[f]) for (const v of [r[0], r[1], r[0] - 1e-9, r[1] + 1e-9]) push(Object.assign({}, base, { [f]: v }));
  }
  // JSON cannot carry NaN: encode as the string "NaN"
  const enc = (d) => { const o = {}; for (const k of Object.keys(d)) o[k] = (typeof d[k] === 'number' && isNaN(d[k])) ? 'NaN' : d[k]; return o; };
  manifest.autoid_verdicts = verdicts.map((v) => ({ data: enc(v.data), result: v.result }));
}

fs.writeFileSync(path.join(OUT, 'manifest.json'), JSON.stringify(manifest, null, 1));
let bytes = 0;
for (const f of fs.readdirSync(OUT)) bytes += fs.statSync(path.join(OUT, f)).size;
console.log(`wrote ${manifest.cases.length} cases, ${Object.keys(signals).length} signals, ${(bytes / 1024).toFixed(0)} KiB into ${OUT}`);
