#!/usr/bin/env node
'use strict';
// Parity of the Node host side against the reference's golden outputs (tests/golden, produced by
// tests/golden/gen_golden.js).  `node node/test_parity.js cpu` checks what needs no GPU (host rules,
// option validation); without arguments it also renders on the GPU:
//   exact mode  -> bit-identical RGBA / uint8 (tables computed by this V8, like the reference's)
//   fast mode   -> colour indices within 1 LSB (cyclic at the plugin's 256 -> 0 wrap)
const fs = require('fs');
const path = require('path');
const S = require('./index.js');
const { Spectrogram, createSpectrogramPlugin, noverlapFromPercent } = require('./spectrogram_plugin.js');
const { createInProcessWorker } = require('./spectrogram_worker.js');

const G = path.join(__dirname, '..', 'tests', 'golden');
const manifest = JSON.parse(fs.readFileSync(path.join(G, 'manifest.json'), 'utf8'));
const cpuOnly = process.argv[2] === 'cpu';
let checks = 0, failures = 0;
function ok(cond, what) { checks++; if (!cond) { failures++; console.error('FAIL: ' + what); } }
function load(name, Type) {
  const b = fs.readFileSync(path.join(G, name));
  return new Type(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength));
}
function signal(name) { return load(manifest.signals[name].file, Float32Array); }
function colormap(name) {
  const f = load(manifest.colormaps[name], Float64Array), cm = [];
  for (let i = 0; i < 256; i++) cm.push([f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]]);
  return cm;
}
function equal(a, b) { if (a.length !== b.length) return false; for (let i = 0; i < a.length; i++) if (a[i] !== b[i]) return false; return true; }
function maxCyclicDiff(a, b) { let m = 0; for (let i = 0; i < a.length; i++) { let d = (a[i] - b[i] + 256) % 256; d = Math.min(d, 256 - d); if (d > m) m = d; } return m; }
function audioBuffer(x, sr) { return { sampleRate: sr, length: x.length, numberOfChannels: 1, getChannelData: () => x }; }
const cases = (kind) => manifest.cases.filter((c) => c.kind === kind);

// ---------------------------------------------------------------- host rules (no GPU)
for (const r of cases('plugin_crop_grid')[0].rows) {
  const c = S.native.pluginCrop(r.N / 2, r.sr, r.fmin, r.fmax);
  ok(c.sy === r.sy && c.sh === r.sh, 'crop ' + JSON.stringify(r));
}
for (const r of cases('plugin_scale_crop_grid')[0].rows) {
  const c = S.native.pluginCropScaled(r.N / 2, r.sr, r.fmin, r.fmax, { mel: 1, logarithmic: 2, bark: 3, erb: 4 }[r.scale]);
  ok(c.sy === r.sy && c.sh === r.sh, 'scaled crop ' + JSON.stringify(r));
}
for (const c of cases('plugin_render')) {
  if (c.colormap !== 'default' && c.colormap !== 'rainbow') continue;
  const p = Spectrogram.create({ colorMap: colormap(c.colormap) });
  ok(equal(p._colorMapUint, load(c.lut, Uint8Array)), 'plugin LUT ' + c.id);
}
for (const c of cases('plugin_get_frequencies')) {
  const L = manifest.signals[c.signal].length;
  const ov = S.native.pluginNoverlap(c.N, c.noverlap || 0, L, c.canvas_width);
  const g = S.native.geometryFor(c.N, c.N - ov, S.FRAMES_PLUGIN_STRICT, L);
  ok(g.frames === c.frames && g.bins === c.bins, 'plugin geometry ' + c.id);
}
for (const c of cases('worker_render')) {
  const L = manifest.signals[c.signal].length, n = c.N || 1024;
  const g = S.native.geometryFor(n, S.native.workerHop(n, c.overlap || 0), S.FRAMES_WORKER_COMPAT, L);
  ok(g.width === c.width && g.bins === c.height, 'worker geometry ' + c.id);
}
{
  const msgs = {};
  for (const e of cases('plugin_errors')[0].errors) msgs[e.label] = e.message;
  const expectThrow = (fn, msg, what) => { let got = null; try { fn(); } catch (e) { got = String(e.message || e); } ok(got === msg, what + ': ' + got); };
  expectThrow(() => Spectrogram.create({ colorMap: Array.from({ length: 10 }, () => [0, 0, 0, 1]) }), msgs.short, 'short colormap');
  expectThrow(() => Spectrogram.create({ colorMap: Array.from({ length: 256 }, () => [0, 0, 0]) }), msgs.bad_entry, 'bad entry');
  expectThrow(() => Spectrogram.create({}), msgs.default_name, 'default name');
  expectThrow(() => Spectrogram.create({ colorMap: 'viridis' }), msgs.unknown_name, 'unknown name');
  expectThrow(() => S.windowId('kaiser'), cases('plugin_window_error')[0].message, 'unknown window');
  const p = Spectrogram.create({ colorMap: 'gray' });
  ok(p.fftSamples === 512 && p.height === 200 && p.noverlap === null && p.gainDB === 20 && p.rangeDB === 80 && p.scale === 'mel', 'defaults');
  const q = createSpectrogramPlugin({ colorMap: colormap('default'), noverlap: 768 });
  ok(q.frequencyMin === 10000 && q.frequencyMax === 128000 && q.scale === 'linear' && q.fftSamples === 1024 && q.height === 800, 'wsManager factory');
  ok(noverlapFromPercent(1024, 75) === 768 && noverlapFromPercent(1024, null) === null, 'noverlap from percent');
}
// tables this V8 computes are the reference's (same engine family): bit-identical doubles
for (const c of cases('worker_tables')) {
  const tw = S.workerTwiddles(c.N);
  ok(equal(tw.cos, load(c.twiddle_cos, Float64Array)) && equal(tw.sin, load(c.twiddle_sin, Float64Array)), 'worker twiddles ' + c.N);
  ok(equal(S.workerHann(c.N), load(c.hann, Float32Array)), 'worker hann ' + c.N);
}

// file intake against the reference's own answers (tests/golden/wav)
{
  const wav = require('./wav.js');
  const wm = JSON.parse(fs.readFileSync(path.join(G, 'wav', 'manifest.json'), 'utf8'));
  const same = (a, b) => JSON.stringify(a) === JSON.stringify(b);
  let bad = 0;
  for (const c of wm.files) {
    const data = fs.readFileSync(path.join(G, c.file));
    if (wav.getWavSampleRate(data) !== c.sample_rate) bad++;
    if (wav.getWavDuration(data) !== c.duration) bad++;
    if (wav.extractGuanoMetadata(data) !== c.guano_text) bad++;
    if (!same(wav.parseGuanoMetadata(c.guano_text), c.guano_fields)) bad++;
    for (const k of c.crops) {
      const got = wav.cropWavBlob(data, k.start, k.end);
      if (k.result === null) { if (got !== null) bad++; }
      else if (!got || !equal(got, new Uint8Array(fs.readFileSync(path.join(G, k.result))))) bad++;
    }
  }
  for (const t of wm.guano_texts) if (!same(wav.parseGuanoMetadata(t.text), t.fields)) bad++;
  ok(bad === 0 && wm.files.length >= 9, 'wav / guano / crop against the reference, mismatches: ' + bad);
  ok(wav.getWavSampleRate(null) === 256000 && wav.getWavDuration(null) === 0 && wav.extractGuanoMetadata(null) === '(no file selected)' &&
     wav.acceptFile(1000, 1) === 'small' && wav.acceptFile(1e7, 21) === 'long' && wav.acceptFile(1e7, 21, true) === 'ok', 'wav defaults and filters');
}

// matcher semantics over the reference's own rule table and verdicts
{
  const { autoIdHK, segmentCalls, callToAutoIdData } = require('./autoid.js');
  let bad = 0;
  for (const v of manifest.autoid_verdicts) {
    const d = {};
    for (const k of Object.keys(v.data)) d[k] = v.data[k] === 'NaN' ? NaN : v.data[k];
    if (autoIdHK(d, manifest.species_rules) !== v.result) bad++;
  }
  ok(bad === 0 && manifest.autoid_verdicts.length > 700, 'autoIdHK verdicts, mismatches: ' + bad);
  const idx = new Uint8Array(40), bin = new Int32Array(40).fill(100);
  idx.set([90, 120, 200, 200, 150, 110, 95], 5); bin.set([213, 200, 180, 170, 160, 150, 147], 5);
  const calls = segmentCalls({ peakBin: bin, peakIdx: idx }, 80, 2, 1000);
  ok(calls.length === 1 && calls[0].startFrame === 1005 && calls[0].endFrame === 1011 && calls[0].peakFrame === 1007 &&
     calls[0].peakBin === 180 && calls[0].highBin === 213 && calls[0].lowBin === 147, 'segmentCalls');
  const d = callToAutoIdData(calls[0], 1024, 256, 384000, { callType: 'FM' });
  ok(Math.abs(d.duration - 7 * 256 / 384) < 1e-9 && Math.abs(d.bandwidth - 66 * 0.375) < 1e-9, 'callToAutoIdData');
  // where a call turns (the library's own definitions; same hand-made tracks as tests/test_autoid.py)
  const { callShape } = require('./autoid.js');
  const track = (bins) => {
    const n = bins.length + 6, pb = new Int32Array(n).fill(50), pi = new Uint8Array(n);
    pb.set(bins, 3); pi.fill(200, 3, 3 + bins.length);
    return { peakBin: pb, peakIdx: pi };
  };
  const f1 = track([200, 190, 180, 170, 160, 150, 148, 147, 146, 146, 145, 145, 144, 144, 130, 110]);
  f1.peakIdx[10] = 230;
  const c1 = segmentCalls(f1, 100, 1, 500)[0], s1 = callShape(f1, c1, 500);
  ok(s1.kneeFrame === 508 && s1.kneeBin === 150 && s1.heelFrame === 516 && s1.heelBin === 144 && s1.lowFrame === 518 &&
     s1.cfStartFrame === 509 && s1.cfEndFrame === 512, 'callShape: FM-QCF track');
  const d1 = callToAutoIdData(c1, 1024, 256, 384000, { callType: 'FM-QCF', shape: s1 });
  ok(Math.abs(d1.kneeFreq - 150 * 0.375) < 1e-12 && Math.abs(d1.kneeLowTime - (508 - 518) * 256 / 384) < 1e-9 &&
     Math.abs(d1.kneeHeelBandwidth - 6 * 0.375) < 1e-12, 'shape-derived fields');
  const f2 = track([290, 291, 290, 290, 289, 290, 290, 291, 290, 290, 290, 290, 289, 290, 290, 290, 291, 290, 290, 290, 280, 268, 255, 240, 226]);
  const c2 = segmentCalls(f2, 100, 1)[0], s2 = callShape(f2, c2);
  const d2 = callToAutoIdData(c2, 1024, 256, 384000, { callType: 'CF-FM', shape: s2 });
  ok(s2.cfStartFrame === 3 && s2.cfEndFrame === 22 && Math.abs(d2.bandwidth - 64 * 0.375) < 1e-12 &&
     callToAutoIdData(c2, 1024, 256, 384000, { callType: 'CF-FM' }).bandwidth === undefined, 'CF part and its bandwidth rule');
  // matcher forms the reference's table does not use
  const rules = [
    { name: 'ranges', rules: [{ duration: [[1, 2], [5, 6]] }] },
    { name: 'relation', rules: [{ startFreq: ['> endFreq'], kneeFreq: ['=> heelFreq'] }] },
    { name: 'never', rules: [{ duration: ['about 5'] }] },
    { name: 'typed', rules: [{ callType: 'FM, QCF', harmonic: [0, 2], cfStart: [60, 70] }] },
    { name: 'two', rules: [{ duration: [100, 200] }, { bandwidth: [3, null] }, { lowestFreq: [null, 20] }] },
  ];
  const m = (x) => autoIdHK(x, rules);
  ok(m({ duration: 1.5 }) === 'ranges' && m({ duration: 3 }) === 'No species matched' &&
     m({ startFreq: 50, endFreq: 40, kneeFreq: 45, heelFreq: 45 }) === 'relation' &&
     m({ startFreq: 50, endFreq: 40, kneeFreq: 45 }) === 'No species matched' &&
     m({ callType: 'QCF', harmonic: 2, cfStart: 65 }) === 'typed' && m({ callType: 'QCF', harmonic: 1, cfStart: 65 }) === 'No species matched' &&
     m({ callType: 'FM', cfStart: 65 }) === 'No species matched' && m({ duration: 150 }) === 'two' &&
     m({ bandwidth: 9 }) === 'No species matched' && m({ lowestFreq: 10 }) === 'two' && m({}) === 'No species matched',
     'matcher: lists of ranges, relations, call types, harmonics, null bounds');
  // an empty array is truthy in JavaScript: `harmonic: []` and `duration: []` never match (autoid_HK.js:229-231,211-212)
  const empty = [
    { name: 'no-harmonic', rules: [{ harmonic: [], duration: [1, 9] }] },
    { name: 'no-range', rules: [{ duration: [] }] },
    { name: 'falsy', rules: [{ duration: 0, bandwidth: null, callType: '', lowestFreq: [10, 20] }] },
  ];
  const e = (x) => autoIdHK(x, empty);
  ok(e({ duration: 5, harmonic: 0 }) === 'No species matched' && e({ duration: 5 }) === 'No species matched' &&
     e({ lowestFreq: 15 }) === 'falsy' && e({ lowestFreq: 15, duration: 1e9, bandwidth: -3 }) === 'falsy',
     'matcher: empty arrays are truthy and never match; 0 / null / "" leave a field unconstrained');
}

// ---------------------------------------------------------------- GPU
if (!cpuOnly) {
  if (S.native.deviceCount() < 1) { console.error('no GPU'); process.exit(2); }
  for (const arith of ['exact', 'fast']) {
    for (const c of cases('worker_render')) {
      const w = createInProcessWorker({ arith });
      let msg = null;
      w.onmessage = (e) => { msg = e.data; };
      const canvas = {};
      w.postMessage({ type: 'init', canvas, options: c.colormap ? { colorMap: colormap(c.colormap) } : undefined });
      const m = { type: 'render', buffer: signal(c.signal) };
      if (c.N != null) m.fftSize = c.N;
      if (c.overlap != null) m.overlap = c.overlap;
      if (c.options) m.options = c.options;
      w.postMessage(m);
      const ref = load(c.rgba, Uint8Array);
      ok(msg && msg.type === 'rendered' && msg.width === c.width && msg.height === c.height && canvas.width === c.width, 'worker msg ' + c.id);
      if (arith === 'exact') ok(equal(msg.data, ref), 'worker exact ' + c.id);
      else if (!c.colormap) ok(maxCyclicDiff(msg.data, ref) <= 1, 'worker fast ' + c.id);   // grey: channel value == index
      w.terminate();
    }
    for (const c of cases('plugin_get_frequencies')) {
      if (!c.frames) continue;
      const opts = { fftSamples: c.N, windowFunc: c.window, colorMap: 'gray', scale: 'linear', arith, canvasWidth: c.canvas_width,
                     gainDB: c.gainDB, rangeDB: c.rangeDB };
      if (c.noverlap != null) opts.noverlap = c.noverlap;
      if (c.alpha != null) opts.alpha = c.alpha;
      const p = Spectrogram.create(opts);
      const out = p.getFrequencies(audioBuffer(signal(c.signal), manifest.signals[c.signal].sample_rate))[0];
      const ref = load(c.data, Uint8Array);
      ok(out.length === c.frames && out[0].length === c.bins, 'plugin shape ' + c.id);
      if (arith === 'exact') ok(equal(out.flat, ref), 'plugin exact ' + c.id);
      else ok(maxCyclicDiff(out.flat, ref) <= 1, 'plugin fast ' + c.id);
      p.destroy();
    }
  }
  for (const c of cases('plugin_render')) {
    const cm = (c.colormap === 'default' || c.colormap === 'rainbow') ? colormap(c.colormap) : c.colormap;
    const p = Spectrogram.create({ fftSamples: c.N, noverlap: c.noverlap, colorMap: cm, scale: 'linear', frequencyMin: c.fmin,
                                   frequencyMax: c.fmax, windowFunc: c.window, arith: 'exact', canvasWidth: c.out_width });
    const img = p.render(audioBuffer(signal(c.signal), manifest.signals[c.signal].sample_rate), c.out_width);
    ok(img.width === c.image_width && img.height === c.image_height && equal(img.data, load(c.rgba, Uint8Array)), 'plugin render ' + c.id);
    ok(img.crop.sy === c.crop.sy && img.crop.sh === c.crop.sh, 'plugin crop ' + c.id);
    p.destroy();
  }
  // the canvas step: the reference's image (fixture) stretched by the engine's filter, restated here in JS
  const stretch = (img, W, rows, sy, sh, dy, dh, height) => {
    const out = new Uint8Array(height * W * 4), f = Math.fround;
    for (let y = 0; y < height; y++) {
      const yc = y + 0.5;
      if (!(sh > 0 && dh > 0 && yc >= dy && yc < dy + dh)) continue;
      let v = (yc - dy) * (sh / dh) - 0.5;
      v = Math.min(Math.max(v, 0), sh - 1);
      const i0 = Math.floor(v), i1 = Math.min(i0 + 1, sh - 1), t = f(v - i0), ra = sy + i0, rb = sy + i1;
      for (let x = 0; x < W; x++)
        for (let ch = 0; ch < 4; ch++) {
          const a = ra >= 0 && ra < rows ? img[(ra * W + x) * 4 + ch] : 0, b = rb >= 0 && rb < rows ? img[(rb * W + x) * 4 + ch] : 0;
          out[(y * W + x) * 4 + ch] = Math.floor(f(f(a + f(f(b - a) * t)) + 0.5));
        }
    }
    return out;
  };
  for (const c of cases('plugin_render')) {
    if (c.fmax > manifest.signals[c.signal].sample_rate / 2) continue;     // no background fill in these cases
    const cm = (c.colormap === 'default' || c.colormap === 'rainbow') ? colormap(c.colormap) : c.colormap;
    for (const height of [200, 800]) {
      const p = Spectrogram.create({ fftSamples: c.N, noverlap: c.noverlap, colorMap: cm, scale: 'linear', frequencyMin: c.fmin,
                                     frequencyMax: c.fmax, windowFunc: c.window, arith: 'exact', canvasWidth: c.out_width, height });
      const sr = manifest.signals[c.signal].sample_rate;
      const cv = p.renderCanvas(audioBuffer(signal(c.signal), sr), c.out_width);
      const rect = S.native.pluginDrawRect(sr, c.fmin, c.fmax, 0, height);
      const want = stretch(load(c.rgba, Uint8Array), c.image_width, c.image_height, c.crop.sy, c.crop.sh, rect.dy, rect.dh, height);
      ok(cv.width === c.out_width && cv.height === height && equal(cv.data, want), 'plugin canvas ' + c.id + ' h' + height);
      p.destroy();
    }
  }
  // non-linear frequency scales: exact arithmetic with this V8's filter weights reproduces the reference's bytes
  for (const c of cases('plugin_scale_frequencies')) {
    const sr = manifest.signals[c.signal].sample_rate;
    const ref = load(c.data, Uint8Array);
    for (const arith of ['exact', 'fast']) {
      const p = Spectrogram.create({ fftSamples: c.N, noverlap: c.noverlap, colorMap: 'gray', scale: c.scale, arith });
      const out = p.getFrequencies(audioBuffer(signal(c.signal), sr))[0];
      if (arith === 'exact') ok(equal(out.flat, ref), 'scale ' + c.scale + ' exact ' + c.id);
      else {
        let worst = 0;
        for (let i = 0; i < ref.length; i++) { let d = Math.abs(out.flat[i] - ref[i]); d = Math.min(d, 256 - d); if (d > worst) worst = d; }
        ok(worst <= 1, 'scale ' + c.scale + ' fast ' + c.id + ' worst ' + worst);
      }
      p.destroy();
    }
  }
  // render before init is dropped; setOptions installs the colour map
  {
    const c = cases('worker_contract')[0];
    const w = createInProcessWorker({ arith: 'exact' });
    const got = [];
    w.onmessage = (e) => got.push(e.data);
    w.postMessage({ type: 'render', buffer: signal(c.signal), fftSize: 1024, overlap: 75 });
    ok(got.length === 0, 'render before init dropped');
    w.postMessage({ type: 'init', canvas: {} });
    w.postMessage({ type: 'setOptions', options: { colorMap: colormap('rainbow') } });
    w.postMessage({ type: 'render', buffer: signal(c.signal), fftSize: 1024, overlap: 75 });
    ok(got.length === 1 && equal(got[0].data, load(c.rgba, Uint8Array)), 'worker contract');
    w.terminate();
  }
}

console.log(JSON.stringify({ checks, failures, gpu: !cpuOnly }));
process.exit(failures ? 1 : 0);
