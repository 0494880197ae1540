'use strict';
// Build-container tool (never run on the GPU box): times the reference's own worker render and the JavaScript
// restatement that bench.py uses as its CPU baseline, on the same seeded 10 s x 384 kHz signal, one thread each,
// and writes their ratio to profiles/cpu_baseline_ratio.json.  bench.py divides its restatement figure by that
// ratio to state the reference-equivalent rate.  The reference is read by path at run time
// (/root/reference/spectrogramWorker.js, evaluated under `vm` with a canvas stand-in); nothing of it is stored.
//
//   node tools/time_reference.js [/root/reference] [seconds=10] [repeats=3]
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const { render } = require('../oracle/worker_restatement.js');

const REF = process.argv[2] || '/root/reference';
const seconds = Number(process.argv[3] || 10), repeats = Number(process.argv[4] || 3);
const RATE = 384000, L = seconds * RATE;

const x = new Float32Array(L);
let seed = 12345;
const period = Math.round(0.080 * RATE), plen = Math.round(0.005 * RATE);
for (let i = 0; i < L; i++) {
  seed = (Math.imul(1103515245, seed) + 12345) & 0x7fffffff;
  let v = 0.01 * (seed / 0x40000000 - 1);
  const p = i % period;
  if (p < plen) {
    const t = p / RATE, dur = plen / RATE;
    v += 0.3 * 0.5 * (1 - Math.cos(2 * Math.PI * p / plen)) * Math.sin(2 * Math.PI * (80000 * t + 0.5 * (35000 - 80000) * t * t / dur));
  }
  x[i] = v;
}

const src = fs.readFileSync(path.join(REF, 'spectrogramWorker.js'), 'utf8');
const posted = [];
const self = { postMessage: (m) => posted.push(m) };
vm.runInContext(src, vm.createContext({ self, Math, Map, Float32Array, Uint32Array, Uint8ClampedArray, Uint8Array, Array, Object, console }),
                { filename: 'spectrogramWorker.js' });
let image = null;
const canvas = { width: 0, height: 0, getContext: () => ({
  createImageData: (w, h) => ({ width: w, height: h, data: new Uint8ClampedArray(w * h * 4) }),
  putImageData: (img) => { image = img; },
}) };
self.onmessage({ data: { type: 'init', canvas } });

const best = (fn) => {
  let b = Infinity;
  for (let r = 0; r < repeats; r++) {
    const t0 = process.hrtime.bigint();
    fn();
    b = Math.min(b, Number(process.hrtime.bigint() - t0) / 1e9);
  }
  return b;
};
const tRef = best(() => self.onmessage({ data: { type: 'render', buffer: x, sampleRate: RATE, fftSize: 1024, overlap: 75 } }));
let mine = null;
const tMine = best(() => { mine = render(x, 1024, 75, null); });
let same = image && mine && image.data.length === mine.data.length;
for (let i = 0; same && i < mine.data.length; i++) same = image.data[i] === mine.data[i];
const out = {
  tool: 'tools/time_reference.js', node: process.version, signal: seconds + ' s x 384 kHz seeded bat-call PCM, FFT 1024, 75 % overlap, one thread',
  reference_msamples_per_s: +(L / tRef / 1e6).toFixed(3), restatement_msamples_per_s: +(L / tMine / 1e6).toFixed(3),
  restatement_over_reference: +(tRef / tMine).toFixed(3), images_identical: !!same,
};
fs.writeFileSync(path.join(__dirname, '..', 'profiles', 'cpu_baseline_ratio.json'), JSON.stringify(out, null, 1) + '\n');
console.log(JSON.stringify(out));
