'use strict';
// Build-container tool (never run on the GPU box): times the reference's own worker render and the JavaScript
// restatement that bench.py uses as its CPU baseline, on the same seeded 10 s x 384 kHz signal, one thread each,
// and writes their ratios to profiles/cpu_baseline_ratio.json.  bench.py divides its restatement figure by the
// like-for-like ratio to state the reference-equivalent rate.  The reference is read by path at run time
// (/root/reference/spectrogramWorker.js with a canvas stand-in); nothing of it is stored.
//
// Like for like: a script evaluated in a `vm` sandbox reaches Math and the typed arrays through the sandbox's global
// proxy, which costs V8 a multiple on loops like these; so BOTH implementations are timed BOTH ways -- inside a
// vm.createContext sandbox, and in the main context with the real globals (the reference by vm.runInThisContext with
// a global `self`, the restatement by require) -- and the ratio that is used compares main context with main context.
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
const mineSrc = fs.readFileSync(path.join(__dirname, '..', 'oracle', 'worker_restatement.js'), 'utf8');
const sandbox = () => vm.createContext({ Math, Map, Float32Array, Float64Array, Uint32Array, Uint8ClampedArray, Uint8Array, Array, Object, console });
let image = null;
const canvas = { width: 0, height: 0, getContext: () => ({
  createImageData: (w, h) => ({ width: w, height: h, data: new Uint8ClampedArray(w * h * 4) }),
  putImageData: (img) => { image = img; },
}) };
// the reference in a sandbox
const ctxRef = sandbox();
ctxRef.self = { postMessage: () => {} };
vm.runInContext(src, ctxRef, { filename: 'spectrogramWorker.js' });
ctxRef.self.onmessage({ data: { type: 'init', canvas } });
// the reference in the main context (real globals)
global.self = { postMessage: () => {} };
vm.runInThisContext(src, { filename: 'spectrogramWorker.js' });
global.self.onmessage({ data: { type: 'init', canvas } });
// the restatement in a sandbox (its module wrapper's arguments supplied by hand; only `render` is used)
const ctxMine = sandbox();
ctxMine.require = require; ctxMine.module = { exports: {} }; ctxMine.__filename = 'worker_restatement.js'; ctxMine.process = process;
vm.runInContext(mineSrc.replace('require.main === module', 'false'), ctxMine, { filename: 'worker_restatement.js' });
const renderSandboxed = ctxMine.module.exports.render;

const best = (fn) => {
  let b = Infinity;
  for (let r = 0; r < repeats; r++) {
    const t0 = process.hrtime.bigint();
    fn();
    b = Math.min(b, Number(process.hrtime.bigint() - t0) / 1e9);
  }
  return b;
};
const msg = { data: { type: 'render', buffer: x, sampleRate: RATE, fftSize: 1024, overlap: 75 } };
const tRefVm = best(() => ctxRef.self.onmessage(msg));
const imageVm = image;
const tRefMain = best(() => global.self.onmessage(msg));
const imageMain = image;
let mine = null, mineVm = null;
const tMineMain = best(() => { mine = render(x, 1024, 75, null); });
const tMineVm = best(() => { mineVm = renderSandboxed(x, 1024, 75, null); });
const sameAs = (a, b) => {
  let same = a && b && a.data.length === b.data.length;
  for (let i = 0; same && i < a.data.length; i++) same = a.data[i] === b.data[i];
  return !!same;
};
const rate = (t) => +(L / t / 1e6).toFixed(3);
const out = {
  tool: 'tools/time_reference.js', node: process.version, signal: seconds + ' s x 384 kHz seeded bat-call PCM, FFT 1024, 75 % overlap, one thread',
  main_context: { reference_msamples_per_s: rate(tRefMain), restatement_msamples_per_s: rate(tMineMain), restatement_over_reference: +(tRefMain / tMineMain).toFixed(3) },
  vm_sandbox: { reference_msamples_per_s: rate(tRefVm), restatement_msamples_per_s: rate(tMineVm), restatement_over_reference: +(tRefVm / tMineVm).toFixed(3) },
  // the like-for-like figure bench.py uses: both in the main context with the real globals
  reference_msamples_per_s: rate(tRefMain), restatement_msamples_per_s: rate(tMineMain),
  restatement_over_reference: +(tRefMain / tMineMain).toFixed(3),
  images_identical: sameAs(imageVm, mine) && sameAs(imageMain, mine) && sameAs(mineVm, mine),
};
fs.writeFileSync(path.join(__dirname, '..', 'profiles', 'cpu_baseline_ratio.json'), JSON.stringify(out, null, 1) + '\n');
console.log(JSON.stringify(out));
