'use strict';
// The headline measurement through the JavaScript host: one hour of synthetic 384 kHz bat-call PCM resident in
// HBM, worker render (spectrogramWorker.js semantics) to RGBA -- or colour indices with `--out u8` -- at FFT 1024,
// hop 256, timed with HIP events on the context's stream through the N-API addon.  Same workload and timing method
// as bench.py, which runs this file and reports its line under `variants.node_host_*`.
//
//   node node/bench.js [--seconds 3600] [--steps 20] [--warmup 3] [--out rgba|u8]
const S = require('./index.js');
const N = S.native;

const args = { seconds: 3600, steps: 20, warmup: 3, out: 'rgba' };
for (let i = 2; i + 1 < process.argv.length; i += 2) {
  const k = process.argv[i].replace(/^--/, '');
  args[k] = k === 'out' ? process.argv[i + 1] : Number(process.argv[i + 1]);
}
const RATE = 384000, FFT = 1024, HOP = 256;
const L = args.seconds * RATE;

// one 80 ms period of the bat-call signal's deterministic part (5 ms FM down-sweep 80 -> 35 kHz, Hann envelope,
// amplitude 0.3); white noise 0.01 from an LCG on top, block by block
const period = Math.round(0.080 * RATE), plen = Math.round(0.005 * RATE);
const pulse = new Float32Array(period);
for (let p = 0; p < plen; p++) {
  const t = p / RATE, dur = plen / RATE;
  pulse[p] = 0.3 * 0.5 * (1 - Math.cos(2 * Math.PI * p / plen)) * Math.sin(2 * Math.PI * (80000 * t + 0.5 * (35000 - 80000) * t * t / dur));
}
const blockLen = Math.min(L, 60 * RATE);                 // generated once on the host, uploaded to every 60 s of the recording
const block = new Float32Array(blockLen);
let seed = 12345;
for (let i = 0; i < blockLen; i++) {
  seed = (Math.imul(1103515245, seed) + 12345) & 0x7fffffff;
  block[i] = 0.01 * (seed / 0x40000000 - 1) + pulse[i % period] + (i % (10 * RATE) >= 5 * RATE ? 0.05 * Math.sin(2 * Math.PI * 110000 * i / RATE) : 0);
}

const h = N.create({
  fftN: FFT, hop: HOP, windowId: S.WIN_WORKER_HANN, scaleMode: S.SCALE_WORKER_LOG5, frameRule: S.FRAMES_WORKER_COMPAT,
  twiddleMode: S.TW_ACCURATE, arith: S.ARITH_FAST_F32, outKind: args.out === 'u8' ? S.OUT_U8_INDEX : S.OUT_RGBA8, inKind: S.IN_F32,
}, 0);
const outBytes = N.outBytes(h, L);
const dx = N.deviceMalloc(h, L * 4), dout = N.deviceMalloc(h, outBytes);
for (let at = 0; at < L; at += blockLen) {
  const n = Math.min(blockLen, L - at);
  N.copyToDevice(h, dx + BigInt(at) * 4n, n === blockLen ? block : block.subarray(0, n));
}
for (let i = 0; i < args.warmup; i++) N.stftDevice(h, dx, L, dout, outBytes, false);
N.wait(h);
const t0 = process.hrtime.bigint();
N.timerStart(h);
for (let i = 0; i < args.steps; i++) N.stftDevice(h, dx, L, dout, outBytes, false);
const evMs = N.timerStopMs(h) / args.steps;
const wallMs = Number(process.hrtime.bigint() - t0) / 1e6 / args.steps;
const head = new Uint8Array(4096);
N.copyToHost(h, head, dout);
const algo = L * 4 + outBytes;
console.log(JSON.stringify({
  host: 'node ' + process.version, out: args.out, seconds: args.seconds, samples: L, steps: args.steps, warmup: args.warmup,
  kernel_ms: +evMs.toFixed(4), ms_per_step: +wallMs.toFixed(4), value: +(L / wallMs / 1e3).toFixed(1), unit: 'Msamples/s',
  roofline_frac: +(algo / (evMs * 1e-3) / 1e9 / 8000).toFixed(4), route: N.kernelRoute(h), checksum: head.reduce((a, b) => a + b, 0),
}));
N.deviceFree(h, dx);
N.deviceFree(h, dout);
N.destroy(h);
