'use strict';
// JavaScript face of libsonospec.so: loads the N-API addon and names the enums of include/sonospec.h.
const path = require('path');

let native;
try {
  native = require(path.join(__dirname, 'build', 'sonospec.node'));
} catch (e) {
  // The addon is the product; there is no JavaScript fallback.
  throw new Error('sonospec: native addon not built or not loadable (run `make -C node`): ' + e.message);
}

const WINDOW_IDS = {
  bartlett: 0, bartlettHann: 1, blackman: 2, cosine: 3, gauss: 4,
  hamming: 5, hann: 6, lanczoz: 7, rectangular: 8, triangular: 9,
};

const enums = {
  WINDOW_IDS,
  WIN_WORKER_HANN: 10,
  SCALE_WORKER_LOG5: 0, SCALE_PLUGIN_DB: 1,
  FRAMES_PLUGIN_STRICT: 0, FRAMES_WORKER_COMPAT: 1,
  TW_ACCURATE: 0, TW_PLUGIN_RECURRENCE: 1, TW_HOST_TABLE: 2,
  ARITH_FAST_F32: 0, ARITH_EXACT_F64: 1,
  OUT_U8_INDEX: 0, OUT_RGBA8: 1, OUT_F32_DB: 2, OUT_F32_MAG: 3, OUT_FEATURES: 4,
  IN_F32: 0, IN_I16: 1,
};

// Window name -> id, with the reference's error text (modules/spectrogram.esm.js:149-150).
function windowId(name) {
  if (name === undefined || name === null) name = 'hann';
  if (!Object.prototype.hasOwnProperty.call(WINDOW_IDS, name)) throw Error("No such window function '" + name + "'");
  return WINDOW_IDS[name];
}

// Twiddle tables computed by THIS JavaScript engine, in the flat stage layout the library expects
// (the stage with half size T occupies [T-1, 2T-1)).  Passing them with setTables makes the exact
// arithmetic mode see the very doubles the reference's own code would compute under the same V8.
function workerTwiddles(n) {            // spectrogramWorker.js:110-125
  const c = new Float64Array(n - 1), s = new Float64Array(n - 1);
  for (let n1 = 1; n1 < n; n1 <<= 1) {
    const n2 = n1 << 1;
    for (let j = 0; j < n1; j++) {
      const angle = -2 * Math.PI * j / n2;
      c[n1 - 1 + j] = Math.cos(angle);
      s[n1 - 1 + j] = Math.sin(angle);
    }
  }
  return { cos: c, sin: s };
}

function pluginTwiddles(n) {            // modules/spectrogram.esm.js:158-160,176-190
  const c = new Float64Array(n - 1), s = new Float64Array(n - 1);
  for (let T = 1; T < n; T <<= 1) {
    const d = Math.fround(Math.cos(-Math.PI / T)), w = Math.fround(Math.sin(-Math.PI / T));
    let g = 1, b = 0;
    for (let z = 0; z < T; z++) {
      c[T - 1 + z] = g;
      s[T - 1 + z] = b;
      const v = g;
      g = v * d - b * w;
      b = v * w + b * d;
    }
  }
  return { cos: c, sin: s };
}

function workerHann(n) {                // spectrogramWorker.js:195-203
  const win = new Float32Array(n);
  const invN = 1 / (n - 1), twoPI = 2 * Math.PI;
  for (let i = 0; i < n; i++) win[i] = 0.5 * (1 - Math.cos(twoPI * i * invN));
  return win;
}

module.exports = Object.assign({ native, enums, windowId, workerTwiddles, pluginTwiddles, workerHann }, enums);
