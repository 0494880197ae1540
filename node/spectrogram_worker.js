'use strict';
// The spectrogram Web Worker's message contract (spectrogramWorker.js:17-36,98) on the GPU.
//
// Runs either inside a worker_threads Worker (it then listens on parentPort) or in-process through
// createInProcessWorker().  Messages in: {type:'init', canvas, sampleRate?, options?},
// {type:'setOptions', options}, {type:'render', buffer:Float32Array, sampleRate?, fftSize?=1024,
// overlap?=0 (percent), options?}.  Message out: {type:'rendered'} -- extended, since Node has no
// canvas to paint, with {width, height, data:Uint8ClampedArray} (RGBA, ImageData layout), whose
// ArrayBuffer is transferred.  As in the reference, windowFunc/gainDB/rangeDB are accepted and
// ignored (spectrogramWorker.js:10-15,46,75); render before init is dropped.  Opt-in, not in the reference:
// with options.scale === 'plugin' those three options are honoured -- the frames get the plugin's window
// `windowFunc` (spectrogram.esm.js:105-151; an unknown name throws its error) and the quantiser becomes
// 20*log10(2/N*|X|) with gainDB/rangeDB (:531-535); geometry, FFT and colour map stay the worker's.
const S = require('./index.js');
const N = S.native;

function makeWorkerCore(post, coreOptions) {
  let canvas = null, sampleRate = 44100;                               // spectrogramWorker.js:1
  let currentOptions = { colorMap: null, windowFunc: 'hann', gainDB: 20, rangeDB: 80 };   // :10-15
  let lut = null;                                                      // colorMapUint, :8
  const engines = new Map();                                           // GPU contexts, least recently used first
  const maxEngines = (coreOptions && coreOptions.maxEngines) || 8;     // (a gain slider in 'plugin' scale makes one per step)
  const arith = (coreOptions && coreOptions.arith) || 'fast';
  const device = (coreOptions && coreOptions.device) || 0;

  function setColorMapUint(colorMap) {                                 // :128-143
    if (!colorMap || !colorMap.length) { lut = null; return; }
    const cm = new Float64Array(1024);
    for (let ii = 0; ii < 256; ii++) {
      const cc = colorMap[ii] || [0, 0, 0, 1];
      cm[ii * 4] = cc[0] || 0;
      cm[ii * 4 + 1] = cc[1] || 0;
      cm[ii * 4 + 2] = cc[2] || 0;
      cm[ii * 4 + 3] = cc[3] == null ? 1 : cc[3];
    }
    lut = N.workerLut(cm);
  }

  function engine(fftSize, hop, opts) {
    const plugin = opts && opts.scale === 'plugin';
    const gain = plugin && opts.gainDB != null ? +opts.gainDB : 20, range = plugin && opts.rangeDB != null ? +opts.rangeDB : 80;
    const win = plugin ? S.windowId(opts.windowFunc) : S.WIN_WORKER_HANN;
    const key = fftSize + ':' + hop + (plugin ? ':p' + win + ':' + opts.alpha + ':' + gain + ':' + range : '');
    let h = engines.get(key);
    if (h) {
      engines.delete(key);                  // most recently used last
      engines.set(key, h);
    } else {
      while (engines.size >= maxEngines) {  // every context holds device tables and scratch: keep the most recent few
        const [oldKey, old] = engines.entries().next().value;
        engines.delete(oldKey);
        N.destroy(old);
      }
      const exact = arith === 'exact';
      h = N.create({
        fftN: fftSize, hop, windowId: win, alpha: plugin ? opts.alpha : undefined,
        scaleMode: plugin ? S.SCALE_PLUGIN_DB : S.SCALE_WORKER_LOG5, gainDB: gain, rangeDB: range,
        frameRule: S.FRAMES_WORKER_COMPAT,
        twiddleMode: exact ? S.TW_HOST_TABLE : S.TW_ACCURATE, arith: exact ? S.ARITH_EXACT_F64 : S.ARITH_FAST_F32,
        outKind: S.OUT_RGBA8, inKind: S.IN_F32,
      }, device);
      if (exact) {                          // this V8's Math.cos/sin, like the reference's own tables
        const tw = S.workerTwiddles(fftSize);
        N.setTables(h, plugin ? null : S.workerHann(fftSize), tw.cos, tw.sin);
      }
      engines.set(key, h);
    }
    return h;
  }

  function renderSpectrogram(signal, sr, fftSize, overlapPct, opts) { // :38-99
    const hop = N.workerHop(fftSize, overlapPct);                      // :39
    const h = engine(fftSize, hop, opts);
    N.setColormap(h, lut);
    const data = N.stft(h, signal);
    const g = N.geometry(h, signal.length);
    if (canvas && typeof canvas === 'object') { canvas.width = g.width; canvas.height = g.bins; }   // :42-43
    post({ type: 'rendered', width: g.width, height: g.bins, data }, [data.buffer]);                 // :98
  }

  function onmessage(e) {                                              // :17-36
    const d = e.data;
    const type = d.type;
    if (type === 'init') {
      canvas = d.canvas === undefined ? {} : d.canvas;
      sampleRate = d.sampleRate || sampleRate;
      if (d.options) {
        currentOptions = Object.assign({}, currentOptions, d.options);
        if (currentOptions.colorMap) setColorMapUint(currentOptions.colorMap);
      }
    } else if (type === 'setOptions') {
      currentOptions = Object.assign({}, currentOptions, d.options || {});
      if (d.options && d.options.colorMap) setColorMapUint(d.options.colorMap);
    } else if (type === 'render') {
      if (!canvas) return;                                             // :31
      const opts = Object.assign({}, currentOptions, d.options || {});
      if (opts.colorMap) setColorMapUint(opts.colorMap);
      renderSpectrogram(d.buffer, d.sampleRate || sampleRate, d.fftSize || 1024, d.overlap || 0, opts);
    } else if (type === 'close') {
      for (const h of engines.values()) N.destroy(h);
      engines.clear();
    }
  }
  return { onmessage };
}

// In-process stand-in with the Worker surface the page uses: postMessage(msg) / onmessage(event).
function createInProcessWorker(coreOptions) {
  const w = { onmessage: null };
  const core = makeWorkerCore((msg) => { if (w.onmessage) w.onmessage({ data: msg }); }, coreOptions);
  w.postMessage = (msg) => core.onmessage({ data: msg });
  w.terminate = () => core.onmessage({ data: { type: 'close' } });
  return w;
}

module.exports = { makeWorkerCore, createInProcessWorker };

// When loaded as the entry script of a worker_threads Worker, serve parentPort.
try {
  const wt = require('worker_threads');
  if (!wt.isMainThread && wt.parentPort && wt.workerData && wt.workerData.sonospecWorker) {
    const core = makeWorkerCore((msg, transfer) => wt.parentPort.postMessage(msg, transfer), wt.workerData);
    wt.parentPort.on('message', (data) => core.onmessage({ data }));
  }
} catch (e) { /* worker_threads unavailable: in-process use only */ }
