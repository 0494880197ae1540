#!/usr/bin/env node
'use strict';
// The worker contract over a real worker_threads Worker: the caller transfers the PCM buffer in
// (main.js:1417) and gets the painted pixels transferred back with {type:'rendered'}.
const path = require('path');
const fs = require('fs');
const { Worker } = require('worker_threads');

const G = path.join(__dirname, '..', 'tests', 'golden');
const manifest = JSON.parse(fs.readFileSync(path.join(G, 'manifest.json'), 'utf8'));
const c = manifest.cases.filter((k) => k.kind === 'worker_render' && k.signal === 'bat' && k.N === 1024 && k.overlap === 75 && !k.colormap && !k.options)[0];
const b = fs.readFileSync(path.join(G, manifest.signals[c.signal].file));
const x = new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength));
const ref = fs.readFileSync(path.join(G, c.rgba));

const w = new Worker(path.join(__dirname, 'spectrogram_worker.js'), { workerData: { sonospecWorker: true, arith: 'exact' } });
const timer = setTimeout(() => { console.error('timeout'); process.exit(3); }, 60000);
w.on('error', (e) => { console.error('worker error', e); process.exit(2); });
w.on('message', (m) => {
  clearTimeout(timer);
  let same = m.type === 'rendered' && m.width === c.width && m.height === c.height && m.data.length === ref.length;
  for (let i = 0; same && i < ref.length; i++) same = m.data[i] === ref[i];
  console.log(JSON.stringify({ rendered: m.type === 'rendered', width: m.width, height: m.height, identical: same, input_detached: x.length === 0 }));
  w.postMessage({ type: 'close' });
  w.terminate().then(() => process.exit(same ? 0 : 1));
});
w.postMessage({ type: 'init', canvas: {} });
w.postMessage({ type: 'render', buffer: x, sampleRate: 384000, fftSize: 1024, overlap: 75 }, [x.buffer]);
