'use strict';
// TEST INFRASTRUCTURE, not product code: a JavaScript restatement of the reference's worker render
// (spectrogramWorker.js:38-99 renderSpectrogram, :110-125 twiddles, :177-193 bit reversal, :195-203
// Hann window, :206-253 fftOptimized), written from its arithmetic contract so that the CPU baseline of
// bench.py can be timed in the reference's own language and runtime (Node worker_threads).  Only
// tests/ and bench.py's cpu_baseline leg may run it.  Pinned: tests/test_oracle_golden.py renders the
// golden worker cases through it and compares the RGBA bytes with what the reference produced.
//
//   node worker_restatement.js render <in.f32> <fftSize> <overlapPct> <lut.u8|-> <out.rgba>
//   node worker_restatement.js bench  <in.f32> <fftSize> <overlapPct> <workers> <chunkSeconds> <sampleRate> [passes=1]

const fs = require('fs');
const { Worker, isMainThread, parentPort, workerData } = require('worker_threads');

const tables = new Map();
function tablesFor(n) {
  let t = tables.get(n);
  if (t) return t;
  const win = new Float32Array(n);
  for (let i = 0; i < n; i++) win[i] = 0.5 * (1 - Math.cos(2 * Math.PI * i * (1 / (n - 1))));  // :195-203
  let bits = 0;
  while ((1 << bits) < n) bits++;
  const rev = new Uint32Array(n);
  for (let i = 0; i < n; i++) {
    let x = i, y = 0;
    for (let j = 0; j < bits; j++) { y = (y << 1) | (x & 1); x >>= 1; }
    rev[i] = y;
  }
  const cos = new Float64Array(n), sin = new Float64Array(n);  // stage with half size n1 at [n1-1, 2*n1-1)
  for (let n1 = 1; n1 < n; n1 <<= 1) {
    const n2 = n1 << 1;
    for (let j = 0; j < n1; j++) {
      const angle = -2 * Math.PI * j / n2;                      // :118-119
      cos[n1 - 1 + j] = Math.cos(angle);
      sin[n1 - 1 + j] = Math.sin(angle);
    }
  }
  t = { win, rev, cos, sin, re: new Float32Array(n), im: new Float32Array(n) };
  tables.set(n, t);
  return t;
}

function fft(re, im, n, t) {
  const rev = t.rev;
  for (let i = 0; i < n; i++) {                                  // :210-217
    const y = rev[i];
    if (y > i) {
      let v = re[i]; re[i] = re[y]; re[y] = v;
      v = im[i]; im[i] = im[y]; im[y] = v;
    }
  }
  for (let n1 = 1; n1 < n; n1 <<= 1) {                           // :233-252: doubles, float32 stores
    const n2 = n1 << 1;
    for (let j = 0; j < n1; j++) {
      const c = t.cos[n1 - 1 + j], s = t.sin[n1 - 1 + j];
      for (let i = j; i < n; i += n2) {
        const k = i + n1;
        const tr = c * re[k] - s * im[k];
        const ti = c * im[k] + s * re[k];
        re[k] = re[i] - tr;
        im[k] = im[i] - ti;
        re[i] = re[i] + tr;
        im[i] = im[i] + ti;
      }
    }
  }
}

// signal: Float32Array; lut: Uint8Array(1024) or null.  Returns {width, height, data}.
function render(signal, n, overlapPct, lut) {
  const hop = Math.max(1, Math.floor(n * (1 - overlapPct / 100)));            // :39
  const width = Math.max(1, Math.ceil((signal.length - n) / hop));            // :40
  const height = n / 2;
  const data = new Uint8ClampedArray(width * height * 4);
  const t = tablesFor(n), re = t.re, im = t.im, win = t.win;
  for (let x = 0, i = 0; i + n <= signal.length; i += hop, x++) {              // :59
    for (let j = 0; j < n; j++) { re[j] = signal[i + j] * win[j]; im[j] = 0; }
    fft(re, im, n, t);
    for (let y = 0; y < height; y++) {
      const mag = Math.sqrt(re[y] * re[y] + im[y] * im[y]);
      let val = mag > 1e-12 ? Math.log10(mag) / 5 : -2.4;                      // :71-78
      val = val < 0 ? 0 : (val > 1 ? 1 : val);
      const col = Math.floor(val * 255);
      const p = ((height - 1 - y) * width + x) * 4;                            // :79-80 (x == width wraps into the next row)
      if (lut) {
        data[p] = lut[col * 4]; data[p + 1] = lut[col * 4 + 1]; data[p + 2] = lut[col * 4 + 2]; data[p + 3] = lut[col * 4 + 3];
      } else {
        data[p] = col; data[p + 1] = col; data[p + 2] = col; data[p + 3] = 255;
      }
    }
  }
  return { width, height, data };
}

function readF32(path, first, count) {
  const fd = fs.openSync(path, 'r');
  const buf = Buffer.alloc(count * 4);
  fs.readSync(fd, buf, 0, count * 4, first * 4);
  fs.closeSync(fd);
  return new Float32Array(buf.buffer, buf.byteOffset, count);
}

if (!isMainThread && workerData && workerData.restatementBench) {
  // one bench worker: renders its list of chunks (hop-aligned, with an fftSize - hop halo), reports busy time
  const { path, n, overlap, chunks } = workerData;
  let samples = 0, busy = 0, checksum = 0;
  for (const [first, count, own] of chunks) {
    const sig = readF32(path, first, count);
    const t0 = process.hrtime.bigint();
    const img = render(sig, n, overlap, null);
    busy += Number(process.hrtime.bigint() - t0) / 1e9;
    samples += own;
    checksum = (checksum + img.data[img.data.length >> 1]) | 0;
  }
  parentPort.postMessage({ samples, busy, checksum });
} else if (require.main === module) {
  const [cmd, inPath, nArg, ovArg] = process.argv.slice(2);
  const n = parseInt(nArg, 10), overlap = parseFloat(ovArg);
  const total = fs.statSync(inPath).size / 4;
  if (cmd === 'render') {
    const lutPath = process.argv[6], outPath = process.argv[7];
    const lut = lutPath === '-' ? null : new Uint8Array(fs.readFileSync(lutPath));
    const img = render(readF32(inPath, 0, total), n, overlap, lut);
    fs.writeFileSync(outPath, Buffer.from(img.data.buffer, img.data.byteOffset, img.data.length));
    process.stdout.write(JSON.stringify({ width: img.width, height: img.height }) + '\n');
  } else if (cmd === 'bench') {
    const workers = parseInt(process.argv[6], 10), chunkSeconds = parseFloat(process.argv[7]), sr = parseFloat(process.argv[8]);
    const passes = process.argv[9] ? parseInt(process.argv[9], 10) : 1;  // render the file this many times over
    const hop = Math.max(1, Math.floor(n * (1 - overlap / 100)));
    const own = Math.max(hop, Math.floor(chunkSeconds * sr / hop) * hop);     // samples whose frames a chunk owns
    const lists = Array.from({ length: workers }, () => []);
    let c = 0;
    for (let pass = 0; pass < passes; pass++)
      for (let first = 0; first + n <= total; first += own, c++) {
        const count = Math.min(total - first, own + n - hop);
        lists[c % workers].push([first, count, Math.min(own, total - first)]);
      }
    const t0 = process.hrtime.bigint();
    let done = 0, samples = 0, busyMax = 0;
    for (let w = 0; w < workers; w++) {
      const wk = new Worker(__filename, { workerData: { restatementBench: true, path: inPath, n, overlap, chunks: lists[w] } });
      wk.on('message', (m) => {
        samples += m.samples;
        busyMax = Math.max(busyMax, m.busy);
        if (++done === workers) {
          const wall = Number(process.hrtime.bigint() - t0) / 1e9;
          process.stdout.write(JSON.stringify({ samples, workers, chunks: c, passes, wall_s: wall, busy_max_s: busyMax,
                                                msamples_per_s: samples / wall / 1e6, node: process.version }) + '\n');
        }
      });
      wk.on('error', (e) => { console.error(e); process.exit(1); });
    }
  } else {
    console.error('usage: worker_restatement.js render|bench ...');
    process.exit(2);
  }
}

module.exports = { render };
