'use strict';
// GPU test of the tile cache and the local bridge: tiles fetched over HTTP from an uploaded .wav file equal the same
// columns of plugin.renderCanvas on the decoded samples; a second view of the recording hits the cached plane.
const http = require('http');
const path = require('path');
const fs = require('fs');
const wav = require('./wav.js');
const { Spectrogram } = require('./spectrogram_plugin.js');
const { TileCache } = require('./tiles.js');
const { createBridge } = require('./bridge.js');

let failures = 0, checks = 0;
const ok = (c, m) => { checks++; if (!c) { failures++; console.error('FAIL ' + m); } };
const equal = (a, b) => { if (a.length !== b.length) return false; for (let i = 0; i < a.length; i++) if (a[i] !== b[i]) return false; return true; };
const columns = (img, W, H, x0, n) => {      // columns [x0, x0+n) of an H x W RGBA image
  const out = new Uint8Array(H * n * 4);
  for (let y = 0; y < H; y++) out.set(img.subarray((y * W + x0) * 4, (y * W + x0 + n) * 4), y * n * 4);
  return out;
};

// a synthetic PCM16 recording: noise plus a chirp
let seed = 11;
const rnd = () => (seed = (Math.imul(1103515245, seed) + 12345) & 0x7fffffff) / 0x40000000 - 1;
const n = 200000, rate = 256000, pcm = Buffer.alloc(n * 2);
for (let i = 0; i < n; i++) pcm.writeInt16LE(Math.round(2000 * rnd() + 12000 * Math.sin(i * i * 2e-6)), i * 2);
const fmt = Buffer.alloc(24);
fmt.write('fmt ', 0, 4, 'latin1'); fmt.writeUInt32LE(16, 4); fmt.writeUInt16LE(1, 8); fmt.writeUInt16LE(1, 10); fmt.writeUInt32LE(rate, 12);
fmt.writeUInt32LE(rate * 2, 16); fmt.writeUInt16LE(2, 20); fmt.writeUInt16LE(16, 22);
const dh = Buffer.alloc(8); dh.write('data', 0, 4, 'latin1'); dh.writeUInt32LE(pcm.length, 4);
const head = Buffer.alloc(12); head.write('RIFF', 0, 4, 'latin1'); head.writeUInt32LE(4 + 24 + 8 + pcm.length, 4); head.write('WAVE', 8, 4, 'latin1');
const file = Buffer.concat([head, fmt, dh, pcm]);

const samples = wav.channelData(file, 0);
ok(samples.length === n && samples[5] === pcm.readInt16LE(10) / 32768, 'host decode');
for (const name of ['float32_mono.wav', 'int32_mono.wav']) {   // every sample format decodes to the documented convention
  const f = fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', 'wav', name));
  const x = wav.channelData(f, 0), w = wav.pcmSamples(f).info;
  let fine = x.length === w.frames;
  for (let i = 0; i < x.length; i++) fine = fine && Number.isFinite(x[i]) && Math.abs(x[i]) <= 1;
  ok(fine, 'decode ' + name);
}
const buffer = { sampleRate: rate, length: n, numberOfChannels: 1, getChannelData: () => samples };

// TileCache directly
{
  const opts = { fftSamples: 512, noverlap: 384, colorMap: 'gray', scale: 'linear', frequencyMin: 10000, frequencyMax: 100000, height: 200 };
  const p = Spectrogram.create(opts), cache = new TileCache();
  const W = 500, whole = p.renderCanvas(buffer, W);
  for (const [x0, nc] of [[0, 256], [256, 244], [100, 1]])
    ok(equal(cache.tile('r', p, buffer, W, x0, nc).data, columns(whole.data, W, 200, x0, nc)), 'tile ' + x0 + '+' + nc);
  ok(cache.misses === 1 && cache.hits === 2, 'cache counts ' + cache.misses + '/' + cache.hits);
  const q = Spectrogram.create(Object.assign({}, opts, { colorMap: 'igray', height: 77, frequencyMax: 150000 }));   // above Nyquist: background band
  ok(equal(cache.tile('r', q, buffer, 300).data, q.renderCanvas(buffer, 300).data) && cache.misses === 1, 'other view, same plane');
  let threw = false; try { cache.tile('r', p, buffer, 100, 90, 20); } catch (e) { threw = e instanceof RangeError; }
  ok(threw, 'tile outside the canvas');
  cache.drop();
  ok(cache.used === 0, 'drop releases');
  p.destroy(); q.destroy();
  // a plugin keeps at most maxEngines contexts (automatic overlap makes one per distinct hop); a retired context takes
  // its planes out of the cache first, and a tile of a retired configuration is simply computed again
  const z = Spectrogram.create(Object.assign({}, opts, { noverlap: null, maxEngines: 3 }));
  const first = cache.tile('z', z, buffer, 400).data.slice();
  for (const w of [500, 650, 800, 1000, 1300]) cache.tile('z', z, buffer, w);
  ok(z._engines.size <= 3, 'plugin contexts bounded: ' + z._engines.size);
  ok(equal(cache.tile('z', z, buffer, 400).data, first), 'tile of a retired context, recomputed');
  cache.drop();
  z.destroy();
}

// over HTTP
const server = createBridge();
const request = (method, url, body) => new Promise((resolve, reject) => {
  const req = http.request({ host: '127.0.0.1', port: server.address().port, path: url, method }, (res) => {
    const chunks = [];
    res.on('data', (c) => chunks.push(c));
    res.on('end', () => resolve({ status: res.statusCode, headers: res.headers, body: Buffer.concat(chunks) }));
  });
  req.on('error', reject);
  if (body) req.write(body);
  req.end();
});
server.listen(0, '127.0.0.1', async () => {
  try {
    const up = await request('PUT', '/recording/bat1', file);
    const meta = JSON.parse(up.body.toString());
    ok(up.status === 200 && meta.sampleRate === rate && meta.length === n, 'upload ' + up.body.toString());
    const query = 'fftSamples=1024&noverlap=768&height=300&frequencyMin=10000&frequencyMax=120000&colorMap=gray';
    const p = Spectrogram.create({ fftSamples: 1024, noverlap: 768, height: 300, frequencyMin: 10000, frequencyMax: 120000, colorMap: 'gray', scale: 'linear',
                                   windowFunc: 'hann', gainDB: 20, rangeDB: 80 });
    const W = 640, whole = p.renderCanvas(buffer, W);
    for (const x0 of [0, 256, 512]) {
      const nc = Math.min(256, W - x0);
      const t = await request('GET', '/tile/bat1?' + query + '&width=' + W + '&x0=' + x0 + '&ncols=' + nc);
      ok(t.status === 200 && Number(t.headers['x-width']) === nc && Number(t.headers['x-height']) === 300 &&
         equal(new Uint8Array(t.body), columns(whole.data, W, 300, x0, nc)), 'http tile ' + x0);
    }
    const st = JSON.parse((await request('GET', '/stats')).body.toString());
    ok(st.misses === 1 && st.hits === 2 && st.recordings === 1, 'stats ' + JSON.stringify(st));
    // a custom colour map
    const table = []; for (let i = 0; i < 256; i++) table.push([i / 255, 0.5, 1 - i / 255, 1]);
    ok((await request('PUT', '/colormap/custom', Buffer.from(JSON.stringify(table)))).status === 200, 'colormap upload');
    const pc = Spectrogram.create({ fftSamples: 1024, noverlap: 768, height: 300, frequencyMin: 10000, frequencyMax: 120000, colorMap: table, scale: 'linear' });
    const tc = await request('GET', '/tile/bat1?' + query.replace('colorMap=gray', 'colorMap=custom') + '&width=' + W);
    ok(equal(new Uint8Array(tc.body), pc.renderCanvas(buffer, W).data), 'custom colour map tile');
    ok(JSON.parse((await request('GET', '/stats')).body.toString()).misses === 1, 'colour map change keeps the plane');
    ok((await request('GET', '/tile/nope?width=10')).status === 404, 'unknown recording');
    ok((await request('GET', '/tile/bat1?width=10&x0=8&ncols=5')).status === 400, 'bad tile');
    for (const bad of ['width=10.5', 'width=0', 'width=1e9', 'width=10&x0=-1', 'width=10&ncols=abc', 'width=10&height=0', 'width=10&height=1e6',
                       'width=10&channel=3', 'width=NaN'])
      ok((await request('GET', '/tile/bat1?' + bad)).status === 400, 'rejected before the addon: ' + bad);
    // the same option set (frequencyMax left at 0 = Nyquist) on recordings of two rates: each gets its own Nyquist
    const file2 = Buffer.from(file);
    file2.writeUInt32LE(rate / 2, 12 + 12); file2.writeUInt32LE(rate, 12 + 16);          // same samples declared at half the rate
    ok((await request('PUT', '/recording/bat2', file2)).status === 200, 'second upload');
    const q0 = 'fftSamples=512&noverlap=256&height=100&frequencyMin=10000&colorMap=gray&width=200';
    const b1 = { sampleRate: rate, length: n, numberOfChannels: 1, getChannelData: () => samples };
    const b2 = Object.assign({}, b1, { sampleRate: rate / 2 });
    const mk = () => Spectrogram.create({ fftSamples: 512, noverlap: 256, height: 100, frequencyMin: 10000, colorMap: 'gray', scale: 'linear' });
    const p1 = mk(), p2 = mk();
    const t1 = await request('GET', '/tile/bat1?' + q0), t2 = await request('GET', '/tile/bat2?' + q0);
    ok(equal(new Uint8Array(t1.body), p1.renderCanvas(b1, 200).data) && equal(new Uint8Array(t2.body), p2.renderCanvas(b2, 200).data) &&
       !equal(new Uint8Array(t1.body), new Uint8Array(t2.body)), 'frequencyMax = Nyquist of each recording');
    p1.destroy(); p2.destroy();
    // the plugin cache is bounded: every gain value is a plugin with GPU contexts of its own
    const before = JSON.parse((await request('GET', '/stats')).body.toString()).plugins;
    for (let g = 0; g < 40; g++) ok((await request('GET', '/tile/bat1?' + query + '&width=64&gainDB=' + g)).status === 200, 'gain ' + g);
    const after = JSON.parse((await request('GET', '/stats')).body.toString()).plugins;
    ok(before >= 1 && after === 32, 'plugin cache bounded: ' + before + ' -> ' + after);
    const again = await request('GET', '/tile/bat1?' + query + '&width=' + W + '&x0=0&ncols=256');
    ok(equal(new Uint8Array(again.body), columns(whole.data, W, 300, 0, 256)), 'tile after evictions');
    ok((await request('DELETE', '/recording/bat1')).status === 200 && (await request('DELETE', '/recording/bat2')).status === 200 &&
       server.cache.used === 0, 'delete');
    p.destroy(); pc.destroy();
  } catch (e) {
    failures++;
    console.error(e);
  }
  server.closeBridge();
  console.log(JSON.stringify({ checks, failures }));
  process.exit(failures ? 1 : 0);
});
