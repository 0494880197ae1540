'use strict';
// Local bridge between the browser UI and the GPU engine: the page asks for canvas tiles over HTTP
// instead of running getFrequencies on its main thread (modules/wsManager.js:91-110 -> plugin.render()).
// Node's built-in http only; binds to 127.0.0.1.
//
//   PUT  /recording/<id>            body: a RIFF/WAVE file                        -> {id, sampleRate, length, duration}
//   GET  /tile/<id>?width=&x0=&ncols=&height=&fftSamples=&noverlap=&windowFunc=&frequencyMin=&frequencyMax=
//                   &gainDB=&rangeDB=&colorMap=gray|igray                          -> raw RGBA bytes, headers X-Width / X-Height
//   DELETE /recording/<id>
// A colour map other than the two built-in names is installed with PUT /colormap/<name> (JSON: 256 x [r,g,b,a], 0..1).
const http = require('http');
const { URL } = require('url');
const wav = require('./wav.js');
const { Spectrogram } = require('./spectrogram_plugin.js');
const { TileCache } = require('./tiles.js');

function createBridge({ capacityBytes, maxUploadBytes = 1 << 30, maxPlugins = 32 } = {}) {
  const cache = new TileCache(capacityBytes);
  const recordings = new Map();   // id -> { sampleRate, length, numberOfChannels, getChannelData }
  const colormaps = new Map();    // name -> 256 x [r, g, b, a]
  const plugins = new Map();      // option string -> Spectrogram (keeps its engines alive); least recently used first

  const body = (req) => new Promise((resolve, reject) => {
    const chunks = [];
    let n = 0;
    req.on('data', (c) => { n += c.length; if (n > maxUploadBytes) { reject(new Error('upload too large')); req.destroy(); } else chunks.push(c); });
    req.on('end', () => resolve(Buffer.concat(chunks)));
    req.on('error', reject);
  });
  const json = (res, code, obj) => { res.writeHead(code, { 'Content-Type': 'application/json' }); res.end(JSON.stringify(obj)); };
  const num = (q, name, dflt) => (q.has(name) && q.get(name) !== '' ? Number(q.get(name)) : dflt);

  function pluginFor(q) {
    const cmName = q.get('colorMap') || 'gray';
    const opts = {
      fftSamples: num(q, 'fftSamples', 1024), height: num(q, 'height', 800), windowFunc: q.get('windowFunc') || 'hann',
      frequencyMin: num(q, 'frequencyMin', 0), frequencyMax: num(q, 'frequencyMax', 0), gainDB: num(q, 'gainDB', 20),
      rangeDB: num(q, 'rangeDB', 80), scale: q.get('scale') || 'linear', noverlap: num(q, 'noverlap', null),
    };
    if (q.has('alpha')) opts.alpha = Number(q.get('alpha'));
    const key = JSON.stringify(opts) + '|' + cmName + '|' + (colormaps.has(cmName) ? colormaps.get(cmName).version : 0);
    let p = plugins.get(key);
    if (p) {
      plugins.delete(key);          // most recently used last
    } else {
      // every distinct option set (each step of a gain slider, say) is a plugin with GPU contexts of its own:
      // keep the most recent `maxPlugins`; an evicted plugin's planes go first, then its contexts
      while (plugins.size >= maxPlugins) {
        const [oldKey, old] = plugins.entries().next().value;
        cache.dropEngines(old._engines.values());
        old.destroy();
        plugins.delete(oldKey);
      }
      opts.colorMap = colormaps.has(cmName) ? colormaps.get(cmName).table : cmName;
      p = Spectrogram.create(opts);
    }
    plugins.set(key, p);
    return p;
  }

  const server = http.createServer(async (req, res) => {
    try {
      const url = new URL(req.url, 'http://127.0.0.1');
      const parts = url.pathname.split('/').filter(Boolean);
      if (parts[0] === 'recording' && parts[1] && req.method === 'PUT') {
        const file = await body(req);
        const info = wav.pcmSamples(file).info;
        const channels = [];
        for (let ch = 0; ch < info.channels; ch++) channels.push(wav.channelData(file, ch));
        const length = channels[0].length;
        cache.drop(parts[1]);
        recordings.set(parts[1], { sampleRate: info.sampleRate, length, numberOfChannels: info.channels, getChannelData: (i) => channels[i] });
        return json(res, 200, { id: parts[1], sampleRate: info.sampleRate, length, duration: length / info.sampleRate });
      }
      if (parts[0] === 'recording' && parts[1] && req.method === 'DELETE') {
        cache.drop(parts[1]);
        return json(res, recordings.delete(parts[1]) ? 200 : 404, { id: parts[1] });
      }
      if (parts[0] === 'colormap' && parts[1] && req.method === 'PUT') {
        const table = JSON.parse((await body(req)).toString('utf8'));
        const prev = colormaps.get(parts[1]);
        colormaps.set(parts[1], { table, version: prev ? prev.version + 1 : 1 });
        return json(res, 200, { name: parts[1], entries: table.length });
      }
      if (parts[0] === 'tile' && parts[1] && req.method === 'GET') {
        const rec = recordings.get(parts[1]);
        if (!rec) return json(res, 404, { error: 'unknown recording' });
        const q = url.searchParams;
        const width = num(q, 'width', 300), x0 = num(q, 'x0', 0), ncols = num(q, 'ncols', width - x0), channel = num(q, 'channel', 0);
        // (what reaches the addon sizes device buffers: whole numbers inside the canvas, a canvas of a sane size)
        const whole = (v, lo, hi) => Number.isInteger(v) && v >= lo && v <= hi;
        if (!whole(width, 1, 1 << 24) || !whole(x0, 0, width) || !whole(ncols, 0, width - x0) || !whole(channel, 0, rec.numberOfChannels - 1) ||
            !whole(num(q, 'height', 800), 1, 16384))
          return json(res, 400, { error: 'width, x0, ncols, height, channel: whole numbers with 1 <= width <= 2^24, 0 <= x0, x0 + ncols <= width, ' +
                                         '1 <= height <= 16384, channel < ' + rec.numberOfChannels });
        const t = cache.tile(parts[1], pluginFor(q), rec, width, x0, ncols, channel);
        res.writeHead(200, { 'Content-Type': 'application/octet-stream', 'X-Width': String(t.width), 'X-Height': String(t.height),
                             'X-Cache-Hits': String(cache.hits), 'X-Cache-Misses': String(cache.misses) });
        return res.end(Buffer.from(t.data.buffer, t.data.byteOffset, t.data.byteLength));
      }
      if (parts[0] === 'stats') return json(res, 200, { hits: cache.hits, misses: cache.misses, bytes: cache.used, recordings: recordings.size, plugins: plugins.size });
      json(res, 404, { error: 'no such route' });
    } catch (e) {
      json(res, 400, { error: String((e && e.message) || e) });
    }
  });
  server.cache = cache;
  server.closeBridge = () => { cache.drop(); for (const p of plugins.values()) p.destroy(); plugins.clear(); server.close(); };
  return server;
}

module.exports = { createBridge };

if (require.main === module) {
  const port = Number(process.argv[2] || 8177);
  createBridge().listen(port, '127.0.0.1', () => console.log('sonospec bridge on http://127.0.0.1:' + port));
}
