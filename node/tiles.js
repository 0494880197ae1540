'use strict';
// Tile cache in front of the plugin's render path.  The reference recomputes getFrequencies on every
// render() (modules/spectrogram.esm.js:336-344; modules/wsManager.js:91-110 calls it again for every zoom,
// colour-map or frequency-range change).  Here the colour-index plane [frames][fftSamples/2] of a recording
// stays in device memory, keyed by what it depends on, and a view is rendered from it: resample (:595-640)
// of the requested columns only, then LUT + flip + crop + canvas placement (:345-388), all on the device;
// only the finished canvas tile crosses PCIe.
const S = require('./index.js');
const N = S.native;
const SCALE_IDS = { linear: 0, mel: 1, logarithmic: 2, bark: 3, erb: 4 };

class TileCache {
  constructor(capacityBytes = 4 * 2 ** 30) {
    this.capacity = capacityBytes;
    this.used = 0;
    this.hits = 0;
    this.misses = 0;
    this._planes = new Map();     // key -> plane, least recently used first (Map keeps insertion order)
  }

  _key(recId, channel, p, hop, sampleRate) {
    return [recId, channel, p.fftSamples, hop, p.windowFunc, p.alpha, p.gainDB, p.rangeDB, p.scale, p.arith, sampleRate, p.device].join('|');
  }

  // The device-resident getFrequencies result of one channel (computed on a miss).
  plane(recId, plugin, buffer, width, channel = 0) {
    // (:496 sets frequencyMax on the plugin once; a plugin object that serves recordings of several rates must
    // take each recording's own Nyquist when the option was left at 0)
    if (!plugin._frequencyMaxGiven) plugin._frequencyMaxGiven = plugin.frequencyMax ? 'option' : 'nyquist';
    if (plugin._frequencyMaxGiven === 'nyquist') plugin.frequencyMax = buffer.sampleRate / 2;
    plugin.buffer = buffer;
    const n = plugin.fftSamples;
    const hop = n - N.pluginNoverlap(n, plugin.noverlap || 0, buffer.length, width);           // :502-506
    if (hop <= 0) throw new Error('noverlap must be smaller than fftSamples');
    const key = this._key(recId, channel, plugin, hop, buffer.sampleRate);
    let pl = this._planes.get(key);
    if (pl) {
      this.hits++;
      this._planes.delete(key);
      this._planes.set(key, pl);
      return pl;
    }
    this.misses++;
    if (!this._hook) this._hook = (handle) => this.dropEngines([handle]);
    if (plugin._evictHooks) plugin._evictHooks.add(this._hook);   // a context the plugin retires takes its planes with it
    const h = plugin._engine(hop, buffer.sampleRate);
    const x = buffer.getChannelData(channel);
    const g = N.geometry(h, x.length);
    pl = { h, recId, frames: g.frames, bins: g.bins, sampleRate: buffer.sampleRate, nbytes: Math.max(4, g.frames * g.bins), ptr: null };
    this._evict(pl.nbytes);
    pl.ptr = N.deviceMalloc(h, pl.nbytes);
    const dpcm = N.deviceMalloc(h, Math.max(4, x.length * 4));
    try {
      if (x.length) N.copyToDevice(h, dpcm, x);
      N.stftDevice(h, dpcm, x.length, pl.ptr, pl.nbytes, true);
    } catch (e) {
      N.deviceFree(h, pl.ptr);
      throw e;
    } finally {
      N.deviceFree(h, dpcm);
    }
    this._planes.set(key, pl);
    this.used += pl.nbytes;
    return pl;
  }

  _evict(incoming) {
    for (const [key, old] of this._planes) {
      if (this.used + incoming <= this.capacity) break;
      this._planes.delete(key);
      N.deviceFree(old.h, old.ptr);
      this.used -= old.nbytes;
    }
  }

  // Columns [x0, x0 + ncols) of the width-pixel-wide, plugin.height-pixel-high canvas drawSpectrogram paints:
  // ImageData-shaped { width: ncols, height, data: Uint8ClampedArray }.
  tile(recId, plugin, buffer, width, x0 = 0, ncols = width - x0, channel = 0) {
    if (x0 < 0 || ncols < 0 || x0 + ncols > width) throw new RangeError('tile outside the canvas');
    const pl = this.plane(recId, plugin, buffer, width, channel);
    const h = pl.h, height = plugin.height;
    const data = new Uint8ClampedArray(height * ncols * 4);
    if (!ncols || !height || !pl.frames) return { width: ncols, height, data };
    const scale = SCALE_IDS[plugin.scale];
    const crop = N.pluginCropScaled(pl.bins, pl.sampleRate, plugin.frequencyMin, plugin.frequencyMax, scale);
    const rect = N.pluginDrawRect(pl.sampleRate, plugin.frequencyMin, plugin.frequencyMax, scale, height);
    const bg = rect.fill ? N.pluginBackground(Float64Array.from(plugin.colorMap[plugin.colorMap.length - 1])) : null;
    N.setColormap(h, plugin._colorMapUint);
    const cols = N.deviceMalloc(h, ncols * pl.bins), canvas = N.deviceMalloc(h, data.length);
    try {
      N.resampleColsDevice(h, pl.ptr, pl.frames, pl.bins, width, x0, ncols, cols);
      N.canvasDrawDevice(h, cols, ncols, pl.bins, crop.sy, crop.sh, rect.dy, rect.dh, height, bg, canvas);
      N.copyToHost(h, data, canvas);
    } finally {
      N.deviceFree(h, cols);
      N.deviceFree(h, canvas);
    }
    return { width: ncols, height, data };
  }

  // Forget the planes that live on the given engine handles (a plugin about to be destroyed): their device
  // memory belongs to those contexts and must be released before them.
  dropEngines(handles) {
    const set = new Set(handles);
    for (const [key, pl] of this._planes) {
      if (!set.has(pl.h)) continue;
      this._planes.delete(key);
      N.deviceFree(pl.h, pl.ptr);
      this.used -= pl.nbytes;
    }
  }

  // Forget one recording's planes (or all) and release their device memory.
  drop(recId) {
    for (const [key, pl] of this._planes) {
      if (recId !== undefined && pl.recId !== recId) continue;
      this._planes.delete(key);
      N.deviceFree(pl.h, pl.ptr);
      this.used -= pl.nbytes;
    }
  }
}

module.exports = { TileCache };
