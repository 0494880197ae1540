"""Tile cache in front of the plugin's render path.

The reference recomputes getFrequencies on every render() (modules/spectrogram.esm.js:336-344; called
again by modules/wsManager.js:91-110 for every zoom step, colour-map or frequency-range change).  Here the
colour-index plane [frames][fftSamples/2] of a recording stays in device memory, keyed by what it depends on
(recording, channel, fftSamples, hop, window, alpha, gainDB, rangeDB, scale, arithmetic), and a view is
rendered from it: resample (:595-640) of the requested columns only, then LUT + flip + crop + canvas
placement (:345-388), all on the device; only the finished canvas tile crosses PCIe.
"""
import collections

import numpy as np

from . import engine as E


class _Plane:
    __slots__ = ("eng", "ptr", "frames", "bins", "nbytes", "sample_rate")


class TileCache:
    def __init__(self, capacity_bytes=4 << 30):
        self.capacity = int(capacity_bytes)
        self.used = 0
        self.hits = 0
        self.misses = 0
        self._planes = collections.OrderedDict()        # key -> _Plane, least recently used first

    # -------------------------------------------------------------------------------------
    @staticmethod
    def _key(rec_id, channel, plugin, hop, sample_rate):
        return (rec_id, channel, plugin.fftSamples, hop, plugin.windowFunc, plugin.alpha, plugin.gainDB, plugin.rangeDB,
                plugin.scale, plugin.arith, plugin.twiddles, sample_rate, plugin.device)

    def _hop(self, plugin, buffer, width):
        n = plugin.fftSamples
        hop = n - E.plugin_noverlap(n, plugin.noverlap or 0, buffer.length, width)          # :502-506
        if hop <= 0:
            raise ValueError("noverlap must be smaller than fftSamples")
        return hop

    def plane(self, rec_id, plugin, buffer, width, channel=0):
        """The device-resident getFrequencies result of one channel (computed on a miss)."""
        plugin.frequencyMax = plugin.frequencyMax or buffer.sampleRate / 2                   # :496
        plugin.buffer = buffer
        hop = self._hop(plugin, buffer, width)
        key = self._key(rec_id, channel, plugin, hop, buffer.sampleRate)
        pl = self._planes.get(key)
        if pl is not None:
            self.hits += 1
            self._planes.move_to_end(key)
            return pl
        self.misses += 1
        eng = plugin._engine(hop, buffer.sampleRate)
        x = np.ascontiguousarray(buffer.getChannelData(channel), np.float32)
        g = eng.geometry(x.shape[0])
        pl = _Plane()
        pl.eng, pl.frames, pl.bins, pl.sample_rate = eng, int(g.frames), int(g.bins), buffer.sampleRate
        pl.nbytes = max(4, pl.frames * pl.bins)
        self._evict(pl.nbytes)
        pl.ptr = eng.device_malloc(pl.nbytes)
        dpcm = eng.device_malloc(max(4, x.nbytes))
        try:
            if x.nbytes:
                eng.copy_to_device(dpcm, x)
            eng.stft_device(dpcm, x.shape[0], pl.ptr, pl.nbytes)
        except Exception:
            eng.device_free(pl.ptr)
            raise
        finally:
            eng.device_free(dpcm)
        self._planes[key] = pl
        self.used += pl.nbytes
        return pl

    def _evict(self, incoming):
        while self._planes and self.used + incoming > self.capacity:
            _, old = self._planes.popitem(last=False)
            old.eng.device_free(old.ptr)
            self.used -= old.nbytes

    # -------------------------------------------------------------------------------------
    def tile(self, rec_id, plugin, buffer, width, x0=0, ncols=None, channel=0):
        """Columns [x0, x0 + ncols) of the `width`-pixel-wide, plugin.height-pixel-high canvas that
        drawSpectrogram paints for this recording: [height][ncols][4] uint8."""
        ncols = width - x0 if ncols is None else ncols
        if x0 < 0 or ncols < 0 or x0 + ncols > width:
            raise ValueError("tile outside the canvas")
        pl = self.plane(rec_id, plugin, buffer, width, channel)
        eng, height = pl.eng, int(plugin.height)
        out = np.zeros((height, ncols, 4), np.uint8)
        if ncols == 0 or height == 0:
            return out
        sy, sh = E.plugin_crop_scaled(pl.bins, pl.sample_rate, plugin.frequencyMin, plugin.frequencyMax, plugin.scale)
        dy, dh, fill = E.plugin_draw_rect(pl.sample_rate, plugin.frequencyMin, plugin.frequencyMax, plugin.scale, height)
        bg = E.plugin_background(plugin.colorMap[len(plugin.colorMap) - 1]) if fill else None
        eng.set_colormap(plugin._colorMapUint)
        cols = eng.device_malloc(ncols * pl.bins)
        canvas = eng.device_malloc(out.nbytes)
        try:
            if pl.frames:
                eng.resample_cols_device(pl.ptr, pl.frames, pl.bins, width, x0, ncols, cols)
                eng.canvas_draw_device(cols, ncols, pl.bins, sy, sh, dy, dh, height, bg, canvas)
                eng.copy_to_host(out, canvas)
        finally:
            eng.device_free(cols)
            eng.device_free(canvas)
        return out

    def drop(self, rec_id=None):
        """Forget one recording's planes (or all of them) and release their device memory."""
        for key in [k for k in self._planes if rec_id is None or k[0] == rec_id]:
            pl = self._planes.pop(key)
            pl.eng.device_free(pl.ptr)
            self.used -= pl.nbytes
