"""Host-side mirror of the spectrogram Web Worker's message contract
(reference: spectrogramWorker.js:17-36,98; caller main.js:78-82,1417), rendering on the GPU.

    w = SpectrogramWorker()
    w.onmessage = lambda e: ...                       # receives {"data": {"type": "rendered", ...}}
    w.postMessage({"type": "init", "canvas": canvas, "options": {"colorMap": cmap}})
    w.postMessage({"type": "render", "buffer": samples, "sampleRate": 384000,
                   "fftSize": 1024, "overlap": 75, "options": {...}})

`canvas` is any object; like the OffscreenCanvas in the reference it receives `width` and
`height`, and (there being no 2-D context outside a browser) the painted pixels as `data`, an
(height, width, 4) uint8 array.  The `rendered` message additionally carries width/height/data.
As in the reference, windowFunc/gainDB/rangeDB options are accepted and ignored: the worker always
uses its Hann window and the log10(|X|)/5 scale (spectrogramWorker.js:10-15,46,75).

Opt-in, not in the reference: with ``options["scale"] == "plugin"`` those three options are honoured --
the frames get the plugin's window ``windowFunc`` (spectrogram.esm.js:105-151; an unknown name raises its
error) and the quantiser becomes 20*log10(2/N*|X|) with gainDB/rangeDB (:531-535); geometry, FFT and
colour map stay the worker's.
"""
import numpy as np

from . import _ffi as F
from . import engine as E


class SpectrogramWorker:
    def __init__(self, device=0, arith="fast"):
        self.onmessage = None
        self._canvas = None
        self._ready = False
        self._sample_rate = 44100                                   # spectrogramWorker.js:1
        self._options = {"colorMap": None, "windowFunc": "hann", "gainDB": 20, "rangeDB": 80}   # :10-15
        self._lut = None                                            # colorMapUint, :8
        self._device = device
        self._arith = arith
        self._engines = {}

    # -- the contract ------------------------------------------------------------------------
    def postMessage(self, data, transfer=None):
        t = data.get("type")
        if t == "init":                                             # :19-26
            self._canvas = data.get("canvas")
            self._sample_rate = data.get("sampleRate") or self._sample_rate
            self._ready = self._canvas is not None
            if data.get("options"):
                self._options = {**self._options, **data["options"]}
                if self._options.get("colorMap") is not None and len(self._options["colorMap"]):
                    self._set_color_map(self._options["colorMap"])
        elif t == "setOptions":                                     # :27-29
            opts = data.get("options") or {}
            self._options = {**self._options, **opts}
            if opts.get("colorMap") is not None and len(opts["colorMap"]):
                self._set_color_map(opts["colorMap"])
        elif t == "render":                                         # :30-35
            if not self._ready:
                return                                              # silently dropped before init (:31)
            opts = {**self._options, **(data.get("options") or {})}
            if opts.get("colorMap") is not None and len(opts["colorMap"]):
                self._set_color_map(opts["colorMap"])
            self._render(data["buffer"], data.get("sampleRate") or self._sample_rate, data.get("fftSize") or 1024,
                         data.get("overlap") or 0, opts)

    def terminate(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}

    # -- internals ---------------------------------------------------------------------------
    def _set_color_map(self, color_map):                            # setColorMapUint, :128-143
        cm = np.full((256, 4), np.nan, np.float64)
        for i in range(min(256, len(color_map))):
            cc = color_map[i]
            if cc is None:
                cm[i] = [0, 0, 0, 1]
                continue
            for ch in range(4):
                v = cc[ch] if ch < len(cc) else None
                if v is not None:
                    cm[i, ch] = float(v)
        if len(color_map) < 256:
            cm[len(color_map):] = [0, 0, 0, 1]
        self._lut = E.worker_lut(cm)

    def _engine(self, fft_size, hop, opts=None):
        plugin = bool(opts) and opts.get("scale") == "plugin"
        win = E.window_id(opts.get("windowFunc")) if plugin else F.WIN_WORKER_HANN
        gain = float(opts["gainDB"]) if plugin and opts.get("gainDB") is not None else 20.0
        rng = float(opts["rangeDB"]) if plugin and opts.get("rangeDB") is not None else 80.0
        alpha = (opts.get("alpha") or 0.0) if plugin else 0.0
        key = (fft_size, hop, win, alpha, gain, rng) if plugin else (fft_size, hop)
        eng = self._engines.get(key)
        if eng is None:
            eng = E.Engine(fft_n=fft_size, hop=hop, window_id=win, alpha=alpha, gain_db=gain, range_db=rng,
                           scale_mode=F.SCALE_PLUGIN_DB if plugin else F.SCALE_WORKER_LOG5,
                           frame_rule=F.FRAMES_WORKER_COMPAT, twiddle_mode=F.TW_ACCURATE,
                           arith=F.ARITH_EXACT_F64 if self._arith == "exact" else F.ARITH_FAST_F32,
                           out_kind=F.OUT_RGBA8, device=self._device)
            self._engines[key] = eng
        return eng

    def _render(self, signal, sr, fft_size, overlap_pct, opts=None):  # renderSpectrogram, :38-99
        hop = E.worker_hop(fft_size, overlap_pct)
        eng = self._engine(fft_size, hop, opts)
        eng.set_colormap(self._lut)
        img = eng.stft(signal)
        height, width = img.shape[0], img.shape[1]
        for name, val in (("width", width), ("height", height), ("data", img)):
            try:
                setattr(self._canvas, name, val)
            except Exception:
                if isinstance(self._canvas, dict):
                    self._canvas[name] = val
        if self.onmessage:
            self.onmessage({"data": {"type": "rendered", "width": width, "height": height, "data": img}})
