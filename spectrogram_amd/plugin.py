"""Host-side mirror of the wavesurfer Spectrogram plugin's option set and methods
(reference: modules/spectrogram.esm.js:203-272,336-388,492-542,595-640; factory
modules/wsManager.js:29-54), computing on the GPU through libsonospec.so.

The option names, defaults, validation messages and return layouts are the reference's, so code
that drives the plugin (wsManager.createSpectrogramPlugin / replacePlugin) maps one to one:

    p = Spectrogram.create({"fftSamples": 1024, "noverlap": 768, "windowFunc": "hann",
                            "colorMap": cmap, "frequencyMin": 10e3, "frequencyMax": 128e3,
                            "scale": "linear", "height": 800})
    frames = p.getFrequencies(audio_buffer)      # list[channel] of uint8 [frames][fftSamples/2]
    image  = p.render(audio_buffer, width)       # cropped RGBA rows as drawSpectrogram hands to the canvas
"""
import numpy as np

from . import _ffi as F
from . import engine as E


class AudioBuffer:
    """The slice of the WebAudio AudioBuffer interface the plugin reads
    (sampleRate, length, numberOfChannels, getChannelData)."""

    def __init__(self, channels, sample_rate):
        if isinstance(channels, np.ndarray) and channels.ndim == 1:
            channels = [channels]
        self._ch = [np.ascontiguousarray(c, np.float32) for c in channels]
        self.sampleRate = sample_rate
        self.length = len(self._ch[0]) if self._ch else 0
        self.numberOfChannels = len(self._ch)

    def getChannelData(self, i):
        return self._ch[i]


class Spectrogram:
    @staticmethod
    def create(options=None):
        return Spectrogram(options or {})

    def __init__(self, options):
        t = dict(options)
        self.options = t
        cm = t.get("colorMap")
        if cm is not None and not isinstance(cm, str):          # spectrogram.esm.js:212-219
            if len(cm) < 256:
                raise ValueError("Colormap must contain 256 elements")
            for e in cm:
                if len(e) != 4:
                    raise ValueError("ColorMap entries must contain 4 values")
            self.colorMap = [list(map(float, e)) for e in cm]
        else:                                                   # :221-239
            name = cm or "roseus"
            if name == "gray":
                self.colorMap = [[(255 - i) / 256] * 3 + [1] for i in range(256)]
            elif name == "igray":
                self.colorMap = [[i / 256] * 3 + [1] for i in range(256)]
            else:
                raise ValueError("No such colormap '%s'" % name)
        self.fftSamples = t.get("fftSamples") or 512            # :240-249
        self.height = t.get("height") or 200
        self.noverlap = t.get("noverlap") or None
        self.windowFunc = t.get("windowFunc") or "hann"
        self.alpha = t.get("alpha")
        self.frequencyMin = t.get("frequencyMin") or 0
        self.frequencyMax = t.get("frequencyMax") or 0
        self.gainDB = 20 if t.get("gainDB") is None else t["gainDB"]
        self.rangeDB = 80 if t.get("rangeDB") is None else t["rangeDB"]
        self.scale = t.get("scale") or "mel"
        self.splitChannels = bool(t.get("splitChannels"))
        # engine options that have no counterpart in the reference
        self.arith = t.get("arith", "fast")                     # 'fast' (float32) | 'exact' (double, bit-identical)
        self.twiddles = t.get("twiddles", "plugin")             # 'plugin' (recurrence, :176-190) | 'accurate'
        self.device = t.get("device", 0)
        self._colorMapUint = E.plugin_lut(np.array(self.colorMap[:256], np.float64))   # :262-271
        self.buffer = None
        self._engines = {}

    # -------------------------------------------------------------------------------------
    def _engine(self, hop, sample_rate=None):
        if self.scale not in F.FREQ_SCALES:
            raise ValueError("unknown scale '%s'" % self.scale)
        fb_rate = sample_rate if self.scale != "linear" else None
        key = (self.fftSamples, hop, self.windowFunc, self.alpha, self.gainDB, self.rangeDB, self.arith, self.twiddles,
               self.scale, fb_rate)
        eng = self._engines.get(key)
        if eng is None:
            n = self.fftSamples
            if n <= 0 or (n & (n - 1)) != 0:
                raise ValueError("Invalid buffer size, must be a power of 2.")      # :168-169
            eng = E.Engine(
                fft_n=n, hop=hop, window_id=E.window_id(self.windowFunc), alpha=self.alpha or 0.0,
                gain_db=self.gainDB, range_db=self.rangeDB, scale_mode=F.SCALE_PLUGIN_DB,
                frame_rule=F.FRAMES_PLUGIN_STRICT,
                twiddle_mode=F.TW_PLUGIN_RECURRENCE if self.twiddles == "plugin" else F.TW_ACCURATE,
                arith=F.ARITH_EXACT_F64 if self.arith == "exact" else F.ARITH_FAST_F32,
                out_kind=F.OUT_U8_INDEX, device=self.device)
            eng.set_colormap(self._colorMapUint)
            if fb_rate is not None:      # :509-521: numMelFilters = ... = fftSamples / 2 filters over [0, sampleRate / 2]
                eng.set_filterbank(*E.filterbank_table(self.scale, n, fb_rate))
            self._engines[key] = eng
        return eng

    def getFrequencies(self, buffer, canvas_width=None):
        """spectrogram.esm.js:492-542.  Returns list[channels] of uint8 [F][N/2]."""
        self.frequencyMax = self.frequencyMax or buffer.sampleRate / 2          # :496
        self.buffer = buffer
        n = self.fftSamples
        channels = buffer.numberOfChannels if self.splitChannels else 1         # :495
        width = canvas_width if canvas_width is not None else self.options.get("canvasWidth", 300)
        noverlap = E.plugin_noverlap(n, self.noverlap or 0, buffer.length, width)   # :502-506
        hop = n - noverlap
        if hop <= 0:
            raise ValueError("noverlap must be smaller than fftSamples")
        eng = self._engine(hop, buffer.sampleRate)
        return [eng.stft(buffer.getChannelData(ch)) for ch in range(channels)]

    def resample(self, frames, width):
        """spectrogram.esm.js:595-640: [F][B] uint8 -> [width][B] uint8."""
        frames = np.ascontiguousarray(frames, np.uint8)
        return self._any_engine().resample(frames, width)

    def _any_engine(self):
        if self._engines:
            return next(iter(self._engines.values()))
        return self._engine(self.fftSamples)

    def crop_rows(self, bins):
        """Source-row window drawSpectrogram hands to createImageBitmap (:377-380): (sy, sh)."""
        sr = self.buffer.sampleRate
        return E.plugin_crop_scaled(bins, sr, self.frequencyMin, self.frequencyMax, self.scale)

    def drawSpectrogram(self, frames, width):
        """LUT + flip + crop of spectrogram.esm.js:345-388 for one channel.  Returns
        (full RGBA image [bins][width][4], (sy, sh)); the browser's final drawImage stretch to
        `height` pixels is left to the consumer."""
        cols = self.resample(frames, width)
        eng = self._any_engine()
        eng.set_colormap(self._colorMapUint)
        rgba = eng.colorize(cols)
        return rgba, self.crop_rows(cols.shape[1])

    def render(self, buffer, width):
        """render() of the plugin (:336-344) for channel 0: the cropped RGBA rows."""
        frames = self.getFrequencies(buffer, canvas_width=width)[0]
        rgba, (sy, sh) = self.drawSpectrogram(frames, width)
        sy0, sy1 = max(0, sy), max(0, min(rgba.shape[0], sy + sh))
        return rgba[sy0:sy1]

    def drawCanvas(self, frames, width):
        """The whole of drawSpectrogram (:345-388) for one channel: the [height][width][4] canvas.  The
        browser's drawImage resampling filter is implementation-defined; the engine's is bilinear along
        y between pixel centres (see sono_canvas_draw)."""
        cols = self.resample(frames, width)
        eng = self._any_engine()
        eng.set_colormap(self._colorMapUint)
        sr = self.buffer.sampleRate
        sy, sh = self.crop_rows(cols.shape[1])
        dy, dh, fill = E.plugin_draw_rect(sr, self.frequencyMin, self.frequencyMax, self.scale, self.height)
        bg = E.plugin_background(self.colorMap[len(self.colorMap) - 1]) if fill else None      # :357-361
        return eng.canvas_draw(cols, sy, sh, dy, dh, self.height, bg)

    def renderCanvas(self, buffer, width):
        """render() (:336-344) down to the canvas pixels, channel 0."""
        frames = self.getFrequencies(buffer, canvas_width=width)[0]
        return self.drawCanvas(frames, width)

    def destroy(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}


def createSpectrogramPlugin(colorMap, height=800, frequencyMin=10, frequencyMax=128, fftSamples=1024, noverlap=None,
                            windowFunc="hann", **engine_options):
    """modules/wsManager.js:29-54: kHz -> Hz, scale 'linear', labels off."""
    base = dict(labels=False, height=height, fftSamples=fftSamples, frequencyMin=frequencyMin * 1000,
                frequencyMax=frequencyMax * 1000, scale="linear", windowFunc=windowFunc, colorMap=colorMap)
    if noverlap is not None:
        base["noverlap"] = noverlap
    base.update(engine_options)
    return Spectrogram.create(base)


def noverlap_from_percent(fftSamples, overlapPercent):
    """modules/wsManager.js:87-89."""
    return None if overlapPercent is None else E.overlap_from_percent(fftSamples, overlapPercent)
