"""ctypes front end of the CPU checker ``libsono_oracle.so`` (test infrastructure).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  It restates the reference's JavaScript (``spectrogramWorker.js``,
``modules/spectrogram.esm.js``); see ``sono_oracle.h`` for the per-function citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsono_oracle.so")

WINDOW_IDS = {
    "bartlett": 0, "bartlettHann": 1, "blackman": 2, "cosine": 3, "gauss": 4,
    "hamming": 5, "hann": 6, "lanczoz": 7, "rectangular": 8, "triangular": 9,
}


def build(force=False):
    """Compile the checker with gcc (no FMA contraction).  Cheap; safe to call repeatedly."""
    src = os.path.join(_HERE, "sono_oracle.c")
    hdr = os.path.join(_HERE, "sono_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        f32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
        u8p, u32p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
        L.orc_plugin_window.argtypes = [C.c_int, C.c_int, C.c_double, f32p]
        L.orc_plugin_window.restype = C.c_int
        L.orc_plugin_tables.argtypes = [C.c_int, f32p, f32p, u32p]
        L.orc_plugin_spectrum.argtypes = [f32p, C.c_int, f32p, f32p, f32p, u32p, f32p, f64p, C.POINTER(C.c_int)]
        L.orc_plugin_spectrum.restype = C.c_int
        L.orc_plugin_noverlap.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_int]
        L.orc_plugin_noverlap.restype = C.c_int
        L.orc_plugin_frame_count.argtypes = [C.c_size_t, C.c_int, C.c_int]
        L.orc_plugin_frame_count.restype = C.c_size_t
        L.orc_plugin_get_frequencies.argtypes = [f32p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                                 C.c_double, C.c_double, u8p, f32p, f64p, C.POINTER(C.c_int)]
        L.orc_plugin_get_frequencies.restype = C.c_size_t
        L.orc_plugin_resample.argtypes = [u8p, C.c_size_t, C.c_int, C.c_size_t, u8p]
        L.orc_plugin_lut.argtypes = [f64p, u8p]
        L.orc_plugin_crop.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        L.orc_plugin_draw.argtypes = [u8p, C.c_size_t, C.c_int, u8p, u8p]
        L.orc_worker_hann.argtypes = [C.c_int, f32p]
        L.orc_worker_twiddles.argtypes = [C.c_int, f64p, f64p]
        L.orc_worker_fft.argtypes = [f32p, f32p, C.c_int, f64p, f64p]
        L.orc_worker_hop.argtypes = [C.c_int, C.c_double]
        L.orc_worker_hop.restype = C.c_int
        L.orc_worker_width.argtypes = [C.c_size_t, C.c_int, C.c_int]
        L.orc_worker_width.restype = C.c_size_t
        L.orc_worker_frames.argtypes = [C.c_size_t, C.c_int, C.c_int]
        L.orc_worker_frames.restype = C.c_size_t
        L.orc_worker_lut.argtypes = [f64p, u8p]
        L.orc_worker_render.argtypes = [f32p, C.c_size_t, C.c_int, C.c_double, u8p, f64p, f64p, u8p, u8p]
        L.orc_worker_render_mt.argtypes = [f32p, C.c_size_t, C.c_int, C.c_double, u8p, u8p, C.c_int]
        L.orc_frame_peaks.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, i32p, u8p]
        L.orc_worker_magnitudes.argtypes = [f32p, C.c_size_t, C.c_int, C.c_int, C.c_size_t, f32p, u8p]
        L.orc_frame_peaks_f32.argtypes = [f32p, C.c_size_t, C.c_int, C.c_int, C.c_int, i32p]
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


# ---------------------------------------------------------------- plugin (spectrogram.esm.js)
def plugin_window(name, n, alpha=None):
    if name not in WINDOW_IDS:
        raise ValueError("No such window function '%s'" % name)
    out = np.empty(n, np.float32)
    a = float(alpha) if alpha else 0.0
    rc = lib().orc_plugin_window(WINDOW_IDS[name], n, a, _p(out, C.c_float))
    assert rc == 0
    return out


def plugin_tables(n):
    s, c, r = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, np.uint32)
    lib().orc_plugin_tables(n, _p(s, C.c_float), _p(c, C.c_float), _p(r, C.c_uint32))
    return s, c, r


class PluginFFT:
    """The plugin's FFT object: tables once, running peak across calls."""

    def __init__(self, n, window="hann", alpha=None):
        self.n = n
        self.win = plugin_window(window, n, alpha)
        self.sin_t, self.cos_t, self.rev = plugin_tables(n)
        self.peak = C.c_double(0.0)
        self.peak_band = C.c_int(0)

    def calculate_spectrum(self, frame):
        frame = _f32(frame)
        if frame.shape[0] != self.n:
            raise ValueError("Supplied buffer is not the same size as defined FFT. FFT Size: %d Buffer Size: %d"
                             % (self.n, frame.shape[0]))
        mag = np.empty(self.n // 2, np.float32)
        rc = lib().orc_plugin_spectrum(_p(frame, C.c_float), self.n, _p(self.win, C.c_float), _p(self.sin_t, C.c_float),
                                       _p(self.cos_t, C.c_float), _p(self.rev, C.c_uint32), _p(mag, C.c_float),
                                       C.byref(self.peak), C.byref(self.peak_band))
        if rc != 0:
            raise ValueError("Invalid buffer size, must be a power of 2.")
        return mag


def plugin_get_frequencies(x, n, noverlap=None, canvas_width=300, window="hann", alpha=None,
                           gain_db=20.0, range_db=80.0, want_mags=False):
    x = _f32(x)
    L = lib()
    ov = L.orc_plugin_noverlap(n, int(noverlap or 0), x.shape[0], canvas_width)
    hop = n - ov
    if hop <= 0:
        raise ValueError("noverlap must be smaller than fftSamples")
    frames = L.orc_plugin_frame_count(x.shape[0], n, hop)
    out = np.empty((frames, n // 2), np.uint8)
    mags = np.empty((frames, n // 2), np.float32) if want_mags else None
    peak, band = C.c_double(0), C.c_int(0)
    got = L.orc_plugin_get_frequencies(_p(x, C.c_float), x.shape[0], n, int(noverlap or 0), canvas_width,
                                       WINDOW_IDS[window], float(alpha) if alpha else 0.0, gain_db, range_db,
                                       _p(out, C.c_uint8), _p(mags, C.c_float), C.byref(peak), C.byref(band))
    assert got == frames, (got, frames)
    if want_mags:
        return out, mags, peak.value, band.value
    return out


SCALE_IDS = {"linear": 0, "mel": 1, "logarithmic": 2, "bark": 3, "erb": 4}


def hz_to_scale(scale, hz):
    L = lib()
    L.orc_hz_to_scale.restype = C.c_double
    return L.orc_hz_to_scale(SCALE_IDS[scale], C.c_double(hz))


def scale_to_hz(scale, v):
    L = lib()
    L.orc_scale_to_hz.restype = C.c_double
    return L.orc_scale_to_hz(SCALE_IDS[scale], C.c_double(v))


def plugin_filterbank(scale, n, sample_rate):
    b, w0, w1 = np.empty(n // 2, np.int32), np.empty(n // 2, np.float64), np.empty(n // 2, np.float64)
    lib().orc_plugin_filterbank(SCALE_IDS[scale], n, C.c_double(sample_rate), n // 2, _p(b, C.c_int32), _p(w0, C.c_double),
                                _p(w1, C.c_double))
    return b, w0, w1


def plugin_get_frequencies_scaled(x, n, noverlap, scale, sample_rate, canvas_width=300, window="hann", alpha=None,
                                  gain_db=20.0, range_db=80.0):
    x = _f32(x)
    L = lib()
    ov = L.orc_plugin_noverlap(n, int(noverlap or 0), x.shape[0], canvas_width)
    frames = L.orc_plugin_frame_count(x.shape[0], n, n - ov)
    out = np.empty((frames, n // 2), np.uint8)
    L.orc_plugin_get_frequencies_scaled.restype = C.c_size_t
    got = L.orc_plugin_get_frequencies_scaled(_p(x, C.c_float), C.c_size_t(x.shape[0]), n, int(noverlap or 0), canvas_width,
                                              WINDOW_IDS[window], C.c_double(float(alpha) if alpha else 0.0), C.c_double(gain_db),
                                              C.c_double(range_db), SCALE_IDS[scale], C.c_double(sample_rate), _p(out, C.c_uint8))
    assert got == frames, (got, frames)
    return out


def plugin_crop_scaled(bins, sample_rate, fmin, fmax, scale):
    sy, sh = C.c_long(0), C.c_long(0)
    lib().orc_plugin_crop_scaled(bins, C.c_double(sample_rate), C.c_double(fmin), C.c_double(fmax), SCALE_IDS[scale],
                                 C.byref(sy), C.byref(sh))
    return sy.value, sh.value


def plugin_draw_rect(sample_rate, fmin, fmax, scale, height):
    dy, dh, fill = C.c_double(0), C.c_double(0), C.c_int(0)
    lib().orc_plugin_draw_rect(C.c_double(sample_rate), C.c_double(fmin), C.c_double(fmax), SCALE_IDS[scale], int(height),
                               C.byref(dy), C.byref(dh), C.byref(fill))
    return dy.value, dh.value, bool(fill.value)


def plugin_background(last):
    last = np.ascontiguousarray(last, np.float64).reshape(4)
    out = np.empty(4, np.uint8)
    lib().orc_plugin_background(_p(last, C.c_double), _p(out, C.c_uint8))
    return out


def canvas_draw(image, sy, sh, dy, dh, height, background=None):
    """The engine's own drawImage filter on a rows x width x 4 image (parity unpinned: browser-defined step)."""
    image = np.ascontiguousarray(image, np.uint8)
    rows, width = image.shape[0], image.shape[1]
    out = np.empty((int(height), width, 4), np.uint8)
    bg = None if background is None else np.ascontiguousarray(background, np.uint8).reshape(4)
    lib().orc_canvas_draw(_p(image, C.c_uint8), C.c_size_t(width), int(rows), C.c_long(int(sy)), C.c_long(int(sh)),
                          C.c_double(dy), C.c_double(dh), int(height), _p(bg, C.c_uint8), _p(out, C.c_uint8))
    return out


def plugin_resample(frames, out_width):
    frames = np.ascontiguousarray(frames, np.uint8)
    F, B = frames.shape
    out = np.zeros((out_width, B), np.uint8)
    lib().orc_plugin_resample(_p(frames, C.c_uint8), F, B, out_width, _p(out, C.c_uint8))
    return out


def plugin_lut(colormap):
    cm = np.ascontiguousarray(colormap, np.float64).reshape(1024)
    lut = np.empty(1024, np.uint8)
    lib().orc_plugin_lut(_p(cm, C.c_double), _p(lut, C.c_uint8))
    return lut


def plugin_crop(bins, sample_rate, fmin_hz, fmax_hz):
    sy, sh = C.c_long(0), C.c_long(0)
    lib().orc_plugin_crop(bins, sample_rate, fmin_hz, fmax_hz, C.byref(sy), C.byref(sh))
    return sy.value, sh.value


def plugin_draw(frames, lut):
    frames = np.ascontiguousarray(frames, np.uint8)
    W, B = frames.shape
    lut = np.ascontiguousarray(lut, np.uint8)
    rgba = np.empty((B, W, 4), np.uint8)
    lib().orc_plugin_draw(_p(frames, C.c_uint8), W, B, _p(lut, C.c_uint8), _p(rgba, C.c_uint8))
    return rgba


# ---------------------------------------------------------------- worker (spectrogramWorker.js)
def worker_hann(n):
    out = np.empty(n, np.float32)
    lib().orc_worker_hann(n, _p(out, C.c_float))
    return out


def worker_twiddles(n):
    c, s = np.zeros(n, np.float64), np.zeros(n, np.float64)
    lib().orc_worker_twiddles(n, _p(c, C.c_double), _p(s, C.c_double))
    return c[:n - 1], s[:n - 1]


def worker_fft(re, im, cos_t=None, sin_t=None):
    re, im = _f32(re).copy(), _f32(im).copy()
    n = re.shape[0]
    if cos_t is None:
        cos_t, sin_t = worker_twiddles(n)
    cos_t = np.ascontiguousarray(cos_t, np.float64)
    sin_t = np.ascontiguousarray(sin_t, np.float64)
    lib().orc_worker_fft(_p(re, C.c_float), _p(im, C.c_float), n, _p(cos_t, C.c_double), _p(sin_t, C.c_double))
    return re, im


def worker_geometry(length, n, overlap_pct):
    L = lib()
    hop = L.orc_worker_hop(n, float(overlap_pct))
    return hop, L.orc_worker_width(length, n, hop), L.orc_worker_frames(length, n, hop)


def worker_lut(colormap):
    cm = np.ascontiguousarray(colormap, np.float64).reshape(1024)
    lut = np.empty(1024, np.uint8)
    lib().orc_worker_lut(_p(cm, C.c_double), _p(lut, C.c_uint8))
    return lut


def worker_render(x, n=1024, overlap_pct=0.0, lut=None, cos_t=None, sin_t=None, want_index=False, threads=1):
    """Returns the RGBA image (n/2, width, 4) [and the frames x bins index matrix]."""
    x = _f32(x)
    hop, width, frames = worker_geometry(x.shape[0], n, overlap_pct)
    rgba = np.empty((n // 2, width, 4), np.uint8)
    lut_a = np.ascontiguousarray(lut, np.uint8) if lut is not None else None
    if threads > 1:
        lib().orc_worker_render_mt(_p(x, C.c_float), x.shape[0], n, float(overlap_pct), _p(lut_a, C.c_uint8),
                                   _p(rgba, C.c_uint8), threads)
        return rgba
    idx = np.empty((frames, n // 2), np.uint8) if want_index else None
    ct = np.ascontiguousarray(cos_t, np.float64) if cos_t is not None else None
    st = np.ascontiguousarray(sin_t, np.float64) if sin_t is not None else None
    lib().orc_worker_render(_p(x, C.c_float), x.shape[0], n, float(overlap_pct), _p(lut_a, C.c_uint8),
                            _p(ct, C.c_double), _p(st, C.c_double), _p(rgba, C.c_uint8), _p(idx, C.c_uint8))
    return (rgba, idx) if want_index else rgba


def frame_peaks(idx, lo=0, hi=None):
    idx = np.ascontiguousarray(idx, np.uint8)
    F, B = idx.shape
    hi = B if hi is None else hi
    pb, pv = np.empty(F, np.int32), np.empty(F, np.uint8)
    lib().orc_frame_peaks(_p(idx, C.c_uint8), F, B, lo, hi, _p(pb, C.c_int32), _p(pv, C.c_uint8))
    return pb, pv


def worker_magnitudes(x, n, hop, frames):
    """float32 |X| and colour index of every worker frame x bin."""
    x = _f32(x)
    mag = np.empty((frames, n // 2), np.float32)
    idx = np.empty((frames, n // 2), np.uint8)
    lib().orc_worker_magnitudes(_p(x, C.c_float), x.shape[0], n, hop, frames, _p(mag, C.c_float), _p(idx, C.c_uint8))
    return mag, idx


def frame_peaks_f32(mag, lo=0, hi=None):
    mag = np.ascontiguousarray(mag, np.float32)
    F, B = mag.shape
    hi = B if hi is None else hi
    pb = np.empty(F, np.int32)
    lib().orc_frame_peaks_f32(_p(mag, C.c_float), F, B, lo, hi, _p(pb, C.c_int32))
    return pb
