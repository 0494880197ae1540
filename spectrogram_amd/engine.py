"""Engine: one GPU context of libsonospec.so with numpy-friendly calls.

This is plumbing over the C ABI; all arithmetic happens in the HIP kernels.
"""
import ctypes as C

import numpy as np

from . import _ffi as F


def window_id(name):
    """Window name -> id; the error text is the reference's (spectrogram.esm.js:149-150)."""
    if name is None:
        name = "hann"
    if name not in F.WINDOW_IDS:
        raise ValueError("No such window function '%s'" % name)
    return F.WINDOW_IDS[name]


class Engine:
    def __init__(self, fft_n=1024, hop=256, window_id=F.WINDOW_IDS["hann"], alpha=0.0, gain_db=20.0, range_db=80.0,
                 scale_mode=F.SCALE_PLUGIN_DB, frame_rule=F.FRAMES_PLUGIN_STRICT, twiddle_mode=F.TW_ACCURATE,
                 arith=F.ARITH_FAST_F32, out_kind=F.OUT_U8_INDEX, in_kind=F.IN_F32, bin_lo=0, bin_hi=0, device=0):
        self._lib = F.lib()
        cfg = F.SonoConfig()
        cfg.fft_n, cfg.hop, cfg.window_id, cfg.alpha = int(fft_n), int(hop), int(window_id), float(alpha or 0.0)
        cfg.gain_db, cfg.range_db = float(gain_db), float(range_db)
        cfg.scale_mode, cfg.frame_rule, cfg.twiddle_mode = scale_mode, frame_rule, twiddle_mode
        cfg.arith, cfg.out_kind, cfg.in_kind = arith, out_kind, in_kind
        cfg.bin_lo, cfg.bin_hi = int(bin_lo), int(bin_hi)
        self.cfg = cfg
        self._ctx = self._lib.sono_create(int(device), C.byref(cfg))
        if not self._ctx:
            msg = self._lib.sono_create_error().decode()
            raise F.SonoError(F.ERR_ARG, msg)

    # -- lifetime
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.sono_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != F.OK:
            raise F.SonoError(rc, self._lib.sono_last_error(self._ctx).decode())

    # -- configuration
    def set_stream(self, hip_stream):
        self._check(self._lib.sono_set_stream(self._ctx, C.c_void_p(hip_stream or 0)))

    def set_colormap(self, lut):
        """lut: 1024 uint8 (256 x RGBA) or None for the worker's grey fallback."""
        if lut is None:
            self._check(self._lib.sono_set_colormap(self._ctx, None))
            return
        lut = np.ascontiguousarray(lut, np.uint8).reshape(-1)
        if lut.shape[0] != 1024:
            raise ValueError("LUT must hold 256 RGBA entries")
        self._check(self._lib.sono_set_colormap(self._ctx, F.ptr(lut)))

    def set_tables(self, window=None, tw_cos=None, tw_sin=None):
        n = self.cfg.fft_n
        w = np.ascontiguousarray(window, np.float32) if window is not None else None
        c = np.ascontiguousarray(tw_cos, np.float64) if tw_cos is not None else None
        s = np.ascontiguousarray(tw_sin, np.float64) if tw_sin is not None else None
        if w is not None and w.shape[0] != n:
            raise ValueError("window must have fft_n entries")
        if c is not None and (c.shape[0] != n - 1 or s is None or s.shape[0] != n - 1):
            raise ValueError("twiddle tables must have fft_n-1 entries each")
        self._check(self._lib.sono_set_tables(self._ctx, F.ptr(w), F.ptr(c), F.ptr(s)))

    def set_filterbank(self, bin=None, w0=None, w1=None):
        """Two-tap filter bank of a non-linear frequency scale (fft_n/2 filters); None removes it."""
        if bin is None:
            self._check(self._lib.sono_set_filterbank(self._ctx, None, None, None))
            return
        b = np.ascontiguousarray(bin, np.int32)
        a0, a1 = np.ascontiguousarray(w0, np.float64), np.ascontiguousarray(w1, np.float64)
        if not (b.shape[0] == a0.shape[0] == a1.shape[0] == self.cfg.fft_n // 2):
            raise ValueError("a filter bank has fft_n/2 filters")
        self._check(self._lib.sono_set_filterbank(self._ctx, F.ptr(b), F.ptr(a0), F.ptr(a1)))

    # -- geometry
    def geometry(self, n_samples):
        g = F.SonoGeometry()
        self._check(self._lib.sono_geometry_of(self._ctx, int(n_samples), C.byref(g)))
        return g

    def out_bytes(self, n_samples):
        return int(self._lib.sono_out_bytes(self._ctx, int(n_samples)))

    def _out_array(self, n_samples):
        g = self.geometry(n_samples)
        k = self.cfg.out_kind
        if k == F.OUT_U8_INDEX:
            return np.empty((g.frames, g.bins), np.uint8)
        if k == F.OUT_RGBA8:
            return np.empty((g.bins, g.width, 4), np.uint8)
        if k == F.OUT_FEATURES:
            return np.empty(g.frames, F.FEATURE_DTYPE)
        return np.empty((g.frames, g.bins), np.float32)

    def _in_array(self, pcm):
        dt = np.int16 if self.cfg.in_kind == F.IN_I16 else np.float32
        return np.ascontiguousarray(pcm, dt).reshape(-1)

    # -- transforms on host buffers (staged through the context)
    def stft(self, pcm):
        x = self._in_array(pcm)
        out = self._out_array(x.shape[0])
        self._check(self._lib.sono_stft(self._ctx, F.ptr(x), x.shape[0], F.ptr(out), out.nbytes, F.MEM_HOST))
        return out

    # -- transforms on device pointers (ints): the caller owns the memory
    def stft_device(self, pcm_ptr, n_samples, out_ptr, out_capacity, wait=True):
        rc = self._lib.sono_stft_async(self._ctx, C.c_void_p(pcm_ptr), int(n_samples), C.c_void_p(out_ptr),
                                       int(out_capacity))
        self._check(rc)
        if wait:
            self.wait()

    def wait(self):
        self._check(self._lib.sono_wait(self._ctx))

    def stft_static_device(self, pcm_ptr, n_samples, out_ptr, out_capacity, wait=True):
        """sono_stft_static: the RGBA transform through the library's static index plane (no allocation)."""
        rc = self._lib.sono_stft_static(self._ctx, C.c_void_p(pcm_ptr), int(n_samples), C.c_void_p(out_ptr),
                                        int(out_capacity))
        self._check(rc)
        if wait:
            self.wait()

    def kernel_route(self):
        """Kernels the most recent transform call enqueued (sono_kernel_route)."""
        return self._lib.sono_kernel_route(self._ctx).decode()

    def features_recomputed(self):
        """Records of the last float32 feature launch that were recomputed in the exact arithmetic."""
        return int(self._lib.sono_features_recomputed(self._ctx))

    def reserve(self, n_samples):
        """Size the staging buffers for host calls of up to n_samples (sono_reserve)."""
        self._check(self._lib.sono_reserve(self._ctx, int(n_samples)))

    # -- device memory owned by the caller
    def device_malloc(self, nbytes):
        p = self._lib.sono_device_malloc(self._ctx, int(nbytes))
        if not p:
            raise F.SonoError(F.ERR_DEVICE, self._lib.sono_last_error(self._ctx).decode())
        return p

    def device_free(self, ptr):
        self._lib.sono_device_free(self._ctx, C.c_void_p(ptr))

    def copy_to_host(self, arr, ptr):
        self._check(self._lib.sono_copy_to_host(self._ctx, F.ptr(arr), C.c_void_p(ptr), arr.nbytes))
        return arr

    def copy_to_device(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._check(self._lib.sono_copy_to_device(self._ctx, C.c_void_p(ptr), F.ptr(arr), arr.nbytes))

    def resample_cols_device(self, in_ptr, in_frames, bins, out_width, col0, ncols, out_ptr):
        """Columns [col0, col0 + ncols) of resample's out_width-column result, device to device."""
        self._check(self._lib.sono_resample_cols(self._ctx, C.c_void_p(in_ptr), int(in_frames), int(bins), int(out_width),
                                                 int(col0), int(ncols), C.c_void_p(out_ptr), F.MEM_DEVICE))

    def canvas_draw_device(self, cols_ptr, width, bins, sy, sh, dy, dh, height, background, out_ptr):
        bg = None if background is None else np.ascontiguousarray(background, np.uint8).reshape(4)
        self._check(self._lib.sono_canvas_draw(self._ctx, C.c_void_p(cols_ptr), int(width), int(bins), int(sy), int(sh), float(dy),
                                               float(dh), int(height), None if bg is None else F.ptr(bg), C.c_void_p(out_ptr),
                                               F.MEM_DEVICE))

    def pcm_to_device(self, raw, format_tag, bits, channels=1, channel=0):
        """Interleaved PCM bytes (host) -> float32 samples of one channel in device memory.  Returns
        (device pointer, frames); free it with device_free."""
        raw = np.ascontiguousarray(np.frombuffer(raw, np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8))
        frames = raw.size // (int(channels) * (int(bits) // 8))
        dst = self.device_malloc(max(4, frames * 4))
        rc = self._lib.sono_pcm_to_f32(self._ctx, F.ptr(raw), F.MEM_HOST, int(format_tag), int(bits), int(channels), int(channel),
                                       frames, C.c_void_p(dst))
        if rc != F.OK:
            self.device_free(dst)
            self._check(rc)
        return dst, frames

    def stft_wav(self, data, channel=0):
        """One .wav file (bytes) -> this context's output: chunk walk on the host, raw PCM over PCIe,
        conversion and transform on the device.  Contexts for this call take float32 input."""
        from . import intake
        if self.cfg.in_kind != F.IN_F32:
            raise ValueError("stft_wav converts to float32 on the device: create the context with in_kind = IN_F32")
        raw, w = intake.pcm_samples(data)
        dpcm, frames = self.pcm_to_device(raw, w.format_tag, w.bits_per_sample, w.channels, channel)
        try:
            out = self._out_array(frames)
            dout = self.device_malloc(max(4, out.nbytes))
            try:
                self.stft_device(dpcm, frames, dout, out.nbytes)
                if out.nbytes:
                    self.copy_to_host(out, dout)
            finally:
                self.device_free(dout)
        finally:
            self.device_free(dpcm)
        return out, w

    def resample(self, frames, out_width):
        frames = np.ascontiguousarray(frames, np.uint8)
        Fr, B = frames.shape
        out = np.empty((int(out_width), B), np.uint8)
        self._check(self._lib.sono_resample(self._ctx, F.ptr(frames), Fr, B, int(out_width), F.ptr(out), F.MEM_HOST))
        return out

    def colorize(self, frames, row0=0, rows=None):
        frames = np.ascontiguousarray(frames, np.uint8)
        W, B = frames.shape
        rows = B - row0 if rows is None else rows
        out = np.empty((rows, W, 4), np.uint8)
        self._check(self._lib.sono_colorize(self._ctx, F.ptr(frames), W, B, int(row0), int(rows), F.ptr(out), F.MEM_HOST))
        return out

    def canvas_draw(self, frames, sy, sh, dy, dh, height, background=None):
        """[width][bins] uint8 columns -> [height][width][4] canvas: LUT + flip, rows [sy, sy + sh) stretched
        over canvas rows [dy, dy + dh) (spectrogram.esm.js:357-381 with the engine's bilinear filter)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        W, B = frames.shape
        out = np.empty((int(height), W, 4), np.uint8)
        bg = None if background is None else np.ascontiguousarray(background, np.uint8).reshape(4)
        self._check(self._lib.sono_canvas_draw(self._ctx, F.ptr(frames), W, B, int(sy), int(sh), float(dy), float(dh),
                                               int(height), None if bg is None else F.ptr(bg), F.ptr(out), F.MEM_HOST))
        return out

    def frame_peaks(self, idx):
        idx = np.ascontiguousarray(idx, np.uint8)
        Fr, B = idx.shape
        pb, pv = np.empty(Fr, np.int32), np.empty(Fr, np.uint8)
        self._check(self._lib.sono_frame_peaks(self._ctx, F.ptr(idx), Fr, B, F.ptr(pb), F.ptr(pv), F.MEM_HOST))
        return pb, pv

    def measure_valu_peak(self):
        """10^9 wavefront-level v_fma_f32 per second this GPU issues at 16 wavefronts per CU (sono_measure_valu_peak)."""
        return float(self._lib.sono_measure_valu_peak(self._ctx))

    # -- timing with HIP events on the context's stream
    def timer_start(self):
        self._check(self._lib.sono_timer_start(self._ctx))

    def timer_stop_ms(self):
        ms = self._lib.sono_timer_stop_ms(self._ctx)
        if ms < 0:
            raise F.SonoError(F.ERR_DEVICE, self._lib.sono_last_error(self._ctx).decode())
        return ms


# ---- host-side rules (no GPU) ------------------------------------------------------------------
def worker_hop(fft_n, overlap_pct):
    return int(F.lib().sono_worker_hop(int(fft_n), float(overlap_pct)))


def plugin_noverlap(fft_n, noverlap, n_samples, canvas_width):
    return int(F.lib().sono_plugin_noverlap(int(fft_n), int(noverlap or 0), int(n_samples), int(canvas_width)))


def overlap_from_percent(fft_n, pct):
    return int(F.lib().sono_overlap_from_percent(int(fft_n), float(pct)))


def geometry_for(fft_n, hop, frame_rule, n_samples, bin_lo=0, bin_hi=0):
    g = F.SonoGeometry()
    rc = F.lib().sono_geometry_for(int(fft_n), int(hop), int(frame_rule), int(bin_lo), int(bin_hi), int(n_samples),
                                   C.byref(g))
    if rc != F.OK:
        raise F.SonoError(rc, "bad geometry arguments")
    return g


def filterbank_table(scale, fft_n, sample_rate, filters=None):
    """createFilterBank (spectrogram.esm.js:389-411) as (first tap, weight of it, weight of the next bin)."""
    filters = fft_n // 2 if filters is None else int(filters)
    b, w0, w1 = np.empty(filters, np.int32), np.empty(filters, np.float64), np.empty(filters, np.float64)
    rc = F.lib().sono_filterbank_table(F.FREQ_SCALES[scale], int(fft_n), float(sample_rate), filters, F.ptr(b), F.ptr(w0), F.ptr(w1))
    if rc != F.OK:
        raise F.SonoError(rc, "bad filter bank arguments")
    return b, w0, w1


def hz_to_scale(scale, hz):
    return float(F.lib().sono_hz_to_scale(F.FREQ_SCALES[scale], float(hz)))


def scale_to_hz(scale, v):
    return float(F.lib().sono_scale_to_hz(F.FREQ_SCALES[scale], float(v)))


def plugin_crop_scaled(bins, sample_rate, fmin_hz, fmax_hz, scale):
    sy, sh = C.c_int64(0), C.c_int64(0)
    rc = F.lib().sono_plugin_crop_scaled(int(bins), float(sample_rate), float(fmin_hz), float(fmax_hz), F.FREQ_SCALES[scale],
                                         C.byref(sy), C.byref(sh))
    if rc != F.OK:
        raise F.SonoError(rc, "bad crop arguments")
    return sy.value, sh.value


def plugin_draw_rect(sample_rate, fmin_hz, fmax_hz, scale, height):
    """(dy, dh, fill_background) of drawSpectrogram's drawImage call, spectrogram.esm.js:357,377-381."""
    dy, dh, fill = C.c_double(0), C.c_double(0), C.c_int(0)
    rc = F.lib().sono_plugin_draw_rect(float(sample_rate), float(fmin_hz), float(fmax_hz), F.FREQ_SCALES[scale], int(height),
                                       C.byref(dy), C.byref(dh), C.byref(fill))
    if rc != F.OK:
        raise F.SonoError(rc, "bad draw_rect arguments")
    return dy.value, dh.value, bool(fill.value)


def plugin_background(colormap_last):
    """The fill colour of spectrogram.esm.js:358-360 as four bytes."""
    last = np.ascontiguousarray(colormap_last, np.float64).reshape(4)
    out = np.empty(4, np.uint8)
    rc = F.lib().sono_plugin_background(F.ptr(last), F.ptr(out))
    if rc != F.OK:
        raise F.SonoError(rc, "bad background arguments")
    return out


def plugin_crop(bins, sample_rate, fmin_hz, fmax_hz):
    sy, sh = C.c_int64(0), C.c_int64(0)
    rc = F.lib().sono_plugin_crop(int(bins), float(sample_rate), float(fmin_hz), float(fmax_hz), C.byref(sy), C.byref(sh))
    if rc != F.OK:
        raise F.SonoError(rc, "bad crop arguments")
    return sy.value, sh.value


def plugin_lut(colormap):
    cm = np.ascontiguousarray(colormap, np.float64).reshape(1024)
    lut = np.empty(1024, np.uint8)
    F.lib().sono_plugin_lut(F.ptr(cm), F.ptr(lut))
    return lut


def worker_lut(colormap):
    cm = np.ascontiguousarray(colormap, np.float64).reshape(1024)
    lut = np.empty(1024, np.uint8)
    F.lib().sono_worker_lut(F.ptr(cm), F.ptr(lut))
    return lut


def segment_calls(features, threshold, min_frames=1, first_frame=0):
    """Maximal runs of frames with peak_idx >= threshold (host side of the feature pass)."""
    f = np.ascontiguousarray(features, F.FEATURE_DTYPE)
    n = int(F.lib().sono_segment_calls(F.ptr(f), len(f), int(threshold), int(min_frames), int(first_frame), None, 0))
    out = np.zeros(max(n, 0), F.CALL_DTYPE)
    if n > 0:
        F.lib().sono_segment_calls(F.ptr(f), len(f), int(threshold), int(min_frames), int(first_frame), F.ptr(out), n)
    return out


def window_table(window_id_, fft_n, alpha=0.0):
    out = np.empty(int(fft_n), np.float32)
    rc = F.lib().sono_window_table(int(window_id_), int(fft_n), float(alpha or 0.0), F.ptr(out))
    if rc != F.OK:
        raise F.SonoError(rc, "No such window function '%s'" % window_id_)
    return out


def twiddle_table(mode, fft_n):
    c, s = np.zeros(int(fft_n), np.float64), np.zeros(int(fft_n), np.float64)
    rc = F.lib().sono_twiddle_table(int(mode), int(fft_n), F.ptr(c), F.ptr(s))
    if rc != F.OK:
        raise F.SonoError(rc, "bad twiddle arguments")
    return c[:fft_n - 1], s[:fft_n - 1]
