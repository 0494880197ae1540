"""ctypes binding of libsonospec.so (the C ABI declared in include/sonospec.h).

The library is the product: if it is missing or does not load, importing this module raises --
there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsonospec.so")

# enums of include/sonospec.h
OK = 0
ERR_ARG, ERR_WINDOW, ERR_FFT_SIZE, ERR_COLORMAP, ERR_CAPACITY, ERR_DEVICE, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6, -7
ERR_FORMAT = -8
ERR_BUSY = -9
WINDOW_IDS = {
    "bartlett": 0, "bartlettHann": 1, "blackman": 2, "cosine": 3, "gauss": 4,
    "hamming": 5, "hann": 6, "lanczoz": 7, "rectangular": 8, "triangular": 9,
}
WIN_WORKER_HANN = 10
SCALE_WORKER_LOG5, SCALE_PLUGIN_DB = 0, 1
FRAMES_PLUGIN_STRICT, FRAMES_WORKER_COMPAT = 0, 1
TW_ACCURATE, TW_PLUGIN_RECURRENCE, TW_HOST_TABLE = 0, 1, 2
ARITH_FAST_F32, ARITH_EXACT_F64 = 0, 1
OUT_U8_INDEX, OUT_RGBA8, OUT_F32_DB, OUT_F32_MAG, OUT_FEATURES = 0, 1, 2, 3, 4
IN_F32, IN_I16 = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
STATIC_SCRATCH_BYTES = 32 << 20
FREQ_SCALES = {"linear": 0, "mel": 1, "logarithmic": 2, "bark": 3, "erb": 4}


class SonoConfig(C.Structure):
    _fields_ = [
        ("fft_n", C.c_uint32), ("hop", C.c_uint32), ("window_id", C.c_uint32), ("scale_mode", C.c_uint32),
        ("frame_rule", C.c_uint32), ("twiddle_mode", C.c_uint32), ("arith", C.c_uint32), ("out_kind", C.c_uint32),
        ("in_kind", C.c_uint32), ("bin_lo", C.c_uint32), ("bin_hi", C.c_uint32), ("reserved0", C.c_uint32),
        ("alpha", C.c_double), ("gain_db", C.c_double), ("range_db", C.c_double), ("reserved", C.c_uint32 * 4),
    ]


FEATURE_DTYPE = np.dtype([("peak_bin", np.int32), ("peak_idx", np.uint8), ("reserved", np.uint8, (3,))])
CALL_DTYPE = np.dtype([("start_frame", np.uint64), ("end_frame", np.uint64), ("peak_frame", np.uint64),
                       ("start_bin", np.int32), ("end_bin", np.int32), ("high_bin", np.int32), ("low_bin", np.int32),
                       ("peak_bin", np.int32), ("peak_idx", np.uint32)])


SHAPE_DTYPE = np.dtype([("high_frame", np.int64), ("low_frame", np.int64), ("knee_frame", np.int64), ("heel_frame", np.int64),
                        ("cf_start_frame", np.int64), ("cf_end_frame", np.int64), ("knee_bin", np.int32), ("heel_bin", np.int32),
                        ("cf_start_bin", np.int32), ("cf_end_bin", np.int32)])
AUTOID_RULE_DTYPE = np.dtype([("call_type_mask", np.uint64), ("harmonic_first", np.uint32), ("harmonic_count", np.uint32),
                              ("cond_first", np.uint32), ("cond_count", np.uint32), ("species", np.uint32), ("reserved", np.uint32)])


class AutoidData(C.Structure):
    _fields_ = [("field", C.c_double * 14), ("call_type", C.c_int32), ("harmonic", C.c_int32)]


class SonoWavInfo(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_uint32), ("has_fmt", C.c_uint32), ("has_data", C.c_uint32), ("fmt_sample_rate", C.c_uint32),
        ("format_tag", C.c_uint16), ("channels", C.c_uint16), ("bits_per_sample", C.c_uint16), ("reserved0", C.c_uint16),
        ("block_align", C.c_uint32), ("fmt_bytes", C.c_uint32), ("fmt_offset", C.c_uint64), ("data_offset", C.c_uint64),
        ("data_bytes", C.c_uint64), ("data_bytes_present", C.c_uint64), ("frames", C.c_uint64), ("duration_s", C.c_double),
    ]


class SonoGuanoFields(C.Structure):
    _fields_ = [("present", C.c_int32), ("date", C.c_char * 64), ("time", C.c_char * 16), ("latitude", C.c_char * 64),
                ("longitude", C.c_char * 64)]


class SonoGeometry(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("width", C.c_uint64), ("bins", C.c_uint32), ("hop", C.c_uint32)]


# every symbol include/sonospec.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "sono_hz_to_scale": (C.c_double, [C.c_uint32, C.c_double]),
    "sono_scale_to_hz": (C.c_double, [C.c_uint32, C.c_double]),
    "sono_filterbank_table": (C.c_int, [C.c_uint32, C.c_uint32, C.c_double, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sono_plugin_crop_scaled": (C.c_int, [C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint32, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64)]),
    "sono_set_filterbank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sono_stft_host_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "sono_host_malloc": (C.c_void_p, [C.c_uint64]),
    "sono_host_free": (C.c_int, [C.c_void_p]),
    "sono_wav_parse": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(SonoWavInfo)]),
    "sono_wav_guano": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "sono_guano_fields": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(SonoGuanoFields)]),
    "sono_wav_crop": (C.c_int64, [C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_void_p, C.c_uint64]),
    "sono_wav_accept": (C.c_int, [C.c_uint64, C.c_double, C.c_int]),
    "sono_pcm_to_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                  C.c_void_p]),
    "sono_abi_version": (C.c_int, []),
    "sono_device_count": (C.c_int, []),
    "sono_device_numa_node": (C.c_int, [C.c_int]),
    "sono_worker_hop": (C.c_uint32, [C.c_uint32, C.c_double]),
    "sono_plugin_noverlap": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]),
    "sono_overlap_from_percent": (C.c_uint32, [C.c_uint32, C.c_double]),
    "sono_geometry_for": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                    C.POINTER(SonoGeometry)]),
    "sono_plugin_crop": (C.c_int, [C.c_uint32, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64)]),
    "sono_plugin_lut": (C.c_int, [_VP, _VP]),
    "sono_worker_lut": (C.c_int, [_VP, _VP]),
    "sono_segment_calls": (C.c_int64, [_VP, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, _VP, C.c_uint64]),
    "sono_autoid_match": (C.c_int, [_VP, C.c_uint32, _VP, _VP, _VP, C.POINTER(AutoidData), _VP, C.c_uint32]),
    "sono_call_shape": (C.c_int, [_VP, C.c_uint64, C.c_uint64, _VP, _VP]),
    "sono_autoid_data_of_call": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_int,
                                           C.POINTER(AutoidData)]),
    "sono_window_table": (C.c_int, [C.c_uint32, C.c_uint32, C.c_double, _VP]),
    "sono_twiddle_table": (C.c_int, [C.c_uint32, C.c_uint32, _VP, _VP]),
    "sono_create": (_VP, [C.c_int, C.POINTER(SonoConfig)]),
    "sono_create_error": (C.c_char_p, []),
    "sono_destroy": (None, [_VP]),
    "sono_last_error": (C.c_char_p, [_VP]),
    "sono_get_config": (C.c_int, [_VP, C.POINTER(SonoConfig)]),
    "sono_set_stream": (C.c_int, [_VP, _VP]),
    "sono_set_colormap": (C.c_int, [_VP, _VP]),
    "sono_set_tables": (C.c_int, [_VP, _VP, _VP, _VP]),
    "sono_geometry_of": (C.c_int, [_VP, C.c_uint64, C.POINTER(SonoGeometry)]),
    "sono_out_bytes": (C.c_uint64, [_VP, C.c_uint64]),
    "sono_stft": (C.c_int, [_VP, _VP, C.c_uint64, _VP, C.c_uint64, C.c_int]),
    "sono_stft_async": (C.c_int, [_VP, _VP, C.c_uint64, _VP, C.c_uint64]),
    "sono_wait": (C.c_int, [_VP]),
    "sono_stft_static": (C.c_int, [_VP, _VP, C.c_uint64, _VP, C.c_uint64]),
    "sono_kernel_route": (C.c_char_p, [_VP]),
    "sono_features_recomputed": (C.c_int64, [_VP]),
    "sono_reserve": (C.c_int, [_VP, C.c_uint64]),
    "sono_resample": (C.c_int, [_VP, _VP, C.c_uint64, C.c_uint32, C.c_uint64, _VP, C.c_int]),
    "sono_resample_cols": (C.c_int, [_VP, _VP, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, _VP, C.c_int]),
    "sono_colorize": (C.c_int, [_VP, _VP, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _VP, C.c_int]),
    "sono_frame_peaks": (C.c_int, [_VP, _VP, C.c_uint64, C.c_uint32, _VP, _VP, C.c_int]),
    "sono_canvas_draw": (C.c_int, [_VP, _VP, C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_uint32,
                                   _VP, _VP, C.c_int]),
    "sono_plugin_draw_rect": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "sono_plugin_background": (C.c_int, [_VP, _VP]),
    "sono_device_malloc": (_VP, [_VP, C.c_uint64]),
    "sono_device_free": (C.c_int, [_VP, _VP]),
    "sono_copy_to_device": (C.c_int, [_VP, _VP, _VP, C.c_uint64]),
    "sono_copy_to_host": (C.c_int, [_VP, _VP, _VP, C.c_uint64]),
    "sono_measure_valu_peak": (C.c_double, [_VP]),
    "sono_timer_start": (C.c_int, [_VP]),
    "sono_timer_stop_ms": (C.c_double, [_VP]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch wheels carry their own libamdhip64 (same SONAME as the
    system one); if libsonospec.so pulled in the system copy first, a later `import torch` would load
    a second runtime and find no GPU.  When torch is installed, its copy is loaded first so that both
    resolve to it.  Without torch (Node addon, C++ hosts) the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libsonospec.so (once).  Raises if the library is absent: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libsonospec.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C spectrogram_amd/csrc`" % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.sono_abi_version() != 1:
            raise ImportError("libsonospec.so ABI version mismatch")
        _lib = L
    return _lib


class SonoError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def ptr(a):
    """Raw address of a numpy array (or pass an int device pointer through)."""
    if a is None:
        return None
    if isinstance(a, (int,)):
        return C.c_void_p(a)
    return C.c_void_p(a.ctypes.data)
