"""sono_stft_static -- the entry point that allocates nothing -- against sono_stft_async, through the C ABI.

Two calls in a row on one context, the second with a smaller input (the static index plane keeps the first call's
bytes), in both arithmetics and on both reference paths; then the size limit.  Bar: the image must equal the
existing entry point's byte for byte (both run the same index kernels and the same colour-map kernel).
"""
import ctypes as C

import numpy as np
import pytest

from spectrogram_amd import _ffi as F
from spectrogram_amd import engine as E

pytestmark = pytest.mark.gpu


def _signal(L, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(L) * 0.05).astype(np.float32)
    t = np.arange(L, dtype=np.float64)
    x += (0.4 * np.sin(0.9 * t - 2e-6 * t * t)).astype(np.float32)
    return x


def _both(eng, x):
    """(image from sono_stft_async, image from sono_stft_static) for host samples x, device buffers of the test's own."""
    L = len(x)
    nbytes = eng.out_bytes(L)
    g = eng.geometry(L)
    dpcm = eng.device_malloc(max(4, x.nbytes))
    da = eng.device_malloc(max(4, nbytes))
    db = eng.device_malloc(max(4, nbytes))
    try:
        eng.copy_to_device(dpcm, x)
        a = np.full((g.bins, g.width, 4), 0xA5, np.uint8)
        b = np.full((g.bins, g.width, 4), 0x5A, np.uint8)
        eng.copy_to_device(da, a)
        eng.copy_to_device(db, b)
        eng.stft_device(dpcm, L, da, nbytes)
        route_a = eng.kernel_route()
        eng.stft_static_device(dpcm, L, db, nbytes)
        route_b = eng.kernel_route()
        eng.copy_to_host(a, da)
        eng.copy_to_host(b, db)
    finally:
        for p in (dpcm, da, db):
            eng.device_free(p)
    return a, b, route_a, route_b


@pytest.mark.parametrize("arith", [F.ARITH_FAST_F32, F.ARITH_EXACT_F64], ids=["fast", "exact"])
@pytest.mark.parametrize("path", ["worker", "plugin_crop"])
def test_static_entry_equals_async_twice(arith, path):
    lut = (np.arange(1024, dtype=np.uint32) * 11 % 253).astype(np.uint8)
    if path == "worker":
        kw = dict(fft_n=1024, hop=256, window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5,
                  frame_rule=F.FRAMES_WORKER_COMPAT, twiddle_mode=F.TW_ACCURATE)
    else:   # the viewer's default band at 384 kHz (10-128 kHz = bins 26..340): not made of four-bin groups
        kw = dict(fft_n=1024, hop=256, window_id=F.WINDOW_IDS["hann"], scale_mode=F.SCALE_PLUGIN_DB,
                  frame_rule=F.FRAMES_PLUGIN_STRICT, twiddle_mode=F.TW_PLUGIN_RECURRENCE, bin_lo=26, bin_hi=341)
    with E.Engine(arith=arith, out_kind=F.OUT_RGBA8, **kw) as eng:
        eng.set_colormap(lut)
        # (L - N) % hop == 0 on the first call: the worker's wrapped last frame is part of it
        for L in (1024 + 256 * 700, 1024 + 256 * 123 + 77):
            a, b, ra, rb = _both(eng, _signal(L, L))
            assert a.shape == b.shape and np.array_equal(a, b), (L, ra, rb)
            assert "colorize_kernel" in rb, rb          # the static route is the two-kernel one
            assert (a != 0xA5).any()


def test_static_entry_limits():
    with E.Engine(fft_n=1024, hop=256, window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5,
                  frame_rule=F.FRAMES_WORKER_COMPAT, out_kind=F.OUT_RGBA8) as eng:
        # frames x bins beyond the static plane: rejected before anything is launched (pointers are never touched)
        L = 1024 + 256 * (F.STATIC_SCRATCH_BYTES // 512 + 1)
        rc = eng._lib.sono_stft_static(eng._ctx, C.c_void_p(0x1000), L, C.c_void_p(0x1000), eng.out_bytes(L))
        assert rc == F.ERR_CAPACITY and b"static index plane" in eng._lib.sono_last_error(eng._ctx)
    with E.Engine(fft_n=1024, hop=256, out_kind=F.OUT_U8_INDEX) as eng:
        rc = eng._lib.sono_stft_static(eng._ctx, C.c_void_p(0x1000), 4096, C.c_void_p(0x1000), 1 << 20)
        assert rc == F.ERR_ARG
