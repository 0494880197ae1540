"""sono_stft_static -- the entry point that allocates nothing -- against sono_stft_async, through the C ABI.

Two calls in a row on one context, the second with a smaller input (the static index plane keeps the first call's
bytes), in both arithmetics and on both reference paths; then the size limit.  Bar: the image must equal the
existing entry point's byte for byte (both run the same index kernels and the same colour-map kernel), and -- on the
worker path -- the CPU checker's render of the same samples (bit-identical in the exact arithmetic, within one colour
index in float32).  The free device memory must not move across a warm call, and a call that arrives on a second
stream while the first one's kernels still run must be refused, not run.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
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
            if path == "worker":   # ... and the checker's render (spectrogramWorker.js:38-99) of the same samples
                ref, idx = oracle.worker_render(_signal(L, L), 1024, 75.0, lut, want_index=True)
                if arith == F.ARITH_EXACT_F64:
                    assert np.array_equal(b, ref), L
                else:               # float32: every pixel is the colour of the checker's index or of a neighbouring one
                    lut4 = lut.reshape(256, 4)
                    at = idx[: b.shape[1]].T[::-1].astype(np.int64)            # [bin row][column] (column 0 may hold the
                    near = [(b == lut4[np.clip(at + d, 0, 255)]).all(-1) for d in (0, 1, -1)]   # worker's wrapped frame)
                    assert (near[0] | near[1] | near[2])[:, 1:].all(), L
                    assert near[0][:, 1:].mean() > 0.98


def test_static_entry_moves_no_memory_and_refuses_overlap():
    torch = pytest.importorskip("torch")
    lut = (np.arange(1024, dtype=np.uint32) * 11 % 253).astype(np.uint8)
    kw = dict(fft_n=1024, hop=256, window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5,
              frame_rule=F.FRAMES_WORKER_COMPAT, twiddle_mode=F.TW_ACCURATE, out_kind=F.OUT_RGBA8)
    L = 1024 + 256 * 60000                      # just under the static plane's 65 536 frames: a launch of about 0.1 ms
    x = _signal(L, 5)
    with E.Engine(**kw) as e1, E.Engine(**kw) as e2:
        e1.set_colormap(lut)
        e2.set_colormap(lut)
        nbytes = e1.out_bytes(L)
        dpcm, d1, d2 = e1.device_malloc(x.nbytes), e1.device_malloc(nbytes), e1.device_malloc(nbytes)
        try:
            e1.copy_to_device(dpcm, x)
            e1.stft_static_device(dpcm, L, d1, nbytes)               # warm: code objects loaded, nothing left to set up
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
            for _ in range(3):
                e1.stft_static_device(dpcm, L, d1, nbytes, wait=False)
            free1 = torch.cuda.mem_get_info()[0]                     # (hipMemGetInfo, while the calls are in flight)
            e1.wait()
            assert free1 == free0 == torch.cuda.mem_get_info()[0]
            # two contexts = two streams, one GPU, one static plane: the second stream's call is refused while the
            # first stream's kernels run, and accepted once they are done
            busy = 0
            for _ in range(20):
                e1.stft_static_device(dpcm, L, d1, nbytes, wait=False)
                rc = e2._lib.sono_stft_static(e2._ctx, C.c_void_p(dpcm), L, C.c_void_p(d2), nbytes)
                if rc == F.ERR_BUSY:
                    busy += 1
                    assert b"in use" in e2._lib.sono_last_error(e2._ctx)
                else:
                    assert rc == F.OK
                    e2.wait()
                e1.wait()
            assert busy >= 1
            e2.stft_static_device(dpcm, L, d2, nbytes)               # nothing in flight: accepted
            a, b = np.empty(nbytes, np.uint8), np.empty(nbytes, np.uint8)
            e1.copy_to_host(a, d1)
            e2.copy_to_host(b, d2)
            assert np.array_equal(a, b)
        finally:
            for p in (dpcm, d1, d2):
                e1.device_free(p)


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
