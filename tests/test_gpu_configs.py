"""The benchmark configurations at their own sizes, against the CPU checker.

The checker cannot transform an hour of audio in seconds, so each test pins a head (and a tail) of the result
to the checker and carries that to the full size through a size-independent property: a periodic signal gives
periodic results; the image is the colour map applied to the indices the index output of the same arithmetic
gives; spans of a recording reproduce the whole.  Tolerances as everywhere: colour indices within 1 LSB in the
float32 arithmetic, integer feature records bit-exact.
"""
import numpy as np
import pytest

import oracle
from conftest import golden_store
from spectrogram_amd import _ffi as F
from spectrogram_amd import engine as E
from spectrogram_amd import sharding

pytestmark = pytest.mark.gpu
G = golden_store()


def plugin_engine(n, hop, arith=F.ARITH_FAST_F32, out_kind=F.OUT_U8_INDEX, **kw):
    return E.Engine(fft_n=n, hop=hop, window_id=F.WINDOW_IDS["hann"], gain_db=20.0, range_db=80.0, scale_mode=F.SCALE_PLUGIN_DB,
                    frame_rule=F.FRAMES_PLUGIN_STRICT, twiddle_mode=F.TW_PLUGIN_RECURRENCE, arith=arith, out_kind=out_kind, **kw)


def worker_engine(n, hop, arith=F.ARITH_FAST_F32, out_kind=F.OUT_RGBA8, **kw):
    return E.Engine(fft_n=n, hop=hop, window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5,
                    frame_rule=F.FRAMES_WORKER_COMPAT, twiddle_mode=F.TW_ACCURATE, arith=arith, out_kind=out_kind, **kw)


def within_one_step(img, want_idx, lut):
    """Every pixel is the colour of the checker's index or of a neighbouring one (cyclic at the plugin's wrap)."""
    l4 = lut.reshape(256, 4)
    w = want_idx.astype(np.int16)
    ok = np.zeros(w.shape, bool)
    for dlt in (-1, 0, 1):
        ok |= np.all(img == l4[(w + dlt) % 256], axis=-1)
    return ok.all(), np.all(img == l4[w], axis=-1).mean()


# ------------------------------------------------------------------ FFT 4096, 93.75 % overlap, 15-120 kHz, RGBA
def test_fft4096_band_crop_rgba_fast_against_checker():
    """FFT 4096, hop 256, bins 160..1279 (15-120 kHz at 384 kHz), float32 arithmetic, RGBA through the default
    colour map: the route the benchmark variant times.  On the reference's bat signal and on a seeded signal of
    2 100 frames every pixel must be the colour of the checker's index +-1, and >= 99 % equal."""
    n, hop, lo, hi = 4096, 256, 160, 1280
    sy, sh = E.plugin_crop(n // 2, 384000, 15000, 120000)
    assert (n // 2 - sy - sh, n // 2 - sy) == (lo, hi)
    lut = E.plugin_lut(G.colormap("default"))
    rng = np.random.default_rng(4096)
    L = n + hop * 2100 + 5
    t = np.arange(L, dtype=np.float64)
    seeded = (rng.standard_normal(L) * 0.02).astype(np.float32)
    seeded += (0.5 * np.sin(0.35 * t + 3e-7 * t * t) * (np.sin(2e-4 * t) > 0)).astype(np.float32)
    for x in (G.signal("bat"), seeded):
        ref = oracle.plugin_get_frequencies(x, n, n - hop)
        with plugin_engine(n, hop, out_kind=F.OUT_RGBA8, bin_lo=lo, bin_hi=hi) as eng:
            eng.set_colormap(lut)
            img = eng.stft(x)
            route = eng.kernel_route()
        assert "4096" in route and "general" not in route, route
        want = ref[:, lo:hi].T[::-1]
        assert img.shape == want.shape + (4,)
        ok, same = within_one_step(img, want, lut)
        assert ok and same >= 0.99, (route, same)


@pytest.mark.gpu
@pytest.mark.parametrize("hop,in_kind", [(256, "f32"), (1024, "i16"), (204, "f32"), (1022, "f32")])
def test_fft4096_kernel_indices_and_image(hop, in_kind):
    """The 4096-point kernel (one wavefront per frame: complex 2048-point transform of even/odd samples + the reference's
    last stage): [frame][bin] indices within 1 LSB of the checker; the image it writes itself (rounds of 32 frames at
    this length; test_fft4096_image_rounds_of_64_frames covers the long-recording flavour), with
    a band that starts and ends inside a group of four bins, equals the colour map applied to those indices.  A hop that
    is not a multiple of four samples cannot take that kernel's 16-byte sample loads; the three-pass kernel would serve
    it, but with two frames per transform the recurrence factors at 4096 points let a full-scale frame leave a trace
    of -96 dB in its partner, above the display floor: such a context takes that kernel's one-frame-per-transform
    flavour (same indices within 1 LSB)."""
    n = 4096
    rng = np.random.default_rng(7000 + hop)
    L = n + hop * 700 + 3
    x = (rng.standard_normal(L) * 0.1).astype(np.float32)
    t = np.arange(L, dtype=np.float64)
    x += (0.4 * np.sin(0.9 * t + 2e-6 * t * t)).astype(np.float32)
    kw, src = {}, x
    if in_kind == "i16":
        src = np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)
        x = (src.astype(np.float32) / 32768.0).astype(np.float32)
        kw["in_kind"] = F.IN_I16
    ref = oracle.plugin_get_frequencies(x, n, n - hop)
    with plugin_engine(n, hop, **kw) as eng:
        got = eng.stft(src)
        route = eng.kernel_route()
    # (the kernel splits a frame into its even and odd samples; under the recurrence table that split is safe only below a
    # sample level, so it transforms the even and the odd samples of a loud frame apart -- this signal is loud)
    assert route == ("stft4096_kernel<redo>" if hop % 4 == 0 else "stft_mw_kernel<4096,single>"), route
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    d = np.minimum(d, 256 - d)
    assert got.shape == ref.shape and d.max() <= 1 and (d != 0).mean() < 0.02
    lut = (np.arange(1024, dtype=np.uint32) * 11 % 256).astype(np.uint8)
    for lo, hi in ((161, 1279), (0, 2048), (1030, 1034)):
        with plugin_engine(n, hop, out_kind=F.OUT_RGBA8, bin_lo=lo, bin_hi=hi, **kw) as eng:
            eng.set_colormap(lut)
            img = eng.stft(src)
            route = eng.kernel_route()
        want = lut.reshape(256, 4)[got[:, lo:hi].T[::-1]]
        if hop % 4 == 0:
            assert route == "stft4096_kernel<image,redo>:rgba", route
            assert img.shape == want.shape and np.array_equal(img, want)          # same arithmetic, same indices
        else:
            assert route == "stft_mw_kernel<4096,single>:u8 + colorize_kernel", route
            assert img.shape == want.shape and np.array_equal(img, want)


def test_fft4096_image_rounds_of_64_frames():
    """A recording long enough that the 4096-point image writer takes rounds of 64 frames (every workgroup gets at
    least 16 of them): the image of a band equals the colour map applied to the index output of the same frames, the
    indices of a few stretches of frames are within 1 LSB of the checker, and the last, partial round is written."""
    n, hop = 4096, 256
    cus = 256
    frames = 64 * 16 * cus + 37
    L = n + hop * (frames - 1) + 9
    rng = np.random.default_rng(4096)
    x = (rng.standard_normal(L) * 0.003).astype(np.float32)
    t = np.arange(hop * 4000, dtype=np.float64)
    call = (0.3 * np.sin(0.9 * t + 2e-6 * t * t)).astype(np.float32)
    for start in (0, L // 3, L - call.size):       # three loud stretches, one at each end
        x[start:start + call.size] += call
    lo, hi = 161, 291
    lut = (np.arange(1024, dtype=np.uint32) * 11 % 256).astype(np.uint8)
    # (the index output takes whole groups of four bins; the image band starts and ends inside one)
    with plugin_engine(n, hop, bin_lo=lo - 1, bin_hi=hi + 1) as e8, plugin_engine(n, hop, out_kind=F.OUT_RGBA8, bin_lo=lo, bin_hi=hi) as er:
        er.set_colormap(lut)
        idx, img = e8.stft(x)[:, 1:-1], er.stft(x)
        assert er.kernel_route() == "stft4096_kernel<image,redo>:rgba"
    assert idx.shape == (frames, hi - lo) and img.shape == (hi - lo, frames, 4)
    assert np.array_equal(img, lut.reshape(256, 4)[idx.T[::-1]])
    for f0 in (0, 3000, frames // 2, frames - 60):
        ref = oracle.plugin_get_frequencies(x[f0 * hop:(f0 + 60) * hop + n], n, n - hop)[:, lo:hi]
        assert ref.shape[0] >= 59
        d = np.abs(idx[f0:f0 + ref.shape[0]].astype(np.int16) - ref.astype(np.int16))
        assert np.minimum(d, 256 - d).max() <= 1, f0


@pytest.mark.parametrize("n", [1024, 512])
def test_image_writer_sector_alignment_cases(n):
    """The 1024- and 512-point image writers cut every row's runs at 64-byte boundaries of the output, which depend on the
    output address, the image width and the row: images of every width residue mod 16 (the pitch in pixels), written at
    every 4-byte offset inside a sector, with enough frames for several rounds per workgroup's neighbour and a band that
    is not a whole number of quads, must equal the colour map applied to the [frame][bin] indices of the same arithmetic;
    nothing may be written outside the image."""
    rng = np.random.default_rng(64)
    lut = (np.arange(1024, dtype=np.uint32) * 13 % 256).astype(np.uint8)
    hop, rf, H = n // 4, 128 * 1024 // n, n // 2     # frames per round: 128 at 1024 points, 256 at 512
    # (rounds per workgroup: one in the small cases, so that every run ends at a range boundary; four in the large ones,
    # so that most runs take the unmasked path)
    for case, (extra, off_px, lo, hi, rounds) in enumerate([(0, 0, 0, H, 9), (1, 1, 0, H, 9), (5, 7, 26, H - 170, 9), (9, 15, 0, H, 9),
                                                            (14, 3, 100, 101, 9), (3, 8, 0, H, 9), (11, 13, 4, H - 4, 9),
                                                            (7, 5, 0, H, 1024 - 3), (10, 11, 30, H - 177, 1024 - 50)]):
        frames = rounds * rf + 40 + extra           # widths of different residues mod 16
        L = n + hop * (frames - 1) + 1 + (hop - 2 if case % 2 else 0)
        x = (rng.standard_normal(L) * 0.2).astype(np.float32)
        with worker_engine(n, hop, out_kind=F.OUT_U8_INDEX) as eng:
            idx = eng.stft(x)
        with worker_engine(n, hop, bin_lo=lo, bin_hi=hi) as eng:
            eng.set_colormap(lut)
            g = eng.geometry(len(x))
            need = eng.out_bytes(len(x))
            pad = 256
            d_in = eng.device_malloc(x.nbytes)
            d_out = eng.device_malloc(need + 2 * pad)
            try:
                eng.copy_to_device(d_in, x)
                guard = np.full(need + 2 * pad, 0xA5, np.uint8)
                eng.copy_to_device(d_out, guard)
                eng.stft_device(d_in, len(x), d_out + pad + 4 * off_px, need)
                assert eng.kernel_route().startswith("stft%d_rgba_kernel" % n), eng.kernel_route()
                got = np.empty(need + 2 * pad, np.uint8)
                eng.copy_to_host(got, d_out)
            finally:
                eng.device_free(d_in)
                eng.device_free(d_out)
        start = pad + 4 * off_px
        img = got[start:start + need].reshape(hi - lo, g.width, 4)
        want = lut.reshape(256, 4)[idx[:g.width, lo:hi].T[::-1]]
        assert idx.shape[0] == g.width == frames, (idx.shape, g.width, frames)
        assert np.array_equal(img, want), case
        assert (got[:start] == 0xA5).all() and (got[start + need:] == 0xA5).all(), case


def test_fft4096_band_crop_rgba_at_benchmark_size():
    """The same route at the benchmark's size (60 s x 384 kHz: 89 984 frames, 100 MB of indices = several chunks of
    the index buffer): the image must be the colour map applied to the [frame][bin] indices the same arithmetic
    writes (flipped, transposed) everywhere; a signal of period 64 hops gives columns of period 64; the head
    equals the checker's within one step."""
    torch = pytest.importorskip("torch")
    n, hop, lo, hi, pf = 4096, 256, 160, 1280, 64
    period = hop * pf
    L = 60 * 384000
    rng = np.random.default_rng(41)
    base = (rng.standard_normal(period) * 0.05).astype(np.float32)
    base[2000:9000] += (0.5 * np.sin(np.arange(7000) * 0.6 + 2e-5 * np.arange(7000) ** 2)).astype(np.float32)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(base).to(dev).repeat(L // period + 1)[:L].contiguous()
    lut = E.plugin_lut(G.colormap("default"))
    with plugin_engine(n, hop, out_kind=F.OUT_RGBA8, bin_lo=lo, bin_hi=hi) as eng, \
            plugin_engine(n, hop, out_kind=F.OUT_U8_INDEX, bin_lo=lo, bin_hi=hi) as eng8:
        eng.set_colormap(lut)
        g = eng.geometry(L)
        assert (g.frames, g.bins) == (89984, 1120)
        img = torch.empty((g.bins, g.width, 4), dtype=torch.uint8, device=dev)
        idx = torch.empty((g.frames, g.bins), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.stft_device(x.data_ptr(), L, img.data_ptr(), img.numel())
        eng8.stft_device(x.data_ptr(), L, idx.data_ptr(), idx.numel())
        px = img.view(torch.int32).view(g.bins, g.width)
        lut32 = torch.from_numpy(lut.view(np.int32).copy()).to(dev)
        assert bool((px == lut32[idx.long()].t().flip(0)).all())
        whole = (g.frames // pf) * pf
        v = idx[:whole].view(-1, pf, g.bins)
        assert bool((v == v[0:1]).all())
        head = img[:, :pf].cpu().numpy()
    ref = oracle.plugin_get_frequencies(np.tile(base, 2)[:period + n + 1], n, n - hop)[:pf]
    ok, same = within_one_step(head, ref[:, lo:hi].T[::-1], lut)
    assert ok and same >= 0.99


# ------------------------------------------------------------------ the feature pass on the 1 h recording
def test_feature_pass_at_recording_size_and_span_merge():
    """1 h x 384 kHz, FFT 1024, hop 256, SONO_OUT_FEATURES in the float32 arithmetic (with its exact fix-up pass):
    records of a period-64-hops signal repeat every 64 frames and the first 64 are the checker's, bit for bit;
    segmenting all 5.4 M records gives what the periodic structure predicts; the recording cut into 8 spans
    (split_recording), each transformed on its own and segmented with min_frames = 1, then merged with the real
    minimum (merge_calls), gives the same calls."""
    torch = pytest.importorskip("torch")
    n, hop, pf = 1024, 256, 64
    period = hop * pf
    L = 3600 * 384000
    rng = np.random.default_rng(55)
    base = (rng.standard_normal(period) * 0.01).astype(np.float32)
    t = np.arange(1920, dtype=np.float64)          # one 5 ms down-sweeping pulse per period, Hann-shaped
    base[4000:5920] += (0.3 * 0.5 * (1 - np.cos(2 * np.pi * t / 1920)) * np.sin(1.2 * t - 1.6e-4 * t * t)).astype(np.float32)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(base).to(dev).repeat(L // period + 1)[:L].contiguous()
    thr, min_frames = 190, 3
    with plugin_engine(n, hop, out_kind=F.OUT_FEATURES, bin_lo=40, bin_hi=400) as eng:
        g = eng.geometry(L)
        assert g.frames == 5399996
        out = torch.empty((g.frames, 8), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.stft_device(x.data_ptr(), L, out.data_ptr(), out.numel())
        assert "fixup" in eng.kernel_route()
        whole = (g.frames // pf) * pf
        v = out[:whole].view(-1, pf, 8)
        assert bool((v == v[0:1]).all())
        ft = out.cpu().numpy().view(F.FEATURE_DTYPE).reshape(-1)
        # spans on their own (pointer offsets into the resident recording)
        spans = sharding.split_recording(L, n, hop, 8, F.FRAMES_PLUGIN_STRICT)
        parts = []
        for s in spans:
            so = torch.empty((s["frames"], 8), dtype=torch.uint8, device=dev)
            eng.stft_device(x.data_ptr() + 4 * s["sample_start"], s["sample_count"], so.data_ptr(), so.numel())
            fs = so.cpu().numpy().view(F.FEATURE_DTYPE).reshape(-1)
            parts.append(E.segment_calls(fs, thr, 1, first_frame=s["first_frame"]))
    assert not ft["reserved"].any()
    # head against the checker
    idx, mags, _, _ = oracle.plugin_get_frequencies(np.tile(base, 2)[:period + n + 1], n, n - hop, want_mags=True)
    refp = oracle.frame_peaks_f32(mags[:pf], 40, 400)
    assert np.array_equal(ft["peak_bin"][:pf], refp)
    assert np.array_equal(ft["peak_idx"][:pf], idx[np.arange(pf), refp])
    # segmentation of the whole, predicted from the head
    calls = E.segment_calls(ft, thr, min_frames)
    head_calls = E.segment_calls(ft[:pf], thr, min_frames)
    assert len(head_calls) >= 1 and head_calls["start_frame"][0] > 0 and head_calls["end_frame"][-1] < pf - 1
    assert len(calls) == len(head_calls) * (g.frames // pf) + len(E.segment_calls(ft[whole:], thr, min_frames))
    k = len(head_calls)
    assert np.array_equal(calls["peak_bin"][:k], head_calls["peak_bin"])
    assert np.array_equal(calls["start_frame"][k:2 * k], head_calls["start_frame"] + pf)
    # 8 spans, stitched
    merged = sharding.merge_calls(parts, min_frames=min_frames)
    assert merged.tobytes() == calls.tobytes()


# ------------------------------------------------------------------ per-file batch: 5 s x 500 kHz int16 files
def test_file_pipeline_at_file_size_against_checker():
    """Three 2.5 M-sample int16 files (5 s x 500 kHz) through the overlapped per-file pipeline, worker render to RGBA:
    the first and the last 64 columns of every image are within one step of the checker's render of the same
    samples, and each image equals the synchronous call's."""
    from spectrogram_amd.sharding import FilePipeline
    n, hop, L = 1024, 256, 2_500_000
    kw = dict(fft_n=n, hop=hop, window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5, frame_rule=F.FRAMES_WORKER_COMPAT,
              twiddle_mode=F.TW_ACCURATE, arith=F.ARITH_FAST_F32, out_kind=F.OUT_RGBA8, in_kind=F.IN_I16)
    files = []
    for i in range(3):
        rng = np.random.default_rng(500 + i)
        x = rng.standard_normal(L) * 0.01
        for p0 in range(20000 * (i + 1), L - 3000, 40000):          # 5 ms pulses every 80 ms, 80 -> 35 kHz
            t = np.arange(2500, dtype=np.float64)
            x[p0:p0 + 2500] += 0.3 * np.hanning(2500) * np.sin(2 * np.pi * (80e3 * t / 5e5 - 0.5 * 45e3 / 2500 * t * t / 5e5))
        files.append(("file%d" % i, np.round(np.clip(x, -1, 1) * 32767).astype(np.int16)))
    lut = E.worker_lut(G.colormap("default"))
    with FilePipeline(L, slots=3, **kw) as pipe:
        for e in pipe.engines:
            e.set_colormap(lut)
        got = list(pipe.run(files))
    width = -(-(L - n) // hop)
    frames = (L - n) // hop + 1
    assert frames == width == 9762
    with E.Engine(**kw) as eng:
        eng.set_colormap(lut)
        for (tag, img), (tag0, q) in zip(got, files):
            assert tag == tag0 and img.shape == (n // 2, width, 4)
            assert np.array_equal(img, eng.stft(q))
            xf = (q.astype(np.float32) / 32768.0).astype(np.float32)
            head = oracle.worker_render(xf[:n + hop * 64 + 1], n, 75.0, None)[..., 0]      # grey render = index image
            tail = oracle.worker_render(xf[(frames - 64) * hop:], n, 75.0, None)[..., 0]
            assert head.shape[1] >= 64 and tail.shape[1] == 64
            for part, ref_idx in ((img[:, :64], head[:, :64]), (img[:, -64:], tail)):
                ok, same = within_one_step(part, ref_idx, lut)
                assert ok and same >= 0.99, tag


def test_feature_fixup_when_most_frames_are_doubtful():
    """A tone half-way between two bins makes the two top bins equal up to rounding in EVERY frame, so the float32
    feature kernel marks all 1.8 M frames of a 20-minute recording -- more than the fix-up pass can list (2^20), which
    exercises its scan for the rest.  Records must repeat with the signal's period (8 hops) and equal the checker's."""
    torch = pytest.importorskip("torch")
    n, hop, k = 1024, 256, 200
    L = 1200 * 384000
    period = 8 * hop
    t = np.arange(period, dtype=np.float64)
    base = (0.5 * np.sin(2 * np.pi * (k + 0.5) / n * t)).astype(np.float32)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(base).to(dev).repeat(L // period + 1)[:L].contiguous()
    for eng in (plugin_engine(n, hop, out_kind=F.OUT_FEATURES), worker_engine(n, hop, out_kind=F.OUT_FEATURES)):
        with eng:
            g = eng.geometry(L)
            out = torch.empty((g.frames, 8), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            eng.stft_device(x.data_ptr(), L, out.data_ptr(), out.numel())
            assert eng.features_recomputed() > (1 << 20)
            whole = (g.frames // 8) * 8
            v = out[:whole].view(-1, 8, 8)
            assert bool((v == v[0:1]).all())
            ft = out[:8].cpu().numpy().view(F.FEATURE_DTYPE).reshape(-1)
        assert not ft["reserved"].any()
        head = np.tile(base, 3)[:period + n + 1]
        if eng.cfg.scale_mode == F.SCALE_PLUGIN_DB:
            idx, mags, _, _ = oracle.plugin_get_frequencies(head, n, n - hop, want_mags=True)
        else:
            mags, idx = oracle.worker_magnitudes(head, n, hop, 8)
        refp = oracle.frame_peaks_f32(mags[:8])
        assert set(refp) <= {k, k + 1}
        assert np.array_equal(ft["peak_bin"], refp) and np.array_equal(ft["peak_idx"], idx[np.arange(8), refp])


def test_device_numa_node_and_binding():
    """The per-file batch places a rank beside its GPU: the library names the GPU's NUMA node (or -1 when the
    platform does not say) and the binding moves the process only onto CPUs it was allowed to use."""
    import os
    lib = F.lib()
    node = lib.sono_device_numa_node(0)
    assert isinstance(node, int) and node >= -1
    assert lib.sono_device_numa_node(10 ** 6) == -1          # no such device: unknown, not an error
    before = os.sched_getaffinity(0)
    try:
        got = sharding.bind_to_gpu_numa(0)
        assert got is None or got == node
        assert os.sched_getaffinity(0) <= before and os.sched_getaffinity(0)
    finally:
        os.sched_setaffinity(0, before)


def test_recording_beyond_32_bit_sample_offsets():
    """A recording of more than 2^32 samples (3.7 h at 384 kHz, int16, 10 GB in device memory; 19.9 M frames, an image
    wider than 2^24 columns): sample offsets, frame numbers times the hop, output offsets and the image's row pitch
    all pass 32 bits.  Periodic signal, so: feature records and colour indices repeat every 64 frames and their heads
    are the checker's; the worker image is the colour map applied to the index output of the same arithmetic, flipped
    and transposed, in every column."""
    torch = pytest.importorskip("torch")
    n, hop, period_frames = 1024, 256, 64
    period = hop * period_frames
    L = (1 << 32) + 3 * (1 << 28) + 12345
    rng = np.random.default_rng(13)
    base = rng.standard_normal(period) * 0.05
    base[5000:9000] += 0.5 * np.sin(np.arange(4000) * 0.8 + 1e-4 * np.arange(4000) ** 2)
    q = np.round(np.clip(base, -1, 1) * 32767).astype(np.int16)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(q).to(dev).repeat(L // period + 1)[:L].contiguous()
    head_pcm = (np.tile(q, 3)[:2 * period + n + 1].astype(np.float32) / np.float32(32768.0))
    ref_idx, ref_mag, _, _ = oracle.plugin_get_frequencies(head_pcm, n, n - hop, want_mags=True)
    # feature records (plugin path)
    with plugin_engine(n, hop, out_kind=F.OUT_FEATURES, in_kind=F.IN_I16) as eng:
        g = eng.geometry(L)
        assert g.frames == (L - n + hop - 1) // hop > (1 << 24)
        ft = torch.empty((g.frames, 8), dtype=torch.uint8, device=dev)
        eng.stft_device(x.data_ptr(), L, ft.data_ptr(), ft.numel())
        whole = (g.frames // period_frames) * period_frames
        rec = ft[:whole, :5].view(-1, period_frames, 5)                    # peak_bin (4 bytes) + peak_idx
        assert bool((rec == rec[0:1]).all())
        head = ft[:period_frames].cpu().numpy().view(F.FEATURE_DTYPE).reshape(-1)
        tail = ft[g.frames - 3:].cpu().numpy().view(F.FEATURE_DTYPE).reshape(-1)
    want_bin = oracle.frame_peaks_f32(ref_mag[:period_frames])
    assert np.array_equal(head["peak_bin"], want_bin)
    assert np.array_equal(head["peak_idx"], ref_idx[np.arange(period_frames), want_bin])
    k = (g.frames - 3) % period_frames                                    # the last frames, past the whole periods
    assert np.array_equal(tail["peak_bin"], np.resize(want_bin, k + 3 + period_frames)[k:k + 3])
    del ft, rec
    # colour indices and the image (worker path)
    lut = (np.arange(1024, dtype=np.uint32) * 13 % 251).astype(np.uint8)
    with worker_engine(n, hop, out_kind=F.OUT_RGBA8, in_kind=F.IN_I16) as eng, \
            worker_engine(n, hop, out_kind=F.OUT_U8_INDEX, in_kind=F.IN_I16) as eng8:
        eng.set_colormap(lut)
        g = eng.geometry(L)
        assert g.width > (1 << 24) and g.bins == 512
        idx = torch.empty((g.frames, g.bins), dtype=torch.uint8, device=dev)
        eng8.stft_device(x.data_ptr(), L, idx.data_ptr(), idx.numel())
        whole = (g.frames // period_frames) * period_frames
        v = idx[:whole].view(-1, period_frames, g.bins)
        assert bool((v == v[0:1]).all())
        head8 = idx[:period_frames].cpu().numpy()
        img = torch.empty((g.bins, g.width, 4), dtype=torch.uint8, device=dev)
        eng.stft_device(x.data_ptr(), L, img.data_ptr(), img.numel())
        route = eng.kernel_route()
        px = img.view(torch.int32).view(g.bins, g.width)
        lut32 = torch.from_numpy(lut.view(np.int32).copy()).to(dev)
        for c0 in range(1, g.width, 1 << 18):                                # (column 0 may hold the wrapped last frame)
            c1 = min(g.width, c0 + (1 << 18))
            want = lut32[idx[c0:c1].long()].t().flip(0)
            assert bool((px[:, c0:c1] == want).all()), (c0, route)
    refw = oracle.worker_render(head_pcm, n, 75.0, None, want_index=True)[1]
    assert np.abs(head8.astype(np.int16) - refw[:period_frames]).max() <= 1


@pytest.mark.parametrize("n,hop", [(512, 128), (2048, 512), (4096, 1024), (4096, 1022)])
def test_other_sizes_beyond_32_bit_sample_offsets(n, hop):
    """The same at the other transform sizes (their own kernels; hop 1022 at 4096 points is the general kernel): colour
    indices of a periodic float32 recording of more than 2^32 samples repeat every period and start like the checker's."""
    torch = pytest.importorskip("torch")
    period_frames = 64
    period = hop * period_frames
    L = (1 << 32) + (1 << 27) + 4321
    rng = np.random.default_rng(n + hop)
    base = (rng.standard_normal(period) * 0.05).astype(np.float32)
    base[period // 3:period // 2] += (0.5 * np.sin(np.arange(period // 2 - period // 3) * 0.6)).astype(np.float32)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(base).to(dev).repeat(L // period + 1)[:L].contiguous()
    with plugin_engine(n, hop) as eng:
        g = eng.geometry(L)
        out = torch.empty((g.frames, g.bins), dtype=torch.uint8, device=dev)
        eng.stft_device(x.data_ptr(), L, out.data_ptr(), out.numel())
        route = eng.kernel_route()
        whole = (g.frames // period_frames) * period_frames
        v = out[:whole].view(-1, period_frames, g.bins)
        assert bool((v == v[0:1]).all()), route
        head = out[:period_frames].cpu().numpy()
        tail = out[g.frames - 2:].cpu().numpy()
    ref = oracle.plugin_get_frequencies(np.tile(base, 3)[:period + n + 1 + hop], n, n - hop)[:period_frames]
    d = np.abs(head.astype(np.int16) - ref.astype(np.int16))
    d = np.minimum(d, 256 - d)
    assert d.max() <= 1, route
    k = (g.frames - 2) % period_frames
    assert np.array_equal(tail, np.concatenate([head, head])[k:k + 2]), route


def test_stream_switch_orders_launches_that_share_the_contexts_scratch():
    """sono_set_stream while the previous stream still runs the context's kernels: the 2048-point image writer parks
    every round's colour indices in context-owned scratch, so a launch on the new stream must not overtake the one
    before it.  The library makes the new stream wait for the old one (an event, no host wait); two recordings
    enqueued back to back on alternating streams come out as they do one at a time."""
    torch = pytest.importorskip("torch")
    n, hop = 2048, 512
    rng = np.random.default_rng(77)
    La, Lb = n + hop * 30000, n + hop * 29000 + 13
    xa = (rng.standard_normal(La) * 0.1).astype(np.float32)
    xb = (rng.standard_normal(Lb) * 0.3).astype(np.float32)
    lut = (np.arange(1024, dtype=np.uint32) * 7 % 251).astype(np.uint8)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with E.Engine(fft_n=n, hop=hop, window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5,
                  frame_rule=F.FRAMES_WORKER_COMPAT, twiddle_mode=F.TW_ACCURATE, arith=F.ARITH_FAST_F32,
                  out_kind=F.OUT_RGBA8) as eng:
        eng.set_colormap(lut)
        na, nb = eng.out_bytes(La), eng.out_bytes(Lb)
        da, db, oa, ob = eng.device_malloc(xa.nbytes), eng.device_malloc(xb.nbytes), eng.device_malloc(na), eng.device_malloc(nb)
        try:
            eng.copy_to_device(da, xa)
            eng.copy_to_device(db, xb)
            eng.stft_device(da, La, oa, na)
            assert "stft2048_rgba_kernel" in eng.kernel_route()
            eng.stft_device(db, Lb, ob, nb)
            want_a, want_b = eng.copy_to_host(np.empty(na, np.uint8), oa), eng.copy_to_host(np.empty(nb, np.uint8), ob)
            for rep in range(6):
                eng.copy_to_device(oa, np.zeros(na, np.uint8))
                eng.copy_to_device(ob, np.zeros(nb, np.uint8))
                eng.wait()
                eng.set_stream(s1.cuda_stream)
                eng.stft_device(da, La, oa, na, wait=False)
                eng.set_stream(s2.cuda_stream)
                eng.stft_device(db, Lb, ob, nb, wait=False)
                eng.wait()                                   # the second stream, which waited for the first
                got_a, got_b = eng.copy_to_host(np.empty(na, np.uint8), oa), eng.copy_to_host(np.empty(nb, np.uint8), ob)
                assert np.array_equal(got_a, want_a) and np.array_equal(got_b, want_b), rep
            eng.set_stream(None)
        finally:
            torch.cuda.synchronize()
            for p in (da, db, oa, ob):
                eng.device_free(p)
