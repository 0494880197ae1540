"""Seeded random configurations of the hot path against the CPU checker.

The kernels have many routes (four transform sizes, exact-root and table-twiddle flavours, hops with and without
shared sample loads, bands inside and across groups of four bins, float32 and int16 samples, colour indices, images
and feature records, frame counts around the block and round sizes of the image writers).  The fixed cases of the
other files pin the routes one by one; this file draws combinations nobody thought of.  Tolerances as everywhere:
float32 arithmetic within 1 LSB of the checker's colour index (cyclic at the plugin's 256 -> 0 wrap, at most 2 % of the
bins off), feature records and everything in the exact arithmetic bit-identical.
"""
import numpy as np
import pytest

import oracle
from spectrogram_amd import _ffi as F
from spectrogram_amd import engine as E

pytestmark = pytest.mark.gpu

# a colour table whose red channel is a permutation of 0..255, so that a pixel names its index
_PERM = np.random.default_rng(99).permutation(256).astype(np.uint8)
_INV = np.argsort(_PERM).astype(np.int16)
LUT = np.stack([_PERM, (np.arange(256) * 7 % 256).astype(np.uint8), (255 - np.arange(256)).astype(np.uint8),
                np.full(256, 255, np.uint8)], axis=1).reshape(-1)


def cyc_diff(a, b):
    d = (a.astype(np.int16) - b.astype(np.int16)) % 256
    return np.minimum(d, 256 - d)


def draw_config(rng):
    n = int(rng.choice([512, 1024, 1024, 2048, 4096]))
    hop = int(rng.choice([n // 4, n // 2, n // 8, n // 16, int(rng.integers(n // 16, n + 1))]))
    path = str(rng.choice(["worker", "plugin"]))
    if path == "worker":                                   # the worker derives its hop from a percentage
        hop = E.worker_hop(n, 100.0 * (1 - hop / n))
    elif hop == n:                                         # the plugin cannot ask for it: noverlap 0 means "choose for me"
        hop = n - 1                                        # (spectrogram.esm.js:502-506)
    out = str(rng.choice(["u8", "rgba", "features"]))
    frames = int(rng.choice([int(rng.integers(3, 40)), int(rng.integers(40, 700)), int(rng.integers(700, 1600))]))
    extra = int(rng.choice([0, int(rng.integers(0, hop))]))
    bins = n // 2
    if rng.random() < 0.5:
        lo, hi = 0, bins
    else:
        lo = int(rng.integers(0, bins - 8))
        hi = int(rng.integers(lo + 5, bins + 1))
    c = dict(n=n, hop=hop, path=path, out=out, L=n + hop * (frames - 1) + extra, lo=lo, hi=hi,
             i16=bool(rng.random() < 0.3), exact=bool(rng.random() < 0.15), window="hann", alpha=None, gain=20.0, range=80.0)
    if path == "plugin":                                   # the plugin's other options (the worker ignores them)
        c["window"] = str(rng.choice(list(oracle.WINDOW_IDS)))
        if c["window"] in ("blackman", "gauss") and rng.random() < 0.5:
            c["alpha"] = float(rng.choice([0.16, 0.25, 0.4]))
        c["gain"], c["range"] = [(20.0, 80.0), (0.0, 100.0), (40.0, 60.0), (-10.0, 30.0), (20.0, 100.0)][int(rng.integers(0, 5))]
        if rng.random() < 0.2 and not c["exact"]:
            c["out"] = "db"                                # float32 decibels, [frame][bin]
        elif c["out"] in ("u8", "rgba") and rng.random() < 0.25:
            c["scale"] = str(rng.choice(["mel", "logarithmic", "bark", "erb"]))      # a non-linear frequency scale
            c["rate"] = int(rng.choice([44100, 256000, 384000, 500000]))
    return c


def signal(rng, L):
    x = rng.standard_normal(L) * 10.0 ** rng.uniform(-3, -1)
    t = np.arange(L, dtype=np.float64)
    f0, f1 = rng.uniform(0.01, 0.45, 2)
    x += 10.0 ** rng.uniform(-2, -0.2) * np.sin(2 * np.pi * (f0 * t + (f1 - f0) * t * t / (2 * L)))     # a sweep
    if rng.random() < 0.5:
        a = int(rng.integers(0, L // 2))
        x[a:a + L // 5] = 0.0                                                                            # a stretch of silence
    return np.clip(x, -1, 1).astype(np.float32)


def engine_of(c, out_kind):
    kw = dict(fft_n=c["n"], hop=c["hop"], arith=F.ARITH_EXACT_F64 if c["exact"] else F.ARITH_FAST_F32, out_kind=out_kind,
              in_kind=F.IN_I16 if c["i16"] else F.IN_F32, bin_lo=c["lo"], bin_hi=c["hi"] if (c["lo"], c["hi"]) != (0, c["n"] // 2) else 0)
    if c["path"] == "worker":
        return E.Engine(window_id=F.WIN_WORKER_HANN, scale_mode=F.SCALE_WORKER_LOG5, frame_rule=F.FRAMES_WORKER_COMPAT,
                        twiddle_mode=F.TW_ACCURATE, **kw)
    eng = E.Engine(window_id=F.WINDOW_IDS[c.get("window", "hann")], alpha=c.get("alpha") or 0.0, gain_db=c.get("gain", 20.0),
                   range_db=c.get("range", 80.0), scale_mode=F.SCALE_PLUGIN_DB, frame_rule=F.FRAMES_PLUGIN_STRICT,
                   twiddle_mode=F.TW_PLUGIN_RECURRENCE, **kw)
    if "scale" in c:
        eng.set_filterbank(*E.filterbank_table(c["scale"], c["n"], c["rate"]))
    return eng


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SONO_RANDOM_CASES", "96"))))
def test_random_configuration_against_checker(seed):
    rng = np.random.default_rng(int(__import__("os").environ.get("SONO_RANDOM_SEED0", "1000")) + seed)
    c = draw_config(rng)
    x = signal(rng, c["L"])
    if c["i16"]:
        q = np.round(x * 32767).astype(np.int16)
        x_in, x = q, (q.astype(np.float32) / np.float32(32768.0))        # the value of an int16 sample is s / 32768
    else:
        x_in = x
    n, hop, lo, hi = c["n"], c["hop"], c["lo"], c["hi"]
    # the checker: indices [frame][bin] (and, for the features, float32 magnitudes)
    if c["path"] == "worker":
        pct = 100.0 * (1 - hop / n)
        assert E.worker_hop(n, pct) == hop
        ref_img, ref_idx = oracle.worker_render(x, n, pct, LUT, want_index=True)
        mags = None
    elif "scale" in c:
        ref_idx = oracle.plugin_get_frequencies_scaled(x, n, n - hop, c["scale"], c["rate"], window=c["window"], alpha=c["alpha"],
                                                       gain_db=c["gain"], range_db=c["range"])
        ref_img = mags = None
    else:
        ref_idx, mags, _, _ = oracle.plugin_get_frequencies(x, n, n - hop, window=c["window"], alpha=c["alpha"],
                                                            gain_db=c["gain"], range_db=c["range"], want_mags=True)
        ref_img = None
    if c["out"] == "features" and c["path"] == "worker":
        mags, _ = oracle.worker_magnitudes(x, n, hop, ref_idx.shape[0])
    exact = c["exact"]
    tag = repr(c)
    if c["out"] == "u8":
        with engine_of(c, F.OUT_U8_INDEX) as eng:
            got = eng.stft(x_in)
        want = ref_idx[:, lo:hi]
        assert got.shape == want.shape, tag
        if exact:
            assert np.array_equal(got, want), tag
        else:
            d = cyc_diff(got, want)
            assert d.max(initial=0) <= 1 and (d != 0).mean() < 0.02, tag
    elif c["out"] == "rgba":
        with engine_of(c, F.OUT_RGBA8) as eng:
            eng.set_colormap(LUT)
            got = eng.stft(x_in)
        if c["path"] == "worker":
            want_img = ref_img[n // 2 - hi:n // 2 - lo]                       # the band is a crop of the image
        else:
            want_img = LUT.reshape(256, 4)[ref_idx[:, lo:hi].T[::-1]]
        assert got.shape == want_img.shape, tag
        if exact:
            assert np.array_equal(got, want_img), tag
        else:
            gi, wi = _INV[got[..., 0]], _INV[want_img[..., 0]]
            assert np.array_equal(got, LUT.reshape(256, 4)[gi]), tag           # every pixel is an entry of the table
            d = cyc_diff(gi, wi) if c["path"] == "plugin" else np.abs(gi - wi)
            assert d.max(initial=0) <= 1 and (d != 0).mean() < 0.02, tag
    elif c["out"] == "db":
        with engine_of(c, F.OUT_F32_DB) as eng:
            got = eng.stft(x_in)
        # float32 decibels within 0.01 dB of the reference arithmetic above the display floor (-gain-range), for the
        # bins within 75 dB of the frame's peak: float32 round-off sits about 135 dB below the peak (2e-7 of it), and
        # 0.01 dB is a relative error of 1.2e-3
        refd = mags[:, lo:hi].astype(np.float64)
        assert got.shape == refd.shape, tag
        peak = mags.astype(np.float64).max(axis=1, keepdims=True)
        vis = (refd > 10 ** ((-c["gain"] - c["range"]) / 20)) & (refd > peak * 10 ** (-75 / 20))
        if vis.any():
            assert np.abs(got[vis] - 20 * np.log10(refd[vis])).max() <= 0.01, tag
    else:
        with engine_of(c, F.OUT_FEATURES) as eng:
            ft = eng.stft(x_in)
        want_bin = oracle.frame_peaks_f32(mags, lo, hi)
        assert np.array_equal(ft["peak_bin"], want_bin), tag
        assert np.array_equal(ft["peak_idx"], ref_idx[np.arange(len(want_bin)), want_bin]), tag


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
@pytest.mark.parametrize("path", ["worker", "plugin"])
def test_features_of_silent_frames_beside_loud_ones(n, path):
    """The float32 kernels transform two frames at once, so a frame of digital silence (or a band that leaves out a
    loud tone) holds the other frame's rounding noise, about 2e-7 of its peak: such records must come from the
    exact-arithmetic pass, not from the noise.  Blocks of two frame lengths -- a loud tone, exact zeros, very quiet
    noise, the tone again -- at 50 % overlap, so that whole frames are silent and every pairing occurs; once over the
    whole spectrum, once over a band beside the tone."""
    hop = n // 2
    assert E.worker_hop(n, 50.0) == hop
    rng = np.random.default_rng(n)
    blocks = 48
    x = np.zeros(2 * n * blocks + 7, np.float32)
    t = np.arange(2 * n, dtype=np.float64)
    tone_bin = n // 8
    for b in range(blocks):
        kind = b % 4
        if kind in (0, 3):
            x[2 * n * b:2 * n * (b + 1)] = (0.8 * np.sin(2 * np.pi * (tone_bin + 0.3) / n * t + b)).astype(np.float32)
        elif kind == 2:
            x[2 * n * b:2 * n * (b + 1)] = (rng.standard_normal(2 * n) * 1e-5).astype(np.float32)
    if path == "worker":
        _, ref_idx = oracle.worker_render(x, n, 50.0, None, want_index=True)
        mags, _ = oracle.worker_magnitudes(x, n, hop, ref_idx.shape[0])
    else:
        ref_idx, mags, _, _ = oracle.plugin_get_frequencies(x, n, n - hop, want_mags=True)
    for lo, hi in ((0, n // 2), (tone_bin + n // 16 + 1, n // 2 - 3)):
        c = dict(n=n, hop=hop, path=path, lo=lo, hi=hi, i16=False, exact=False)
        with engine_of(c, F.OUT_FEATURES) as eng:
            ft = eng.stft(x)
            redone = eng.features_recomputed()
        want_bin = oracle.frame_peaks_f32(mags, lo, hi)
        assert len(ft) == len(want_bin) and np.array_equal(ft["peak_bin"], want_bin), (n, path, lo, hi)
        assert np.array_equal(ft["peak_idx"], ref_idx[np.arange(len(want_bin)), want_bin]), (n, path, lo, hi)
        assert redone >= blocks // 4, (n, path, lo, hi, redone)     # silent frames beside loud ones were recomputed


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
@pytest.mark.parametrize("out", ["u8", "rgba", "features"])
def test_device_pointers_at_odd_offsets(n, out):
    """Callers hand over device pointers of any alignment the element type allows (a view into a longer recording,
    an image inside a larger buffer): samples at any 4-byte (float32) / 2-byte (int16) offset, colour indices at any
    byte, pixels at any 4 bytes, feature records at any 8 bytes.  The kernels that need wider alignment must step
    aside; the result is held to the checker like that of the host-pointer call (indices within 1 LSB, feature records
    equal), and the bytes around the output stay untouched."""
    rng = np.random.default_rng(n + len(out))
    hop = n // 4
    L = n + hop * 333 + 5
    x = signal(rng, L)
    kinds = {"u8": F.OUT_U8_INDEX, "rgba": F.OUT_RGBA8, "features": F.OUT_FEATURES}
    out_step = {"u8": 1, "rgba": 4, "features": 8}[out]
    for i16 in (False, True):
        xin = np.round(x * 32767).astype(np.int16) if i16 else x
        xval = (xin.astype(np.float32) / np.float32(32768.0)) if i16 else x
        ref_idx = oracle.plugin_get_frequencies(xval, n, n - hop)
        c = dict(n=n, hop=hop, path="plugin", lo=0, hi=n // 2, i16=i16, exact=False)
        with engine_of(c, kinds[out]) as eng:
            if out == "rgba":
                eng.set_colormap(LUT)
            want = eng.stft(xin)
            nbytes = want.nbytes
            guard = 64
            for s_off, o_off in ((1, out_step), (2, 3 * out_step), (3, 0), (5, 5 * out_step)):
                dpcm = eng.device_malloc(xin.nbytes + 64)
                dout = eng.device_malloc(nbytes + 2 * guard + 64)
                try:
                    eng.copy_to_device(dpcm + s_off * xin.itemsize, xin)
                    fill = np.full(nbytes + 2 * guard + 64, 0xA5, np.uint8)
                    eng.copy_to_device(dout, fill)
                    eng.stft_device(dpcm + s_off * xin.itemsize, L, dout + guard + o_off, nbytes)
                    back = eng.copy_to_host(np.empty_like(fill), dout)
                finally:
                    eng.device_free(dpcm)
                    eng.device_free(dout)
                tag = (n, out, i16, s_off, o_off)
                assert (back[:guard + o_off] == 0xA5).all() and (back[guard + o_off + nbytes:] == 0xA5).all(), tag
                got = back[guard + o_off:guard + o_off + nbytes].view(want.dtype).reshape(want.shape)
                if out == "features":
                    assert np.array_equal(got["peak_bin"], want["peak_bin"]) and np.array_equal(got["peak_idx"], want["peak_idx"]), tag
                elif out == "u8":
                    assert cyc_diff(got, ref_idx).max() <= 1 and cyc_diff(want, ref_idx).max() <= 1, tag
                else:
                    assert np.array_equal(got, LUT.reshape(256, 4)[_INV[got[..., 0]]]), tag
                    assert cyc_diff(_INV[got[..., 0]], ref_idx.T[::-1]).max() <= 1, tag
                    assert cyc_diff(_INV[want[..., 0]], ref_idx.T[::-1]).max() <= 1, tag


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
def test_full_scale_burst_beside_silence_at_low_display_floors(n):
    """The float32 arithmetic resolves about 100 dB below the loudest value of a transform; with the plugin's recurrence
    factors two frames share a transform, so a full-scale burst leaves a trace of about -100 dB in a silent partner.
    The reference's defaults (gain 20, range 80: floor -100 dB) and every floor down to -100 dB stay within 1 LSB;
    a context with a lower floor computes in the exact arithmetic and is bit-identical."""
    hop = n // 2
    blocks = 24
    x = np.zeros(2 * n * blocks + 8, np.float32)
    t = np.arange(2 * n, dtype=np.float64)
    for b in range(0, blocks, 2):
        x[2 * n * b:2 * n * (b + 1)] = (0.95 * np.sin(2 * np.pi * (n // 8 + 0.37) / n * t)).astype(np.float32)
    for gain, rng_db in ((20.0, 80.0), (0.0, 100.0), (20.0, 100.0), (20.0, 120.0)):
        ref = oracle.plugin_get_frequencies(x, n, n - hop, gain_db=gain, range_db=rng_db)
        c = dict(n=n, hop=hop, path="plugin", lo=0, hi=n // 2, i16=False, exact=False, gain=gain, range=rng_db)
        for s_off in (0, 1):                                   # 1: a pointer the 16-byte sample loads cannot take
            with engine_of(c, F.OUT_U8_INDEX) as eng:
                dpcm = eng.device_malloc(x.nbytes + 64)
                dout = eng.device_malloc(ref.nbytes + 64)
                try:
                    eng.copy_to_device(dpcm + 4 * s_off, x)
                    eng.stft_device(dpcm + 4 * s_off, len(x), dout, ref.nbytes)
                    got = eng.copy_to_host(np.empty(ref.nbytes, np.uint8), dout).reshape(ref.shape)
                    route = eng.kernel_route()
                finally:
                    eng.device_free(dpcm)
                    eng.device_free(dout)
            if gain + rng_db > 100.0:
                assert np.array_equal(got, ref) and ("exact" in route or "double" in route), (n, gain, rng_db, s_off, route)
            else:
                assert cyc_diff(got, ref).max() <= 1, (n, gain, rng_db, s_off, route)


@pytest.mark.parametrize("seed", range(24))
def test_random_resample_and_colour_steps(seed):
    """The steps behind getFrequencies with random shapes: resample (spectrogram.esm.js:595-640) to any width -- fewer,
    equal, more columns, one column, one frame -- bit-identical to the checker, any window of its columns equal to the
    same columns of the whole, and the colour-map step (:366-376) over any row window equal to the table applied to
    the flipped indices."""
    rng = np.random.default_rng(7000 + seed)
    Fr = int(rng.choice([1, 2, int(rng.integers(3, 60)), int(rng.integers(60, 900))]))
    B = int(rng.choice([1, 7, 64, 256, 512, 1120, 2048]))
    W = int(rng.choice([1, Fr, max(1, Fr // 2), 2 * Fr + 1, int(rng.integers(1, 1500))]))
    frames = rng.integers(0, 256, (Fr, B)).astype(np.uint8)
    if rng.random() < 0.3:
        frames[rng.integers(0, Fr)] = 255                                        # a saturated frame
    want = oracle.plugin_resample(frames, W)
    with E.Engine() as eng:
        got = eng.resample(frames, W)
        assert np.array_equal(got, want), (Fr, B, W)
        # a window of columns, device to device
        col0 = int(rng.integers(0, W))
        ncols = int(rng.integers(1, W - col0 + 1))
        din = eng.device_malloc(frames.nbytes)
        dout = eng.device_malloc(ncols * B + 16)
        try:
            eng.copy_to_device(din, frames)
            eng.resample_cols_device(din, Fr, B, W, col0, ncols, dout)
            part = eng.copy_to_host(np.empty((ncols, B), np.uint8), dout)
        finally:
            eng.device_free(din)
            eng.device_free(dout)
        assert np.array_equal(part, want[col0:col0 + ncols]), (Fr, B, W, col0, ncols)
        # colour map over a row window: image row r shows bin B - 1 - r
        eng.set_colormap(LUT)
        row0 = int(rng.integers(0, B))
        rows = int(rng.integers(1, B - row0 + 1))
        img = eng.colorize(want, row0, rows)
        full = LUT.reshape(256, 4)[want.T[::-1]]
        assert np.array_equal(img, full[row0:row0 + rows]), (Fr, B, W, row0, rows)


@pytest.mark.parametrize("seed", range(24))
def test_random_plugin_option_sets_through_the_host(seed):
    """The wavesurfer plugin's option surface through the Python mirror of the host (the same C ABI calls as the Node
    host): random fftSamples / noverlap (given or derived from the canvas width, spectrogram.esm.js:502-506) / window
    / gain / range / frequency range / colour map / canvas width, against the checker's getFrequencies -> resample
    -> crop -> draw.  Exact arithmetic: bit-identical; float32: every pixel the colour of the checker's index or a
    neighbouring one."""
    from spectrogram_amd.plugin import AudioBuffer, Spectrogram
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([256, 512, 1024, 2048]))
    sr = int(rng.choice([44100, 256000, 384000, 500000]))
    L = int(rng.integers(n + 50, 60000))
    x = signal(rng, L)
    width = int(rng.integers(20, 900))
    nov = None if rng.random() < 0.4 else int(rng.choice([n // 2, 3 * n // 4, n - n // 8, int(rng.integers(1, n))]))
    window = str(rng.choice(list(oracle.WINDOW_IDS)))
    gain, rng_db = [(20.0, 80.0), (0.0, 100.0), (40.0, 60.0)][int(rng.integers(0, 3))]
    fmin = float(rng.choice([0.0, rng.uniform(0, 0.2) * sr]))
    fmax = float(rng.choice([0.0, rng.uniform(0.25, 0.5) * sr, 0.7 * sr]))            # 0: Nyquist; above Nyquist: background rows
    cm = rng.random((256, 4))
    exact = bool(rng.random() < 0.4)
    opts = {"fftSamples": n, "colorMap": cm.tolist(), "scale": "linear", "frequencyMin": fmin, "frequencyMax": fmax,
            "windowFunc": window, "gainDB": gain, "rangeDB": rng_db, "arith": "exact" if exact else "fast"}
    if nov is not None:
        opts["noverlap"] = nov
    p = Spectrogram.create(opts)
    try:
        buf = AudioBuffer(x, sr)
        frames = p.getFrequencies(buf, canvas_width=width)[0]
        ref_frames = oracle.plugin_get_frequencies(x, n, nov, canvas_width=width, window=window, gain_db=gain, range_db=rng_db)
        tag = (n, sr, L, width, nov, window, gain, rng_db, fmin, fmax, exact)
        assert frames.shape == ref_frames.shape, tag
        lut = oracle.plugin_lut(cm)
        assert np.array_equal(p._colorMapUint, lut), tag
        if exact:
            assert np.array_equal(frames, ref_frames), tag
        else:
            d = cyc_diff(frames, ref_frames)
            assert d.max(initial=0) <= 1 and (d != 0).mean() < 0.02, tag
        # resample + draw from the host's own frames (so that the float32 case is compared step by step)
        rgba, (sy, sh) = p.drawSpectrogram(frames, width)
        want = oracle.plugin_draw(oracle.plugin_resample(frames, width), lut)
        assert np.array_equal(rgba, want), tag
        assert (sy, sh) == oracle.plugin_crop(n // 2, sr, fmin, fmax if fmax else sr / 2), tag
    finally:
        p.destroy()


@pytest.mark.parametrize("seed", range(16))
def test_random_worker_messages(seed):
    """The worker's message contract (spectrogramWorker.js:17-36) with random render messages through the Python
    mirror of the host: fftSize, overlap percentage (any number: the hop is max(1, floor(N (1 - pct/100)))), buffer
    lengths around a whole number of hops (the wrapped last frame), with and without a colour map, options that the
    reference ignores.  Exact arithmetic: the checker's image byte for byte; float32: every pixel within one step."""
    from spectrogram_amd.worker import SpectrogramWorker
    rng = np.random.default_rng(12000 + seed)
    n = int(rng.choice([256, 512, 1024, 2048, 4096]))
    pct = float(rng.choice([0, 50, 75, 87.5, 93.75, round(rng.uniform(0, 99), 2)]))
    hop = E.worker_hop(n, pct)
    frames = int(rng.integers(2, 400))
    L = n + hop * (frames - 1) + int(rng.choice([0, int(rng.integers(0, max(1, hop)))]))
    x = signal(rng, L)
    with_map = bool(rng.random() < 0.6)
    cm = rng.random((256, 4))
    arith = "exact" if rng.random() < 0.4 else "fast"
    lut = E.worker_lut(cm) if with_map else None
    want = oracle.worker_render(x, n, pct, lut)
    got = []
    wk = SpectrogramWorker(arith=arith)
    wk.onmessage = lambda e: got.append(e["data"])
    canvas = {}
    wk.postMessage({"type": "init", "canvas": canvas, "options": {"colorMap": cm.tolist()} if with_map else {}})
    wk.postMessage({"type": "render", "buffer": x, "fftSize": n, "overlap": pct,
                    "options": {"windowFunc": "hamming", "gainDB": 7, "rangeDB": 33}})       # accepted, ignored
    wk.terminate()
    tag = (n, pct, hop, L, with_map, arith)
    assert len(got) == 1 and got[0]["type"] == "rendered", tag
    img = got[0]["data"]
    assert img.shape == want.shape, tag
    if arith == "exact":
        assert np.array_equal(img, want), tag
    elif not with_map:
        assert np.abs(img[..., :3].astype(np.int16) - want[..., :3]).max() <= 1 and (img[..., 3] == 255).all(), tag
    else:
        # the checker's indices, and the colours of the neighbouring ones
        idx = oracle.worker_render(x, n, pct, None)[..., 0].astype(np.int16)
        l4 = lut.reshape(256, 4)
        ok = np.zeros(idx.shape, bool)
        for dlt in (-1, 0, 1):
            ok |= np.all(img == l4[np.clip(idx + dlt, 0, 255)], axis=-1)
        assert ok.all(), tag
