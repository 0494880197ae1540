"""The CPU checker (oracle/) against the reference's own outputs (tests/golden/).

Everything here runs on the CPU.  uint8 / RGBA / table outputs must be bit-identical; double
twiddles may differ from V8's by one ulp (glibc libm vs V8's fdlibm port).
"""
import numpy as np
import pytest

import oracle
from conftest import golden_store

G = golden_store()
WINDOWS = ["bartlett", "bartlettHann", "blackman", "cosine", "gauss", "hamming", "hann", "lanczoz",
           "rectangular", "triangular"]


def _ids(cases, key="id"):
    return [str(c.get(key, i)) for i, c in enumerate(cases)]


# ------------------------------------------------------------------ plugin tables / windows
@pytest.mark.parametrize("case", G.cases("plugin_tables"), ids=lambda c: "N%d" % c["N"])
def test_plugin_tables(case):
    n = case["N"]
    assert case["windows"] == WINDOWS
    ref = G.load(case["window_values"], np.float32).reshape(len(WINDOWS), n)
    for i, w in enumerate(WINDOWS):
        got = oracle.plugin_window(w, n)
        assert np.array_equal(got.view(np.uint32), ref[i].view(np.uint32)), w
    s, c, r = oracle.plugin_tables(n)
    gs, gc = G.load(case["sin_table"], np.float32), G.load(case["cos_table"], np.float32)
    assert np.isnan(gs[0]) and np.isnan(s[0]) and np.isnan(gc[0]) and np.isnan(c[0])
    assert np.array_equal(s[1:].view(np.uint32), gs[1:].view(np.uint32))
    assert np.array_equal(c[1:].view(np.uint32), gc[1:].view(np.uint32))
    assert np.array_equal(r, G.load(case["reverse_table"], np.uint32))


def test_plugin_alpha_windows():
    (case,) = G.cases("plugin_alpha_windows")
    n = case["N"]
    ref = G.load(case["window_values"], np.float32).reshape(len(case["alphas"]), 2, n)
    for i, al in enumerate(case["alphas"]):
        assert np.array_equal(oracle.plugin_window("blackman", n, al), ref[i, 0])
        assert np.array_equal(oracle.plugin_window("gauss", n, al), ref[i, 1])


def test_plugin_window_error():
    (case,) = G.cases("plugin_window_error")
    with pytest.raises(ValueError) as e:
        oracle.plugin_window(case["name"], 512)
    assert str(e.value) == case["message"]


# ------------------------------------------------------------------ plugin calculateSpectrum
@pytest.mark.parametrize("case", G.cases("plugin_spectrum"),
                         ids=lambda c: "%s-%d-%s" % (c["signal"], c["N"], c["window"]))
def test_plugin_spectrum(case):
    x = G.signal(case["signal"])
    n, hop = case["N"], case["hop"]
    ref = G.load(case["mags"], np.float32).reshape(case["frames"], n // 2)
    fft = oracle.PluginFFT(n, case["window"], case["alpha"])
    for f in range(case["frames"]):
        got = fft.calculate_spectrum(x[f * hop:f * hop + n])
        assert np.array_equal(got.view(np.uint32), ref[f].view(np.uint32)), "frame %d" % f
        assert fft.peak.value == case["running_peak"][f]["peak"]
        assert fft.peak_band.value == case["running_peak"][f]["peakBand"]


def test_plugin_spectrum_errors():
    (case,) = G.cases("plugin_errors")
    msgs = {e["label"]: e["message"] for e in case["errors"]}
    with pytest.raises(ValueError) as e:
        oracle.PluginFFT(1024).calculate_spectrum(np.zeros(512, np.float32))
    assert str(e.value) == msgs["length_mismatch"]


# ------------------------------------------------------------------ plugin getFrequencies
@pytest.mark.parametrize("case", G.cases("plugin_get_frequencies"), ids=lambda c: c["id"])
def test_plugin_get_frequencies(case):
    x = G.signal(case["signal"])
    got = oracle.plugin_get_frequencies(x, case["N"], case["noverlap"], case["canvas_width"], case["window"],
                                        case["alpha"], case["gainDB"], case["rangeDB"])
    assert got.shape == (case["frames"], case["bins"])
    if case["frames"]:
        ref = G.load(case["data"], np.uint8).reshape(case["frames"], case["bins"])
        assert np.array_equal(got, ref)


def test_plugin_uint8_wrap_probe():
    """r == -gainDB exactly stores 256, which a Uint8Array wraps to 0 (spectrogram.esm.js:534)."""
    c = [c for c in G.cases("plugin_get_frequencies") if c["signal"] == "dc_half" and c["gainDB"] == 0][0]
    ref = G.load(c["data"], np.uint8).reshape(c["frames"], c["bins"])
    assert ref[0, 0] == 0
    got = oracle.plugin_get_frequencies(G.signal("dc_half"), 1024, 768, 300, "rectangular", None, 0.0, 100.0)
    assert got[0, 0] == 0 and np.array_equal(got, ref)


# ------------------------------------------------------------------ plugin resample / draw / crop
@pytest.mark.parametrize("case", G.cases("plugin_resample"), ids=lambda c: "%s-%d-W%d" % (c["signal"], c["N"], c["out_width"]))
def test_plugin_resample(case):
    x = G.signal(case["signal"])
    frames = oracle.plugin_get_frequencies(x, case["N"], case["noverlap"])
    assert frames.shape[0] == case["in_frames"]
    got = oracle.plugin_resample(frames, case["out_width"])
    ref = G.load(case["data"], np.uint8).reshape(case["out_width"], case["bins"])
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("case", G.cases("plugin_render"), ids=lambda c: c["id"])
def test_plugin_render(case):
    x = G.signal(case["signal"])
    sr = G.sample_rate(case["signal"])
    n = case["N"]
    name = case["colormap"]
    if name in ("default", "rainbow"):
        cm = G.colormap(name)
    elif name == "gray":
        cm = np.array([[(255 - t) / 256] * 3 + [1] for t in range(256)], np.float64)
    else:
        cm = np.array([[t / 256] * 3 + [1] for t in range(256)], np.float64)
    lut = oracle.plugin_lut(cm)
    assert np.array_equal(lut, G.load(case["lut"], np.uint8))
    frames = oracle.plugin_get_frequencies(x, n, case["noverlap"], window=case["window"])
    cols = oracle.plugin_resample(frames, case["out_width"])
    rgba = oracle.plugin_draw(cols, lut)
    ref = G.load(case["rgba"], np.uint8).reshape(case["image_height"], case["image_width"], 4)
    assert rgba.shape == ref.shape
    assert np.array_equal(rgba, ref)
    sy, sh = oracle.plugin_crop(n // 2, sr, case["fmin"], case["fmax"])
    assert (sy, sh) == (case["crop"]["sy"], case["crop"]["sh"])
    assert case["crop"]["sx"] == 0 and case["crop"]["sw"] == case["out_width"]


def test_plugin_crop_grid():
    (case,) = G.cases("plugin_crop_grid")
    for r in case["rows"]:
        assert oracle.plugin_crop(r["N"] // 2, r["sr"], r["fmin"], r["fmax"]) == (r["sy"], r["sh"]), r


# ------------------------------------------------------------------ worker
@pytest.mark.parametrize("case", G.cases("worker_tables"), ids=lambda c: "N%d" % c["N"])
def test_worker_tables(case):
    n = case["N"]
    assert np.array_equal(oracle.worker_hann(n).view(np.uint32), G.load(case["hann"], np.float32).view(np.uint32))
    c, s = oracle.worker_twiddles(n)
    gc, gs = G.load(case["twiddle_cos"], np.float64), G.load(case["twiddle_sin"], np.float64)
    # glibc vs V8 (fdlibm): at most one unit in the last place apart
    for a, b in ((c, gc), (s, gs)):
        ulp = np.abs(a.view(np.int64) - b.view(np.int64))
        assert ulp.max() <= 1
        assert (ulp != 0).mean() < 0.08


def test_worker_fft_raw():
    (case,) = G.cases("worker_fft")
    n, hop = case["N"], case["hop"]
    x = G.signal(case["signal"])
    ref = G.load(case["data"], np.float32).reshape(case["frames"], 2, n)
    tabs = [c for c in G.cases("worker_tables") if c["N"] == n][0]
    gc, gs = G.load(tabs["twiddle_cos"], np.float64), G.load(tabs["twiddle_sin"], np.float64)
    win = oracle.worker_hann(n)
    for f in range(case["frames"]):
        re0 = (x[f * hop:f * hop + n].astype(np.float64) * win.astype(np.float64)).astype(np.float32)
        for tw in ((gc, gs), (None, None)):          # V8's twiddles, then libm's
            re, im = oracle.worker_fft(re0, np.zeros(n, np.float32), *tw)
            assert np.array_equal(re.view(np.uint32), ref[f, 0].view(np.uint32))
            assert np.array_equal(im.view(np.uint32), ref[f, 1].view(np.uint32))


@pytest.mark.parametrize("case", G.cases("worker_render"), ids=lambda c: c["id"])
def test_worker_render(case):
    x = G.signal(case["signal"])
    n = case["N"] or 1024
    ov = case["overlap"] or 0
    lut = None
    if case["colormap"]:
        lut = oracle.worker_lut(G.colormap(case["colormap"]))
        assert np.array_equal(lut, G.load(case["lut"], np.uint8))
    hop, width, frames = oracle.worker_geometry(len(x), n, ov)
    assert (width, n // 2) == (case["width"], case["height"])
    rgba = oracle.worker_render(x, n, ov, lut)
    ref = G.load(case["rgba"], np.uint8).reshape(case["height"], case["width"], 4)
    assert np.array_equal(rgba, ref)
    assert np.array_equal(oracle.worker_render(x, n, ov, lut, threads=3), ref)


def test_worker_frame_count_quirk():
    """(L-N) % hop == 0: the loop runs one frame more than the image is wide and that frame lands
    in column 0 one row down (spectrogramWorker.js:40,59,79)."""
    x = G.signal("bat")
    hop, width, frames = oracle.worker_geometry(len(x), 1024, 75)
    assert (len(x) - 1024) % hop == 0 and frames == width + 1
    rgba, idx = oracle.worker_render(x, 1024, 75, want_index=True)
    H = 512
    # column 0: row H-y (y >= 1) shows the extra frame's bin y; the top row keeps frame 0
    assert np.array_equal(rgba[1:, 0, 0], idx[width, 1:][::-1])
    assert rgba[0, 0, 0] == idx[0, H - 1]
    assert np.array_equal(rgba[:, 1, 0], idx[1, ::-1])


def test_worker_contract_case():
    (case,) = G.cases("worker_contract")
    assert case["render_before_init_dropped"] is True
    lut = oracle.worker_lut(G.colormap("rainbow"))
    rgba = oracle.worker_render(G.signal(case["signal"]), case["N"], case["overlap"], lut)
    assert np.array_equal(rgba, G.load(case["rgba"], np.uint8).reshape(case["height"], case["width"], 4))


# ------------------------------------------------------------------ the JavaScript restatement (CPU baseline of bench.py)
@pytest.mark.skipif(__import__("shutil").which("node") is None, reason="node is not installed")
@pytest.mark.parametrize("case", G.cases("worker_render"), ids=lambda c: c["id"])
def test_js_worker_restatement_matches_reference_bytes(case, tmp_path):
    """oracle/worker_restatement.js (timed under worker_threads as bench.py's CPU baseline) renders the
    golden worker cases to the very bytes the reference produced."""
    import json
    import os
    import subprocess
    x = G.signal(case["signal"])
    n, ov = case["N"] or 1024, case["overlap"] or 0
    inp, outp, lutp = tmp_path / "x.f32", tmp_path / "o.rgba", "-"
    x.astype(np.float32).tofile(inp)
    if case["colormap"]:
        lutp = tmp_path / "lut.u8"
        G.load(case["lut"], np.uint8).tofile(lutp)
    js = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "worker_restatement.js")
    r = subprocess.run(["node", js, "render", str(inp), str(n), str(ov), str(lutp), str(outp)], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    dims = json.loads(r.stdout)
    assert (dims["width"], dims["height"]) == (case["width"], case["height"])
    got = np.fromfile(outp, np.uint8).reshape(case["height"], case["width"], 4)
    assert np.array_equal(got, G.load(case["rgba"], np.uint8).reshape(case["height"], case["width"], 4))


@pytest.mark.skipif(__import__("shutil").which("node") is None, reason="node is not installed")
def test_js_worker_restatement_bench_mode(tmp_path):
    import json
    import os
    import subprocess
    x = np.tile(G.signal("bat"), 8)
    inp = tmp_path / "x.f32"
    x.astype(np.float32).tofile(inp)
    js = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "worker_restatement.js")
    r = subprocess.run(["node", js, "bench", str(inp), "1024", "75", "2", "0.05", "384000"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["workers"] == 2 and d["chunks"] >= 2 and d["msamples_per_s"] > 0
    hop = 256
    assert d["samples"] == (len(x) - 1024) // hop * hop + hop or d["samples"] <= len(x)


# ------------------------------------------------------------------ non-linear frequency scales (filter banks)
SCALES = ("mel", "logarithmic", "bark", "erb")


def test_scale_conversions_and_filterbanks():
    (conv,) = G.cases("plugin_scale_conversions")
    for scale in SCALES:
        for row in conv["conv"][scale]:
            to = oracle.hz_to_scale(scale, row["hz"])
            assert abs(to - row["to"]) <= 4e-16 * abs(row["to"]) + 1e-300            # libm vs V8: at most an ulp or two
            assert abs(oracle.scale_to_hz(scale, row["to"]) - row["back"]) <= 2e-13 * abs(row["back"]) + 1e-12
    (banks,) = G.cases("plugin_filterbanks")
    for b in banks["banks"]:
        bin_, w0, w1 = oracle.plugin_filterbank(b["scale"], b["N"], b["sr"])
        rb, r0, r1 = G.load(b["bin"], np.int32), G.load(b["w0"], np.float64), G.load(b["w1"], np.float64)
        assert np.array_equal(bin_, rb), (b["scale"], b["N"])
        inside = rb + 1 <= b["N"] // 2                      # the reference's rows are N/2 + 1 long: taps past that are not stored
        assert np.abs(w0 - r0)[rb <= b["N"] // 2].max() <= 1e-9 and np.abs(w1 - r1)[inside].max() <= 1e-9


@pytest.mark.parametrize("case", G.cases("plugin_scale_frequencies"), ids=lambda c: c["id"])
def test_plugin_get_frequencies_scaled(case):
    x = G.signal(case["signal"])
    got = oracle.plugin_get_frequencies_scaled(x, case["N"], case["noverlap"], case["scale"], G.sample_rate(case["signal"]))
    ref = G.load(case["data"], np.uint8).reshape(case["frames"], case["bins"])
    assert got.shape == ref.shape
    # the filter weights go through log10/pow, where glibc and V8 may differ in the last bit of a double;
    # after the float32 accumulation and the quantiser that could move an index by one at most
    d = np.abs(got.astype(int) - ref.astype(int))
    d = np.minimum(d, 256 - d)
    assert d.max() <= 1 and (d != 0).mean() <= 1e-4


def test_scaled_crop_grid():
    (grid,) = G.cases("plugin_scale_crop_grid")
    for r in grid["rows"]:
        assert oracle.plugin_crop_scaled(r["N"] // 2, r["sr"], r["fmin"], r["fmax"], r["scale"]) == (r["sy"], r["sh"]), r
