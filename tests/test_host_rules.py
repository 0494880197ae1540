"""CPU-only checks of libsonospec.so's host side: the library loads without a GPU, exports every
symbol of include/sonospec.h, and its host rules (geometry, overlap, crop, LUTs, tables) agree
with the reference outputs in tests/golden/."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden_store
from spectrogram_amd import _ffi as F
from spectrogram_amd import engine as E
from spectrogram_amd.plugin import Spectrogram, createSpectrogramPlugin, noverlap_from_percent

G = golden_store()
WINDOWS = ["bartlett", "bartlettHann", "blackman", "cosine", "gauss", "hamming", "hann", "lanczoz",
           "rectangular", "triangular"]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sonospec.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sono_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = F.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(F.SYMBOLS), declared ^ set(F.SYMBOLS)
    assert lib.sono_abi_version() == 1
    assert lib.sono_device_count() >= 0


def test_config_struct_layout():
    import ctypes as C
    assert C.sizeof(F.SonoConfig) == 88 and F.SonoConfig.alpha.offset == 48
    assert C.sizeof(F.SonoGeometry) == 24


def test_windows_match_reference():
    for case in G.cases("plugin_tables"):
        n = case["N"]
        ref = G.load(case["window_values"], np.float32).reshape(len(WINDOWS), n)
        for i, w in enumerate(WINDOWS):
            got = E.window_table(F.WINDOW_IDS[w], n)
            assert np.array_equal(got.view(np.uint32), ref[i].view(np.uint32)), (w, n)
    for case in G.cases("worker_tables"):
        got = E.window_table(F.WIN_WORKER_HANN, case["N"])
        assert np.array_equal(got.view(np.uint32), G.load(case["hann"], np.float32).view(np.uint32))
    (case,) = G.cases("plugin_alpha_windows")
    ref = G.load(case["window_values"], np.float32).reshape(len(case["alphas"]), 2, case["N"])
    for i, al in enumerate(case["alphas"]):
        assert np.array_equal(E.window_table(F.WINDOW_IDS["blackman"], case["N"], al), ref[i, 0])
        assert np.array_equal(E.window_table(F.WINDOW_IDS["gauss"], case["N"], al), ref[i, 1])
    with pytest.raises(F.SonoError):
        E.window_table(99, 512)


def test_unknown_window_message():
    (case,) = G.cases("plugin_window_error")
    with pytest.raises(ValueError) as e:
        E.window_id(case["name"])
    assert str(e.value) == case["message"]


def test_twiddle_tables():
    for case in G.cases("worker_tables"):
        n = case["N"]
        c, s = E.twiddle_table(F.TW_ACCURATE, n)
        for a, b in ((c, G.load(case["twiddle_cos"], np.float64)), (s, G.load(case["twiddle_sin"], np.float64))):
            assert np.abs(a.view(np.int64) - b.view(np.int64)).max() <= 1     # libm vs V8: one ulp
    # plugin recurrence: seeds are the float32 tables, first two entries of each stage are exact
    for case in G.cases("plugin_tables"):
        n = case["N"]
        c, s = E.twiddle_table(F.TW_PLUGIN_RECURRENCE, n)
        gs, gc = G.load(case["sin_table"], np.float32), G.load(case["cos_table"], np.float32)
        T = 1
        while T < n:
            assert c[T - 1] == 1.0 and s[T - 1] == 0.0
            if T > 1:
                assert c[T] == float(gc[T]) and s[T] == float(gs[T])
            T *= 2


def test_geometry_rules():
    for case in G.cases("plugin_get_frequencies"):
        L = G.manifest["signals"][case["signal"]]["length"]
        ov = E.plugin_noverlap(case["N"], case["noverlap"] or 0, L, case["canvas_width"])
        hop = case["N"] - ov
        g = E.geometry_for(case["N"], hop, F.FRAMES_PLUGIN_STRICT, L)
        assert (g.frames, g.width, g.bins) == (case["frames"], case["frames"], case["bins"]), case["id"]
    for case in G.cases("worker_render"):
        L = G.manifest["signals"][case["signal"]]["length"]
        n = case["N"] or 1024
        hop = E.worker_hop(n, case["overlap"] or 0)
        g = E.geometry_for(n, hop, F.FRAMES_WORKER_COMPAT, L)
        assert (g.width, g.bins) == (case["width"], case["height"]), case["id"]
        assert g.frames == ((L - n) // hop + 1 if L >= n else 0)
    assert E.worker_hop(1024, 99.95) == 1 and E.worker_hop(1024, 0) == 1024 and E.worker_hop(1024, 75) == 256
    assert noverlap_from_percent(1024, 75) == 768 and noverlap_from_percent(1024, None) is None
    assert noverlap_from_percent(1024, 99) == 1013
    with pytest.raises(F.SonoError):
        E.geometry_for(1000, 250, 0, 5000)
    with pytest.raises(F.SonoError):
        E.geometry_for(1024, 0, 0, 5000)


def test_crop_grid():
    (case,) = G.cases("plugin_crop_grid")
    for r in case["rows"]:
        assert E.plugin_crop(r["N"] // 2, r["sr"], r["fmin"], r["fmax"]) == (r["sy"], r["sh"]), r
    assert E.plugin_crop(2048, 384000, 15000, 120000) == (768, 1120)


def test_luts():
    for case in G.cases("plugin_render"):
        if case["colormap"] in ("default", "rainbow"):
            assert np.array_equal(E.plugin_lut(G.colormap(case["colormap"])), G.load(case["lut"], np.uint8))
    for case in G.cases("worker_render"):
        if case["colormap"]:
            assert np.array_equal(E.worker_lut(G.colormap(case["colormap"])), G.load(case["lut"], np.uint8))
    cm = G.colormap("default").copy()
    cm[3, 0] = np.nan
    cm[5, 3] = np.nan
    lut = E.worker_lut(cm).reshape(256, 4)
    assert lut[3, 0] == 0 and lut[5, 3] == 255


def test_plugin_option_validation():
    (case,) = G.cases("plugin_errors")
    msgs = {e["label"]: e["message"] for e in case["errors"]}
    with pytest.raises(ValueError) as e:
        Spectrogram.create({"colorMap": [[0, 0, 0, 1]] * 10})
    assert str(e.value) == msgs["short"]
    with pytest.raises(ValueError) as e:
        Spectrogram.create({"colorMap": [[0, 0, 0]] * 256})
    assert str(e.value) == msgs["bad_entry"]
    with pytest.raises(ValueError) as e:
        Spectrogram.create({})
    assert str(e.value) == msgs["default_name"]
    with pytest.raises(ValueError) as e:
        Spectrogram.create({"colorMap": "viridis"})
    assert str(e.value) == msgs["unknown_name"]
    p = Spectrogram.create({"colorMap": "gray"})
    assert (p.fftSamples, p.height, p.noverlap, p.windowFunc, p.gainDB, p.rangeDB, p.scale) == \
        (512, 200, None, "hann", 20, 80, "mel")
    p = Spectrogram.create({"colorMap": "gray", "gainDB": 0, "rangeDB": 100, "noverlap": 0})
    assert (p.gainDB, p.rangeDB, p.noverlap) == (0, 100, None)
    q = createSpectrogramPlugin(G.colormap("default").tolist(), noverlap=768)
    assert (q.frequencyMin, q.frequencyMax, q.scale, q.fftSamples, q.height, q.noverlap) == \
        (10000, 128000, "linear", 1024, 800, 768)


def test_create_without_gpu_reports_cleanly():
    if F.lib().sono_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(F.SonoError) as e:
        E.Engine()
    assert "no HIP device" in str(e.value)
    # argument validation comes before device probing
    with pytest.raises(F.SonoError) as e:
        E.Engine(fft_n=1000)
    assert str(e.value) == "Invalid buffer size, must be a power of 2."
    with pytest.raises(F.SonoError):
        E.Engine(hop=0)
    with pytest.raises(F.SonoError):
        E.Engine(window_id=77)


# ------------------------------------------------------------------ non-linear frequency scales
def test_scale_rules_match_reference():
    (conv,) = G.cases("plugin_scale_conversions")
    for scale, rows in conv["conv"].items():
        for row in rows:
            assert abs(E.hz_to_scale(scale, row["hz"]) - row["to"]) <= 4e-16 * abs(row["to"]) + 1e-300
            assert abs(E.scale_to_hz(scale, row["to"]) - row["back"]) <= 2e-13 * abs(row["back"]) + 1e-12
    assert E.hz_to_scale("linear", 1234.5) == 1234.5
    (banks,) = G.cases("plugin_filterbanks")
    for b in banks["banks"]:
        bin_, w0, w1 = E.filterbank_table(b["scale"], b["N"], b["sr"])
        rb, r0, r1 = G.load(b["bin"], np.int32), G.load(b["w0"], np.float64), G.load(b["w1"], np.float64)
        assert np.array_equal(bin_, rb)
        assert np.abs(w0 - r0)[rb <= b["N"] // 2].max() <= 1e-9 and np.abs(w1 - r1)[rb + 1 <= b["N"] // 2].max() <= 1e-9
    (grid,) = G.cases("plugin_scale_crop_grid")
    for r in grid["rows"]:
        assert E.plugin_crop_scaled(r["N"] // 2, r["sr"], r["fmin"], r["fmax"], r["scale"]) == (r["sy"], r["sh"]), r
    assert E.plugin_crop_scaled(512, 384000, 10000, 128000, "linear") == E.plugin_crop(512, 384000, 10000, 128000)


def test_canvas_placement_rules():
    """drawImage rectangle and background of drawSpectrogram (spectrogram.esm.js:357-361,377-381)."""
    import oracle
    # frequencyMax at or below Nyquist: the rows fill the canvas, no fill
    f = 128000 / 192000
    assert E.plugin_draw_rect(384000, 10000, 128000, "linear", 800) == (800 * (1 - f / f), 800 * f / f, False)   # s*p/f, left to right
    assert E.plugin_draw_rect(384000, 0, 0, "linear", 200) == (0.0, 200.0, False)
    # above Nyquist: p = 1, f = fmax / nyquist; the image sits at the bottom, the top band keeps the fill
    dy, dh, fill = E.plugin_draw_rect(256000, 10000, 160000, "linear", 800)
    assert fill and dh == 800 * 1 / (160000 / 128000) and dy == 800 * (1 - 1 / (160000 / 128000))
    for scale in ("linear", "mel", "logarithmic", "bark", "erb"):
        for sr, f0, f1, h in [(384000, 10000, 128000, 800), (256000, 0, 200000, 512), (44100, 100, 30000, 200), (48000, 0, 0, 37)]:
            assert E.plugin_draw_rect(sr, f0, f1, scale, h) == oracle.plugin_draw_rect(sr, f0, f1, scale, h), (scale, sr, f1)
    # "rgba(r, g, b, a)" built from 0..1 colour-map numbers: channels round to 0 or 1
    assert E.plugin_background([0.98, 0.5, 0.2, 1]).tolist() == [1, 1, 0, 255]
    assert E.plugin_background([0.49, 300, -4, 0.5]).tolist() == [0, 255, 0, 128]
    assert E.plugin_background([float("nan"), 0, 0, 1]).tolist() == [0, 0, 0, 255]
    for last in ([0.98, 0.5, 0.2, 1], [0.49, 300, -4, 0.5], [float("nan"), 0, 0, 1], [0, 0, 0, 0]):
        assert np.array_equal(E.plugin_background(last), oracle.plugin_background(last))


def test_canvas_filter_definition_properties():
    """The checker's restatement of the canvas filter: identity placement copies rows, a two-fold stretch
    interpolates half way between neighbouring rows, uncovered rows keep the background."""
    import oracle
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (40, 9, 4), dtype=np.uint8)
    img[..., 3] = 255
    assert np.array_equal(oracle.canvas_draw(img, 5, 30, 0.0, 30.0, 30), img[5:35])
    bg = np.array([9, 8, 7, 255], np.uint8)
    out = oracle.canvas_draw(img, 0, 40, 10.0, 80.0, 100, bg)
    assert (out[:10] == bg).all() and (out[90:] == bg).all()
    # canvas row 10 + 2k + 1 has its centre 0.25 / 0.75 of the way between rows k and k + 1
    a, b = img[3].astype(np.float32), img[4].astype(np.float32)
    assert np.array_equal(out[10 + 7], np.floor(a + (b - a) * np.float32(0.25) + np.float32(0.5)).astype(np.uint8))
    assert np.array_equal(out[10 + 8], np.floor(a + (b - a) * np.float32(0.75) + np.float32(0.5)).astype(np.uint8))
    assert np.array_equal(out[10], img[0]) and np.array_equal(out[89], img[39])     # clamped edges
