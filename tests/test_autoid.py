"""Species-matcher semantics (modules/autoid_HK.js) and the feature -> data conversion (CPU only)."""
import numpy as np

from conftest import golden_store
from spectrogram_amd import _ffi as F
from spectrogram_amd import engine as E
from spectrogram_amd.autoid import auto_id_hk, call_to_autoid_data

G = golden_store()


def test_matcher_reproduces_reference_verdicts():
    rules = G.manifest["species_rules"]
    assert len(rules) == 13
    n = 0
    for v in G.manifest["autoid_verdicts"]:
        data = {k: (float("nan") if x == "NaN" else x) for k, x in v["data"].items()}
        assert auto_id_hk(data, rules) == v["result"], v
        n += 1
    assert n > 700


def test_segment_calls_and_units():
    f = np.zeros(40, F.FEATURE_DTYPE)
    f["peak_bin"] = 100
    idx = np.zeros(40, np.uint8)
    idx[5:12] = [90, 120, 200, 200, 150, 110, 95]        # one call, peak at frame 7 (first of the two 200s)
    idx[20] = 99                                         # a one-frame blip
    idx[36:40] = 130                                     # a run touching the last frame
    f["peak_idx"] = idx
    f["peak_bin"][5:12] = [213, 200, 180, 170, 160, 150, 147]
    calls = E.segment_calls(f, threshold=80, min_frames=2, first_frame=1000)
    assert len(calls) == 2
    c = calls[0]
    assert (c["start_frame"], c["end_frame"], c["peak_frame"]) == (1005, 1011, 1007)
    assert (c["start_bin"], c["end_bin"], c["high_bin"], c["low_bin"], c["peak_bin"], c["peak_idx"]) == (213, 147, 213, 147, 180, 200)
    assert (calls[1]["start_frame"], calls[1]["end_frame"]) == (1036, 1039)
    assert len(E.segment_calls(f, threshold=80, min_frames=1)) == 3
    d = call_to_autoid_data(c, 1024, 256, 384000, call_type="FM", harmonic=0)
    assert abs(d["highestFreq"] - 213 * 0.375) < 1e-12 and abs(d["lowestFreq"] - 147 * 0.375) < 1e-12
    assert abs(d["duration"] - 7 * 256 / 384000 * 1000) < 1e-12
    assert abs(d["bandwidth"] - (213 - 147) * 0.375) < 1e-12
    # what the engine measures is a valid matcher input
    out = auto_id_hk(d, G.manifest["species_rules"])
    assert isinstance(out, str) and out
    # time expansion x10: frequencies scale up, durations down (autoIdPanel.js:1439-1462)
    d10 = call_to_autoid_data(c, 1024, 256, 38400, time_expansion=10)
    assert abs(d10["highestFreq"] - d["highestFreq"]) < 1e-9 and abs(d10["duration"] - d["duration"]) < 1e-9
