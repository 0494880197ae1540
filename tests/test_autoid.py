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


def _track(bins, idx=200, pad=3):
    """Feature records of one call (peak bins `bins`) with `pad` quiet frames on either side."""
    n = len(bins) + 2 * pad
    f = np.zeros(n, F.FEATURE_DTYPE)
    f["peak_bin"] = 50
    f["peak_bin"][pad:pad + len(bins)] = bins
    f["peak_idx"][pad:pad + len(bins)] = idx
    return f


def test_call_shape_knee_heel_and_cf_part():
    from spectrogram_amd.autoid import call_shape
    # FM-QCF: a steep sweep (10 bins a frame) that flattens at frame 5 and droops again from frame 13 on
    bins = [200, 190, 180, 170, 160, 150, 148, 147, 146, 146, 145, 145, 144, 144, 130, 110]
    f = _track(bins)
    f["peak_idx"][3 + 7] = 230                            # the loudest frame sits on the flat part
    (c,) = E.segment_calls(f, 100, 1, first_frame=500)
    s = call_shape(f, c, first_frame=500)
    assert (s["high_frame"], s["low_frame"]) == (503, 518)
    # slopes over w = 16 // 8 = 2 frames: |before| - |after| is largest at frame 5 of the call (20 - 3), first wins
    assert (s["knee_frame"], s["knee_bin"]) == (503 + 5, 150)
    # steepening |after| - |before| at or after the knee: frame 13 (34 - 1)
    assert (s["heel_frame"], s["heel_bin"]) == (503 + 13, 144)
    # the run within one bin of the loudest frame's peak (147): frames 6..9 = 4 frames = a quarter of 16
    assert (s["cf_start_frame"], s["cf_end_frame"], s["cf_start_bin"], s["cf_end_bin"]) == (509, 512, 148, 146)
    d = call_to_autoid_data(c, 1024, 256, 384000, call_type="FM-QCF", shape=s)
    k = 0.375
    assert abs(d["kneeFreq"] - 150 * k) < 1e-12 and abs(d["heelFreq"] - 144 * k) < 1e-12
    assert abs(d["kneeLowBandwidth"] - (150 - 110) * k) < 1e-12 and abs(d["heelLowBandwidth"] - (144 - 110) * k) < 1e-12
    assert abs(d["kneeHeelBandwidth"] - 6 * k) < 1e-12
    assert abs(d["kneeLowTime"] - (508 - 518) * 256 / 384000 * 1000) < 1e-12      # knee.time - low.time (autoIdPanel.js:1423-1426)
    assert abs(d["bandwidth"] - 90 * k) < 1e-12                                   # high - low for a non-CF type
    # CF-FM: 20 frames at bin 290 (+-1), then a five-frame sweep down; bandwidth = cfStart - end (:1416-1418)
    bins = [290, 291, 290, 290, 289, 290, 290, 291, 290, 290, 290, 290, 289, 290, 290, 290, 291, 290, 290, 290, 280, 268, 255, 240, 226]
    f = _track(bins)
    (c,) = E.segment_calls(f, 100, 1)
    s = call_shape(f, c)
    assert (s["cf_start_frame"], s["cf_end_frame"], s["cf_start_bin"], s["cf_end_bin"]) == (3, 22, 290, 290)
    d = call_to_autoid_data(c, 1024, 256, 384000, call_type="CF-FM", shape=s)
    assert abs(d["cfStart"] - 290 * k) < 1e-12 and abs(d["bandwidth"] - (290 - 226) * k) < 1e-12
    d_fm = call_to_autoid_data(c, 1024, 256, 384000, call_type="FM", shape=s)
    assert abs(d_fm["bandwidth"] - (291 - 226) * k) < 1e-12
    # without a shape the eight shape-derived fields stay absent, and so does a CF type's bandwidth
    d0 = call_to_autoid_data(c, 1024, 256, 384000, call_type="CF-FM")
    assert "cfStart" not in d0 and "kneeFreq" not in d0 and "bandwidth" not in d0
    # a four-frame call has no knee; a call outside the records is refused
    f = _track([200, 180, 160, 150])
    (c,) = E.segment_calls(f, 100, 1)
    s = call_shape(f, c)
    assert s["knee_frame"] == -1 and s["heel_frame"] == -1 and s["cf_start_frame"] == -1
    import pytest
    with pytest.raises(F.SonoError):
        call_shape(f[:5], c)


def test_matcher_semantics_beyond_the_reference_table():
    """Forms autoIdHK accepts that the reference's own table does not use: lists of ranges, relations between
    fields, unparsable one-string conditions, absent values, harmonic lists (autoid_HK.js:200-259)."""
    rules = [
        {"name": "ranges", "rules": [{"duration": [[1, 2], [5, 6]]}]},
        {"name": "relation", "rules": [{"startFreq": ["> endFreq"], "kneeFreq": ["=> heelFreq"]}]},
        {"name": "never", "rules": [{"duration": ["about 5"]}]},
        {"name": "typed", "rules": [{"callType": "FM, QCF", "harmonic": [0, 2], "cfStart": [60, 70]}]},
        {"name": "two", "rules": [{"duration": [100, 200]}, {"bandwidth": [3, None]}, {"lowestFreq": [None, 20]}]},
        {"name": "unknown-field", "rules": [{"duration": ["< nothing"]}]},
    ]
    m = lambda d: auto_id_hk(d, rules)
    assert m({"duration": 1.5}) == "ranges" and m({"duration": 5}) == "ranges" and m({"duration": 3}) == "No species matched"
    assert m({"startFreq": 50, "endFreq": 40, "kneeFreq": 45, "heelFreq": 45}) == "relation"
    assert m({"startFreq": 50, "endFreq": 40, "kneeFreq": 44, "heelFreq": 45}) == "No species matched"
    assert m({"startFreq": 50, "endFreq": 40, "kneeFreq": 45}) == "No species matched"                 # absent other field
    assert m({"startFreq": 50, "endFreq": float("nan"), "kneeFreq": 45, "heelFreq": 45}) == "No species matched"
    assert m({"duration": 5.0}) == "ranges"                                                            # "never" never matches
    assert m({"callType": "QCF", "harmonic": 2, "cfStart": 65}) == "typed"
    assert m({"callType": "QCF", "harmonic": 1, "cfStart": 65}) == "No species matched"
    assert m({"callType": "CF", "harmonic": 2, "cfStart": 65}) == "No species matched"
    assert m({"harmonic": 2, "cfStart": 65}) == "No species matched"                                   # rule has callType, data has none
    assert m({"callType": "FM", "cfStart": 65}) == "No species matched"                                # harmonic absent
    assert m({"duration": 150}) == "two"
    assert m({"bandwidth": 9}) == "No species matched"          # [3, null]: val <= null is val <= 0
    assert m({"lowestFreq": 10}) == "two" and m({"lowestFreq": -1}) == "No species matched"           # [null, 20]: 0 <= val <= 20
    assert m({}) == "No species matched"
    # JavaScript truthiness: an empty array is truthy, so `harmonic: []` makes `[].includes(x)` false for every x and
    # `duration: []` compares against undefined bounds -- such rules never match (autoid_HK.js:229-231,211-212); a 0, a
    # null or an empty string in their place is falsy and leaves the field unconstrained
    empty = [
        {"name": "no-harmonic", "rules": [{"harmonic": [], "duration": [1, 9]}]},
        {"name": "no-range", "rules": [{"duration": []}]},
        {"name": "falsy", "rules": [{"duration": 0, "bandwidth": None, "callType": "", "lowestFreq": [10, 20]}]},
    ]
    e = lambda d: auto_id_hk(d, empty)
    assert e({"duration": 5, "harmonic": 0}) == "No species matched"
    assert e({"duration": 5}) == "No species matched"
    assert e({"lowestFreq": 15}) == "falsy" and e({"lowestFreq": 15, "duration": 1e9, "bandwidth": -3}) == "falsy"
