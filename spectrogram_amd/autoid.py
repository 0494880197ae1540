"""From per-frame spectral features to the data object SonoRadar's species matcher consumes.

The reference's ``autoIdHK(data)`` (modules/autoid_HK.js:215-264) matches a plain object with kHz /
ms fields against a rule table; in the browser those numbers come from markers the user places
(modules/autoIdPanel.js:1388-1468).  The batch engine measures them instead: the GPU emits one peak
per frame (``SONO_OUT_FEATURES``), ``segment_calls`` groups frames into calls, and
``call_to_autoid_data`` converts bins and frames to the reference's units with its own definitions
(duration = (t_max - t_min) * 1000, bandwidth = high - low, autoIdPanel.js:1406-1421).

``auto_id_hk`` restates the matcher's semantics over a rule table passed in as data (the table
itself stays with the reference), so hosts can check that what they produce is a valid input.
"""
import math
import re

FIELDS = ["highestFreq", "lowestFreq", "kneeFreq", "heelFreq", "startFreq", "endFreq", "cfStart", "cfEnd", "duration",
          "bandwidth", "kneeLowTime", "kneeLowBandwidth", "heelLowBandwidth", "kneeHeelBandwidth"]   # autoid_HK.js:216-221
_REL = re.compile(r"^(=|=>|>=|<|<=|>)\s*(\w+)$")                                                       # :203,233


def bin_to_khz(bin_index, fft_n, sample_rate):
    return bin_index * sample_rate / fft_n / 1000.0


def call_to_autoid_data(call, fft_n, hop, sample_rate, call_type=None, harmonic=0, time_expansion=1.0):
    """One sono_call record -> the object autoIdHK reads.  Frequencies in kHz, times in ms.
    `time_expansion` scales like the panel does for time-expanded recordings (autoIdPanel.js:1439-1462):
    frequencies multiply, durations divide."""
    te = float(time_expansion)
    frames = int(call["end_frame"]) - int(call["start_frame"]) + 1
    high = bin_to_khz(int(call["high_bin"]), fft_n, sample_rate) * te
    low = bin_to_khz(int(call["low_bin"]), fft_n, sample_rate) * te
    data = {
        "highestFreq": high,
        "lowestFreq": low,
        "startFreq": bin_to_khz(int(call["start_bin"]), fft_n, sample_rate) * te,
        "endFreq": bin_to_khz(int(call["end_bin"]), fft_n, sample_rate) * te,
        "peakFreq": bin_to_khz(int(call["peak_bin"]), fft_n, sample_rate) * te,
        "duration": frames * hop / sample_rate * 1000.0 / te,
        "bandwidth": high - low,
        "harmonic": harmonic,
        "startTime": int(call["start_frame"]) * hop / sample_rate * 1000.0 / te,
    }
    if call_type is not None:
        data["callType"] = call_type
    return data


def _is_missing(v):
    return v is None or (isinstance(v, float) and math.isnan(v))


def _in_range(val, rng):                                            # inRange, autoid_HK.js:200-213
    if _is_missing(val):
        return False
    if isinstance(rng[0], (list, tuple)):
        return any(_in_range(val, r) for r in rng)
    lo, hi = rng[0], rng[1]
    return lo <= val <= hi


def auto_id_hk(data, species_rules):
    """autoIdHK, autoid_HK.js:215-264, over a caller-supplied rule table."""
    data = data or {}

    def rule_ok(rule):
        if rule.get("callType"):                                     # :225-228
            types = [s.strip() for s in rule["callType"].split(",")]
            if data.get("callType") not in types:
                return False
        if rule.get("harmonic") and data.get("harmonic") not in rule["harmonic"]:   # :229
            return False
        for f in FIELDS:                                             # :230-259
            r = rule.get(f)
            if not r:
                continue
            if isinstance(r[0], str) and len(r) == 1:
                m = _REL.match(r[0])
                if m:
                    op, ref = m.group(1), m.group(2)
                    val, ref_val = data.get(f), data.get(ref)
                    if _is_missing(val) or _is_missing(ref_val):
                        return False
                    ok = {"=": val == ref_val, ">": val > ref_val, "<": val < ref_val,
                          ">=": val >= ref_val, "=>": val >= ref_val, "<=": val <= ref_val}[op]
                    if not ok:
                        return False
                    continue
            if not _in_range(data.get(f), r):
                return False
        return True

    names = [sp["name"] for sp in species_rules if any(rule_ok(rule) for rule in sp["rules"])]
    return " / ".join(names) if names else "No species matched"     # :261-263
