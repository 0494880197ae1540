"""From per-frame spectral features to the data object SonoRadar's species matcher consumes.

The reference's ``autoIdHK(data)`` (modules/autoid_HK.js:215-264) matches a plain object with kHz /
ms fields against a rule table; in the browser those numbers come from markers the user places
(modules/autoIdPanel.js:1388-1468).  The batch engine measures them instead: the GPU emits one peak
per frame (``SONO_OUT_FEATURES``), ``segment_calls`` groups frames into calls, ``call_shape`` finds
where a call turns, and ``call_to_autoid_data`` converts bins and frames to the reference's units
with the panel's definitions (``sono_autoid_data_of_call``).

The matcher itself is one C function, ``sono_autoid_match``, shared with the Node host; this module
only flattens a rule table (passed in as data -- the table stays with the reference) into the
records it takes and joins the names it returns.
"""
import ctypes as C
import math
import re

import numpy as np

from . import _ffi as F

FIELDS = ["highestFreq", "lowestFreq", "kneeFreq", "heelFreq", "startFreq", "endFreq", "cfStart", "cfEnd", "duration",
          "bandwidth", "kneeLowTime", "kneeLowBandwidth", "heelLowBandwidth", "kneeHeelBandwidth"]   # autoid_HK.js:216-221
CF_CALL_TYPES = ("FM-CF-FM", "CF-FM")                                                                 # autoIdPanel.js:1417
_REL = re.compile(r"^(=|=>|>=|<|<=|>)\s*(\w+)$")                                                     # autoid_HK.js:204,233
_REL_CODE = {"=": 0, ">": 1, "<": 2, ">=": 3, "=>": 3, "<=": 4}
_ABSENT = -2 ** 31


def bin_to_khz(bin_index, fft_n, sample_rate):
    return bin_index * sample_rate / fft_n / 1000.0


def call_shape(features, call, first_frame=0):
    """sono_call_shape: knee, heel, CF part, first highest / lowest frame of one call (numpy record of
    F.SHAPE_DTYPE; frames are absolute, -1 = none).  `features` are the records from `first_frame` on."""
    f = np.ascontiguousarray(features, F.FEATURE_DTYPE)
    c = np.ascontiguousarray(call, F.CALL_DTYPE).reshape(1)
    out = np.zeros(1, F.SHAPE_DTYPE)
    rc = F.lib().sono_call_shape(F.ptr(f), len(f), int(first_frame), F.ptr(c), F.ptr(out))
    if rc != F.OK:
        raise F.SonoError(rc, "call lies outside the feature records")
    return out[0]


def call_to_autoid_data(call, fft_n, hop, sample_rate, call_type=None, harmonic=0, time_expansion=1.0, shape=None):
    """One sono_call record (and, optionally, its shape) -> the object autoIdHK reads.  Frequencies in kHz,
    times in ms; absent fields are left out.  `time_expansion` scales like the panel does for time-expanded
    recordings (autoIdPanel.js:1439-1462): frequencies multiply, durations divide."""
    c = np.ascontiguousarray(call, F.CALL_DTYPE).reshape(1)
    sh = None if shape is None else np.ascontiguousarray(shape, F.SHAPE_DTYPE).reshape(1)
    raw = F.AutoidData()
    rc = F.lib().sono_autoid_data_of_call(F.ptr(c), None if sh is None else F.ptr(sh), int(fft_n), int(hop), float(sample_rate),
                                          float(time_expansion), 1 if call_type in CF_CALL_TYPES else 0, C.byref(raw))
    if rc != F.OK:
        raise F.SonoError(rc, "bad call conversion arguments")
    data = {name: raw.field[i] for i, name in enumerate(FIELDS) if not math.isnan(raw.field[i])}
    te = float(time_expansion)
    data["peakFreq"] = bin_to_khz(int(c[0]["peak_bin"]), fft_n, sample_rate) * te
    data["startTime"] = int(c[0]["start_frame"]) * hop / sample_rate * 1000.0 / te
    data["harmonic"] = harmonic
    if call_type is not None:
        data["callType"] = call_type
    return data


def _js_truthy(v):
    """JavaScript truthiness of a rule entry: an empty array is truthy (autoid_HK.js:229-231 test `rule.harmonic &&` and
    `!rule[f]`), null / undefined / false / 0 / NaN / "" are not."""
    if v is None or v is False:
        return False
    if isinstance(v, (int, float)) and not isinstance(v, bool):
        return v == v and v != 0
    if isinstance(v, str):
        return v != ""
    return True


def _num(v):
    """A JavaScript value in a numeric comparison: null -> 0, undefined / non-numbers -> NaN."""
    if v is None:
        return 0.0
    if isinstance(v, bool):
        return float(v)
    if isinstance(v, (int, float)):
        return float(v)
    return float("nan")


class RuleTable:
    """A species rule table (the shape of autoid_HK.js:3-198) flattened for sono_autoid_match."""

    def __init__(self, species_rules):
        self.names = [sp["name"] for sp in species_rules]
        self.call_types = []
        rules, conds, intervals, harmonics = [], [], [], []
        for si, sp in enumerate(species_rules):
            for rule in sp["rules"]:
                mask = 0
                if rule.get("callType"):                                   # :225-228
                    for t in (s.strip() for s in rule["callType"].split(",")):
                        if t not in self.call_types:
                            self.call_types.append(t)
                        mask |= 1 << self.call_types.index(t)
                h_first, h_count = len(harmonics), 0
                c_first = len(conds)
                if _js_truthy(rule.get("harmonic")):                       # :229
                    harmonics.extend(int(h) for h in rule["harmonic"])
                    h_count = len(rule["harmonic"])
                    if h_count == 0:   # `[].includes(x)` is false for every x: such a rule never matches
                        conds.append((0, 2, 0, 0))
                for fi, name in enumerate(FIELDS):                         # :230-259
                    r = rule.get(name)
                    if not _js_truthy(r):
                        continue
                    if not isinstance(r, (list, tuple)) or len(r) == 0:
                        # an empty array (or a truthy non-array): `val >= undefined` is false for every value
                        conds.append((fi, 2, 0, 0))
                        continue
                    if isinstance(r[0], str) and len(r) == 1:
                        m = _REL.match(r[0])
                        if m and m.group(2) in FIELDS:
                            conds.append((fi, 1, _REL_CODE[m.group(1)], FIELDS.index(m.group(2))))
                        else:          # unknown other field: undefined fails; unparsed string: the comparison fails
                            conds.append((fi, 2, 0, 0))
                        continue
                    first = len(intervals)
                    for iv in (r if isinstance(r[0], (list, tuple)) else [r]):
                        lo = _num(iv[0]) if len(iv) > 0 else float("nan")
                        hi = _num(iv[1]) if len(iv) > 1 else float("nan")
                        intervals.append((lo, hi))
                    conds.append((fi, 0, first, len(intervals) - first))
                rules.append((mask, h_first, h_count, c_first, len(conds) - c_first, si, 0))
        if len(self.call_types) > 64:
            raise ValueError("more than 64 distinct call types")
        self._rules = np.array(rules, F.AUTOID_RULE_DTYPE) if rules else np.zeros(0, F.AUTOID_RULE_DTYPE)
        self._conds = np.array(conds, np.uint32).reshape(-1, 4) if conds else np.zeros((0, 4), np.uint32)
        self._intervals = np.array(intervals, np.float64).reshape(-1, 2) if intervals else np.zeros((0, 2), np.float64)
        self._harmonics = np.array(harmonics or [0], np.int32)

    def match(self, data):
        """autoIdHK(data): names joined with ' / ', or 'No species matched' (:261-263)."""
        data = data or {}
        raw = F.AutoidData()
        for i, name in enumerate(FIELDS):
            v = data.get(name)
            raw.field[i] = float("nan") if v is None or isinstance(v, str) else float(v)
        ct = data.get("callType")
        raw.call_type = self.call_types.index(ct) if ct in self.call_types else -1
        h = data.get("harmonic")
        ok_h = isinstance(h, (int, float)) and not isinstance(h, bool) and not (isinstance(h, float) and math.isnan(h)) \
            and float(h) == int(h)
        raw.harmonic = int(h) if ok_h else _ABSENT
        matched = np.zeros(max(1, len(self.names)), np.uint8)
        rc = F.lib().sono_autoid_match(F.ptr(self._rules), len(self._rules), F.ptr(self._conds), F.ptr(self._intervals),
                                       F.ptr(self._harmonics), C.byref(raw), F.ptr(matched), len(self.names))
        if rc != F.OK:
            raise F.SonoError(rc, "bad rule table")
        names = [n for n, m in zip(self.names, matched) if m]
        return " / ".join(names) if names else "No species matched"


_TABLES = {}


def auto_id_hk(data, species_rules):
    """autoIdHK, autoid_HK.js:215-264, over a caller-supplied rule table (flattened once per table object)."""
    key = id(species_rules)
    t = _TABLES.get(key)
    if t is None or t[0] is not species_rules:
        t = (species_rules, RuleTable(species_rules))
        _TABLES[key] = t
    return t[1].match(data)
