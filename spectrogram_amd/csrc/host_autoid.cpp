// Host side of the feature pass: the shape of a call from its per-frame peak track, the unit conversions of the
// reference's auto-ID panel, and the species matcher over a rule table passed in as data.  No GPU code here.
//
// The matcher follows autoIdHK (modules/autoid_HK.js:200-264); the field definitions follow runPulseId
// (modules/autoIdPanel.js:1388-1468).  What the reference does NOT have is a way to find knee, heel and the
// constant-frequency part of a call in a spectrum -- its panel reads markers the user drags -- so
// sono_call_shape's definitions are this library's own (parity unpinned; tested against hand-made tracks).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limits>

#include "sonospec.h"

namespace {

inline bool missing(double v) { return v != v; }  // NaN stands for JavaScript's null / undefined / NaN

// inRange's [min, max] test and autoIdHK's relation operators on present values
bool relation_holds(uint32_t op, double val, double ref) {
  switch (op) {
    case SONO_REL_EQ: return val == ref;
    case SONO_REL_GT: return val > ref;
    case SONO_REL_LT: return val < ref;
    case SONO_REL_GE: return val >= ref;
    case SONO_REL_LE: return val <= ref;
    default: return false;
  }
}

bool cond_holds(const sono_autoid_cond& c, const double* intervals, const sono_autoid_data& d) {
  if (c.field >= SONO_AUTOID_FIELDS) return false;
  const double val = d.field[c.field];
  switch (c.kind) {
    case SONO_COND_RANGES: {  // inRange with one [min, max] or a list of them (autoid_HK.js:200-213)
      if (missing(val)) return false;
      for (uint32_t i = 0; i < c.count; ++i) {
        const double lo = intervals[2 * (c.first + i)], hi = intervals[2 * (c.first + i) + 1];
        if (val >= lo && val <= hi) return true;  // a NaN bound fails like JavaScript's comparison with undefined
      }
      return false;
    }
    case SONO_COND_RELATION: {  // "op otherField" (:232-256): both values must be present
      if (c.count >= SONO_AUTOID_FIELDS) return false;
      const double ref = d.field[c.count];
      if (missing(val) || missing(ref)) return false;
      return relation_holds(c.first, val, ref);
    }
    default:  // SONO_COND_NEVER: a one-string condition the reference's pattern does not parse compares a number
      return false;  // with that string and fails
  }
}

}  // namespace

extern "C" {

int sono_autoid_match(const sono_autoid_rule* rules, uint32_t n_rules, const sono_autoid_cond* conds, const double* intervals,
                      const int32_t* harmonics, const sono_autoid_data* data, uint8_t* matched, uint32_t n_species) {
  if ((!rules && n_rules) || !data || (!matched && n_species)) return SONO_ERR_ARG;
  for (uint32_t s = 0; s < n_species; ++s) matched[s] = 0;
  for (uint32_t r = 0; r < n_rules; ++r) {
    const sono_autoid_rule& rule = rules[r];
    if (rule.species >= n_species) return SONO_ERR_ARG;
    if (matched[rule.species]) continue;  // species.rules.some(...)
    if (rule.call_type_mask) {            // rule.callType: data.callType must be one of the listed types
      if (data->call_type < 0 || data->call_type >= 64 || !((rule.call_type_mask >> data->call_type) & 1u)) continue;
    }
    if (rule.harmonic_count) {            // rule.harmonic.includes(data.harmonic)
      bool in = false;
      for (uint32_t i = 0; i < rule.harmonic_count && !in; ++i)
        in = data->harmonic != SONO_AUTOID_ABSENT && harmonics[rule.harmonic_first + i] == data->harmonic;
      if (!in) continue;
    }
    bool ok = true;                       // fields.every(...)
    for (uint32_t i = 0; i < rule.cond_count && ok; ++i) ok = cond_holds(conds[rule.cond_first + i], intervals, *data);
    if (ok) matched[rule.species] = 1;
  }
  return SONO_OK;
}

int sono_call_shape(const sono_frame_feature* f, uint64_t frames, uint64_t first_frame, const sono_call* call, sono_call_shape_t* out) {
  if (!f || !call || !out) return SONO_ERR_ARG;
  if (call->start_frame < first_frame || call->end_frame < call->start_frame || call->end_frame - first_frame >= frames)
    return SONO_ERR_ARG;
  const sono_frame_feature* t = f + (call->start_frame - first_frame);   // the call's track
  const int64_t n = (int64_t)(call->end_frame - call->start_frame) + 1;
  sono_call_shape_t s;
  s.high_frame = s.low_frame = -1;
  s.knee_frame = s.heel_frame = s.cf_start_frame = s.cf_end_frame = -1;
  s.knee_bin = s.heel_bin = s.cf_start_bin = s.cf_end_bin = 0;
  // first frames holding the highest and the lowest peak
  int32_t hi = t[0].peak_bin, lo = t[0].peak_bin;
  int64_t hi_i = 0, lo_i = 0;
  for (int64_t i = 1; i < n; ++i) {
    if (t[i].peak_bin > hi) { hi = t[i].peak_bin; hi_i = i; }
    if (t[i].peak_bin < lo) { lo = t[i].peak_bin; lo_i = i; }
  }
  s.high_frame = (int64_t)call->start_frame + hi_i;
  s.low_frame = (int64_t)call->start_frame + lo_i;
  // constant-frequency part: the longest run of frames whose peak stays within one bin of the peak at the call's
  // loudest frame (the first such run wins); it counts as a CF part if it is at least 3 frames and a quarter of the call
  {
    const int32_t ref = call->peak_bin;
    int64_t best_len = 0, best_at = 0, at = 0, len = 0;
    for (int64_t i = 0; i < n; ++i) {
      if (std::abs(t[i].peak_bin - ref) <= 1) {
        if (len == 0) at = i;
        if (++len > best_len) { best_len = len; best_at = at; }
      } else {
        len = 0;
      }
    }
    if (best_len >= 3 && 4 * best_len >= n) {
      s.cf_start_frame = (int64_t)call->start_frame + best_at;
      s.cf_end_frame = (int64_t)call->start_frame + best_at + best_len - 1;
      s.cf_start_bin = t[best_at].peak_bin;
      s.cf_end_bin = t[best_at + best_len - 1].peak_bin;
    }
  }
  // knee: the frame where the sweep flattens most; heel: the frame at or after the knee where it steepens most.
  // Slopes are peak-bin differences over w = max(1, n / 8) frames on either side (integers, first frame wins).
  if (n >= 5) {
    const int64_t w = n / 8 > 1 ? n / 8 : 1;
    int64_t knee = -1, heel = -1;
    int32_t knee_gain = 0, heel_gain = 0;
    for (int64_t i = w; i + w < n; ++i) {
      const int32_t before = std::abs(t[i].peak_bin - t[i - w].peak_bin), after = std::abs(t[i + w].peak_bin - t[i].peak_bin);
      if (before - after > knee_gain) { knee_gain = before - after; knee = i; }
    }
    if (knee >= 0) {
      for (int64_t i = knee; i + w < n; ++i) {
        if (i < w) continue;
        const int32_t before = std::abs(t[i].peak_bin - t[i - w].peak_bin), after = std::abs(t[i + w].peak_bin - t[i].peak_bin);
        if (after - before > heel_gain) { heel_gain = after - before; heel = i; }
      }
      s.knee_frame = (int64_t)call->start_frame + knee;
      s.knee_bin = t[knee].peak_bin;
    }
    if (heel >= 0) {
      s.heel_frame = (int64_t)call->start_frame + heel;
      s.heel_bin = t[heel].peak_bin;
    }
  }
  *out = s;
  return SONO_OK;
}

int sono_autoid_data_of_call(const sono_call* call, const sono_call_shape_t* shape, uint32_t fft_n, uint32_t hop, double sample_rate,
                             double time_expansion, int cf_call_type, sono_autoid_data* out) {
  if (!call || !out || !fft_n || !(sample_rate > 0)) return SONO_ERR_ARG;
  const double nan = std::numeric_limits<double>::quiet_NaN();
  const double te = time_expansion > 0 ? time_expansion : 1.0;
  // the panel's displayed units (autoIdPanel.js:1439-1462): kHz, multiplied under time expansion; ms, divided
  auto khz = [&](int32_t bin) { return (double)bin * sample_rate / (double)fft_n / 1000.0 * te; };
  auto ms = [&](int64_t frames) { return (double)frames * (double)hop / sample_rate * 1000.0 / te; };
  for (double& v : out->field) v = nan;
  out->call_type = -1;
  out->harmonic = SONO_AUTOID_ABSENT;
  double* d = out->field;
  d[SONO_F_HIGHEST] = khz(call->high_bin);
  d[SONO_F_LOWEST] = khz(call->low_bin);
  d[SONO_F_START] = khz(call->start_bin);
  d[SONO_F_END] = khz(call->end_bin);
  // duration = (latest marker time - earliest) * 1000 (:1406-1414): the frames of the call, one hop each
  d[SONO_F_DURATION] = ms((int64_t)(call->end_frame - call->start_frame) + 1);
  if (shape) {
    if (shape->knee_frame >= 0) d[SONO_F_KNEE] = khz(shape->knee_bin);
    if (shape->heel_frame >= 0) d[SONO_F_HEEL] = khz(shape->heel_bin);
    if (shape->cf_start_frame >= 0) {
      d[SONO_F_CF_START] = khz(shape->cf_start_bin);
      d[SONO_F_CF_END] = khz(shape->cf_end_bin);
    }
    // (knee.time - low.time) * 1000, knee - low, heel - low, knee - heel (:1423-1429)
    if (shape->knee_frame >= 0 && shape->low_frame >= 0) d[SONO_F_KNEE_LOW_TIME] = ms(shape->knee_frame - shape->low_frame);
    if (shape->knee_frame >= 0) d[SONO_F_KNEE_LOW_BW] = d[SONO_F_KNEE] - d[SONO_F_LOWEST];
    if (shape->heel_frame >= 0) d[SONO_F_HEEL_LOW_BW] = d[SONO_F_HEEL] - d[SONO_F_LOWEST];
    if (shape->knee_frame >= 0 && shape->heel_frame >= 0) d[SONO_F_KNEE_HEEL_BW] = d[SONO_F_KNEE] - d[SONO_F_HEEL];
  }
  // bandwidth: cfStart - end for the CF call types, else high - low (:1416-1421)
  if (cf_call_type) {
    if (!missing(d[SONO_F_CF_START])) d[SONO_F_BANDWIDTH] = d[SONO_F_CF_START] - d[SONO_F_END];
  } else {
    d[SONO_F_BANDWIDTH] = d[SONO_F_HIGHEST] - d[SONO_F_LOWEST];
  }
  return SONO_OK;
}

}  // extern "C"
