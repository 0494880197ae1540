// Device-side helpers shared by the STFT kernels: the two quantisers in both arithmetics.
#ifndef SONO_DEVICE_H
#define SONO_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sonospec.h"

namespace sono {

// ---------------------------------------------------------------- exact (double) quantisers
// Worker pixel value, spectrogramWorker.js:71-78.
__device__ __forceinline__ uint32_t worker_index_exact(float re, float im) {
  const double magSq = (double)re * (double)re + (double)im * (double)im;
  // sqrt and log10 are the long part: magnitudes up to 1 clamp to 0 and magnitudes safely above 1e5 clamp to 1
  // whatever the last bits of log10 are; sqrt is monotone and exact at 1, so the test can be made on the square
  // (a NaN takes the general route and ends as 0 like the reference's failed comparisons)
  if (magSq <= 1.0) return 0;
  if (magSq > 1.00002e10) return 255;
  const double mag = sqrt(magSq);
  double val = (mag > 1e-12) ? log10(mag) / 5 : -2.4;
  val = val < 0 ? 0 : (val > 1 ? 1 : val);
  return (uint32_t)floor(val * 255);
}

// Plugin magnitude, spectrogram.esm.js:195-198: float32 of (2/N)*sqrt(re^2+im^2).
__device__ __forceinline__ float plugin_mag_exact(float re, float im, double two_over_n) {
  const double e = re, s = im;
  return (float)(two_over_n * sqrt(e * e + s * s));
}

// Plugin colour index, spectrogram.esm.js:531-535 (Uint8Array store = truncate, modulo 256).
__device__ __forceinline__ uint32_t plugin_index_exact(float m, double gain, double range) {
  const double s = ((double)m > 1e-12) ? (double)m : 1e-12;
  const double r = 20 * log10(s);
  if (r < -gain - range) return 0;
  if (r > -gain) return 255;
  const double v = (r + gain) / range * 255 + 256;
  if (!(fabs(v) < 1e15)) return 0;  // NaN / infinity store 0
  return (uint32_t)((long long)trunc(v) & 255);
}

// The exact quantisers behind a float32 predictor.  Both are floor() of a smooth function t of the bin's float32
// (re, im): a float32 evaluation of t (|X|^2 by one FMA, v_log_f32, one FMA) lies within `tol` of the double one --
// 2 ulp of |X|^2, 1 ulp of a logarithm of magnitude < 64 (4e-6), times the slope a, plus the rounding of t itself
// (below 3e-5 at t = 256) -- so whenever that t is further than `tol` from every integer, floor() of both is the same
// integer and the double sqrt / log10 (the long half of the exact kernels) are skipped; `tol` carries a fivefold
// margin.  Everything else -- t near an integer (which includes both clamps and the plugin's 256 -> 0 wrap at
// r == -gain), zero, infinite and NaN inputs -- takes the double route unchanged, so the result is the reference's
// integer in every case.  Not used where the plugin's magnitude floor of 1e-12 (-240 dB) lies inside the display range.
struct ExactPred {
  float a, b, tol;
  uint32_t on;
};
__host__ __device__ inline ExactPred make_exact_pred(uint32_t mode, uint32_t n, double gain, double range) {
  ExactPred q;
  const double l10_2 = 0.30102999566398119521;
  if (mode == SONO_SCALE_WORKER_LOG5) {
    q.a = (float)(255.0 * 0.5 * l10_2 / 5.0);  // t = 255 * log10|X| / 5
    q.b = 0.0f;
    q.on = 1;
  } else {
    const double lg = 2.0 * (1.0 - log2((double)n));  // t = (20 log10(2/N |X|) + gain) / range * 255 + 256
    q.a = (float)(10.0 * l10_2 * 255.0 / range);
    q.b = (float)((10.0 * l10_2 * lg + gain) * 255.0 / range + 256.0);
    q.on = (gain + range < 230.0 && range > 0.0) ? 1u : 0u;
  }
  q.tol = 1e-3f + 2e-5f * (q.a < 0 ? -q.a : q.a);
  if (!(q.tol < 0.25f)) q.on = 0;
  return q;
}
__device__ __forceinline__ uint32_t worker_index_exact_p(float re, float im, const ExactPred& q) {
  const float t = __log2f(__builtin_fmaf(re, re, im * im)) * q.a;
  const float fr = __builtin_amdgcn_fractf(t);
  if (t > q.tol && t < 255.0f - q.tol && fr > q.tol && fr < 1.0f - q.tol) return (uint32_t)t;
  if (t < -q.tol) return 0;            // |X| < 1
  if (t > 255.0f + q.tol) return 255;  // |X| > 1e5
  return worker_index_exact(re, im);
}
__device__ __forceinline__ uint32_t plugin_index_exact_p(float re, float im, double two_over_n, double gain, double range,
                                                         const ExactPred& q) {
  if (q.on) {
    const float t = __builtin_fmaf(__log2f(__builtin_fmaf(re, re, im * im)), q.a, q.b);
    const float fr = __builtin_amdgcn_fractf(t);
    if (t > 1.0f + q.tol && t < 256.0f - q.tol && fr > q.tol && fr < 1.0f - q.tol) return (uint32_t)t;
    if (t < 1.0f - q.tol) return 0;      // r < -gain - range (and the 1e-12 floor lies below the display range)
    if (t > 256.0f + q.tol) return 255;  // r > -gain
  }
  return plugin_index_exact(plugin_mag_exact(re, im, two_over_n), gain, range);
}

// Feature pass, reduction of per-lane (largest |X|^2, its bin, runner-up) over the `width` lanes that share a frame:
// larger value wins, equal values go to the lower bin (the strict-> first-wins rule over ascending bins); the
// frame's runner-up is the largest of the losers' bests and the winner's own runner-up.
// A peak found in the band rounded outwards to groups of four bins is the band's own peak unless it sits in the margin.
__device__ __forceinline__ bool peak_outside(int kbest, uint32_t feat_lo, uint32_t feat_hi) {
  return kbest < (int)feat_lo || kbest >= (int)feat_hi;
}

// Cross-lane moves of the peak reductions, all on the vector ALU (a shuffle through LDS costs an LDS round trip per
// step, ten dependent ones per frame): lane ^ 1 and lane ^ 2 by DPP quad permutations, then rotations by 4 and 8 within
// the row of 16 lanes (what a quad holds is uniform by then, so a rotation serves as well as an exchange), then
// v_permlane16_swap / v_permlane32_swap, which hand a lane its own value in one result and the partner row's / half's in
// the other.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppRor4 = 0x124, kDppRor8 = 0x128;

// larger value wins, equal values go to the lower bin
__device__ __forceinline__ void peak_merge(float& best, int& kbest, float o, int ko) {
  const bool g = o > best || (o == best && ko < kbest);
  best = g ? o : best;
  kbest = g ? ko : kbest;
}

// Over the WIDTH (32 or 64) lanes of a transform: the largest value and its bin in every lane, and in `next` the
// runner-up (the largest value of any other bin).
template <int WIDTH>
__device__ __forceinline__ void reduce_peak(float& best, int& kbest, float& next) {
  const float mine = best;
  const int kmine = kbest;
  peak_merge(best, kbest, dpp_f<kDppXor1>(best), dpp_i<kDppXor1>(kbest));
  peak_merge(best, kbest, dpp_f<kDppXor2>(best), dpp_i<kDppXor2>(kbest));
  peak_merge(best, kbest, dpp_f<kDppRor4>(best), dpp_i<kDppRor4>(kbest));
  peak_merge(best, kbest, dpp_f<kDppRor8>(best), dpp_i<kDppRor8>(kbest));
  {
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    const auto k = __builtin_amdgcn_permlane16_swap((uint32_t)kbest, (uint32_t)kbest, false, false);
    best = __uint_as_float(b[0]);
    kbest = (int)k[0];
    peak_merge(best, kbest, __uint_as_float(b[1]), (int)k[1]);
  }
  if constexpr (WIDTH == 64) {
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    const auto k = __builtin_amdgcn_permlane32_swap((uint32_t)kbest, (uint32_t)kbest, false, false);
    best = __uint_as_float(b[0]);
    kbest = (int)k[0];
    peak_merge(best, kbest, __uint_as_float(b[1]), (int)k[1]);
  }
  float c = (kmine == kbest) ? next : mine;   // the winning lane offers its runner-up, every other lane its best
  c = __builtin_fmaxf(c, dpp_f<kDppXor1>(c));
  c = __builtin_fmaxf(c, dpp_f<kDppXor2>(c));
  c = __builtin_fmaxf(c, dpp_f<kDppRor4>(c));
  c = __builtin_fmaxf(c, dpp_f<kDppRor8>(c));
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), __float_as_uint(c), false, false);
    c = __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  if constexpr (WIDTH == 64) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(c), false, false);
    c = __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  next = c;
}

// ---------------------------------------------------------------- fast (float32) quantisers
// Both work from |X|^2 so no square root is needed: log10|X| = 0.5*log10(2)*log2(|X|^2).
struct FastScale {
  float a, b;     // t = log2(|X|^2) * a + b
  float t_floor;  // plugin: t at the 1e-12 magnitude floor (r = -240 dB)
  float db_off;   // dB = 10*log10(2) * log2(|X|^2) + db_off
  float xtalk;    // feature pass: share of the partner frame's |X| that may leak into a bin (two frames per transform)
  float tscale;   // feature pass: |d t / d ln(|X|^2)| = |a| / ln 2
  uint32_t mode;
};

// Feature pass of the float32 kernels: a frame's record is exact unless the float32 arithmetic cannot tell.  With
// P1 >= P2 the two largest |X|^2 of the band, Tp and To the largest |X|^2 ANYWHERE in the spectrum of the frame
// transformed with it and of the frame itself, and t the quantiser argument of P1: the float32 powers are within
// kTieRel (measured error < 1e-6; 40x margin) of the reference's, plus the pair's cross-talk under an asymmetric
// twiddle table (a share of the partner's largest magnitude: a perturbed butterfly spreads over all bins), plus the
// transform's rounding noise, which scales with the LARGEST value in the transform and not with the band's own -- a
// silent frame paired with a loud one, or a band that leaves out the loud bins, holds nothing but that (kLeak times
// the largest magnitude).  The argmax is safe if P1 - P2 exceeds all of that, and the colour index is safe if t is
// further from an integer than the same error scaled by d t / d ln P, plus the error of v_log_f32.  Anything else is
// marked and recomputed in the exact arithmetic (launch_features_fixup).  A band that is exactly zero in a pair that
// is not is marked as well: the reference's values may be tiny rather than zero.
// (When the band is the whole spectrum Tp and To are the two frames' band peaks; otherwise the kernels keep the
// largest power of any bin they compute.)
constexpr float kTieRel = 4e-5f;
constexpr float kLeak = 1e-6f;  // of the larger of the two largest magnitudes: measured 1.5e-7 .. 1.8e-7 at 512 .. 4096 points (tools/leak_probe.py)
__device__ __forceinline__ bool feature_uncertain(float p1, float p2, float tp, float to, float t, const FastScale& s) {
  const float big = fmaxf(tp, to);
  if (!(p1 > 0.0f)) return p1 == 0.0f && big > 0.0f;  // all-zero everywhere: first bin, index from |X| = 0, in every arithmetic
  const float q = fmaxf(s.xtalk * __builtin_amdgcn_sqrtf(tp), kLeak * __builtin_amdgcn_sqrtf(big)) *
                  __builtin_amdgcn_rsqf(p1);                 // |error| / |X|: power error 2q + q^2
  const float rel = __builtin_fmaf(q, 2.0f + q, kTieRel);  // relative error bound of P1 (and, in units of P1, of P2)
  const bool tie = p1 - p2 <= 2.0f * rel * p1;
  const float fr = __builtin_amdgcn_fractf(t);
  const float dt = __builtin_fmaf(s.tscale, rel, 2e-4f);
  const bool edge = !(fr > dt && fr < 1.0f - dt) && t > -1.0f && t < 257.0f;
  return tie || edge;
}

// The largest value over the WIDTH (32 or 64) lanes of a transform, in every lane (vector ALU only, like reduce_peak).
template <int WIDTH>
__device__ __forceinline__ float reduce_max(float c) {
  c = __builtin_fmaxf(c, dpp_f<kDppXor1>(c));
  c = __builtin_fmaxf(c, dpp_f<kDppXor2>(c));
  c = __builtin_fmaxf(c, dpp_f<kDppRor4>(c));
  c = __builtin_fmaxf(c, dpp_f<kDppRor8>(c));
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), __float_as_uint(c), false, false);
    c = __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  if constexpr (WIDTH == 64) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(c), false, false);
    c = __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  return c;
}

__host__ __device__ inline FastScale make_fast_scale(uint32_t mode, uint32_t n, double gain, double range) {
  FastScale s;
  s.mode = mode;
  s.xtalk = 0.0f;
  const double l10_2 = 0.30102999566398119521;
  if (mode == SONO_SCALE_WORKER_LOG5) {
    s.a = (float)(255.0 * 0.5 * l10_2 / 5.0);  // 255 * log10|X| / 5
    s.b = 0.0f;
    s.t_floor = 0.0f;
    s.db_off = 0.0f;
  } else {
    // r = 20*log10(2/N*|X|) = 10*log10(2)*(log2|X|^2 + 2*log2(2/N));  t = (r+gain)/range*255 + 256
    const double lg = 2.0 * (1.0 - log2((double)n));
    s.a = (float)(10.0 * l10_2 * 255.0 / range);
    s.b = (float)((10.0 * l10_2 * lg + gain) * 255.0 / range + 256.0);
    s.t_floor = (float)((-240.0 + gain) * 255.0 / range + 256.0);
    s.db_off = (float)(10.0 * l10_2 * lg);
  }
  s.tscale = 1.4427f * (s.a < 0 ? -s.a : s.a);
  return s;
}

__device__ __forceinline__ uint32_t fast_index(float mag_sq, const FastScale& s) {
  const float lg = __log2f(mag_sq);
  if (s.mode == SONO_SCALE_WORKER_LOG5) {
    float t = lg * s.a;
    t = fminf(fmaxf(t, 0.0f), 255.0f);  // NaN -> 0 like the reference's failed comparison
    return (uint32_t)t;
  }
  // NaN takes the floor, as `n > 1e-12` fails for NaN.  The conversion saturates: t < 1 (r < -gain-range) gives 0,
  // t >= 256 (r > -gain) gives 255.  (The reference stores 256 -> 0 when r == -gain exactly, spectrogram.esm.js:534;
  // in float32 a t of exactly 256 says no more than "r is within rounding of -gain", where the reference gives 255 on
  // both sides, so 255 it is.  The bit-identical arithmetic keeps the wrap.)
  const float t = __builtin_floorf(fmaxf(__builtin_fmaf(lg, s.a, s.b), s.t_floor));
  return __builtin_amdgcn_cvt_pk_u8_f32(t, 0, 0u);
}

// The quantisers of fast_index packed into byte BYTE of `acc`, as the transform kernels use them.  The saturating byte
// conversion rounds to nearest (ties to even), so floor(t) is taken as round(t - 0.5): the same for every t that is not
// an integer, and at an integer t float32 cannot tell which side of the step the reference is on anyway (the feature
// pass marks such records and has them recomputed in the exact arithmetic).  One instruction less per bin.
// (The plugin's magnitude floor of 1e-12 needs no instruction here: it is r = -240 dB, below -gain-range for every
// context that computes in float32 -- those have gain + range <= 100 -- so it is index 0, and that is what the
// saturating conversion makes of any t < 0.5, of -infinity from a zero magnitude and of a NaN, which the reference
// sends to the floor as well.)
template <int BYTE>
__device__ __forceinline__ uint32_t fast_plugin_byte(float mag_sq, const FastScale& s, uint32_t acc) {
  const float t = __builtin_fmaf(__log2f(mag_sq), s.a, s.b - 0.5f);
  return __builtin_amdgcn_cvt_pk_u8_f32(t, BYTE, acc);
}
// The colour index of a frame's peak in the feature records: here the magnitude floor is kept, because a feature
// context computes in float32 whatever its gain and range are, and with gain + range > 240 the floor is not index 0.
__device__ __forceinline__ uint32_t fast_plugin_peak_index(float mag_sq, const FastScale& s) {
  const float t = fmaxf(__builtin_fmaf(__log2f(mag_sq), s.a, s.b - 0.5f), s.t_floor - 0.5f);
  return __builtin_amdgcn_cvt_pk_u8_f32(t, 0, 0u);
}
template <int BYTE>
__device__ __forceinline__ uint32_t fast_worker_byte(float mag_sq, const FastScale& s, uint32_t acc) {
  const float t = __builtin_fmaf(__log2f(mag_sq), s.a, s.b - 0.5f);
  return __builtin_amdgcn_cvt_pk_u8_f32(t, BYTE, acc);  // saturates to [0,255]; NaN -> 0
}

__device__ __forceinline__ float fast_db(float mag_sq, const FastScale& s) {
  // 20*log10(max(m, 1e-12))
  const float db = 3.0102999566f * __log2f(mag_sq) + s.db_off;
  return fmaxf(db, -240.0f);
}

}  // namespace sono
#endif
