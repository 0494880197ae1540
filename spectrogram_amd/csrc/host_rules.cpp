// Host-side rules of the spectrogram paths: geometry, overlap, crop window, colour LUTs, window
// and twiddle tables.  Pure C++ (no GPU); compiled without FMA contraction because the tables
// must be the IEEE double values the reference's JavaScript computes before its float32 store.
#include <cmath>
#include <cstring>
#include <vector>

#include "sonospec.h"
#include "sono_internal.h"

namespace sono {

// ECMAScript Math.round: nearest integer, ties toward +infinity.
static double ecma_round(double x) {
  if (std::isnan(x) || std::isinf(x)) return x;
  const double f = std::floor(x);
  return (x - f >= 0.5) ? f + 1.0 : f;
}

// ECMAScript ToUint8Clamp: clamp to [0,255], ties to even.
static uint8_t ecma_uint8_clamp(double v) {
  if (std::isnan(v) || v <= 0) return 0;
  if (v >= 255) return 255;
  const double f = std::floor(v), d = v - f;
  if (d < 0.5) return (uint8_t)f;
  if (d > 0.5) return (uint8_t)(f + 1);
  return (uint8_t)((((long)f) & 1) ? f + 1 : f);
}

bool is_pow2(uint32_t n) { return n != 0 && (n & (n - 1)) == 0; }

uint32_t ilog2(uint32_t n) {
  uint32_t l = 0;
  while ((1u << l) < n) l++;
  return l;
}

// spectrogram.esm.js:105-151 (plugin windows), spectrogramWorker.js:195-203 (worker Hann)
int fill_window(uint32_t id, uint32_t n, double alpha, float* w) {
  const double t = (double)n;
  const double pi = 3.141592653589793;
  const double e = 2.718281828459045;
  switch (id) {
    case SONO_WIN_BARTLETT:
      for (uint32_t i = 0; i < n; i++) w[i] = (float)(2 / (t - 1) * ((t - 1) / 2 - std::fabs(i - (t - 1) / 2)));
      return SONO_OK;
    case SONO_WIN_BARTLETT_HANN:
      for (uint32_t i = 0; i < n; i++)
        w[i] = (float)(.62 - .48 * std::fabs(i / (t - 1) - .5) - .38 * std::cos(2 * pi * i / (t - 1)));
      return SONO_OK;
    case SONO_WIN_BLACKMAN: {
      const double r = (alpha > 0) ? alpha : .16;
      for (uint32_t i = 0; i < n; i++)
        w[i] = (float)((1 - r) / 2 - .5 * std::cos(2 * pi * i / (t - 1)) + r / 2 * std::cos(4 * pi * i / (t - 1)));
      return SONO_OK;
    }
    case SONO_WIN_COSINE:
      for (uint32_t i = 0; i < n; i++) w[i] = (float)std::cos(pi * i / (t - 1) - pi / 2);
      return SONO_OK;
    case SONO_WIN_GAUSS: {
      const double r = (alpha > 0) ? alpha : .25;
      for (uint32_t i = 0; i < n; i++)
        w[i] = (float)std::pow(e, -.5 * std::pow((i - (t - 1) / 2) / (r * (t - 1) / 2), 2));
      return SONO_OK;
    }
    case SONO_WIN_HAMMING:
      for (uint32_t i = 0; i < n; i++) w[i] = (float)(.54 - .46 * std::cos(2 * pi * i / (t - 1)));
      return SONO_OK;
    case SONO_WIN_HANN:
      for (uint32_t i = 0; i < n; i++) w[i] = (float)(.5 * (1 - std::cos(2 * pi * i / (t - 1))));
      return SONO_OK;
    case SONO_WIN_LANCZOZ:
      for (uint32_t i = 0; i < n; i++)
        w[i] = (float)(std::sin(pi * (2 * i / (t - 1) - 1)) / (pi * (2 * i / (t - 1) - 1)));
      return SONO_OK;
    case SONO_WIN_RECTANGULAR:
      for (uint32_t i = 0; i < n; i++) w[i] = 1.0f;
      return SONO_OK;
    case SONO_WIN_TRIANGULAR:
      for (uint32_t i = 0; i < n; i++) w[i] = (float)(2 / t * (t / 2 - std::fabs(i - (t - 1) / 2)));
      return SONO_OK;
    case SONO_WIN_WORKER_HANN: {
      const double invN = 1.0 / (t - 1), twoPI = 2 * pi;
      for (uint32_t i = 0; i < n; i++) w[i] = (float)(0.5 * (1 - std::cos(twoPI * i * invN)));
      return SONO_OK;
    }
    default:
      return SONO_ERR_WINDOW;
  }
}

// Flat stage layout: the stage whose butterflies span half size T uses entries [T-1, 2T-1).
int fill_twiddles(uint32_t mode, uint32_t n, double* c, double* s) {
  const double pi = 3.141592653589793;
  if (mode == SONO_TW_ACCURATE) {  // spectrogramWorker.js:110-125
    for (uint32_t n1 = 1; n1 < n; n1 <<= 1) {
      const uint32_t n2 = n1 << 1;
      for (uint32_t j = 0; j < n1; j++) {
        const double angle = -2 * pi * j / n2;
        c[n1 - 1 + j] = std::cos(angle);
        s[n1 - 1 + j] = std::sin(angle);
      }
    }
    return SONO_OK;
  }
  if (mode == SONO_TW_PLUGIN_RECURRENCE) {  // spectrogram.esm.js:158-160,176-190
    for (uint32_t T = 1; T < n; T <<= 1) {
      const double d = (double)(float)std::cos(-pi / T);
      const double w = (double)(float)std::sin(-pi / T);
      double g = 1, b = 0;
      for (uint32_t z = 0; z < T; z++) {
        c[T - 1 + z] = g;
        s[T - 1 + z] = b;
        const double v = g;
        g = v * d - b * w;
        b = v * w + b * d;
      }
    }
    return SONO_OK;
  }
  return SONO_ERR_ARG;
}

// How far the spectrum of a REAL frame, transformed by the radix-2 network with the twiddle table (c, s), can be from
// Hermitian symmetry.  The float32 kernels transform two real frames a, b as one complex frame z = a + i b and take frame
// a's bin k as (Z[k] + conj Z[N-k]) / 2, which is exact only if conj A[N-k] == A[k]; the plugin's recurrence factors are
// not conjugate-symmetric, so A carries a defect alpha[k] = conj A[N-k] - A[k] and the separated spectrum is off by
// (alpha[k] - i beta[k]) / 2.  alpha is linear in the samples: alpha[k] = sum_n H[k][n] x[n] w[n] with
// H[k][n] = conj F[N-k][n] - F[k][n], F the network's matrix.  For samples bounded by m that gives
// |alpha[k]| <= m * sum_n |H[k][n]| w[n]; g_raw is that row sum maximised over k (bound in terms of the largest raw
// sample), g_win the same without the window (bound in terms of the largest windowed sample).  F's columns come from
// unit impulses, whose support doubles per stage: 2N operations per column, N columns.
void pair_defect_gains(uint32_t n, const double* c, const double* s, const float* window, double* g_raw, double* g_win) {
  std::vector<double> fr(n), fi(n), acc_raw(n / 2, 0.0), acc_win(n / 2, 0.0);
  uint32_t bits = 0;
  while ((1u << bits) < n) ++bits;
  for (uint32_t j = 0; j < n; ++j) {
    uint32_t pos = 0;
    for (uint32_t b = 0; b < bits; ++b) pos |= ((j >> b) & 1u) << (bits - 1 - b);
    // the impulse's image after each stage lives in the block of 2T entries around `pos`
    uint32_t lo = pos;  // first entry of the support
    fr[pos] = 1.0;
    fi[pos] = 0.0;
    for (uint32_t T = 1; T < n; T <<= 1) {
      const uint32_t blk = lo & ~(2 * T - 1);  // the support [lo, lo + T) lies in one half of this block
      const bool upper = (lo & T) != 0;
      for (uint32_t q = 0; q < T; ++q) {
        if (!upper) {  // the a[k] side: out[k] = out[k + T] = v
          fr[blk + T + q] = fr[blk + q];
          fi[blk + T + q] = fi[blk + q];
        } else {       // the a[k + T] side: out[k] = w v, out[k + T] = -w v
          const double vr = fr[blk + T + q], vi = fi[blk + T + q];
          const double tr = c[T - 1 + q] * vr - s[T - 1 + q] * vi, ti = c[T - 1 + q] * vi + s[T - 1 + q] * vr;
          fr[blk + q] = tr;
          fi[blk + q] = ti;
          fr[blk + T + q] = -tr;
          fi[blk + T + q] = -ti;
        }
      }
      lo = blk;
    }
    const double wj = window ? std::fabs((double)window[j]) : 1.0;
    for (uint32_t k = 1; k < n / 2; ++k) {
      const double hr = fr[n - k] - fr[k], hi = -fi[n - k] - fi[k];
      const double h = std::sqrt(hr * hr + hi * hi);
      acc_raw[k] += h * wj;
      acc_win[k] += h;
    }
  }
  double a = 0, b = 0;
  for (uint32_t k = 1; k < n / 2; ++k) {
    if (acc_raw[k] > a) a = acc_raw[k];
    if (acc_win[k] > b) b = acc_win[k];
  }
  *g_raw = a;
  *g_win = b;
}

int compute_geometry(uint32_t n, uint32_t hop, uint32_t rule, uint32_t bin_lo, uint32_t bin_hi, uint64_t L,
                     sono_geometry* g) {
  if (!is_pow2(n) || n < 2) return SONO_ERR_FFT_SIZE;
  if (hop == 0) return SONO_ERR_ARG;
  const uint32_t half = n / 2;
  if (bin_hi == 0) bin_hi = half;
  if (bin_hi > half || bin_lo >= bin_hi) return SONO_ERR_ARG;
  g->bins = bin_hi - bin_lo;
  g->hop = hop;
  const uint64_t cols = (L > n) ? (L - n + hop - 1) / hop : 0;  // ceil((L-N)/hop), 0 when L <= N
  if (rule == SONO_FRAMES_PLUGIN_STRICT) {  // spectrogram.esm.js:526
    g->frames = cols;
    g->width = cols;
  } else if (rule == SONO_FRAMES_WORKER_COMPAT) {  // spectrogramWorker.js:40,59
    g->frames = (L >= n) ? (L - n) / hop + 1 : 0;
    g->width = cols > 1 ? cols : 1;
  } else {
    return SONO_ERR_ARG;
  }
  return SONO_OK;
}

}  // namespace sono

using namespace sono;

extern "C" {

int sono_abi_version(void) { return SONOSPEC_ABI_VERSION; }

uint32_t sono_worker_hop(uint32_t n, double pct) {  // spectrogramWorker.js:39
  const double h = std::floor((double)n * (1 - pct / 100));
  return h > 1 ? (uint32_t)h : 1u;
}

uint32_t sono_plugin_noverlap(uint32_t n, uint32_t noverlap, uint64_t L, uint32_t canvas_width) {
  if (noverlap > 0) return noverlap;  // spectrogram.esm.js:502-506 (0 / null / undefined are all "auto")
  if (canvas_width == 0) return 0;
  const double e = (double)L / (double)canvas_width;
  const double o = ecma_round((double)n - e);
  return o > 0 ? (uint32_t)o : 0u;
}

uint32_t sono_overlap_from_percent(uint32_t n, double pct) {  // wsManager.js:87-89
  const double v = std::floor((double)n * (pct / 100));
  return v > 0 ? (uint32_t)v : 0u;
}

int sono_geometry_for(uint32_t n, uint32_t hop, uint32_t rule, uint32_t bin_lo, uint32_t bin_hi, uint64_t L,
                      sono_geometry* out) {
  if (!out) return SONO_ERR_ARG;
  return compute_geometry(n, hop, rule, bin_lo, bin_hi, L, out);
}

int sono_plugin_crop(uint32_t bins, double sample_rate, double fmin, double fmax, int64_t* sy, int64_t* sh) {
  if (!sy || !sh || !(sample_rate > 0)) return SONO_ERR_ARG;
  const double nyq = sample_rate / 2;  // spectrogram.esm.js:353
  if (fmax == 0) fmax = nyq;           // :496
  const double u = fmin / nyq, f = fmax / nyq;
  const double p = f < 1 ? f : 1;      // :377-379 with hzToScale == identity (:464)
  *sy = (int64_t)ecma_round((double)bins * (1 - p));
  *sh = (int64_t)ecma_round((double)bins * (p - u));
  return SONO_OK;
}

// ---- non-linear frequency scales, spectrogram.esm.js:412-478 (erb constant :202)
static const double kErbA = 1e3 * std::log(10.0) / 107.939;

double sono_hz_to_scale(uint32_t scale, double t) {
  switch (scale) {
    case SONO_FSCALE_MEL: return 2595 * std::log10(1 + t / 700);
    case SONO_FSCALE_LOGARITHMIC: return std::log10(t > 1 ? t : 1);
    case SONO_FSCALE_BARK: {
      double e = 26.81 * t / (1960 + t) - .53;
      if (e < 2) e += .15 * (2 - e);
      if (e > 20.1) e += .22 * (e - 20.1);
      return e;
    }
    case SONO_FSCALE_ERB: return kErbA * std::log10(1 + .00437 * t);
    default: return t;
  }
}

double sono_scale_to_hz(uint32_t scale, double t) {
  switch (scale) {
    case SONO_FSCALE_MEL: return 700 * (std::pow(10.0, t / 2595) - 1);
    case SONO_FSCALE_LOGARITHMIC: return std::pow(10.0, t);
    case SONO_FSCALE_BARK:
      if (t < 2) t = (t - .3) / .85;
      if (t > 20.1) t = (t + 4.422) / 1.22;
      return (t + .53) / (26.28 - t) * 1960;
    case SONO_FSCALE_ERB: return (std::pow(10.0, t / kErbA) - 1) / .00437;
    default: return t;
  }
}

int sono_filterbank_table(uint32_t scale, uint32_t fft_n, double sample_rate, uint32_t filters, int32_t* bin, double* w0, double* w1) {
  if (!bin || !w0 || !w1 || scale > SONO_FSCALE_ERB || scale == SONO_FSCALE_LINEAR || !sono::is_pow2(fft_n) || !(sample_rate > 0) || !filters)
    return SONO_ERR_ARG;
  // createFilterBank, spectrogram.esm.js:395-408
  const double i = sono_hz_to_scale(scale, 0), a = sono_hz_to_scale(scale, sample_rate / 2), h = sample_rate / fft_n;
  for (uint32_t e = 0; e < filters; e++) {
    const double s = sono_scale_to_hz(scale, i + (double)e / filters * (a - i));
    const double o = std::floor(s / h), l = o * h, c = (s - l) / ((o + 1) * h - l);
    bin[e] = (int32_t)o;
    w0[e] = 1 - c;
    w1[e] = c;
  }
  return SONO_OK;
}

int sono_plugin_crop_scaled(uint32_t bins, double sample_rate, double fmin, double fmax, uint32_t scale, int64_t* sy, int64_t* sh) {
  if (!sy || !sh || !(sample_rate > 0) || scale > SONO_FSCALE_ERB) return SONO_ERR_ARG;
  const double nyq = sample_rate / 2;
  if (fmax == 0) fmax = nyq;
  const double u = sono_hz_to_scale(scale, fmin) / sono_hz_to_scale(scale, nyq);  // :377-379
  const double f = sono_hz_to_scale(scale, fmax) / sono_hz_to_scale(scale, nyq);
  const double p = f < 1 ? f : 1;
  *sy = (int64_t)ecma_round((double)bins * (1 - p));
  *sh = (int64_t)ecma_round((double)bins * (p - u));
  return SONO_OK;
}

int sono_plugin_draw_rect(double sample_rate, double fmin, double fmax, uint32_t scale, uint32_t height, double* dy, double* dh,
                          int* fill_background) {
  if (!dy || !dh || !fill_background || !(sample_rate > 0) || scale > SONO_FSCALE_ERB) return SONO_ERR_ARG;
  (void)fmin;
  const double nyq = sample_rate / 2;
  if (fmax == 0) fmax = nyq;
  const double f = sono_hz_to_scale(scale, fmax) / sono_hz_to_scale(scale, nyq);  // :378-379
  const double p = f < 1 ? f : 1;
  *dy = (double)height * (1 - p / f);  // :381 for channel h = 0
  *dh = (double)height * p / f;
  *fill_background = fmax > nyq ? 1 : 0;  // :357
  return SONO_OK;
}

int sono_plugin_background(const double* last, uint8_t* rgba) {  // spectrogram.esm.js:358-360
  if (!last || !rgba) return SONO_ERR_ARG;
  if (std::isnan(last[0]) || std::isnan(last[1]) || std::isnan(last[2]) || std::isnan(last[3])) {
    rgba[0] = rgba[1] = rgba[2] = 0;  // "rgba(NaN, ...)" does not parse: fillStyle keeps its default, opaque black
    rgba[3] = 255;
    return SONO_OK;
  }
  for (int ch = 0; ch < 3; ch++) {
    const double v = last[ch] < 0 ? 0 : last[ch] > 255 ? 255 : last[ch];
    rgba[ch] = (uint8_t)ecma_round(v);
  }
  const double a = last[3] < 0 ? 0 : last[3] > 1 ? 1 : last[3];
  rgba[3] = (uint8_t)ecma_round(a * 255);
  return SONO_OK;
}

int sono_plugin_lut(const double* cm, uint8_t* lut) {  // spectrogram.esm.js:262-271
  if (!cm || !lut) return SONO_ERR_ARG;
  for (int i = 0; i < 1024; i++) lut[i] = ecma_uint8_clamp(ecma_round(255 * cm[i]));
  return SONO_OK;
}

int sono_worker_lut(const double* cm, uint8_t* lut) {  // spectrogramWorker.js:128-143
  if (!cm || !lut) return SONO_ERR_ARG;
  for (int i = 0; i < 256; i++)
    for (int ch = 0; ch < 4; ch++) {
      double v = cm[i * 4 + ch];
      if (std::isnan(v)) v = (ch < 3) ? 0.0 : 1.0;
      lut[i * 4 + ch] = ecma_uint8_clamp(ecma_round(255 * v));
    }
  return SONO_OK;
}

int64_t sono_segment_calls(const sono_frame_feature* f, uint64_t frames, uint32_t thr, uint32_t min_frames,
                           uint64_t first_frame, sono_call* out, uint64_t cap) {
  if (!f && frames) return SONO_ERR_ARG;
  if (min_frames == 0) min_frames = 1;
  int64_t found = 0;
  uint64_t i = 0;
  while (i < frames) {
    if (f[i].peak_idx < thr) {
      i++;
      continue;
    }
    uint64_t j = i;
    sono_call c;
    c.start_frame = first_frame + i;
    c.start_bin = c.high_bin = c.low_bin = c.peak_bin = f[i].peak_bin;
    c.peak_idx = f[i].peak_idx;
    c.peak_frame = first_frame + i;
    while (j < frames && f[j].peak_idx >= thr) {
      if (f[j].peak_bin > c.high_bin) c.high_bin = f[j].peak_bin;
      if (f[j].peak_bin < c.low_bin) c.low_bin = f[j].peak_bin;
      if (f[j].peak_idx > c.peak_idx) {  // strict: the first frame holding the maximum wins
        c.peak_idx = f[j].peak_idx;
        c.peak_bin = f[j].peak_bin;
        c.peak_frame = first_frame + j;
      }
      j++;
    }
    c.end_frame = first_frame + j - 1;
    c.end_bin = f[j - 1].peak_bin;
    if (j - i >= min_frames) {
      if (out && (uint64_t)found < cap) out[found] = c;
      found++;
    }
    i = j;
  }
  return found;
}

int sono_window_table(uint32_t id, uint32_t n, double alpha, float* out) {
  if (!out || n < 2) return SONO_ERR_ARG;
  return fill_window(id, n, alpha, out);
}

int sono_twiddle_table(uint32_t mode, uint32_t n, double* c, double* s) {
  if (!c || !s) return SONO_ERR_ARG;
  if (!is_pow2(n)) return SONO_ERR_FFT_SIZE;
  return fill_twiddles(mode, n, c, s);
}

}  // extern "C"
