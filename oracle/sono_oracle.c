/*
 * sono_oracle.c -- CPU restatement of SonoRadar's spectrogram hot path (test infrastructure).
 * See sono_oracle.h for the rules this file follows and for who may call it.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread ... -lm
 */
#include "sono_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_E
#define M_E 2.7182818284590452354
#endif

/* ---- ECMAScript conversions -------------------------------------------------------------- */

/* Math.round: nearest integer, ties toward +infinity. */
static double js_round(double x) {
  if (isnan(x) || isinf(x)) return x;
  double f = floor(x);
  return (x - f >= 0.5) ? f + 1.0 : f;
}

/* ToUint8: truncate toward zero, modulo 256; NaN and infinities give 0. */
static uint8_t js_to_uint8(double v) {
  if (isnan(v) || isinf(v)) return 0;
  double t = trunc(v);
  double m = fmod(t, 256.0);
  if (m < 0) m += 256.0;
  return (uint8_t)m;
}

/* ToUint8Clamp: clamp to [0,255], round half to even; NaN gives 0. */
static uint8_t js_to_uint8_clamp(double v) {
  if (isnan(v)) return 0;
  if (v <= 0) return 0;
  if (v >= 255) return 255;
  double f = floor(v);
  double d = v - f;
  if (d < 0.5) return (uint8_t)f;
  if (d > 0.5) return (uint8_t)(f + 1);
  return (uint8_t)((((long)f) & 1) ? f + 1 : f);
}

static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

/* ---- plugin windows: modules/spectrogram.esm.js:105-151 --------------------------------- */

int orc_plugin_window(int id, int n, double alpha, float *w) {
  const double t = (double)n;
  int i;
  switch (id) {
    case ORC_WIN_BARTLETT: /* :106-109 */
      for (i = 0; i < n; i++) w[i] = (float)(2 / (t - 1) * ((t - 1) / 2 - fabs(i - (t - 1) / 2)));
      break;
    case ORC_WIN_BARTLETT_HANN: /* :110-113 */
      for (i = 0; i < n; i++)
        w[i] = (float)(.62 - .48 * fabs(i / (t - 1) - .5) - .38 * cos(2 * M_PI * i / (t - 1)));
      break;
    case ORC_WIN_BLACKMAN: { /* :114-118, r = r || .16 */
      double r = (alpha > 0) ? alpha : .16;
      for (i = 0; i < n; i++)
        w[i] = (float)((1 - r) / 2 - .5 * cos(2 * M_PI * i / (t - 1)) + r / 2 * cos(4 * M_PI * i / (t - 1)));
      break;
    }
    case ORC_WIN_COSINE: /* :119-122 */
      for (i = 0; i < n; i++) w[i] = (float)cos(M_PI * i / (t - 1) - M_PI / 2);
      break;
    case ORC_WIN_GAUSS: { /* :123-127, r = r || .25 */
      double r = (alpha > 0) ? alpha : .25;
      for (i = 0; i < n; i++)
        w[i] = (float)pow(M_E, -.5 * pow((i - (t - 1) / 2) / (r * (t - 1) / 2), 2));
      break;
    }
    case ORC_WIN_HAMMING: /* :128-131 */
      for (i = 0; i < n; i++) w[i] = (float)(.54 - .46 * cos(2 * M_PI * i / (t - 1)));
      break;
    case ORC_WIN_HANN: /* :132-136 */
      for (i = 0; i < n; i++) w[i] = (float)(.5 * (1 - cos(2 * M_PI * i / (t - 1))));
      break;
    case ORC_WIN_LANCZOZ: /* :137-140 */
      for (i = 0; i < n; i++)
        w[i] = (float)(sin(M_PI * (2 * i / (t - 1) - 1)) / (M_PI * (2 * i / (t - 1) - 1)));
      break;
    case ORC_WIN_RECTANGULAR: /* :141-144 */
      for (i = 0; i < n; i++) w[i] = 1.0f;
      break;
    case ORC_WIN_TRIANGULAR: /* :145-148 */
      for (i = 0; i < n; i++) w[i] = (float)(2 / t * (t / 2 - fabs(i - (t - 1) / 2)));
      break;
    default: /* :149-150 throws */
      return -1;
  }
  return 0;
}

/* :152-160 */
void orc_plugin_tables(int n, float *sin_t, float *cos_t, uint32_t *rev) {
  int i, a = 1, half = n >> 1;
  memset(rev, 0, sizeof(uint32_t) * (size_t)n);
  while (a < n) {
    for (i = 0; i < a; i++) rev[i + a] = rev[i] + (uint32_t)half;
    a <<= 1;
    half >>= 1;
  }
  for (i = 0; i < n; i++) {
    sin_t[i] = (float)sin(-M_PI / i);
    cos_t[i] = (float)cos(-M_PI / i);
  }
}

/* :166-200 */
int orc_plugin_spectrum(const float *frame, int n, const float *win, const float *sin_t,
                        const float *cos_t, const uint32_t *rev, float *mag, double *peak,
                        int *peak_band) {
  if (!is_pow2(n)) return -1; /* :168-169 */
  float *o = (float *)malloc(sizeof(float) * (size_t)n);
  float *l = (float *)malloc(sizeof(float) * (size_t)n);
  int k, T, z;
  const double c = 2.0 / n;
  for (k = 0; k < n; k++) { /* :172-174 */
    o[k] = (float)((double)frame[rev[k]] * (double)win[rev[k]]);
    l[k] = 0.0f;
  }
  for (T = 1; T < n; T <<= 1) { /* :175-193 */
    const double d = cos_t[T], w = sin_t[T];
    double g = 1, b = 0;
    for (z = 0; z < T; z++) {
      for (k = z; k < n; k += T << 1) {
        const int M = k + T;
        const double m = g * (double)o[M] - b * (double)l[M];
        const double y = g * (double)l[M] + b * (double)o[M];
        o[M] = (float)((double)o[k] - m);
        l[M] = (float)((double)l[k] - y);
        o[k] = (float)((double)o[k] + m);
        l[k] = (float)((double)l[k] + y);
      }
      const double v = g;
      g = v * d - b * w;
      b = v * w + b * d;
    }
  }
  for (k = 0; k < n / 2; k++) { /* :195-198 */
    const double e = o[k], s = l[k];
    const double r = c * sqrt(e * e + s * s);
    if (r > *peak) {
      *peak_band = k;
      *peak = r;
    }
    mag[k] = (float)r;
  }
  free(o);
  free(l);
  return 0;
}

/* :502-506 */
int orc_plugin_noverlap(int n, int noverlap, size_t length, int canvas_width) {
  if (noverlap > 0) return noverlap;
  double e = (double)length / (double)canvas_width;
  double o = js_round((double)n - e);
  if (!(o > 0)) o = 0; /* Math.max(0, .) ; NaN cannot occur for canvas_width > 0 */
  return (int)o;
}

/* :526,537 */
size_t orc_plugin_frame_count(size_t length, int n, int hop) {
  if (hop <= 0 || length <= (size_t)n) return 0;
  return (length - (size_t)n + (size_t)hop - 1) / (size_t)hop;
}

/* :531-535 */
static uint8_t plugin_quantise(float m, double gain, double range) {
  const double s = ((double)m > 1e-12) ? (double)m : 1e-12;
  const double r = 20 * log10(s);
  if (r < -gain - range) return 0;
  if (r > -gain) return 255;
  return js_to_uint8((r + gain) / range * 255 + 256);
}

size_t orc_plugin_get_frequencies(const float *x, size_t length, int n, int noverlap,
                                  int canvas_width, int window_id, double alpha, double gain,
                                  double range, uint8_t *out, float *mag_out, double *peak,
                                  int *peak_band) {
  if (!is_pow2(n)) return (size_t)-1;
  const int ov = orc_plugin_noverlap(n, noverlap, length, canvas_width);
  const int hop = n - ov;
  if (hop <= 0) return (size_t)-1;
  float *win = (float *)malloc(sizeof(float) * (size_t)n);
  if (orc_plugin_window(window_id, n, alpha, win) != 0) {
    free(win);
    return (size_t)-1;
  }
  float *sin_t = (float *)malloc(sizeof(float) * (size_t)n);
  float *cos_t = (float *)malloc(sizeof(float) * (size_t)n);
  uint32_t *rev = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
  float *mag = (float *)malloc(sizeof(float) * (size_t)(n / 2));
  orc_plugin_tables(n, sin_t, cos_t, rev);
  double pk = 0;
  int pkb = 0;
  size_t f = 0, a = 0;
  const int bins = n / 2;
  for (; a + (size_t)n < length; a += (size_t)hop, f++) { /* :526 */
    orc_plugin_spectrum(x + a, n, win, sin_t, cos_t, rev, mag, &pk, &pkb);
    if (mag_out) memcpy(mag_out + f * (size_t)bins, mag, sizeof(float) * (size_t)bins);
    if (out)
      for (int t = 0; t < bins; t++) out[f * (size_t)bins + (size_t)t] = plugin_quantise(mag[t], gain, range);
  }
  if (peak) *peak = pk;
  if (peak_band) *peak_band = pkb;
  free(win);
  free(sin_t);
  free(cos_t);
  free(rev);
  free(mag);
  return f;
}

/* ---- non-linear frequency scales ------------------------------------------------------------ */

static const double ERB_A = 1e3 * 2.302585092994046 / 107.939; /* :202, 1e3 * Math.log(10) / 107.939 */

double orc_hz_to_scale(int scale, double t) { /* :412-465 */
  switch (scale) {
    case 1: return 2595 * log10(1 + t / 700);
    case 2: return log10(t > 1 ? t : 1);
    case 3: {
      double e = 26.81 * t / (1960 + t) - .53;
      if (e < 2) e += .15 * (2 - e);
      if (e > 20.1) e += .22 * (e - 20.1);
      return e;
    }
    case 4: return ERB_A * log10(1 + .00437 * t);
  }
  return t;
}

double orc_scale_to_hz(int scale, double t) { /* :415-478 */
  switch (scale) {
    case 1: return 700 * (pow(10, t / 2595) - 1);
    case 2: return pow(10, t);
    case 3:
      if (t < 2) t = (t - .3) / .85;
      if (t > 20.1) t = (t + 4.422) / 1.22;
      return (t + .53) / (26.28 - t) * 1960;
    case 4: return (pow(10, t / ERB_A) - 1) / .00437;
  }
  return t;
}

void orc_plugin_filterbank(int scale, int n, double sr, int filters, int32_t *bin, double *w0, double *w1) {
  const double i = orc_hz_to_scale(scale, 0), a = orc_hz_to_scale(scale, sr / 2), h = sr / n; /* :395-400 */
  for (int e = 0; e < filters; e++) {
    const double s = orc_scale_to_hz(scale, i + (double)e / filters * (a - i)); /* :402-407 */
    const double o = floor(s / h), l = o * h, c = (s - l) / ((o + 1) * h - l);
    bin[e] = (int32_t)o;
    w0[e] = 1 - c;
    w1[e] = c;
  }
}

void orc_plugin_apply_filterbank(const float *mag, int bins, const int32_t *bin, const double *w0,
                                 const double *w1, int filters, float *out) {
  /* :484-486: r[i] += t[s] * e[i][s] over EVERY s, accumulated in a Float32Array.  The rows are zero
   * outside their two taps, so the other terms add t[s] * 0: nothing, unless some t[s] is NaN or
   * infinite, in which case every output of the frame becomes NaN. */
  int bad = 0;
  for (int s = 0; s < bins; s++)
    if (!isfinite(mag[s])) bad = 1;
  for (int i = 0; i < filters; i++) {
    float r = 0;
    const int o = bin[i];
    if (o >= 0 && o < bins) r = (float)((double)r + (double)mag[o] * w0[i]);
    if (o + 1 >= 0 && o + 1 < bins) r = (float)((double)r + (double)mag[o + 1] * w1[i]);
    out[i] = bad ? NAN : r;
  }
}

size_t orc_plugin_get_frequencies_scaled(const float *x, size_t length, int n, int noverlap,
                                         int canvas_width, int window_id, double alpha, double gain,
                                         double range, int scale, double sample_rate, uint8_t *out) {
  if (scale == 0)
    return orc_plugin_get_frequencies(x, length, n, noverlap, canvas_width, window_id, alpha, gain, range, out, NULL, NULL, NULL);
  const int bins = n / 2;
  const int ov = orc_plugin_noverlap(n, noverlap, length, canvas_width);
  const int hop = n - ov;
  if (!is_pow2(n) || hop <= 0) return (size_t)-1;
  const size_t frames = orc_plugin_frame_count(length, n, hop);
  float *mags = (float *)malloc(sizeof(float) * (frames ? frames : 1) * (size_t)bins);
  const size_t got = orc_plugin_get_frequencies(x, length, n, noverlap, canvas_width, window_id, alpha, gain, range, NULL, mags, NULL, NULL);
  int32_t *bin = (int32_t *)malloc(sizeof(int32_t) * (size_t)bins);
  double *w0 = (double *)malloc(sizeof(double) * (size_t)bins), *w1 = (double *)malloc(sizeof(double) * (size_t)bins);
  float *flt = (float *)malloc(sizeof(float) * (size_t)bins);
  orc_plugin_filterbank(scale, n, sample_rate, bins, bin, w0, w1); /* numMelFilters = fftSamples / 2, :250-253 */
  for (size_t f = 0; f < got && got != (size_t)-1; f++) {
    orc_plugin_apply_filterbank(mags + f * (size_t)bins, bins, bin, w0, w1, bins, flt);
    for (int t = 0; t < bins; t++) out[f * (size_t)bins + (size_t)t] = plugin_quantise(flt[t], gain, range);
  }
  free(mags); free(bin); free(w0); free(w1); free(flt);
  return got;
}

/* :595-640.  The reference scans every input frame for every output column; only frames whose
 * interval can touch the column are visited here (two frames of margin on both sides), with the
 * reference's own expressions deciding membership and weight, in the same ascending order. */
void orc_plugin_resample(const uint8_t *in, size_t F, int bins, size_t W, uint8_t *out) {
  if (F == 0 || W == 0) return;
  const double invIn = 1.0 / (double)F, invOut = 1.0 / (double)W;
  double *accum = (double *)malloc(sizeof(double) * (size_t)bins);
  for (size_t a = 0; a < W; a++) {
    int any = 0;
    for (int u = 0; u < bins; u++) accum[u] = 0;
    double lo_f = floor((double)a * (double)F / (double)W) - 2;
    double hi_f = ceil((double)(a + 1) * (double)F / (double)W) + 2;
    size_t n0 = lo_f < 0 ? 0 : (size_t)lo_f;
    size_t n1 = hi_f > (double)F ? F : (size_t)hi_f;
    for (size_t n = n0; n < n1; n++) {
      const double s = (double)n * invIn, h = s + invIn;
      const double o = (double)a * invOut, l = o + invOut;
      const double mn = h < l ? h : l, mx = s > o ? s : o;
      double c = mn - mx;
      if (!(c > 0)) continue; /* Math.max(0, .) then c > 0 */
      const double weight = c / invOut;
      const uint8_t *src = in + n * (size_t)bins;
      for (int u = 0; u < bins; u++) accum[u] += weight * (double)src[u];
      any = 1;
    }
    for (int u = 0; u < bins; u++) out[a * (size_t)bins + (size_t)u] = any ? js_to_uint8(accum[u]) : 0;
  }
  free(accum);
}

/* :262-271 */
void orc_plugin_lut(const double *cm, uint8_t *lut) {
  for (int i = 0; i < 1024; i++) lut[i] = js_to_uint8_clamp(js_round(255 * cm[i]));
}

/* :353-355,377-380 with hzToScale == identity (:464) */
void orc_plugin_crop(int bins, double sample_rate, double fmin, double fmax, long *sy, long *sh) {
  const double nyq = sample_rate / 2;
  if (fmax == 0) fmax = nyq; /* :496 */
  const double u = fmin / nyq, f = fmax / nyq;
  const double p = f < 1 ? f : 1;
  *sy = (long)js_round((double)bins * (1 - p));
  *sh = (long)js_round((double)bins * (p - u));
}

void orc_plugin_crop_scaled(int bins, double sample_rate, double fmin, double fmax, int scale, long *sy, long *sh) {
  const double nyq = sample_rate / 2;
  if (fmax == 0) fmax = nyq; /* :496 */
  const double u = orc_hz_to_scale(scale, fmin) / orc_hz_to_scale(scale, nyq); /* :377-379 */
  const double f = orc_hz_to_scale(scale, fmax) / orc_hz_to_scale(scale, nyq);
  const double p = f < 1 ? f : 1;
  *sy = (long)js_round((double)bins * (1 - p));
  *sh = (long)js_round((double)bins * (p - u));
}

/* :366-376 */
void orc_plugin_draw(const uint8_t *frames, size_t width, int bins, const uint8_t *lut, uint8_t *rgba) {
  for (size_t t = 0; t < width; t++)
    for (int e = 0; e < bins; e++) {
      const unsigned idx = frames[t * (size_t)bins + (size_t)e];
      const size_t i = 4 * ((size_t)(bins - e - 1) * width + t);
      memcpy(rgba + i, lut + 4 * idx, 4);
    }
}

/* :377-381, channel h = 0: the cropped rows cover canvas rows [dy, dy + dh); :357 decides the fill. */
void orc_plugin_draw_rect(double sample_rate, double fmin, double fmax, int scale, int height, double *dy, double *dh,
                          int *fill_background) {
  (void)fmin;
  const double nyq = sample_rate / 2;
  if (fmax == 0) fmax = nyq;
  const double f = orc_hz_to_scale(scale, fmax) / orc_hz_to_scale(scale, nyq);
  const double p = f < 1 ? f : 1;
  *dy = height * (1 - p / f);
  *dh = height * p / f;
  *fill_background = fmax > nyq;
}

/* :358-360: the numbers of the last colour-map entry inside "rgba(r, g, b, a)"; CSS clamps r/g/b to 0..255
 * and rounds them, clamps a to 0..1.  A NaN makes the string unparsable: fillStyle stays opaque black. */
void orc_plugin_background(const double *last, uint8_t *rgba) {
  if (isnan(last[0]) || isnan(last[1]) || isnan(last[2]) || isnan(last[3])) {
    rgba[0] = rgba[1] = rgba[2] = 0;
    rgba[3] = 255;
    return;
  }
  for (int ch = 0; ch < 3; ch++) rgba[ch] = (uint8_t)js_round(fmin(fmax(last[ch], 0), 255));
  rgba[3] = (uint8_t)js_round(fmin(fmax(last[3], 0), 1) * 255);
}

/* drawImage(bitmap, 0, dy, width, dh) of :380-381 on a height-row canvas, with the engine's own filter (the
 * browser's is implementation-defined, so nothing in the reference pins this function): bilinear along y between
 * pixel centres with clamped edges, float32 per channel, round half up; a canvas row is covered when its centre
 * lies in [dy, dy + dh).  image: `rows` x width RGBA (orc_plugin_draw's output); crop rows [sy, sy + sh); rows
 * outside the image are transparent black.  Pixels with alpha < 255 go source-over onto the background. */
void orc_canvas_draw(const uint8_t *image, size_t width, int rows, long sy, long sh, double dy, double dh, int height,
                     const uint8_t *background, uint8_t *canvas) {
  static const uint8_t clear[4] = {0, 0, 0, 0};
  const uint8_t *bg = background ? background : clear;
  const int have_bg = bg[0] || bg[1] || bg[2] || bg[3];
  for (int y = 0; y < height; y++) {
    uint8_t *dst = canvas + (size_t)y * width * 4;
    const double yc = y + 0.5;
    const int covered = sh > 0 && dh > 0 && yc >= dy && yc < dy + dh;
    long i0 = 0, i1 = 0;
    float t = 0;
    if (covered) {
      double v = (yc - dy) * ((double)sh / dh) - 0.5;
      if (v < 0) v = 0;
      if (v > (double)(sh - 1)) v = (double)(sh - 1);
      i0 = (long)floor(v);
      i1 = i0 + 1 < sh ? i0 + 1 : sh - 1;
      t = (float)(v - floor(v));
    }
    for (size_t x = 0; x < width; x++) {
      if (!covered) {
        memcpy(dst + 4 * x, bg, 4);
        continue;
      }
      const long ra = sy + i0, rb = sy + i1;
      const uint8_t *a = (ra >= 0 && ra < rows) ? image + ((size_t)ra * width + x) * 4 : clear;
      const uint8_t *b = (rb >= 0 && rb < rows) ? image + ((size_t)rb * width + x) * 4 : clear;
      uint8_t px[4];
      for (int ch = 0; ch < 4; ch++) {
        const float fa = (float)a[ch], fb = (float)b[ch];
        const float d = (fb - fa) * t;
        px[ch] = (uint8_t)(fa + d + 0.5f);
      }
      if (px[3] != 255 && have_bg) {
        const float fsa = (float)px[3] * (1.0f / 255.0f), fda = (float)bg[3] * (1.0f / 255.0f);
        const float k = fda * (1.0f - fsa), oa = fsa + k;
        if (!(oa > 0.0f)) {
          memset(px, 0, 4);
        } else {
          for (int ch = 0; ch < 3; ch++) {
            const float num = (float)px[ch] * fsa + (float)bg[ch] * k;
            px[ch] = (uint8_t)(num / oa + 0.5f);
          }
          px[3] = (uint8_t)(oa * 255.0f + 0.5f);
        }
      }
      memcpy(dst + 4 * x, px, 4);
    }
  }
}

/* ---- worker: spectrogramWorker.js -------------------------------------------------------- */

/* :195-203 */
void orc_worker_hann(int n, float *w) {
  const double invN = 1.0 / (n - 1), twoPI = 2 * M_PI;
  for (int i = 0; i < n; i++) w[i] = (float)(0.5 * (1 - cos(twoPI * i * invN)));
}

/* :110-125 */
void orc_worker_twiddles(int n, double *c, double *s) {
  for (int n1 = 1; n1 < n; n1 <<= 1) {
    const int n2 = n1 << 1;
    for (int j = 0; j < n1; j++) {
      const double angle = -2 * M_PI * j / n2;
      c[n1 - 1 + j] = cos(angle);
      s[n1 - 1 + j] = sin(angle);
    }
  }
}

/* :206-253 */
void orc_worker_fft(float *re, float *im, int n, const double *ct, const double *st) {
  int bits = 0;
  while ((1 << bits) < n) bits++;
  for (int i = 0; i < n; i++) { /* :177-193, :210-217 */
    unsigned x = (unsigned)i, y = 0;
    for (int j = 0; j < bits; j++) {
      y = (y << 1) | (x & 1);
      x >>= 1;
    }
    if ((int)y > i) {
      float t = re[i]; re[i] = re[y]; re[y] = t;
      t = im[i]; im[i] = im[y]; im[y] = t;
    }
  }
  for (int n1 = 1; n1 < n; n1 <<= 1) { /* :233-252 */
    const int n2 = n1 << 1;
    for (int j = 0; j < n1; j++) {
      const double c = ct[n1 - 1 + j], s = st[n1 - 1 + j];
      for (int i = j; i < n; i += n2) {
        const int k = i + n1;
        const double tr = c * (double)re[k] - s * (double)im[k];
        const double ti = c * (double)im[k] + s * (double)re[k];
        re[k] = (float)((double)re[i] - tr);
        im[k] = (float)((double)im[i] - ti);
        re[i] = (float)((double)re[i] + tr);
        im[i] = (float)((double)im[i] + ti);
      }
    }
  }
}

/* :39 */
int orc_worker_hop(int n, double overlap_pct) {
  double h = floor((double)n * (1 - overlap_pct / 100));
  return h > 1 ? (int)h : 1;
}

/* :40 */
size_t orc_worker_width(size_t length, int n, int hop) {
  if (length <= (size_t)n) return 1;
  size_t w = (length - (size_t)n + (size_t)hop - 1) / (size_t)hop;
  return w > 1 ? w : 1;
}

/* :59 */
size_t orc_worker_frames(size_t length, int n, int hop) {
  if (length < (size_t)n) return 0;
  return (length - (size_t)n) / (size_t)hop + 1;
}

/* :128-143 */
void orc_worker_lut(const double *cm, uint8_t *lut) {
  for (int i = 0; i < 256; i++) {
    for (int ch = 0; ch < 4; ch++) {
      double v = cm[i * 4 + ch];
      if (ch < 3) {
        if (isnan(v)) v = 0; /* cc[ch] || 0 */
      } else {
        if (isnan(v)) v = 1; /* cc[3] == null ? 1 : cc[3] */
      }
      lut[i * 4 + ch] = js_to_uint8_clamp(js_round(255 * v));
    }
  }
}

/* :71-78 */
static uint8_t worker_index(float re, float im) {
  const double magSq = (double)re * (double)re + (double)im * (double)im;
  const double mag = sqrt(magSq);
  double val = (mag > 1e-12) ? log10(mag) / 5 : -2.4;
  val = val < 0 ? 0 : (val > 1 ? 1 : val);
  return (uint8_t)floor(val * 255);
}

typedef struct {
  const float *x;
  int n, hop;
  size_t width, height, f0, f1;
  const uint8_t *lut;
  const float *win;
  const double *ct, *st;
  uint8_t *rgba, *idx_out;
} worker_job;

static void worker_columns(const worker_job *J) {
  const int n = J->n;
  float *re = (float *)malloc(sizeof(float) * (size_t)n);
  float *im = (float *)malloc(sizeof(float) * (size_t)n);
  const size_t W = J->width, H = J->height, total = W * H;
  for (size_t f = J->f0; f < J->f1; f++) { /* :59-95 */
    const float *sig = J->x + f * (size_t)J->hop;
    for (int j = 0; j < n; j++) {
      re[j] = (float)((double)sig[j] * (double)J->win[j]);
      im[j] = 0.0f;
    }
    orc_worker_fft(re, im, n, J->ct, J->st);
    for (size_t y = 0; y < H; y++) {
      const uint8_t col = worker_index(re[y], im[y]);
      if (J->idx_out) J->idx_out[f * H + y] = col;
      const size_t idx = (H - 1 - y) * W + f; /* :79 */
      if (idx >= total) continue;             /* typed-array store past the end is dropped */
      uint8_t *px = J->rgba + idx * 4;
      if (J->lut) {
        memcpy(px, J->lut + 4 * (size_t)col, 4);
      } else {
        px[0] = col; px[1] = col; px[2] = col; px[3] = 255;
      }
    }
  }
  free(re);
  free(im);
}

static void *worker_thread(void *p) {
  worker_columns((const worker_job *)p);
  return NULL;
}

static void worker_render_impl(const float *x, size_t length, int n, double overlap_pct,
                               const uint8_t *lut, const double *ct_in, const double *st_in,
                               uint8_t *rgba, uint8_t *idx_out, int threads) {
  const int hop = orc_worker_hop(n, overlap_pct);
  const size_t W = orc_worker_width(length, n, hop), H = (size_t)n / 2;
  const size_t frames = orc_worker_frames(length, n, hop);
  memset(rgba, 0, W * H * 4);
  float *win = (float *)malloc(sizeof(float) * (size_t)n);
  orc_worker_hann(n, win);
  double *ct = NULL, *st = NULL;
  if (!ct_in || !st_in) {
    ct = (double *)malloc(sizeof(double) * (size_t)n);
    st = (double *)malloc(sizeof(double) * (size_t)n);
    orc_worker_twiddles(n, ct, st);
    ct_in = ct;
    st_in = st;
  }
  worker_job base = {x, n, hop, W, H, 0, 0, lut, win, ct_in, st_in, rgba, idx_out};
  /* frames beyond the image width alias earlier pixels, so they must run last, in order */
  const size_t par = frames < W ? frames : W;
  if (threads <= 1 || par < (size_t)threads * 2) {
    base.f0 = 0;
    base.f1 = frames;
    worker_columns(&base);
  } else {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    worker_job *jobs = (worker_job *)malloc(sizeof(worker_job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
      jobs[t] = base;
      jobs[t].f0 = par * (size_t)t / (size_t)threads;
      jobs[t].f1 = par * (size_t)(t + 1) / (size_t)threads;
      pthread_create(&th[t], NULL, worker_thread, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    if (frames > par) {
      base.f0 = par;
      base.f1 = frames;
      worker_columns(&base);
    }
  }
  free(win);
  free(ct);
  free(st);
}

void orc_worker_render(const float *x, size_t length, int n, double overlap_pct, const uint8_t *lut,
                       const double *ct, const double *st, uint8_t *rgba, uint8_t *idx_out) {
  worker_render_impl(x, length, n, overlap_pct, lut, ct, st, rgba, idx_out, 1);
}

void orc_worker_render_mt(const float *x, size_t length, int n, double overlap_pct,
                          const uint8_t *lut, uint8_t *rgba, int threads) {
  worker_render_impl(x, length, n, overlap_pct, lut, NULL, NULL, rgba, NULL, threads);
}

void orc_worker_magnitudes(const float *x, size_t length, int n, int hop, size_t frames, float *mag_out,
                           uint8_t *idx_out) {
  (void)length;
  float *re = (float *)malloc(sizeof(float) * (size_t)n), *im = (float *)malloc(sizeof(float) * (size_t)n);
  float *win = (float *)malloc(sizeof(float) * (size_t)n);
  double *ct = (double *)malloc(sizeof(double) * (size_t)n), *st = (double *)malloc(sizeof(double) * (size_t)n);
  orc_worker_hann(n, win);
  orc_worker_twiddles(n, ct, st);
  const size_t H = (size_t)n / 2;
  for (size_t f = 0; f < frames; f++) {
    const float *sig = x + f * (size_t)hop;
    for (int j = 0; j < n; j++) {
      re[j] = (float)((double)sig[j] * (double)win[j]);
      im[j] = 0.0f;
    }
    orc_worker_fft(re, im, n, ct, st);
    for (size_t y = 0; y < H; y++) {
      const double e = re[y], s = im[y];
      if (mag_out) mag_out[f * H + y] = (float)sqrt(e * e + s * s);
      if (idx_out) idx_out[f * H + y] = worker_index(re[y], im[y]);
    }
  }
  free(re); free(im); free(win); free(ct); free(st);
}

/* ---- per-frame features ------------------------------------------------------------------- */

void orc_frame_peaks_f32(const float *mag, size_t frames, int bins, int lo, int hi, int32_t *peak_bin) {
  for (size_t f = 0; f < frames; f++) {
    const float *row = mag + f * (size_t)bins;
    float best = -1.0f;
    int bk = lo;
    for (int k = lo; k < hi; k++)
      if (row[k] > best) {
        best = row[k];
        bk = k;
      }
    peak_bin[f] = bk;
  }
}

void orc_frame_peaks(const uint8_t *idx, size_t frames, int bins, int lo, int hi, int32_t *peak_bin,
                     uint8_t *peak_idx) {
  for (size_t f = 0; f < frames; f++) {
    const uint8_t *row = idx + f * (size_t)bins;
    int best = lo;
    uint8_t bv = row[lo];
    for (int k = lo + 1; k < hi; k++)
      if (row[k] > bv) {
        bv = row[k];
        best = k;
      }
    peak_bin[f] = best;
    peak_idx[f] = bv;
  }
}
