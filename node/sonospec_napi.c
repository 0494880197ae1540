/*
 * sonospec_napi.c -- Node.js N-API addon over the C ABI of libsonospec.so (include/sonospec.h).
 *
 * Thin by design: every export is one C-ABI call with typed-array arguments.  The JavaScript side
 * (index.js, spectrogram_plugin.js, spectrogram_worker.js) carries the reference's interfaces.
 * Plain C against node_api.h (N-API 3 or later); no C++ wrapper library is needed.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sonospec.h"

#define NAPI_OK(env, call)                                                  \
  do {                                                                      \
    if ((call) != napi_ok) {                                                \
      napi_throw_error((env), NULL, "N-API call failed: " #call);           \
      return NULL;                                                          \
    }                                                                       \
  } while (0)

/* argv holds up to 8 values; argc is clamped to that (napi_get_cb_info reports the real count) */
#define ARGS(env, info, n)                                                  \
  size_t argc = 12;                                                         \
  napi_value argv[12];                                                      \
  NAPI_OK(env, napi_get_cb_info((env), (info), &argc, argv, NULL, NULL));   \
  if (argc > 12) argc = 12;                                                 \
  if (argc < (n)) {                                                         \
    napi_throw_type_error((env), NULL, "too few arguments");                \
    return NULL;                                                            \
  }

static napi_value throw_msg(napi_env env, const char* msg) {
  napi_throw_error(env, NULL, msg);
  return NULL;
}

static int get_u32(napi_env env, napi_value v, uint32_t* out) {
  double d;
  if (napi_get_value_double(env, v, &d) != napi_ok) return 0;
  *out = (uint32_t)d;
  return 1;
}
static int get_f64(napi_env env, napi_value v, double* out) { return napi_get_value_double(env, v, out) == napi_ok; }
static int get_u64(napi_env env, napi_value v, uint64_t* out) {
  double d;
  if (napi_get_value_double(env, v, &d) != napi_ok || d < 0) return 0;
  *out = (uint64_t)d;
  return 1;
}

static napi_value make_num(napi_env env, double d) {
  napi_value v;
  napi_create_double(env, d, &v);
  return v;
}

static int prop_u32(napi_env env, napi_value obj, const char* name, uint32_t* out) {
  napi_value v;
  bool has = false;
  if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return 0;
  if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
  return get_u32(env, v, out);
}
static int prop_f64(napi_env env, napi_value obj, const char* name, double* out) {
  napi_value v;
  bool has = false;
  if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return 0;
  if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
  return get_f64(env, v, out);
}

/* typed array -> raw pointer; returns element count or -1.  null/undefined -> (NULL, 0). */
static int64_t ta_data(napi_env env, napi_value v, napi_typedarray_type* type, void** data) {
  napi_valuetype vt;
  if (napi_typeof(env, v, &vt) != napi_ok) return -1;
  if (vt == napi_null || vt == napi_undefined) {
    *data = NULL;
    return 0;
  }
  bool is_ta = false;
  if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta) return -1;
  size_t len = 0, off = 0;
  napi_value ab;
  if (napi_get_typedarray_info(env, v, type, &len, data, &ab, &off) != napi_ok) return -1;
  return (int64_t)len;
}

static napi_value new_typed(napi_env env, napi_typedarray_type type, size_t count, size_t elem, void** data) {
  napi_value ab, ta;
  if (napi_create_arraybuffer(env, count * elem, data, &ab) != napi_ok) return throw_msg(env, "allocation failed");
  if (napi_create_typedarray(env, type, count, ab, 0, &ta) != napi_ok) return throw_msg(env, "allocation failed");
  return ta;
}

/* ---- context handle ---------------------------------------------------------------------- */

typedef struct {
  sono_ctx* ctx;
} handle_t;

static void handle_finalize(napi_env env, void* data, void* hint) {
  (void)env;
  (void)hint;
  handle_t* h = (handle_t*)data;
  if (h->ctx) sono_destroy(h->ctx);
  free(h);
}

static handle_t* get_handle(napi_env env, napi_value v) {
  void* p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((handle_t*)p)->ctx) {
    napi_throw_type_error(env, NULL, "not a live sonospec context");
    return NULL;
  }
  return (handle_t*)p;
}

static napi_value throw_ctx(napi_env env, handle_t* h, int code) {
  char buf[600];
  snprintf(buf, sizeof buf, "%s (sonospec status %d)", sono_last_error(h->ctx), code);
  napi_throw_error(env, NULL, buf);
  return NULL;
}

/* create({fftN, hop, windowId, scaleMode, frameRule, twiddleMode, arith, outKind, inKind, binLo, binHi,
 *         alpha, gainDB, rangeDB}, device) */
static napi_value js_create(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  sono_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.fft_n = 1024;
  cfg.hop = 256;
  cfg.window_id = SONO_WIN_HANN;
  cfg.gain_db = 20;
  cfg.range_db = 80;
  napi_value o = argv[0];
  prop_u32(env, o, "fftN", &cfg.fft_n);
  prop_u32(env, o, "hop", &cfg.hop);
  prop_u32(env, o, "windowId", &cfg.window_id);
  prop_u32(env, o, "scaleMode", &cfg.scale_mode);
  prop_u32(env, o, "frameRule", &cfg.frame_rule);
  prop_u32(env, o, "twiddleMode", &cfg.twiddle_mode);
  prop_u32(env, o, "arith", &cfg.arith);
  prop_u32(env, o, "outKind", &cfg.out_kind);
  prop_u32(env, o, "inKind", &cfg.in_kind);
  prop_u32(env, o, "binLo", &cfg.bin_lo);
  prop_u32(env, o, "binHi", &cfg.bin_hi);
  prop_f64(env, o, "alpha", &cfg.alpha);
  prop_f64(env, o, "gainDB", &cfg.gain_db);
  prop_f64(env, o, "rangeDB", &cfg.range_db);
  uint32_t device = 0;
  if (argc > 1) get_u32(env, argv[1], &device);
  sono_ctx* c = sono_create((int)device, &cfg);
  if (!c) return throw_msg(env, sono_create_error());
  handle_t* h = (handle_t*)calloc(1, sizeof *h);
  h->ctx = c;
  napi_value ext;
  if (napi_create_external(env, h, handle_finalize, NULL, &ext) != napi_ok) {
    sono_destroy(c);
    free(h);
    return throw_msg(env, "napi_create_external failed");
  }
  return ext;
}

static napi_value js_destroy(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  void* p = NULL;
  if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
    handle_t* h = (handle_t*)p;
    if (h->ctx) sono_destroy(h->ctx);
    h->ctx = NULL;
  }
  return NULL;
}

static napi_value js_device_count(napi_env env, napi_callback_info info) {
  (void)info;
  return make_num(env, sono_device_count());
}

/* deviceNumaNode(device) -> NUMA node of the host memory next to the GPU, -1 when unknown */
static napi_value js_device_numa_node(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  int32_t dev = 0;
  if (napi_get_value_int32(env, argv[0], &dev) != napi_ok) return throw_msg(env, "device must be a number");
  return make_num(env, sono_device_numa_node(dev));
}

static napi_value js_set_colormap(napi_env env, napi_callback_info info) {
  ARGS(env, info, 2);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  napi_typedarray_type t;
  void* d;
  int64_t n = ta_data(env, argv[1], &t, &d);
  if (n < 0 || (d && (n != 1024 || (t != napi_uint8_array && t != napi_uint8_clamped_array))))
    return throw_msg(env, "colour LUT must be a Uint8Array of 1024 bytes or null");
  int rc = sono_set_colormap(h->ctx, (const uint8_t*)d);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

static napi_value js_set_tables(napi_env env, napi_callback_info info) {
  ARGS(env, info, 4);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  sono_config cfg;
  sono_get_config(h->ctx, &cfg);
  napi_typedarray_type t;
  void *w, *c, *s;
  int64_t nw = ta_data(env, argv[1], &t, &w);
  if (nw < 0 || (w && (t != napi_float32_array || nw != (int64_t)cfg.fft_n))) return throw_msg(env, "window must be Float32Array(fftN)");
  int64_t nc = ta_data(env, argv[2], &t, &c);
  if (nc < 0 || (c && (t != napi_float64_array || nc != (int64_t)cfg.fft_n - 1))) return throw_msg(env, "twiddle cos must be Float64Array(fftN-1)");
  int64_t ns = ta_data(env, argv[3], &t, &s);
  if (ns < 0 || (s && (t != napi_float64_array || ns != (int64_t)cfg.fft_n - 1))) return throw_msg(env, "twiddle sin must be Float64Array(fftN-1)");
  int rc = sono_set_tables(h->ctx, (const float*)w, (const double*)c, (const double*)s);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

static napi_value geometry_object(napi_env env, const sono_geometry* g) {
  napi_value o;
  napi_create_object(env, &o);
  napi_set_named_property(env, o, "frames", make_num(env, (double)g->frames));
  napi_set_named_property(env, o, "width", make_num(env, (double)g->width));
  napi_set_named_property(env, o, "bins", make_num(env, (double)g->bins));
  napi_set_named_property(env, o, "hop", make_num(env, (double)g->hop));
  return o;
}

static napi_value js_geometry(napi_env env, napi_callback_info info) {
  ARGS(env, info, 2);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  uint64_t L;
  if (!get_u64(env, argv[1], &L)) return throw_msg(env, "sample count expected");
  sono_geometry g;
  int rc = sono_geometry_of(h->ctx, L, &g);
  if (rc != SONO_OK) return throw_ctx(env, h, rc);
  return geometry_object(env, &g);
}

/* stft(handle, Float32Array | Int16Array) -> typed array of the context's out_kind */
static napi_value js_stft(napi_env env, napi_callback_info info) {
  ARGS(env, info, 2);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  sono_config cfg;
  sono_get_config(h->ctx, &cfg);
  napi_typedarray_type t;
  void* pcm;
  int64_t n = ta_data(env, argv[1], &t, &pcm);
  if (n < 0) return throw_msg(env, "PCM must be a typed array");
  if (cfg.in_kind == SONO_IN_F32 && n > 0 && t != napi_float32_array) return throw_msg(env, "PCM must be a Float32Array");
  if (cfg.in_kind == SONO_IN_I16 && n > 0 && t != napi_int16_array) return throw_msg(env, "PCM must be an Int16Array");
  const uint64_t bytes = sono_out_bytes(h->ctx, (uint64_t)n);
  void* out = NULL;
  napi_value result;
  if (cfg.out_kind == SONO_OUT_U8_INDEX)
    result = new_typed(env, napi_uint8_array, (size_t)bytes, 1, &out);
  else if (cfg.out_kind == SONO_OUT_RGBA8)
    result = new_typed(env, napi_uint8_clamped_array, (size_t)bytes, 1, &out);
  else if (cfg.out_kind == SONO_OUT_FEATURES)   /* raw sono_frame_feature records, 8 bytes each */
    result = new_typed(env, napi_uint8_array, (size_t)bytes, 1, &out);
  else
    result = new_typed(env, napi_float32_array, (size_t)(bytes / 4), 4, &out);
  if (!result) return NULL;
  static uint8_t dummy_in[4], dummy_out[4];
  int rc = sono_stft(h->ctx, pcm ? pcm : dummy_in, (uint64_t)n, bytes ? out : dummy_out, bytes ? bytes : 0, SONO_MEM_HOST);
  if (rc != SONO_OK) return throw_ctx(env, h, rc);
  return result;
}

static napi_value js_resample(napi_env env, napi_callback_info info) {
  ARGS(env, info, 5);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  napi_typedarray_type t;
  void* in;
  int64_t n = ta_data(env, argv[1], &t, &in);
  uint64_t F, W;
  uint32_t B;
  if (n < 0 || !get_u64(env, argv[2], &F) || !get_u32(env, argv[3], &B) || !get_u64(env, argv[4], &W) || (uint64_t)n < F * B)
    return throw_msg(env, "resample(handle, Uint8Array, frames, bins, outWidth)");
  void* out;
  napi_value result = new_typed(env, napi_uint8_array, (size_t)(W * B), 1, &out);
  if (!result) return NULL;
  if (W * B == 0) return result;
  if (F == 0) {
    memset(out, 0, (size_t)(W * B));
    return result;
  }
  int rc = sono_resample(h->ctx, (const uint8_t*)in, F, B, W, (uint8_t*)out, SONO_MEM_HOST);
  return rc == SONO_OK ? result : throw_ctx(env, h, rc);
}

static napi_value js_colorize(napi_env env, napi_callback_info info) {
  ARGS(env, info, 6);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  napi_typedarray_type t;
  void* in;
  int64_t n = ta_data(env, argv[1], &t, &in);
  uint64_t W;
  uint32_t B, row0, rows;
  if (n < 0 || !get_u64(env, argv[2], &W) || !get_u32(env, argv[3], &B) || !get_u32(env, argv[4], &row0) ||
      !get_u32(env, argv[5], &rows) || (uint64_t)n < W * B)
    return throw_msg(env, "colorize(handle, Uint8Array, width, bins, row0, rows)");
  void* out;
  napi_value result = new_typed(env, napi_uint8_clamped_array, (size_t)(W * rows * 4), 1, &out);
  if (!result) return NULL;
  if (W * rows == 0) return result;
  int rc = sono_colorize(h->ctx, (const uint8_t*)in, W, B, row0, rows, (uint8_t*)out, SONO_MEM_HOST);
  return rc == SONO_OK ? result : throw_ctx(env, h, rc);
}

static napi_value js_frame_peaks(napi_env env, napi_callback_info info) {
  ARGS(env, info, 4);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  napi_typedarray_type t;
  void* in;
  int64_t n = ta_data(env, argv[1], &t, &in);
  uint64_t F;
  uint32_t B;
  if (n < 0 || !get_u64(env, argv[2], &F) || !get_u32(env, argv[3], &B) || (uint64_t)n < F * B)
    return throw_msg(env, "framePeaks(handle, Uint8Array, frames, bins)");
  void *pb, *pv;
  napi_value bins = new_typed(env, napi_int32_array, (size_t)F, 4, &pb);
  napi_value vals = new_typed(env, napi_uint8_array, (size_t)F, 1, &pv);
  if (!bins || !vals) return NULL;
  if (F) {
    int rc = sono_frame_peaks(h->ctx, (const uint8_t*)in, F, B, (int32_t*)pb, (uint8_t*)pv, SONO_MEM_HOST);
    if (rc != SONO_OK) return throw_ctx(env, h, rc);
  }
  napi_value o;
  napi_create_object(env, &o);
  napi_set_named_property(env, o, "bins", bins);
  napi_set_named_property(env, o, "values", vals);
  return o;
}

/* ---- device-resident path (benchmarks, batch hosts): handles are opaque numbers ------------- */

static napi_value js_device_malloc(napi_env env, napi_callback_info info) {
  ARGS(env, info, 2);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  uint64_t bytes;
  if (!get_u64(env, argv[1], &bytes)) return throw_msg(env, "byte count expected");
  void* p = sono_device_malloc(h->ctx, bytes);
  if (!p) return throw_ctx(env, h, SONO_ERR_DEVICE);
  napi_value v;
  napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)p, &v);
  return v;
}

static int get_ptr(napi_env env, napi_value v, void** out) {
  uint64_t u;
  bool lossless;
  if (napi_get_value_bigint_uint64(env, v, &u, &lossless) != napi_ok) return 0;
  *out = (void*)(uintptr_t)u;
  return 1;
}

static napi_value js_device_free(napi_env env, napi_callback_info info) {
  ARGS(env, info, 2);
  handle_t* h = get_handle(env, argv[0]);
  void* p;
  if (!h || !get_ptr(env, argv[1], &p)) return throw_msg(env, "deviceFree(handle, pointer)");
  int rc = sono_device_free(h->ctx, p);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

static napi_value js_copy_to_device(napi_env env, napi_callback_info info) {
  ARGS(env, info, 3);
  handle_t* h = get_handle(env, argv[0]);
  void* p;
  if (!h || !get_ptr(env, argv[1], &p)) return throw_msg(env, "copyToDevice(handle, pointer, typedArray)");
  napi_typedarray_type t;
  void* src;
  size_t len, off;
  napi_value ab;
  if (napi_get_typedarray_info(env, argv[2], &t, &len, &src, &ab, &off) != napi_ok) return throw_msg(env, "typed array expected");
  static const size_t esz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
  int rc = sono_copy_to_device(h->ctx, p, src, (uint64_t)len * esz[t]);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

static napi_value js_copy_to_host(napi_env env, napi_callback_info info) {
  ARGS(env, info, 3);
  handle_t* h = get_handle(env, argv[0]);
  void* p;
  if (!h || !get_ptr(env, argv[2], &p)) return throw_msg(env, "copyToHost(handle, typedArray, pointer)");
  napi_typedarray_type t;
  void* dst;
  size_t len, off;
  napi_value ab;
  if (napi_get_typedarray_info(env, argv[1], &t, &len, &dst, &ab, &off) != napi_ok) return throw_msg(env, "typed array expected");
  static const size_t esz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
  int rc = sono_copy_to_host(h->ctx, dst, p, (uint64_t)len * esz[t]);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

static napi_value js_out_bytes(napi_env env, napi_callback_info info) {
  ARGS(env, info, 2);
  handle_t* h = get_handle(env, argv[0]);
  uint64_t L;
  if (!h || !get_u64(env, argv[1], &L)) return throw_msg(env, "outBytes(handle, samples)");
  return make_num(env, (double)sono_out_bytes(h->ctx, L));
}

/* stftDevice(handle, pcmPtr, samples, outPtr, outCapacity, wait) */
static napi_value js_stft_device(napi_env env, napi_callback_info info) {
  ARGS(env, info, 5);
  handle_t* h = get_handle(env, argv[0]);
  void *in, *out;
  uint64_t L, cap;
  if (!h || !get_ptr(env, argv[1], &in) || !get_u64(env, argv[2], &L) || !get_ptr(env, argv[3], &out) || !get_u64(env, argv[4], &cap))
    return throw_msg(env, "stftDevice(handle, pcmPtr, samples, outPtr, outCapacity[, wait])");
  bool wait = true;
  if (argc > 5) napi_get_value_bool(env, argv[5], &wait);
  int rc = sono_stft_async(h->ctx, in, L, out, cap);
  if (rc == SONO_OK && wait) rc = sono_wait(h->ctx);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

static napi_value js_wait(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  int rc = sono_wait(h->ctx);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

static napi_value js_timer_start(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  int rc = sono_timer_start(h->ctx);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

static napi_value js_timer_stop(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  return make_num(env, sono_timer_stop_ms(h->ctx));
}

/* ---- host rules ------------------------------------------------------------------------------ */

static napi_value js_worker_hop(napi_env env, napi_callback_info info) {
  ARGS(env, info, 2);
  uint32_t n;
  double pct;
  if (!get_u32(env, argv[0], &n) || !get_f64(env, argv[1], &pct)) return throw_msg(env, "workerHop(fftN, overlapPercent)");
  return make_num(env, sono_worker_hop(n, pct));
}

static napi_value js_plugin_noverlap(napi_env env, napi_callback_info info) {
  ARGS(env, info, 4);
  uint32_t n, ov, cw;
  uint64_t L;
  if (!get_u32(env, argv[0], &n) || !get_u32(env, argv[1], &ov) || !get_u64(env, argv[2], &L) || !get_u32(env, argv[3], &cw))
    return throw_msg(env, "pluginNoverlap(fftN, noverlap, samples, canvasWidth)");
  return make_num(env, sono_plugin_noverlap(n, ov, L, cw));
}

static napi_value js_overlap_from_percent(napi_env env, napi_callback_info info) {
  ARGS(env, info, 2);
  uint32_t n;
  double pct;
  if (!get_u32(env, argv[0], &n) || !get_f64(env, argv[1], &pct)) return throw_msg(env, "overlapFromPercent(fftN, percent)");
  return make_num(env, sono_overlap_from_percent(n, pct));
}

static napi_value js_plugin_crop(napi_env env, napi_callback_info info) {
  ARGS(env, info, 4);
  uint32_t bins;
  double sr, fmin, fmax;
  if (!get_u32(env, argv[0], &bins) || !get_f64(env, argv[1], &sr) || !get_f64(env, argv[2], &fmin) || !get_f64(env, argv[3], &fmax))
    return throw_msg(env, "pluginCrop(bins, sampleRate, fminHz, fmaxHz)");
  int64_t sy, sh;
  if (sono_plugin_crop(bins, sr, fmin, fmax, &sy, &sh) != SONO_OK) return throw_msg(env, "bad crop arguments");
  napi_value o;
  napi_create_object(env, &o);
  napi_set_named_property(env, o, "sy", make_num(env, (double)sy));
  napi_set_named_property(env, o, "sh", make_num(env, (double)sh));
  return o;
}

static napi_value lut_common(napi_env env, napi_callback_info info, int worker) {
  ARGS(env, info, 1);
  napi_typedarray_type t;
  void* cm;
  int64_t n = ta_data(env, argv[0], &t, &cm);
  if (n != 1024 || t != napi_float64_array) return throw_msg(env, "colour map must be a Float64Array of 256*4 values");
  void* out;
  napi_value result = new_typed(env, napi_uint8_array, 1024, 1, &out);
  if (!result) return NULL;
  if (worker)
    sono_worker_lut((const double*)cm, (uint8_t*)out);
  else
    sono_plugin_lut((const double*)cm, (uint8_t*)out);
  return result;
}
static napi_value js_plugin_lut(napi_env env, napi_callback_info info) { return lut_common(env, info, 0); }
static napi_value js_worker_lut(napi_env env, napi_callback_info info) { return lut_common(env, info, 1); }

static napi_value js_window_table(napi_env env, napi_callback_info info) {
  ARGS(env, info, 3);
  uint32_t id, n;
  double alpha;
  if (!get_u32(env, argv[0], &id) || !get_u32(env, argv[1], &n) || !get_f64(env, argv[2], &alpha))
    return throw_msg(env, "windowTable(windowId, fftN, alpha)");
  void* out;
  napi_value result = new_typed(env, napi_float32_array, n, 4, &out);
  if (!result) return NULL;
  if (sono_window_table(id, n, alpha, (float*)out) != SONO_OK) return throw_msg(env, "No such window function");
  return result;
}

static napi_value js_geometry_for(napi_env env, napi_callback_info info) {
  ARGS(env, info, 4);
  uint32_t n, hop, rule, lo = 0, hi = 0;
  uint64_t L;
  if (!get_u32(env, argv[0], &n) || !get_u32(env, argv[1], &hop) || !get_u32(env, argv[2], &rule) || !get_u64(env, argv[3], &L))
    return throw_msg(env, "geometryFor(fftN, hop, frameRule, samples[, binLo, binHi])");
  if (argc > 4) get_u32(env, argv[4], &lo);
  if (argc > 5) get_u32(env, argv[5], &hi);
  sono_geometry g;
  if (sono_geometry_for(n, hop, rule, lo, hi, L, &g) != SONO_OK) return throw_msg(env, "bad geometry arguments");
  return geometry_object(env, &g);
}

/* ---- file intake (host side) ----------------------------------------------------------------- */

static void set_num(napi_env env, napi_value obj, const char* name, double v) { napi_set_named_property(env, obj, name, make_num(env, v)); }
static void set_str(napi_env env, napi_value obj, const char* name, const char* v) {
  napi_value sv;
  napi_create_string_utf8(env, v, NAPI_AUTO_LENGTH, &sv);
  napi_set_named_property(env, obj, name, sv);
}

/* wavParse(Uint8Array) -> {sampleRate, hasFmt, hasData, fmtSampleRate, formatTag, channels, bitsPerSample, blockAlign,
 *                          dataOffset, dataBytes, dataBytesPresent, frames, duration} */
static napi_value js_wav_parse(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  napi_typedarray_type t;
  void* data;
  int64_t n = ta_data(env, argv[0], &t, &data);
  if (n < 0 || t != napi_uint8_array) return throw_msg(env, "wavParse(Uint8Array)");
  sono_wav_info w;
  if (sono_wav_parse(data, (uint64_t)n, &w) != SONO_OK) {
    napi_throw_range_error(env, NULL, "Offset is outside the bounds of the DataView");
    return NULL;
  }
  napi_value o;
  napi_create_object(env, &o);
  set_num(env, o, "sampleRate", w.sample_rate);
  set_num(env, o, "hasFmt", w.has_fmt);
  set_num(env, o, "hasData", w.has_data);
  set_num(env, o, "fmtSampleRate", w.fmt_sample_rate);
  set_num(env, o, "formatTag", w.format_tag);
  set_num(env, o, "channels", w.channels);
  set_num(env, o, "bitsPerSample", w.bits_per_sample);
  set_num(env, o, "blockAlign", w.block_align);
  set_num(env, o, "dataOffset", (double)w.data_offset);
  set_num(env, o, "dataBytes", (double)w.data_bytes);
  set_num(env, o, "dataBytesPresent", (double)w.data_bytes_present);
  set_num(env, o, "frames", (double)w.frames);
  set_num(env, o, "duration", w.duration_s);
  return o;
}

/* wavGuano(Uint8Array) -> [offset, length] (length 0: none) */
static napi_value js_wav_guano(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  napi_typedarray_type t;
  void* data;
  int64_t n = ta_data(env, argv[0], &t, &data);
  if (n < 0 || t != napi_uint8_array) return throw_msg(env, "wavGuano(Uint8Array)");
  uint64_t off = 0, len = 0;
  if (sono_wav_guano(data, (uint64_t)n, &off, &len) != SONO_OK) {
    napi_throw_range_error(env, NULL, "Offset is outside the bounds of the DataView");
    return NULL;
  }
  napi_value arr;
  napi_create_array_with_length(env, 2, &arr);
  napi_set_element(env, arr, 0, make_num(env, (double)off));
  napi_set_element(env, arr, 1, make_num(env, (double)len));
  return arr;
}

/* guanoFields(Uint8Array utf8) -> {} | {date, time, latitude, longitude} */
static napi_value js_guano_fields(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  napi_typedarray_type t;
  void* data;
  int64_t n = ta_data(env, argv[0], &t, &data);
  if (n < 0 || (n > 0 && t != napi_uint8_array)) return throw_msg(env, "guanoFields(Uint8Array)");
  sono_guano_fields_t g;
  if (sono_guano_fields((const char*)data, (uint64_t)n, &g) != SONO_OK) return throw_msg(env, "guanoFields failed");
  napi_value o;
  napi_create_object(env, &o);
  if (g.present) {
    set_str(env, o, "date", g.date);
    set_str(env, o, "time", g.time);
    set_str(env, o, "latitude", g.latitude);
    set_str(env, o, "longitude", g.longitude);
  }
  return o;
}

/* wavCrop(Uint8Array, startSeconds, endSeconds) -> Uint8Array | null */
static napi_value js_wav_crop(napi_env env, napi_callback_info info) {
  ARGS(env, info, 3);
  napi_typedarray_type t;
  void* data;
  double t0, t1;
  int64_t n = ta_data(env, argv[0], &t, &data);
  if (n < 0 || t != napi_uint8_array || !get_f64(env, argv[1], &t0) || !get_f64(env, argv[2], &t1))
    return throw_msg(env, "wavCrop(Uint8Array, start, end)");
  int64_t need = sono_wav_crop(data, (uint64_t)n, t0, t1, NULL, 0);
  if (need < 0) return throw_msg(env, "wavCrop: unsupported file or times");
  napi_value nul;
  napi_get_null(env, &nul);
  if (need == 0) return nul;
  void* out;
  napi_value ta = new_typed(env, napi_uint8_array, (size_t)need, 1, &out);
  if (!ta) return NULL;
  if (sono_wav_crop(data, (uint64_t)n, t0, t1, out, (uint64_t)need) != need) return throw_msg(env, "wavCrop failed");
  return ta;
}

/* wavAccept(fileBytes, duration, timeExpansion) -> 0 | 1 | 2 */
static napi_value js_wav_accept(napi_env env, napi_callback_info info) {
  ARGS(env, info, 2);
  uint64_t fb;
  double d;
  bool te = false;
  if (!get_u64(env, argv[0], &fb) || !get_f64(env, argv[1], &d)) return throw_msg(env, "wavAccept(fileBytes, duration[, timeExpansion])");
  if (argc > 2) napi_get_value_bool(env, argv[2], &te);
  return make_num(env, sono_wav_accept(fb, d, te ? 1 : 0));
}

/* pcmToDevice(handle, Uint8Array raw, formatTag, bits, channels, channel, dstPtr) */
static napi_value js_pcm_to_device(napi_env env, napi_callback_info info) {
  ARGS(env, info, 7);
  handle_t* h = get_handle(env, argv[0]);
  napi_typedarray_type t;
  void *data, *dst;
  uint32_t tag, bits, ch, c;
  int64_t n = ta_data(env, argv[1], &t, &data);
  if (!h || n < 0 || t != napi_uint8_array || !get_u32(env, argv[2], &tag) || !get_u32(env, argv[3], &bits) || !get_u32(env, argv[4], &ch) ||
      !get_u32(env, argv[5], &c) || !get_ptr(env, argv[6], &dst) || !ch || bits < 8)
    return throw_msg(env, "pcmToDevice(handle, raw, formatTag, bits, channels, channel, dstPtr)");
  const uint64_t frames = (uint64_t)n / ((uint64_t)ch * (bits / 8));
  int rc = sono_pcm_to_f32(h->ctx, data, SONO_MEM_HOST, tag, bits, ch, c, frames, (float*)dst);
  if (rc != SONO_OK) return throw_ctx(env, h, rc);
  return make_num(env, (double)frames);
}

/* setFilterbank(handle, Int32Array bin | null, Float64Array w0, Float64Array w1) */
static napi_value js_set_filterbank(napi_env env, napi_callback_info info) {
  ARGS(env, info, 2);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  napi_typedarray_type t0, t1, t2;
  void *b = NULL, *w0 = NULL, *w1 = NULL;
  int64_t nb = ta_data(env, argv[1], &t0, &b);
  if (nb < 0) return throw_msg(env, "setFilterbank(handle, Int32Array|null, Float64Array, Float64Array)");
  if (!b) {
    int rc = sono_set_filterbank(h->ctx, NULL, NULL, NULL);
    return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
  }
  if (argc < 4) return throw_msg(env, "setFilterbank: weights missing");
  int64_t n0 = ta_data(env, argv[2], &t1, &w0), n1 = ta_data(env, argv[3], &t2, &w1);
  sono_config cfg;
  sono_get_config(h->ctx, &cfg);
  if (t0 != napi_int32_array || t1 != napi_float64_array || t2 != napi_float64_array || nb != (int64_t)cfg.fft_n / 2 || n0 != nb || n1 != nb)
    return throw_msg(env, "setFilterbank: fft_n/2 entries of Int32Array, Float64Array, Float64Array");
  int rc = sono_set_filterbank(h->ctx, (const int32_t*)b, (const double*)w0, (const double*)w1);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

/* pluginCropScaled(bins, sampleRate, fmin, fmax, scaleId) -> {sy, sh} */
static napi_value js_plugin_crop_scaled(napi_env env, napi_callback_info info) {
  ARGS(env, info, 5);
  uint32_t bins, scale;
  double sr, f0, f1;
  if (!get_u32(env, argv[0], &bins) || !get_f64(env, argv[1], &sr) || !get_f64(env, argv[2], &f0) || !get_f64(env, argv[3], &f1) ||
      !get_u32(env, argv[4], &scale))
    return throw_msg(env, "pluginCropScaled(bins, sampleRate, fmin, fmax, scale)");
  int64_t sy = 0, sh = 0;
  if (sono_plugin_crop_scaled(bins, sr, f0, f1, scale, &sy, &sh) != SONO_OK) return throw_msg(env, "pluginCropScaled: bad arguments");
  napi_value o;
  napi_create_object(env, &o);
  napi_set_named_property(env, o, "sy", make_num(env, (double)sy));
  napi_set_named_property(env, o, "sh", make_num(env, (double)sh));
  return o;
}

/* pluginDrawRect(sampleRate, fmin, fmax, scale, height) -> {dy, dh, fill} */
static napi_value js_plugin_draw_rect(napi_env env, napi_callback_info info) {
  ARGS(env, info, 5);
  uint32_t scale, height;
  double sr, f0, f1;
  if (!get_f64(env, argv[0], &sr) || !get_f64(env, argv[1], &f0) || !get_f64(env, argv[2], &f1) || !get_u32(env, argv[3], &scale) ||
      !get_u32(env, argv[4], &height))
    return throw_msg(env, "pluginDrawRect(sampleRate, fmin, fmax, scale, height)");
  double dy = 0, dh = 0;
  int fill = 0;
  if (sono_plugin_draw_rect(sr, f0, f1, scale, height, &dy, &dh, &fill) != SONO_OK) return throw_msg(env, "pluginDrawRect: bad arguments");
  napi_value o, b;
  napi_create_object(env, &o);
  napi_set_named_property(env, o, "dy", make_num(env, dy));
  napi_set_named_property(env, o, "dh", make_num(env, dh));
  napi_get_boolean(env, fill != 0, &b);
  napi_set_named_property(env, o, "fill", b);
  return o;
}

/* pluginBackground(Float64Array(4)) -> Uint8Array(4) */
static napi_value js_plugin_background(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  napi_typedarray_type t;
  void* in;
  int64_t n = ta_data(env, argv[0], &t, &in);
  if (n < 4 || t != napi_float64_array) return throw_msg(env, "pluginBackground(Float64Array(4))");
  void* out;
  napi_value result = new_typed(env, napi_uint8_array, 4, 1, &out);
  if (!result) return NULL;
  if (sono_plugin_background((const double*)in, (uint8_t*)out) != SONO_OK) return throw_msg(env, "pluginBackground: bad arguments");
  return result;
}

/* canvasDraw(handle, Uint8Array columns, width, bins, sy, sh, dy, dh, height, background Uint8Array(4) | null)
 * -> Uint8ClampedArray(height * width * 4) */
static napi_value js_canvas_draw(napi_env env, napi_callback_info info) {
  ARGS(env, info, 10);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  napi_typedarray_type t;
  void* in;
  int64_t n = ta_data(env, argv[1], &t, &in);
  uint64_t W;
  uint32_t B, height;
  double sy, sh, dy, dh;
  if (n < 0 || !get_u64(env, argv[2], &W) || !get_u32(env, argv[3], &B) || !get_f64(env, argv[4], &sy) || !get_f64(env, argv[5], &sh) ||
      !get_f64(env, argv[6], &dy) || !get_f64(env, argv[7], &dh) || !get_u32(env, argv[8], &height) || (uint64_t)n < W * B)
    return throw_msg(env, "canvasDraw(handle, Uint8Array, width, bins, sy, sh, dy, dh, height, background)");
  const uint8_t* bg = NULL;
  napi_valuetype vt;
  napi_typeof(env, argv[9], &vt);
  if (vt != napi_null && vt != napi_undefined) {
    void* bgp;
    int64_t bn = ta_data(env, argv[9], &t, &bgp);
    if (bn < 4) return throw_msg(env, "canvasDraw: background must hold 4 bytes");
    bg = (const uint8_t*)bgp;
  }
  void* out;
  napi_value result = new_typed(env, napi_uint8_clamped_array, (size_t)(W * height * 4), 1, &out);
  if (!result) return NULL;
  if (W * height == 0) return result;
  int rc = sono_canvas_draw(h->ctx, (const uint8_t*)in, W, B, (int64_t)sy, (int64_t)sh, dy, dh, height, bg, (uint8_t*)out, SONO_MEM_HOST);
  return rc == SONO_OK ? result : throw_ctx(env, h, rc);
}

/* resampleColsDevice(handle, inPtr, frames, bins, outWidth, col0, ncols, outPtr): device to device */
static napi_value js_resample_cols_device(napi_env env, napi_callback_info info) {
  ARGS(env, info, 8);
  handle_t* h = get_handle(env, argv[0]);
  void *in, *out;
  uint64_t F, W, col0, ncols;
  uint32_t B;
  if (!h || !get_ptr(env, argv[1], &in) || !get_u64(env, argv[2], &F) || !get_u32(env, argv[3], &B) || !get_u64(env, argv[4], &W) ||
      !get_u64(env, argv[5], &col0) || !get_u64(env, argv[6], &ncols) || !get_ptr(env, argv[7], &out))
    return throw_msg(env, "resampleColsDevice(handle, inPtr, frames, bins, outWidth, col0, ncols, outPtr)");
  int rc = sono_resample_cols(h->ctx, (const uint8_t*)in, F, B, W, col0, ncols, (uint8_t*)out, SONO_MEM_DEVICE);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

/* canvasDrawDevice(handle, colsPtr, width, bins, sy, sh, dy, dh, height, background Uint8Array(4) | null, outPtr) */
static napi_value js_canvas_draw_device(napi_env env, napi_callback_info info) {
  ARGS(env, info, 11);
  handle_t* h = get_handle(env, argv[0]);
  void *in, *out;
  uint64_t W;
  uint32_t B, height;
  double sy, sh, dy, dh;
  if (!h || !get_ptr(env, argv[1], &in) || !get_u64(env, argv[2], &W) || !get_u32(env, argv[3], &B) || !get_f64(env, argv[4], &sy) ||
      !get_f64(env, argv[5], &sh) || !get_f64(env, argv[6], &dy) || !get_f64(env, argv[7], &dh) || !get_u32(env, argv[8], &height) ||
      !get_ptr(env, argv[10], &out))
    return throw_msg(env, "canvasDrawDevice(handle, colsPtr, width, bins, sy, sh, dy, dh, height, background, outPtr)");
  napi_typedarray_type t;
  void* bgp = NULL;
  int64_t bn = ta_data(env, argv[9], &t, &bgp);
  if (bn < 0 || (bgp && bn < 4)) return throw_msg(env, "canvasDrawDevice: background must hold 4 bytes or be null");
  int rc = sono_canvas_draw(h->ctx, (const uint8_t*)in, W, B, (int64_t)sy, (int64_t)sh, dy, dh, height, (const uint8_t*)bgp, (uint8_t*)out,
                            SONO_MEM_DEVICE);
  return rc == SONO_OK ? NULL : throw_ctx(env, h, rc);
}

/* ---- overlapped per-file batches: page-locked host buffers + host-async transform ------------- */

static void pinned_finalize(napi_env env, void* data, void* hint) {
  (void)env;
  (void)hint;
  sono_host_free(data);
}

/* hostAlloc(bytes) -> ArrayBuffer over page-locked memory (freed when the buffer is collected) */
static napi_value js_host_alloc(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  uint64_t bytes;
  if (!get_u64(env, argv[0], &bytes) || !bytes) return throw_msg(env, "hostAlloc(bytes)");
  void* p = sono_host_malloc(bytes);
  if (!p) return throw_msg(env, "page-locked allocation failed");
  napi_value ab;
  if (napi_create_external_arraybuffer(env, p, (size_t)bytes, pinned_finalize, NULL, &ab) != napi_ok) {
    sono_host_free(p);
    return throw_msg(env, "external ArrayBuffer failed");
  }
  return ab;
}

/* stftHostAsync(handle, inTypedArray, samples, outUint8Array): enqueues copy-in, transform, copy-out; wait(handle) completes it.
 * Both arrays must stay alive and untouched until then (use hostAlloc memory for the copies to overlap). */
static napi_value js_stft_host_async(napi_env env, napi_callback_info info) {
  ARGS(env, info, 4);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  napi_typedarray_type ti, to;
  void *in, *out;
  uint64_t L;
  int64_t nin = ta_data(env, argv[1], &ti, &in), nout = ta_data(env, argv[3], &to, &out);
  if (nin < 0 || nout < 0 || !get_u64(env, argv[2], &L) || (to != napi_uint8_array && to != napi_uint8_clamped_array))
    return throw_msg(env, "stftHostAsync(handle, input, samples, Uint8Array output)");
  sono_config cfg;
  sono_get_config(h->ctx, &cfg);
  const napi_typedarray_type want = cfg.in_kind == SONO_IN_I16 ? napi_int16_array : napi_float32_array;
  if (ti != want || (uint64_t)nin < L) return throw_msg(env, "stftHostAsync: input type or length does not match the context");
  const uint64_t need = sono_out_bytes(h->ctx, L);
  if ((uint64_t)nout < need) return throw_msg(env, "stftHostAsync: output too small");
  int rc = sono_stft_host_async(h->ctx, in, L, out, (uint64_t)nout);
  if (rc != SONO_OK) return throw_ctx(env, h, rc);
  return make_num(env, (double)need);
}

#define EXPORT(name, fn)                                                     \
  do {                                                                       \
    napi_value f_;                                                           \
    if (napi_create_function(env, name, NAPI_AUTO_LENGTH, fn, NULL, &f_) != napi_ok) return NULL; \
    if (napi_set_named_property(env, exports, name, f_) != napi_ok) return NULL;                  \
  } while (0)

/* ---- feature pass: calls, their shape, the panel's fields, the matcher -------------------- */

#define CALL_NUMS 9
static void call_to_nums(const sono_call* c, double* d) {
  d[0] = (double)c->start_frame; d[1] = (double)c->end_frame; d[2] = (double)c->peak_frame;
  d[3] = c->start_bin; d[4] = c->end_bin; d[5] = c->high_bin; d[6] = c->low_bin; d[7] = c->peak_bin; d[8] = c->peak_idx;
}
static void nums_to_call(const double* d, sono_call* c) {
  c->start_frame = (uint64_t)d[0]; c->end_frame = (uint64_t)d[1]; c->peak_frame = (uint64_t)d[2];
  c->start_bin = (int32_t)d[3]; c->end_bin = (int32_t)d[4]; c->high_bin = (int32_t)d[5]; c->low_bin = (int32_t)d[6];
  c->peak_bin = (int32_t)d[7]; c->peak_idx = (uint32_t)d[8];
}

/* segmentCalls(records: Uint8Array of sono_frame_feature, threshold, minFrames, firstFrame) -> Float64Array, 9 numbers a call:
 * startFrame, endFrame, peakFrame, startBin, endBin, highBin, lowBin, peakBin, peakIdx */
static napi_value js_segment_calls(napi_env env, napi_callback_info info) {
  ARGS(env, info, 4);
  napi_typedarray_type t;
  void* rec;
  int64_t n = ta_data(env, argv[0], &t, &rec);
  uint32_t thr, minf;
  uint64_t first;
  if (n < 0 || (n > 0 && t != napi_uint8_array) || n % 8 || !get_u32(env, argv[1], &thr) || !get_u32(env, argv[2], &minf) ||
      !get_u64(env, argv[3], &first))
    return throw_msg(env, "segmentCalls(records Uint8Array, threshold, minFrames, firstFrame)");
  static sono_frame_feature none;
  const sono_frame_feature* f = n ? (const sono_frame_feature*)rec : &none;
  int64_t found = sono_segment_calls(f, (uint64_t)n / 8, thr, minf, first, NULL, 0);
  if (found < 0) return throw_msg(env, "sono_segment_calls failed");
  sono_call* calls = (sono_call*)malloc(sizeof(sono_call) * (size_t)(found ? found : 1));
  if (!calls) return throw_msg(env, "allocation failed");
  sono_segment_calls(f, (uint64_t)n / 8, thr, minf, first, calls, (uint64_t)found);
  void* out = NULL;
  napi_value result = new_typed(env, napi_float64_array, (size_t)found * CALL_NUMS, 8, &out);
  if (result)
    for (int64_t i = 0; i < found; i++) call_to_nums(&calls[i], (double*)out + i * CALL_NUMS);
  free(calls);
  return result;
}

/* callShape(records, firstFrame, call: Float64Array(9)) -> Float64Array(10): highFrame, lowFrame, kneeFrame, heelFrame,
 * cfStartFrame, cfEndFrame, kneeBin, heelBin, cfStartBin, cfEndBin (frames -1 = none) */
static napi_value js_call_shape(napi_env env, napi_callback_info info) {
  ARGS(env, info, 3);
  napi_typedarray_type t, tc;
  void *rec, *cn;
  int64_t n = ta_data(env, argv[0], &t, &rec), nc = ta_data(env, argv[2], &tc, &cn);
  uint64_t first;
  if (n <= 0 || t != napi_uint8_array || n % 8 || nc != CALL_NUMS || tc != napi_float64_array || !get_u64(env, argv[1], &first))
    return throw_msg(env, "callShape(records Uint8Array, firstFrame, call Float64Array(9))");
  sono_call c;
  nums_to_call((const double*)cn, &c);
  sono_call_shape_t s;
  if (sono_call_shape((const sono_frame_feature*)rec, (uint64_t)n / 8, first, &c, &s) != SONO_OK)
    return throw_msg(env, "call lies outside the feature records");
  void* out = NULL;
  napi_value result = new_typed(env, napi_float64_array, 10, 8, &out);
  if (!result) return NULL;
  double* d = (double*)out;
  d[0] = (double)s.high_frame; d[1] = (double)s.low_frame; d[2] = (double)s.knee_frame; d[3] = (double)s.heel_frame;
  d[4] = (double)s.cf_start_frame; d[5] = (double)s.cf_end_frame; d[6] = s.knee_bin; d[7] = s.heel_bin; d[8] = s.cf_start_bin;
  d[9] = s.cf_end_bin;
  return result;
}

/* autoidDataOfCall(call Float64Array(9), shape Float64Array(10) | null, fftN, hop, sampleRate, timeExpansion, cfCallType)
 * -> Float64Array(14) in autoIdHK's field order, NaN = absent */
static napi_value js_autoid_data_of_call(napi_env env, napi_callback_info info) {
  ARGS(env, info, 7);
  napi_typedarray_type tc, ts;
  void *cn, *sn;
  int64_t nc = ta_data(env, argv[0], &tc, &cn), ns = ta_data(env, argv[1], &ts, &sn);
  uint32_t fft_n, hop, cf;
  double sr, te;
  if (nc != CALL_NUMS || tc != napi_float64_array || ns < 0 || (ns != 0 && (ns != 10 || ts != napi_float64_array)) ||
      !get_u32(env, argv[2], &fft_n) || !get_u32(env, argv[3], &hop) || !get_f64(env, argv[4], &sr) || !get_f64(env, argv[5], &te) ||
      !get_u32(env, argv[6], &cf))
    return throw_msg(env, "autoidDataOfCall(call, shape | null, fftN, hop, sampleRate, timeExpansion, cfCallType)");
  sono_call c;
  nums_to_call((const double*)cn, &c);
  sono_call_shape_t s;
  if (ns) {
    const double* d = (const double*)sn;
    s.high_frame = (int64_t)d[0]; s.low_frame = (int64_t)d[1]; s.knee_frame = (int64_t)d[2]; s.heel_frame = (int64_t)d[3];
    s.cf_start_frame = (int64_t)d[4]; s.cf_end_frame = (int64_t)d[5]; s.knee_bin = (int32_t)d[6]; s.heel_bin = (int32_t)d[7];
    s.cf_start_bin = (int32_t)d[8]; s.cf_end_bin = (int32_t)d[9];
  }
  sono_autoid_data data;
  if (sono_autoid_data_of_call(&c, ns ? &s : NULL, fft_n, hop, sr, te, (int)cf, &data) != SONO_OK)
    return throw_msg(env, "bad call conversion arguments");
  void* out = NULL;
  napi_value result = new_typed(env, napi_float64_array, SONO_AUTOID_FIELDS, 8, &out);
  if (result) memcpy(out, data.field, sizeof data.field);
  return result;
}

/* autoidMatch(rules: Uint8Array of sono_autoid_rule, conds: Uint32Array (4 a test), intervals: Float64Array (2 an interval),
 *             harmonics: Int32Array, fields: Float64Array(14), callType (index, -1 = absent), harmonic (null = absent), nSpecies)
 * -> Uint8Array(nSpecies) */
static napi_value js_autoid_match(napi_env env, napi_callback_info info) {
  ARGS(env, info, 8);
  napi_typedarray_type tr, tc, ti, th, tf;
  void *rules, *conds, *iv, *harm, *fields;
  int64_t nr = ta_data(env, argv[0], &tr, &rules), nc = ta_data(env, argv[1], &tc, &conds), ni = ta_data(env, argv[2], &ti, &iv);
  int64_t nh = ta_data(env, argv[3], &th, &harm), nf = ta_data(env, argv[4], &tf, &fields);
  double ct;
  uint32_t nsp;
  if (nr < 0 || nr % (int64_t)sizeof(sono_autoid_rule) || (nr && tr != napi_uint8_array) || nc < 0 || nc % 4 ||
      (nc && tc != napi_uint32_array) || ni < 0 || ni % 2 || (ni && ti != napi_float64_array) || nh < 0 ||
      (nh && th != napi_int32_array) || nf != SONO_AUTOID_FIELDS || tf != napi_float64_array || !get_f64(env, argv[5], &ct) ||
      !get_u32(env, argv[7], &nsp))
    return throw_msg(env, "autoidMatch(rules, conds, intervals, harmonics, fields, callType, harmonic, nSpecies)");
  /* every index in the records must stay inside the arrays passed */
  const sono_autoid_rule* r = (const sono_autoid_rule*)rules;
  const uint32_t n_rules = (uint32_t)(nr / (int64_t)sizeof(sono_autoid_rule));
  const sono_autoid_cond* cd = (const sono_autoid_cond*)conds;
  for (uint32_t i = 0; i < n_rules; i++)
    if ((uint64_t)r[i].cond_first + r[i].cond_count > (uint64_t)nc / 4 || (uint64_t)r[i].harmonic_first + r[i].harmonic_count > (uint64_t)nh)
      return throw_msg(env, "rule record points outside its arrays");
  for (int64_t i = 0; i < nc / 4; i++)
    if (cd[i].kind == SONO_COND_RANGES && (uint64_t)cd[i].first + cd[i].count > (uint64_t)ni / 2)
      return throw_msg(env, "field test points outside the intervals");
  sono_autoid_data data;
  memcpy(data.field, fields, sizeof data.field);
  data.call_type = (int32_t)ct;
  data.harmonic = SONO_AUTOID_ABSENT;
  napi_valuetype vt;
  double hv;
  if (napi_typeof(env, argv[6], &vt) == napi_ok && vt == napi_number && get_f64(env, argv[6], &hv) && hv == (double)(int32_t)hv)
    data.harmonic = (int32_t)hv;
  void* out = NULL;
  napi_value result = new_typed(env, napi_uint8_array, nsp, 1, &out);
  if (!result) return NULL;
  static uint8_t none_out[1];
  if (sono_autoid_match(r, n_rules, cd, (const double*)iv, (const int32_t*)harm, &data, nsp ? (uint8_t*)out : none_out, nsp) != SONO_OK)
    return throw_msg(env, "bad rule table");
  return result;
}

static napi_value js_measure_valu_peak(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  return make_num(env, sono_measure_valu_peak(h->ctx));
}

static napi_value js_kernel_route(napi_env env, napi_callback_info info) {
  ARGS(env, info, 1);
  handle_t* h = get_handle(env, argv[0]);
  if (!h) return NULL;
  napi_value s;
  NAPI_OK(env, napi_create_string_utf8(env, sono_kernel_route(h->ctx), NAPI_AUTO_LENGTH, &s));
  return s;
}

static napi_value init(napi_env env, napi_value exports) {
  EXPORT("deviceCount", js_device_count);
  EXPORT("deviceNumaNode", js_device_numa_node);
  EXPORT("create", js_create);
  EXPORT("destroy", js_destroy);
  EXPORT("setColormap", js_set_colormap);
  EXPORT("setTables", js_set_tables);
  EXPORT("geometry", js_geometry);
  EXPORT("outBytes", js_out_bytes);
  EXPORT("stft", js_stft);
  EXPORT("resample", js_resample);
  EXPORT("colorize", js_colorize);
  EXPORT("framePeaks", js_frame_peaks);
  EXPORT("deviceMalloc", js_device_malloc);
  EXPORT("deviceFree", js_device_free);
  EXPORT("copyToDevice", js_copy_to_device);
  EXPORT("copyToHost", js_copy_to_host);
  EXPORT("stftDevice", js_stft_device);
  EXPORT("wait", js_wait);
  EXPORT("timerStart", js_timer_start);
  EXPORT("timerStopMs", js_timer_stop);
  EXPORT("workerHop", js_worker_hop);
  EXPORT("pluginNoverlap", js_plugin_noverlap);
  EXPORT("overlapFromPercent", js_overlap_from_percent);
  EXPORT("pluginCrop", js_plugin_crop);
  EXPORT("pluginLut", js_plugin_lut);
  EXPORT("workerLut", js_worker_lut);
  EXPORT("windowTable", js_window_table);
  EXPORT("geometryFor", js_geometry_for);
  EXPORT("hostAlloc", js_host_alloc);
  EXPORT("stftHostAsync", js_stft_host_async);
  EXPORT("setFilterbank", js_set_filterbank);
  EXPORT("pluginCropScaled", js_plugin_crop_scaled);
  EXPORT("pluginDrawRect", js_plugin_draw_rect);
  EXPORT("pluginBackground", js_plugin_background);
  EXPORT("canvasDraw", js_canvas_draw);
  EXPORT("resampleColsDevice", js_resample_cols_device);
  EXPORT("canvasDrawDevice", js_canvas_draw_device);
  EXPORT("wavParse", js_wav_parse);
  EXPORT("wavGuano", js_wav_guano);
  EXPORT("guanoFields", js_guano_fields);
  EXPORT("wavCrop", js_wav_crop);
  EXPORT("wavAccept", js_wav_accept);
  EXPORT("pcmToDevice", js_pcm_to_device);
  EXPORT("segmentCalls", js_segment_calls);
  EXPORT("callShape", js_call_shape);
  EXPORT("autoidDataOfCall", js_autoid_data_of_call);
  EXPORT("autoidMatch", js_autoid_match);
  EXPORT("kernelRoute", js_kernel_route);
  EXPORT("measureValuPeak", js_measure_valu_peak);
  napi_set_named_property(env, exports, "abiVersion", make_num(env, sono_abi_version()));
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
