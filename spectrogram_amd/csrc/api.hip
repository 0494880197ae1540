// C ABI of libsonospec.so: context management, host staging and kernel dispatch.
#include <hip/hip_runtime.h>

#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "sono_device.h"
#include "sono_internal.h"
#include "sonospec.h"

using namespace sono;

struct sono_ctx {
  int device = 0;
  sono_config cfg{};
  uint32_t bin_lo = 0, bin_hi = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  hipEvent_t ev_switch = nullptr;  // orders a newly set stream behind the work already enqueued on the previous one
  // device tables
  float* d_window = nullptr;
  void* d_tw = nullptr;       // float2[n-1] (fast) or double2[n-1] (exact)
  void* d_tw_exact = nullptr; // double2[n-1] beside a float2 d_tw: the feature pass recomputes doubtful frames exactly
  void* d_claim = nullptr;    // the queue of that recomputation (features_fixup_queue_bytes(kFixupCapacity))
  uint32_t* d_lut = nullptr;  // 256 packed pixels
  void* d_fast = nullptr;     // twiddle matrix of the fast kernel (accurate twiddles only)
  bool have_lut = false;
  int32_t* d_fb_bin = nullptr;  // filter bank of a non-linear frequency scale (fft_n/2 entries each), or nullptr
  double* d_fb_w = nullptr;     // w0 then w1
  void* d_fb_packed = nullptr;  // the same bank for the 1024-point float32 kernel: int32 tap[n/2], float w0[n/2], float w1[n/2],
                                // taps outside the spectrum folded away (pack_filterbank)
  bool host_twiddles = false;  // sono_set_tables replaced the twiddle table
  // host copies (sono_set_tables may replace part of them)
  std::vector<float> h_window;
  std::vector<double> h_cos, h_sin;
  // staging for host-pointer calls
  void* d_in = nullptr;
  uint64_t d_in_bytes = 0;
  void* d_out = nullptr;
  uint64_t d_out_bytes = 0;
  void* d_aux = nullptr;
  uint64_t d_aux_bytes = 0;
  // facts about the tables that choose the kernel; recomputed whenever a table changes (refresh_routes)
  uint32_t fast_mode = 0;   // StftParams::fast_mode
  bool mw_ok = false;       // StftParams::mw_ok
  uint32_t fb_fast_mode = 0;  // with a filter bank installed: the same two for the magnitude pass in front of it
  bool fb_mw_ok = false;
  bool pair_exact = false;  // StftParams::pair_exact
  double pair_asym = 0;     // StftParams::pair_asym
  bool pair_idx_ok = true;  // StftParams::pair_idx_ok
  // Pairs of frames (512 / 1024 / 2048 points) or a frame's even / odd samples (4096 points) under an asymmetric table
  // are safe only below a sample level: the kernels transform their loud units once more, one frame per transform
  // (StftParams::loud_thr_raw)
  bool loud_redo = false;
  bool split_ok = true;     // StftParams::split_ok
  float loud_thr_raw = 0.0f;
  void* d_redo = nullptr;   // StftParams::redo_scratch (4096 points)
  std::string route;        // kernels of the most recent transform call (sono_kernel_route)
  std::string err;
};

// The index plane of sono_stft_static: one array per GPU, part of the code object, so that the entry point
// needs no allocation.  Host code never uses the array's name as an address; the per-device address comes from
// hipGetSymbolAddress (static_scratch_of).
static __device__ __align__(256) unsigned char scratch[SONO_STATIC_SCRATCH_BYTES];
// Per device: the array's address there (resolved once, with that device current, by whichever thread creates the
// first context on it) and who used the array last -- an event recorded behind that call's kernels and the stream they
// went to -- so that a call arriving on another stream while they still run is refused instead of corrupting them.
struct StaticScratch {
  std::once_flag once;
  void* ptr = nullptr;
  hipEvent_t done = nullptr;
  hipStream_t last = nullptr;
  bool used = false;
};
static StaticScratch g_static[64];
static std::mutex g_static_mu;  // guards `last` / `used` and the record / query pair of sono_stft_static

// Device address of `scratch` on the current device `dev`, or nullptr.  Called from sono_create.
static void* static_scratch_of(int dev) {
  if (dev < 0 || dev >= 64) return nullptr;
  StaticScratch& st = g_static[dev];
  std::call_once(st.once, [&st] {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess) {
      (void)hipGetLastError();
      ptr = nullptr;
    }
    if (ptr && hipEventCreateWithFlags(&st.done, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      st.done = nullptr;
    }
    st.ptr = ptr;
  });
  return st.ptr;
}

static thread_local const char* t_kernel = "";
namespace sono {
void note_kernel(const char* name) { t_kernel = name; }
}

// Frames that the feature fix-up pass can list per launch (4 MiB; beyond it the pass scans for the rest).
constexpr uint32_t kFixupCapacity = 1u << 20;

static thread_local std::string g_create_error;

static int fail(sono_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}

#define HIP_TRY(c, expr)                                                                            \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return fail((c), SONO_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

static int ensure(sono_ctx* c, void** p, uint64_t* have, uint64_t need) {
  if (*have >= need && *p) return SONO_OK;
  if (*p) HIP_TRY(c, hipFree(*p));
  *p = nullptr;
  *have = 0;
  const uint64_t want = need < 4096 ? 4096 : need;
  HIP_TRY(c, hipMalloc(p, want));
  *have = want;
  return SONO_OK;
}

static int upload_tables(sono_ctx* c) {
  const uint32_t n = c->cfg.fft_n;
  HIP_TRY(c, hipMemcpyAsync(c->d_window, c->h_window.data(), sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
  if (c->cfg.arith == SONO_ARITH_EXACT_F64) {
    std::vector<double> t(2 * (size_t)(n - 1));
    for (uint32_t i = 0; i + 1 < n; i++) {
      t[2 * i] = c->h_cos[i];
      t[2 * i + 1] = c->h_sin[i];
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_tw, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    std::vector<float> t(2 * (size_t)(n - 1));
    for (uint32_t i = 0; i + 1 < n; i++) {
      t[2 * i] = (float)c->h_cos[i];
      t[2 * i + 1] = (float)c->h_sin[i];
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_tw, t.data(), sizeof(float) * t.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_tw_exact) {
      std::vector<double> d(2 * (size_t)(n - 1));
      for (uint32_t i = 0; i + 1 < n; i++) {
        d[2 * i] = c->h_cos[i];
        d[2 * i + 1] = c->h_sin[i];
      }
      HIP_TRY(c, hipMemcpy(c->d_tw_exact, d.data(), sizeof(double) * d.size(), hipMemcpyHostToDevice));
    }
  }
  return SONO_OK;
}

static int refresh_routes(sono_ctx* c);

extern "C" {

int sono_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int sono_device_numa_node(int device) {
  char bdf[32] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  for (char* q = bdf; *q; ++q) *q = (char)tolower((unsigned char)*q);  // sysfs spells the address in lower case
  char path[96];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node < 0 ? -1 : node;
}

const char* sono_create_error(void) { return g_create_error.c_str(); }

sono_ctx* sono_create(int device, const sono_config* cfg) {
  g_create_error.clear();
  if (!cfg) {
    fail(nullptr, SONO_ERR_ARG, "config is NULL");
    return nullptr;
  }
  for (uint32_t r : {cfg->reserved0, cfg->reserved[0], cfg->reserved[1], cfg->reserved[2], cfg->reserved[3]})
    if (r != 0) {
      fail(nullptr, SONO_ERR_ARG, "reserved config fields must be zero");
      return nullptr;
    }
  const uint32_t n = cfg->fft_n;
  if (!is_pow2(n)) {
    fail(nullptr, SONO_ERR_FFT_SIZE, "Invalid buffer size, must be a power of 2.");
    return nullptr;
  }
  if (n < 64 || n > 8192) {
    fail(nullptr, SONO_ERR_FFT_SIZE, "fft_n must be between 64 and 8192 (got %u)", n);
    return nullptr;
  }
  if (cfg->hop == 0) {
    fail(nullptr, SONO_ERR_ARG, "hop must be at least 1 (noverlap must be smaller than fftSamples)");
    return nullptr;
  }
  if (cfg->scale_mode > SONO_SCALE_PLUGIN_DB || cfg->frame_rule > SONO_FRAMES_WORKER_COMPAT ||
      cfg->twiddle_mode > SONO_TW_HOST_TABLE || cfg->arith > SONO_ARITH_EXACT_F64 || cfg->out_kind > SONO_OUT_FEATURES ||
      cfg->in_kind > SONO_IN_I16) {
    fail(nullptr, SONO_ERR_ARG, "config enum out of range");
    return nullptr;
  }
  if (cfg->scale_mode == SONO_SCALE_PLUGIN_DB && !(cfg->range_db > 0)) {
    fail(nullptr, SONO_ERR_ARG, "range_db must be positive");
    return nullptr;
  }
  sono_geometry g;
  if (compute_geometry(n, cfg->hop, cfg->frame_rule, cfg->bin_lo, cfg->bin_hi, 0, &g) != SONO_OK) {
    fail(nullptr, SONO_ERR_ARG, "bin window [%u, %u) is not inside [0, %u)", cfg->bin_lo, cfg->bin_hi, n / 2);
    return nullptr;
  }
  std::vector<float> win(n);
  if (fill_window(cfg->window_id, n, cfg->alpha, win.data()) != SONO_OK) {
    fail(nullptr, SONO_ERR_WINDOW, "No such window function '%u'", cfg->window_id);
    return nullptr;
  }
  int count = sono_device_count();
  if (count <= 0) {
    fail(nullptr, SONO_ERR_NO_DEVICE, "no HIP device available");
    return nullptr;
  }
  if (device < 0 || device >= count) {
    fail(nullptr, SONO_ERR_ARG, "device %d out of range (have %d)", device, count);
    return nullptr;
  }
  sono_ctx* c = new (std::nothrow) sono_ctx();
  if (!c) return nullptr;
  c->device = device;
  c->cfg = *cfg;
  // The float32 arithmetic resolves about 100 dB below the loudest value of a transform (its own rounding, and under
  // the plugin's recurrence factors the cross-talk between the two frames transformed together): a full-scale burst
  // beside silence stays within 1 LSB of the reference down to a display floor of -100 dB and not below (measured at
  // 512 .. 4096 points).  A quantiser whose floor -gain-range lies lower would show that noise where the reference
  // shows its own, so such a context computes colour indices and images in the exact arithmetic.
  if (c->cfg.arith == SONO_ARITH_FAST_F32 && c->cfg.scale_mode == SONO_SCALE_PLUGIN_DB &&
      (c->cfg.out_kind == SONO_OUT_U8_INDEX || c->cfg.out_kind == SONO_OUT_RGBA8) &&
      (double)c->cfg.gain_db + (double)c->cfg.range_db > 100.0)
    c->cfg.arith = SONO_ARITH_EXACT_F64;
  cfg = &c->cfg;
  c->bin_lo = cfg->bin_lo;
  c->bin_hi = cfg->bin_hi ? cfg->bin_hi : n / 2;
  c->h_window = std::move(win);
  c->h_cos.assign(n, 0.0);
  c->h_sin.assign(n, 0.0);
  if (cfg->twiddle_mode == SONO_TW_HOST_TABLE)
    fill_twiddles(SONO_TW_ACCURATE, n, c->h_cos.data(), c->h_sin.data());  // until sono_set_tables replaces them
  else
    fill_twiddles(cfg->twiddle_mode, n, c->h_cos.data(), c->h_sin.data());

  auto bail = [&](const char* what, hipError_t e) -> sono_ctx* {
    fail(nullptr, SONO_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
    sono_destroy(c);
    return nullptr;
  };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  c->stream = c->own_stream;
  if ((e = hipEventCreate(&c->ev_start)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&c->ev_stop)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreateWithFlags(&c->ev_switch, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipMalloc((void**)&c->d_window, sizeof(float) * n)) != hipSuccess) return bail("hipMalloc(window)", e);
  if ((e = hipMalloc(&c->d_tw, sizeof(double) * 2 * n)) != hipSuccess) return bail("hipMalloc(twiddles)", e);
  // entries [0, 256): the caller's colour map; [256, 512): the grey ramp used when none is set
  if ((e = hipMalloc((void**)&c->d_lut, 2048)) != hipSuccess) return bail("hipMalloc(lut)", e);
  {
    uint32_t grey[256];
    for (uint32_t i = 0; i < 256; ++i) grey[i] = i * 0x010101u | 0xFF000000u;
    if ((e = hipMemcpy(c->d_lut + 256, grey, 1024, hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy(lut)", e);
  }
  if (cfg->arith == SONO_ARITH_FAST_F32 && cfg->out_kind == SONO_OUT_FEATURES &&
      (e = hipMalloc(&c->d_tw_exact, sizeof(double) * 2 * n)) != hipSuccess)
    return bail("hipMalloc(exact twiddles)", e);
  if (c->d_tw_exact && (e = hipMalloc(&c->d_claim, features_fixup_queue_bytes(kFixupCapacity))) != hipSuccess)
    return bail("hipMalloc(fix-up queue)", e);
  if (upload_tables(c) != SONO_OK) {
    g_create_error = c->err;
    sono_destroy(c);
    return nullptr;
  }
  if (cfg->arith == SONO_ARITH_FAST_F32 && cfg->twiddle_mode == SONO_TW_ACCURATE && (n == 1024 || n == 512 || n == 2048)) {
    std::vector<float> t;
    if (n == 1024) build_fast_table_1024(t); else if (n == 512) build_fast_table_512(t); else build_fast_table_2048(t);
    if ((e = hipMalloc(&c->d_fast, sizeof(float) * t.size())) != hipSuccess) return bail("hipMalloc(fast table)", e);
    if ((e = hipMemcpy(c->d_fast, t.data(), sizeof(float) * t.size(), hipMemcpyHostToDevice)) != hipSuccess)
      return bail("hipMemcpy(fast table)", e);
  }
  (void)static_scratch_of(device);  // the device is current: resolve this GPU's copy of the static index plane
  if (refresh_routes(c) != SONO_OK) {
    g_create_error = c->err;
    sono_destroy(c);
    return nullptr;
  }
  return c;
}

void sono_destroy(sono_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  void* bufs[] = {c->d_window, c->d_tw, c->d_tw_exact, c->d_claim, c->d_lut, c->d_fast, c->d_in, c->d_out, c->d_aux, c->d_fb_bin, c->d_fb_w, c->d_fb_packed, c->d_redo};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
  if (c->ev_switch) (void)hipEventDestroy(c->ev_switch);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* sono_last_error(const sono_ctx* c) { return c ? c->err.c_str() : "context is NULL"; }

int sono_get_config(const sono_ctx* c, sono_config* out) {
  if (!c || !out) return SONO_ERR_ARG;
  *out = c->cfg;
  return SONO_OK;
}

int sono_set_stream(sono_ctx* c, void* s) {
  if (!c) return SONO_ERR_ARG;
  hipStream_t next = s ? reinterpret_cast<hipStream_t>(s) : c->own_stream;
  if (next != c->stream) {
    // the context's scratch (image slots, index chunk, parked half transforms) is shared by all its launches: what
    // the new stream runs must not overtake what the previous one still has in flight
    HIP_TRY(c, hipSetDevice(c->device));
    if (hipEventRecord(c->ev_switch, c->stream) == hipSuccess) {
      HIP_TRY(c, hipStreamWaitEvent(next, c->ev_switch, 0));
    } else {
      (void)hipGetLastError();  // the previous stream is gone (its owner destroyed it): nothing of it is in flight
    }
    c->stream = next;
  }
  return SONO_OK;
}

int sono_set_colormap(sono_ctx* c, const uint8_t* lut) {
  if (!c) return SONO_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!lut) {
    c->have_lut = false;
    return SONO_OK;
  }
  HIP_TRY(c, hipMemcpyAsync(c->d_lut, lut, 1024, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_lut = true;
  return SONO_OK;
}

int sono_set_tables(sono_ctx* c, const float* window, const double* tw_cos, const double* tw_sin) {
  if (!c) return SONO_ERR_ARG;
  if ((tw_cos == nullptr) != (tw_sin == nullptr)) return fail(c, SONO_ERR_ARG, "tw_cos and tw_sin must be given together");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t n = c->cfg.fft_n;
  if (window) std::memcpy(c->h_window.data(), window, sizeof(float) * n);
  if (tw_cos) {
    std::memcpy(c->h_cos.data(), tw_cos, sizeof(double) * (n - 1));
    std::memcpy(c->h_sin.data(), tw_sin, sizeof(double) * (n - 1));
    c->host_twiddles = true;
  }
  int rc = upload_tables(c);
  return rc != SONO_OK ? rc : refresh_routes(c);
}

// out[y] = m[o] * w0[y] + m[o + 1] * w1[y], a tap outside [0, half) contributing nothing, as two taps that are always
// readable when m[half] is a zero: {tap, w0, w1} with o = -1 turned into {0, w1, 0}, o = half - 1 into {o, w0, 0} (its
// second tap reads the zero), anything further out into {0, 0, 0}.
static void pack_filterbank(const int32_t* bin, const double* w0, const double* w1, uint32_t half, std::vector<uint32_t>& packed) {
  packed.assign((size_t)3 * half, 0u);
  for (uint32_t y = 0; y < half; y++) {
    const int64_t o = bin[y];
    const bool v0 = o >= 0 && o < (int64_t)half, v1 = o + 1 >= 0 && o + 1 < (int64_t)half;
    int32_t t = 0;
    float a = 0.0f, b = 0.0f;
    if (v0) {
      t = (int32_t)o;
      a = (float)w0[y];
      if (v1) b = (float)w1[y];
    } else if (v1) {
      t = (int32_t)(o + 1);
      a = (float)w1[y];
    }
    std::memcpy(&packed[y], &t, 4);
    std::memcpy(&packed[half + y], &a, 4);
    std::memcpy(&packed[2 * half + y], &b, 4);
  }
}

int sono_set_filterbank(sono_ctx* c, const int32_t* bin, const double* w0, const double* w1) {
  if (!c) return SONO_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!bin) {
    if (c->d_fb_bin) HIP_TRY(c, hipFree(c->d_fb_bin));
    if (c->d_fb_w) HIP_TRY(c, hipFree(c->d_fb_w));
    if (c->d_fb_packed) HIP_TRY(c, hipFree(c->d_fb_packed));
    c->d_fb_bin = nullptr;
    c->d_fb_w = nullptr;
    c->d_fb_packed = nullptr;
    return refresh_routes(c);
  }
  if (!w0 || !w1) return fail(c, SONO_ERR_ARG, "filter bank weights missing");
  if (c->cfg.scale_mode != SONO_SCALE_PLUGIN_DB) return fail(c, SONO_ERR_ARG, "a filter bank needs the plugin scale mode");
  const uint32_t half = c->cfg.fft_n / 2;
  if (!c->d_fb_bin) HIP_TRY(c, hipMalloc((void**)&c->d_fb_bin, sizeof(int32_t) * half));
  if (!c->d_fb_w) HIP_TRY(c, hipMalloc((void**)&c->d_fb_w, sizeof(double) * 2 * half));
  HIP_TRY(c, hipMemcpy(c->d_fb_bin, bin, sizeof(int32_t) * half, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_fb_w, w0, sizeof(double) * half, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_fb_w + half, w1, sizeof(double) * half, hipMemcpyHostToDevice));
  {
    std::vector<uint32_t> packed;
    pack_filterbank(bin, w0, w1, half, packed);
    if (!c->d_fb_packed) HIP_TRY(c, hipMalloc(&c->d_fb_packed, packed.size() * 4));
    HIP_TRY(c, hipMemcpy(c->d_fb_packed, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  }
  return refresh_routes(c);
}

int sono_geometry_of(const sono_ctx* c, uint64_t L, sono_geometry* out) {
  if (!c || !out) return SONO_ERR_ARG;
  return compute_geometry(c->cfg.fft_n, c->cfg.hop, c->cfg.frame_rule, c->bin_lo, c->bin_hi, L, out);
}

uint64_t sono_out_bytes(const sono_ctx* c, uint64_t L) {
  sono_geometry g;
  if (sono_geometry_of(c, L, &g) != SONO_OK) return 0;
  switch (c->cfg.out_kind) {
    case SONO_OUT_U8_INDEX: return g.frames * g.bins;
    case SONO_OUT_RGBA8: return g.width * g.bins * 4;
    case SONO_OUT_FEATURES: return g.frames * sizeof(sono_frame_feature);
    default: return g.frames * g.bins * 4;
  }
}

static uint64_t in_bytes(const sono_ctx* c, uint64_t L) { return L * (c->cfg.in_kind == SONO_IN_I16 ? 2 : 4); }

// Which flavour of the fast kernels serves this context (StftParams::fast_mode).
static uint32_t compute_fast_mode(const sono_ctx* c, bool with_filterbank = false) {
  const uint32_t fn = c->cfg.fft_n;
  if (c->cfg.arith != SONO_ARITH_FAST_F32 || (fn != 1024 && fn != 512 && fn != 2048 && fn != 4096) || (c->d_fb_bin && !with_filterbank))
    return 0;
  // (the 4096-point kernel has the table flavour only: accurate twiddles reach it as a table too)
  if (fn != 4096 && c->cfg.twiddle_mode == SONO_TW_ACCURATE && !c->host_twiddles) return c->d_fast ? 1 : 0;
  // table twiddles: the kernel folds the first butterfly of every stage, so that factor must be 1
  for (uint32_t T = 1; T < c->cfg.fft_n; T <<= 1)
    if (c->h_cos[T - 1] != 1.0 || c->h_sin[T - 1] != 0.0) return 0;
  return 2;
}
// Largest deviation of the twiddle table from conjugate symmetry, |w[T-j] + conj(w[j])| over every stage of half
// size T (StftParams::pair_exact).
static double table_asymmetry(const sono_ctx* c) {
  double worst = 0;
  for (uint32_t T = 2; T < c->cfg.fft_n; T <<= 1)
    for (uint32_t j = 1; j < T; j++) {
      const double dr = c->h_cos[T - 1 + (T - j)] + c->h_cos[T - 1 + j];
      const double di = c->h_sin[T - 1 + (T - j)] - c->h_sin[T - 1 + j];
      const double d = std::sqrt(dr * dr + di * di);
      if (d > worst) worst = d;
    }
  return worst;
}
// The 512/2048/4096-point three-pass kernel takes any twiddle table whose stages start with the factor (1, 0).
static bool compute_mw_ok(const sono_ctx* c, bool with_filterbank = false) {
  if (c->cfg.arith != SONO_ARITH_FAST_F32 || (c->cfg.fft_n != 512 && c->cfg.fft_n != 2048 && c->cfg.fft_n != 4096) ||
      (c->d_fb_bin && !with_filterbank))
    return false;
  for (uint32_t T = 1; T < c->cfg.fft_n; T <<= 1)
    if (c->h_cos[T - 1] != 1.0 || c->h_sin[T - 1] != 0.0) return false;
  return true;
}

// Default size of the RGBA index-chunk buffer when the context cannot write the image from the transform kernel.
constexpr uint64_t kDefaultChunkBytes = (uint64_t)32 << 20;
constexpr uint64_t kMaxChunkBytes = (uint64_t)96 << 20;

// Recomputes what the kernel choice depends on and sizes the context's RGBA scratch.  Runs in the calls that may
// allocate and wait (sono_create, sono_set_tables, sono_set_filterbank), never in a transform call.
static int refresh_routes(sono_ctx* c) {
  c->fast_mode = compute_fast_mode(c);
  c->mw_ok = compute_mw_ok(c);
  c->fb_fast_mode = c->d_fb_bin ? compute_fast_mode(c, true) : 0;
  c->fb_mw_ok = c->d_fb_bin ? compute_mw_ok(c, true) : false;
  c->pair_asym = table_asymmetry(c);
  c->pair_exact = c->pair_asym <= 5e-7;
  c->loud_redo = false;
  c->split_ok = true;
  if (c->pair_exact) {
    c->pair_idx_ok = true;
  } else {
    // An asymmetric table (the plugin's recurrence factors, or a host table).  Two real frames transformed as one complex
    // frame separate only up to the table's Hermitian defect: a bin is off by at most g times the largest sample of
    // the pair (pair_defect_gains: the exact operator norm for this table, as the kernels hold it in float32, and this
    // window).  A colour index moves by less than one step while |X| is off by less than eps = 10^(step in dB / 20) - 1
    // of its value, and the smallest value that shows is the quantiser's floor; float32 rounding gets its share of that
    // step first (below).  So pairs are safe outright if full-scale samples cannot reach the bound; safe with the loud units
    // transformed once more, one frame per transform, if a plausible level can (the register kernels at 512 / 1024 /
    // 2048 points do that themselves, and so does the 4096-point kernel, which splits a frame into even and odd
    // samples and carries the defect of the 2048-point transform inside it twice); otherwise such a context transforms
    // one frame at a time from the start.
    const uint32_t n = c->cfg.fft_n;
    const bool plugin = c->cfg.scale_mode == SONO_SCALE_PLUGIN_DB;
    const double eps = plugin ? std::pow(10.0, (double)c->cfg.range_db / (255.0 * 20.0)) - 1.0 : std::pow(10.0, 5.0 / 255.0) - 1.0;
    const double floor_mag = plugin ? 0.5 * (double)n * std::pow(10.0, -((double)c->cfg.gain_db + (double)c->cfg.range_db) / 20.0) : 1.0;
    // ... less what float32 rounding may take of the step: the float32 kernels are within 0.04 dB of the reference at a
    // bin 95 dB below its frame's largest (the asserted bound; measured 0.026), i.e. within 8.2e-8 of that largest value,
    // which for samples up to full scale is at most the window's sum; and 5 % of the step in reserve.
    double wsum = 0.0;
    for (uint32_t i = 0; i < n; i++) wsum += std::fabs((double)c->h_window[i]);
    const double crit = 0.95 * eps * floor_mag - 8.2e-8 * wsum;  // (<= 0: no level is safe, one frame per transform throughout)
    std::vector<double> tc(n), ts(n);
    for (uint32_t i = 0; i + 1 < n; i++) {
      tc[i] = (double)(float)c->h_cos[i];
      ts[i] = (double)(float)c->h_sin[i];
    }
    double g_raw = 0, g_win = 0;
    pair_defect_gains(n, tc.data(), ts.data(), c->h_window.data(), &g_raw, &g_win);
    double thr_raw = crit <= 0 ? 0.0 : g_raw > 0 ? crit / g_raw : 1e30;
    c->pair_idx_ok = thr_raw >= 1.0;
    bool redo = !c->pair_idx_ok && (n == 512 || n == 1024 || n == 2048);
    if (n == 4096) {  // the kernel that splits a frame: the inner transform's table is the first eleven stages of this one
      std::vector<float> we(n / 2), wo(n / 2);
      for (uint32_t i = 0; i < n / 2; i++) {
        we[i] = c->h_window[2 * i];
        wo[i] = c->h_window[2 * i + 1];
      }
      double ge = 0, go = 0, gw = 0;
      pair_defect_gains(n / 2, tc.data(), ts.data(), we.data(), &ge, &gw);
      pair_defect_gains(n / 2, tc.data(), ts.data(), wo.data(), &go, &gw);
      const double g2 = 2.0 * (ge > go ? ge : go);  // X[k] = E[k] + w O[k]: both halves' errors add
      thr_raw = crit <= 0 ? 0.0 : g2 > 0 ? crit / g2 : 1e30;
      redo = thr_raw < 1.0;
      c->split_ok = !redo;
    }
    const bool index_out = c->cfg.out_kind == SONO_OUT_U8_INDEX || c->cfg.out_kind == SONO_OUT_RGBA8;
    // (a filter bank: the 1024-point kernel alone applies it to pairs; a threshold below a fiftieth of full scale: nearly
    // every unit would be loud, one frame per transform from the start is cheaper)
    if (redo && index_out && thr_raw >= 0.02 && (!c->d_fb_bin || n == 1024) && c->cfg.arith == SONO_ARITH_FAST_F32) {
      c->loud_redo = true;
      c->loud_thr_raw = (float)thr_raw;
      if (n == 4096 && !c->d_redo) HIP_TRY(c, hipMalloc(&c->d_redo, redo_scratch_bytes()));
    }
    // (the 4096-point kernel without that would be unsafe: see fast4096_available)
  }
  const bool ragged_u8 = c->cfg.out_kind == SONO_OUT_U8_INDEX && ((c->bin_lo | c->bin_hi) & 3u) && (c->fast_mode || c->mw_ok);
  const bool fb_route = (c->fb_fast_mode || c->fb_mw_ok) && (c->cfg.out_kind == SONO_OUT_RGBA8 || c->cfg.out_kind == SONO_OUT_U8_INDEX);
  if (c->cfg.out_kind == SONO_OUT_RGBA8 || ragged_u8 || fb_route) {
    // (the magnitude chunks of the filter-bank route are 4 bytes per bin: the largest chunk that still stays in the
    // Infinity Cache keeps the transform kernel's launches long enough to fill the chip)
    const uint64_t want = fb_route ? kMaxChunkBytes : c->fast_mode && !ragged_u8 ? fast_rgba_scratch_bytes() : kDefaultChunkBytes;
    if (c->d_aux_bytes < want) {
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      int rc = ensure(c, &c->d_aux, &c->d_aux_bytes, want);
      if (rc != SONO_OK) return rc;
    }
  }
  return SONO_OK;
}

// Appends the kernel that the last launch_* call reported to the context's route string.
static void note_route(sono_ctx* c, const char* suffix = nullptr) {
  if (c->route.size() > 200) return;  // (a chunked call repeats itself)
  std::string item = t_kernel;
  if (suffix) item += suffix;
  if (c->route.find(item) != std::string::npos) return;
  if (!c->route.empty()) c->route += " + ";
  c->route += item;
}

// Where the colour indices of the two-kernel RGBA route are parked.
struct IndexScratch {
  void* ptr;
  uint64_t bytes;
  bool single_chunk;  // sono_stft_static: the whole call must fit
};

// One frame range through the kernels.  Allocates nothing and does not wait.
static int launch_range(sono_ctx* c, const StftParams& q_in, const IndexScratch& aux, bool fused_ok) {
  if ((q_in.frame0 + q_in.nframes - 1) * q_in.hop + q_in.n > q_in.n_samples) return fail(c, SONO_ERR_ARG, "frame range exceeds the signal");
  int r = 1;
  const bool exact = c->cfg.arith == SONO_ARITH_EXACT_F64;
  const bool rgba = q_in.out_kind == SONO_OUT_RGBA8;
  // A non-linear frequency scale at 1024 points: the float32 index kernel applies the two taps per output bin itself
  // (whole groups of four bins, as for the linear scale; the image goes through a chunk of indices, whose band is
  // rounded outwards); at 512 / 2048 / 4096 points the three-pass kernel does.  Ragged index bands, and 1024-point
  // tables whose pairs would leak, take the route through a plane of magnitudes further down.
  const bool fb_1024 = q_in.n == 1024 && c->fb_fast_mode;  // (pairs that would leak: its one-frame-per-transform flavour)
  const bool fb_mw = q_in.n != 1024 && c->fb_mw_ok;  // (the three-pass kernel: 512 / 2048 / 4096 points, pairs or single frames)
  const bool fb_fused = q_in.fb_bin && !exact && c->d_fb_packed && (fb_1024 || fb_mw) &&
                        (rgba || (q_in.out_kind == SONO_OUT_U8_INDEX && !((q_in.bin_lo | q_in.bin_hi | q_in.out_bins | q_in.out_bin0) & 3u)));
  StftParams q = q_in;
  // An asymmetric table whose pairs (512 / 1024 / 2048 points) or split frames (4096 points) are safe only below a sample
  // level: the kernels transform their loud units once more, one frame per transform (refresh_routes).
  const bool loud = c->loud_redo && (q_in.n != 4096 || c->d_redo != nullptr) && !exact && q_in.nframes >= 16 &&
                    (q_in.out_kind == SONO_OUT_U8_INDEX || rgba) &&
                    (!q_in.fb_bin || (q_in.n == 1024 && fb_fused && fb_1024));  // (a filter bank: the 1024-point kernel alone)
  if (loud) {
    if (q.n == 4096) q.redo_scratch = c->d_redo;
    q.loud_thr_raw = c->loud_thr_raw;
    if (q.n != 4096) q.pair_idx_ok = 1;  // (4096 points: the pairing three-pass kernel stays off, the splitting kernel serves)
  }
  if (fb_fused) {
    q.fast_mode = fb_1024 ? c->fb_fast_mode : 0;
    q.mw_ok = fb_mw ? 1u : 0u;
    q.fb_packed = c->d_fb_packed;
  }
  StftParams qq = q;
  if (!qq.lut) qq.lut = c->d_lut + 256;
  const bool fast = c->fast_mode != 0 || fb_fused;
  if (fused_ok && fast && rgba && (fast_rgba_available(qq) || fast512_rgba_available(qq) || fast2048_rgba_available(qq) || fast4096_rgba_available(qq))) {
    // the fast kernel writes the image itself; its rounds go through a scratch buffer sized at set-up
    if (!c->d_aux || c->d_aux_bytes < fast_rgba_scratch_bytes())
      return fail(c, SONO_ERR_ARG, "the context's image scratch is missing (tables changed without sono_set_tables?)");
    qq.tile_scratch = c->d_aux;
    r = launch_stft_fast(qq, c->stream);
    if (r == 1) r = launch_stft_fast512(qq, c->stream);
    if (r == 1) r = launch_stft_fast2048(qq, c->stream);
    if (r == 1) r = launch_stft_fast4096(qq, c->stream);
    if (r < 0) return fail(c, r, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    if (r == SONO_OK) {
      note_route(c, ":rgba");
      return SONO_OK;
    }
    // (r == 1: no image writer took the call after all, e.g. an output pointer that is not 4-byte aligned; the
    // two-kernel route below serves it)
  }
  if (rgba && q.row_of_bin0 == (int32_t)q.bin_hi - 1 && (q.nframes >= 16 || aux.single_chunk)) {
    // RGBA in two kernels: colour indices of a chunk of frames go to a [frame][bin] buffer small enough to stay in
    // the Infinity Cache, then a transposing LUT kernel writes the chunk's image columns.  The index kernels want
    // whole groups of four bins, so the chunk holds the band rounded outwards and the LUT kernel takes its rows.
    StftParams u = q;
    u.out_kind = SONO_OUT_U8_INDEX;
    u.bin_lo = q.bin_lo & ~3u;
    u.bin_hi = (q.bin_hi + 3u) & ~3u;
    u.out_bins = u.bin_hi - u.bin_lo;
    u.out_bin0 = u.bin_lo;
    u.lut = nullptr;
    const bool fast_u8 = exact ? exact_available(u) : (fast_u8_available(u) || mw_u8_available(u) || fast);
    if (fast_u8 || aux.single_chunk) {
      const uint64_t nb = u.out_bins;
      if (!aux.ptr || aux.bytes < 64 * nb)
        return fail(c, SONO_ERR_ARG, "no index scratch for the RGBA route");
      uint64_t chunk = (aux.bytes < kMaxChunkBytes ? aux.bytes : kMaxChunkBytes) / nb / 64 * 64;  // frames per chunk
      if (aux.single_chunk) {
        if (q.nframes * nb > aux.bytes)
          return fail(c, SONO_ERR_CAPACITY, "%llu frames x %llu bins exceed the %llu-byte static index plane",
                      (unsigned long long)q.nframes, (unsigned long long)nb, (unsigned long long)aux.bytes);
        chunk = q.nframes;
      }
      for (uint64_t f = 0; f < q.nframes; f += chunk) {
        u.frame0 = q.frame0 + f;
        u.nframes = (q.nframes - f < chunk) ? q.nframes - f : chunk;
        u.col0 = 0;
        u.out = aux.ptr;
        r = 1;
        if (exact) {
          r = launch_stft_exact(u, c->stream);
          if (r == 1) r = launch_stft_general(u, true, c->stream);
        } else if (u.nframes >= 16) {
          if (fast) r = launch_stft_fast(u, c->stream);
          if (r == 1 && fast) r = launch_stft_fast512(u, c->stream);
          if (r == 1 && fast) r = launch_stft_fast2048(u, c->stream);
          if (r == 1 && fast) r = launch_stft_fast4096(u, c->stream);
          if (r == 1 && u.mw_ok) r = launch_stft_mw(u, c->stream);
        }
        if (r == 1) r = launch_stft_general(u, exact, c->stream);
        if (r != SONO_OK) return fail(c, r, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        note_route(c, ":u8");
        r = launch_colorize((const uint8_t*)aux.ptr, u.nframes, (uint32_t)nb, u.bin_hi - q.bin_hi, q.bin_hi - q.bin_lo, q.lut,
                            (uint8_t*)q.out, q.width, q.col0 + f, c->stream);
        if (r != SONO_OK) return fail(c, r, "colorize launch failed");
        sono::note_kernel("colorize_kernel");
        note_route(c);
      }
      return SONO_OK;
    }
  }
  // Non-linear frequency scales in the float32 arithmetic: the float32 transform kernel writes a chunk of magnitudes
  // ([frame][n/2] float32, in the context's scratch), the filter-bank kernel gathers its two taps per output bin from it
  // and quantises; for the image a chunk of indices and the colour-map kernel follow.
  if (q.fb_bin && !fb_fused && !exact && (c->fb_fast_mode || c->fb_mw_ok) && aux.ptr &&
      (q.out_kind == SONO_OUT_U8_INDEX || (rgba && q.row_of_bin0 == (int32_t)q.bin_hi - 1))) {
    const uint32_t half = q.n / 2;
    const uint32_t lo4 = q.bin_lo & ~3u, hi4 = (q.bin_hi + 3u) & ~3u, nb4 = hi4 - lo4;
    const uint64_t per_frame = (uint64_t)half * 4 + (rgba ? nb4 : 0);
    uint64_t chunk = (aux.bytes < kMaxChunkBytes ? aux.bytes : kMaxChunkBytes) / per_frame / 64 * 64;  // frames per chunk
    if (aux.single_chunk) chunk = q.nframes * per_frame <= aux.bytes ? q.nframes : 0;
    if (chunk) {
      float* mags = reinterpret_cast<float*>(aux.ptr);
      uint8_t* idx = reinterpret_cast<uint8_t*>(aux.ptr) + chunk * half * 4;
      const FastScale fs = make_fast_scale(q.scale_mode, q.n, q.gain_db, q.range_db);
      for (uint64_t f = 0; f < q.nframes; f += chunk) {
        StftParams m = q;
        m.fb_bin = nullptr;
        m.fb_w0 = m.fb_w1 = nullptr;
        m.lut = nullptr;
        m.out_kind = SONO_OUT_F32_MAG;
        m.bin_lo = 0;
        m.bin_hi = half;
        m.out_bins = half;
        m.out_bin0 = 0;
        m.frame0 = q.frame0 + f;
        m.nframes = (q.nframes - f < chunk) ? q.nframes - f : chunk;
        m.col0 = 0;
        m.out = mags;
        m.fast_mode = c->fb_fast_mode;
        m.mw_ok = c->fb_mw_ok ? 1u : 0u;
        m.pair_exact = c->pair_idx_ok ? 1 : q.pair_exact;  // (an asymmetric table's cross-talk stays below the quantiser's floor, or no pairs)
        r = 1;
        if (m.nframes >= 16) {
          if (m.fast_mode) r = launch_stft_fast(m, c->stream);
          if (r == 1 && m.mw_ok) r = launch_stft_mw(m, c->stream);
        }
        if (r == 1) r = launch_stft_general(m, false, c->stream);
        if (r != SONO_OK) return fail(c, r, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
        note_route(c, ":mag");
        if (rgba) {
          r = launch_filterbank_index(mags, m.nframes, half, q.fb_bin, q.fb_w0, q.fb_w1, lo4, hi4, nb4, lo4, 0, fs, q.n, idx, c->stream);
          if (r == SONO_OK)
            r = launch_colorize(idx, m.nframes, nb4, hi4 - q.bin_hi, q.bin_hi - q.bin_lo, q.lut, (uint8_t*)q.out, q.width, q.col0 + f,
                                c->stream);
        } else {
          r = launch_filterbank_index(mags, m.nframes, half, q.fb_bin, q.fb_w0, q.fb_w1, q.bin_lo, q.bin_hi, q.out_bins, q.out_bin0,
                                      q.col0 + f, fs, q.n, reinterpret_cast<uint8_t*>(q.out), c->stream);
        }
        if (r != SONO_OK) return fail(c, r, "filter-bank launch failed");
        sono::note_kernel(rgba ? "filterbank_index_kernel + colorize_kernel" : "filterbank_index_kernel");
        note_route(c);
      }
      return SONO_OK;
    }
  }
  // The float32 index and feature kernels work on whole groups of four bins.  A band that starts or ends inside a group
  // is served by the band rounded outwards: feature records whose peak fell into the margin are recomputed by the
  // exact-arithmetic pass below (which searches the band itself); colour indices go to a [frame][rounded bins] chunk of
  // the context's scratch and the band's columns are copied out of it.
  const bool ragged = !exact && (fast || q.mw_ok) && ((q.bin_lo | q.bin_hi) & 3u) && !q.fb_bin;
  StftParams w = q;
  if (ragged) {
    w.bin_lo = q.bin_lo & ~3u;
    w.bin_hi = (q.bin_hi + 3u) & ~3u;
  }
  if (ragged && q.out_kind == SONO_OUT_U8_INDEX && aux.ptr && aux.bytes >= 64 * (uint64_t)(w.bin_hi - w.bin_lo)) {
    const uint64_t nb = w.bin_hi - w.bin_lo;
    const uint64_t chunk = (aux.bytes < kMaxChunkBytes ? aux.bytes : kMaxChunkBytes) / nb / 64 * 64;  // frames per chunk
    w.out_bins = (uint32_t)nb;
    w.out_bin0 = w.bin_lo;
    for (uint64_t f = 0; f < q.nframes; f += chunk) {
      w.frame0 = q.frame0 + f;
      w.nframes = (q.nframes - f < chunk) ? q.nframes - f : chunk;
      w.col0 = 0;
      w.out = aux.ptr;
      r = 1;
      if (w.nframes >= 16) {
        if (fast) r = launch_stft_fast(w, c->stream);
        if (r == 1 && fast) r = launch_stft_fast512(w, c->stream);
        if (r == 1 && fast) r = launch_stft_fast2048(w, c->stream);
        if (r == 1 && fast) r = launch_stft_fast4096(w, c->stream);
        if (r == 1 && w.mw_ok) r = launch_stft_mw(w, c->stream);
      }
      if (r == 1) r = launch_stft_general(w, false, c->stream);
      if (r != SONO_OK) return fail(c, r, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
      note_route(c, ":band");
      uint8_t* dst = reinterpret_cast<uint8_t*>(q.out) + (q.col0 + f) * q.out_bins + (q.bin_lo - q.out_bin0);
      HIP_TRY(c, hipMemcpy2DAsync(dst, q.out_bins, reinterpret_cast<const uint8_t*>(aux.ptr) + (q.bin_lo - w.bin_lo), nb,
                                  q.bin_hi - q.bin_lo, w.nframes, hipMemcpyDeviceToDevice, c->stream));
    }
    return SONO_OK;
  }
  StftParams fq = (ragged && q.out_kind == SONO_OUT_FEATURES) ? w : q;
  // (1024 points: the float32 feature kernel lists the frames it marks itself, in the queue the fix-up pass reads)
  bool listed = false;
  if (fast && q.out_kind == SONO_OUT_FEATURES && q.n == 1024 && c->d_tw_exact && c->d_claim && fast_features_available(fq)) {
    StftParams u = q;
    u.twiddles = c->d_tw_exact;
    u.tile_scratch = c->d_claim;
    u.fixup = kFixupCapacity;
    if (features_fixup_available(u)) {
      HIP_TRY(c, hipMemsetAsync(c->d_claim, 0, 16, c->stream));
      fq.feat_queue = reinterpret_cast<uint32_t*>(c->d_claim);
      fq.feat_queue_cap = kFixupCapacity;
      listed = true;
    }
  }
  if (fast) r = launch_stft_fast(fq, c->stream);
  if (r != SONO_OK) listed = false;
  if (r == 1 && fast) r = launch_stft_fast512(fq, c->stream);
  if (r == 1 && fast) r = launch_stft_fast2048(fq, c->stream);
  if (r == 1 && fast) r = launch_stft_fast4096(fq, c->stream);
  if (r == 1 && q.mw_ok) r = launch_stft_mw(fq, c->stream);
  if (exact) r = launch_stft_exact(q, c->stream);
  // Feature records are integers that segmentation and species rules branch on, so they are the reference's in
  // both arithmetics: a float32 kernel marks the frames it cannot decide (two top bins too close, peak too close
  // to a step of the quantiser) and those records are recomputed in the exact arithmetic right behind it; where no
  // float32 kernel with that marking serves, the pass runs in the exact arithmetic outright.
  const bool feat_fast = q.out_kind == SONO_OUT_FEATURES && !exact && c->d_tw_exact && !q.fb_bin;
  if (r == 1 && feat_fast) {
    StftParams u = q;
    u.twiddles = c->d_tw_exact;
    r = launch_stft_general(u, true, c->stream);
    if (r != SONO_OK) return fail(c, r, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    note_route(c);
    return SONO_OK;
  }
  if (r == 1) r = launch_stft_general(q, exact, c->stream);
  if (r != SONO_OK) return fail(c, r, "kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
  note_route(c);
  if (feat_fast) {
    StftParams u = q;
    u.twiddles = c->d_tw_exact;
    u.tile_scratch = c->d_claim;
    u.fixup = kFixupCapacity;
    r = launch_features_fixup(u, c->stream, listed);
    if (r == 1) {
      u.fixup = 1;
      HIP_TRY(c, hipMemsetAsync(c->d_claim, 0, 16, c->stream));
      r = launch_stft_general(u, true, c->stream);
    }
    if (r != SONO_OK) return fail(c, r, "feature fix-up launch failed: %s", hipGetErrorString(hipGetLastError()));
    note_route(c);
  }
  return SONO_OK;
}

// The transform of one recording on device pointers: the frames with a column of their own, then the worker's
// wrapped last frame.  `aux` is where a two-kernel RGBA route parks its colour indices.
static int transform(sono_ctx* c, const void* pcm, uint64_t L, void* out, uint64_t cap, const IndexScratch& aux, bool fused_ok) {
  if (!out || (!pcm && L)) return fail(c, SONO_ERR_ARG, "NULL buffer");
  sono_geometry g;
  int rc = sono_geometry_of(c, L, &g);
  if (rc != SONO_OK) return fail(c, rc, "bad geometry");
  const uint64_t need = sono_out_bytes(c, L);
  if (cap < need) return fail(c, SONO_ERR_CAPACITY, "output needs %llu bytes, capacity is %llu", (unsigned long long)need,
                              (unsigned long long)cap);
  HIP_TRY(c, hipSetDevice(c->device));
  c->route.clear();
  const uint32_t n = c->cfg.fft_n, H = n / 2;
  const bool rgba = c->cfg.out_kind == SONO_OUT_RGBA8;
  // every image route stores whole pixels (ImageData's Uint8ClampedArray sits on an ArrayBuffer, which is aligned)
  if (rgba && (reinterpret_cast<uintptr_t>(out) & 3)) return fail(c, SONO_ERR_ARG, "an RGBA8 image must start on a 4-byte boundary");
  if (rgba && g.frames < g.width)  // L < N under the worker rule: a one-column, all-zero image (createImageData)
    HIP_TRY(c, hipMemsetAsync(out, 0, need, c->stream));
  if (g.frames == 0) return SONO_OK;

  StftParams p{};
  p.pcm = pcm;
  p.n_samples = L;
  p.n = n;
  p.log2n = ilog2(n);
  p.hop = c->cfg.hop;
  p.width = g.width;
  p.window = c->d_window;
  p.twiddles = c->d_tw;
  p.lut = c->have_lut ? c->d_lut : nullptr;
  p.out = out;
  p.fast_table = c->d_fast;
  p.fast_mode = c->fast_mode;
  p.mw_ok = c->mw_ok ? 1u : 0u;
  p.pair_exact = c->pair_exact ? 1u : 0u;
  p.pair_asym = c->pair_asym;
  p.pair_idx_ok = c->pair_idx_ok ? 1u : 0u;
  p.split_ok = c->split_ok ? 1u : 0u;
  p.fb_bin = c->d_fb_bin;
  p.fb_w0 = c->d_fb_w;
  p.fb_w1 = c->d_fb_w ? c->d_fb_w + n / 2 : nullptr;
  p.tile_scratch = nullptr;
  p.scale_mode = c->cfg.scale_mode;
  p.out_kind = c->cfg.out_kind;
  p.in_kind = c->cfg.in_kind;
  p.gain_db = c->cfg.gain_db;
  p.range_db = c->cfg.range_db;
  p.out_bins = g.bins;
  p.out_bin0 = c->bin_lo;

  // frames that have a column of their own
  const uint64_t main_frames = g.frames < g.width ? g.frames : g.width;
  p.frame0 = 0;
  p.nframes = main_frames;
  p.col0 = 0;
  p.bin_lo = c->bin_lo;
  p.bin_hi = c->bin_hi;
  p.feat_lo = c->bin_lo;
  p.feat_hi = c->bin_hi;
  p.row_of_bin0 = (int32_t)c->bin_hi - 1;
  rc = launch_range(c, p, aux, fused_ok);
  if (rc != SONO_OK) return rc;

  if (g.frames > g.width) {
    // worker quirk, spectrogramWorker.js:40,59,79: frame `width` is written at x == width, i.e. at
    // column 0 one image row further down; its bin 0 falls off the end of the image.
    const IndexScratch none = {nullptr, 0, false};
    if (rgba) {
      StftParams q = p;
      q.frame0 = g.width;
      q.nframes = g.frames - g.width;  // always 1
      q.col0 = 0;
      q.bin_lo = c->bin_lo + 1;
      q.bin_hi = (c->bin_hi + 1 <= H) ? c->bin_hi + 1 : H;
      q.row_of_bin0 = (int32_t)c->bin_hi;  // one row down
      if (q.bin_lo < q.bin_hi) {
        rc = launch_range(c, q, none, false);
        if (rc != SONO_OK) return rc;
      }
    } else {
      // [frame][bin] outputs simply carry the extra frame as one more row
      StftParams q = p;
      q.frame0 = g.width;
      q.nframes = g.frames - g.width;
      q.col0 = g.width;
      rc = launch_range(c, q, none, false);
      if (rc != SONO_OK) return rc;
    }
  }
  return SONO_OK;
}

int sono_stft_async(sono_ctx* c, const void* pcm, uint64_t L, void* out, uint64_t cap) {
  if (!c) return SONO_ERR_ARG;
  const IndexScratch aux = {c->d_aux, c->d_aux_bytes, false};
  return transform(c, pcm, L, out, cap, aux, true);
}

int sono_stft_static(sono_ctx* c, const void* pcm, uint64_t L, void* out, uint64_t cap) {
  if (!c) return SONO_ERR_ARG;
  if (c->cfg.out_kind != SONO_OUT_RGBA8) return fail(c, SONO_ERR_ARG, "sono_stft_static serves SONO_OUT_RGBA8 contexts");
  if (c->device < 0 || c->device >= 64) return fail(c, SONO_ERR_DEVICE, "the static index plane has no address on device %d", c->device);
  StaticScratch& st = g_static[c->device];
  if (!st.ptr) return fail(c, SONO_ERR_DEVICE, "the static index plane has no address on device %d", c->device);
  // Calls on one stream follow each other there; a call on another stream may only start once the kernels of the
  // array's last user have finished.  (hipEventQuery and hipEventRecord allocate nothing.)
  std::lock_guard<std::mutex> lock(g_static_mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (st.used && st.done && st.last != c->stream) {
    const hipError_t q = hipEventQuery(st.done);
    if (q == hipErrorNotReady)
      return fail(c, SONO_ERR_BUSY, "the static index plane of device %d is in use by a call on another stream", c->device);
    if (q != hipSuccess) (void)hipGetLastError();  // (e.g. a query that stream capture does not allow: the caller orders the calls)
  }
  const IndexScratch aux = {st.ptr, (uint64_t)SONO_STATIC_SCRATCH_BYTES, true};
  const int rc = transform(c, pcm, L, out, cap, aux, false);
  if (st.done) {
    if (hipEventRecord(st.done, c->stream) == hipSuccess) {
      st.last = c->stream;
      st.used = true;
    } else {
      (void)hipGetLastError();
    }
  }
  return rc;
}

const char* sono_kernel_route(const sono_ctx* c) { return c ? c->route.c_str() : ""; }

int64_t sono_features_recomputed(sono_ctx* c) {
  if (!c) return SONO_ERR_ARG;
  if (!c->d_claim) return 0;
  uint32_t words[2] = {0, 0};   // [0]: marked records of the 1024-point pass; [1]: of the general pass
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(words, c->d_claim, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return (int64_t)words[0] + (int64_t)words[1];
}

int sono_reserve(sono_ctx* c, uint64_t L) {
  if (!c) return SONO_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int rc = ensure(c, &c->d_in, &c->d_in_bytes, in_bytes(c, L));
  if (rc != SONO_OK) return rc;
  rc = ensure(c, &c->d_out, &c->d_out_bytes, sono_out_bytes(c, L));
  if (rc != SONO_OK) return rc;
  if (c->cfg.out_kind == SONO_OUT_RGBA8 && !c->fast_mode) {
    sono_geometry g;
    if (sono_geometry_of(c, L, &g) == SONO_OK) {
      uint64_t want = g.frames * (((uint64_t)c->bin_hi + 3) / 4 * 4 - (c->bin_lo & ~3u));
      if (want > kMaxChunkBytes) want = kMaxChunkBytes;
      if (want > c->d_aux_bytes) rc = ensure(c, &c->d_aux, &c->d_aux_bytes, want);
    }
  }
  return rc;
}

int sono_pcm_to_f32(sono_ctx* c, const void* src, int mem, uint32_t format_tag, uint32_t bits, uint32_t channels, uint32_t channel,
                    uint64_t frames, float* dst_dev) {
  if (!c) return SONO_ERR_ARG;
  if (!src || !dst_dev || !channels || channel >= channels) return fail(c, SONO_ERR_ARG, "bad PCM layout");
  const bool ok = (format_tag == 3 && bits == 32) || (format_tag == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32));
  if (!ok) return fail(c, SONO_ERR_ARG, "unsupported PCM format (tag %u, %u bits)", format_tag, bits);
  HIP_TRY(c, hipSetDevice(c->device));
  const void* dsrc = src;
  if (mem == SONO_MEM_HOST) {  // raw bytes cross PCIe; the float array exists on the device only
    const uint64_t nbytes = frames * channels * (bits / 8);
    int rc = ensure(c, &c->d_in, &c->d_in_bytes, nbytes);
    if (rc != SONO_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_in, src, nbytes, hipMemcpyHostToDevice, c->stream));
    dsrc = c->d_in;
  }
  int r = launch_pcm_to_f32(dsrc, format_tag, bits, channels, channel, frames, dst_dev, c->stream);
  if (r != SONO_OK) return fail(c, r, "PCM conversion launch failed");
  if (mem == SONO_MEM_HOST) HIP_TRY(c, hipStreamSynchronize(c->stream));  // the staging buffer is reused by the next call
  return SONO_OK;
}

int sono_wait(sono_ctx* c) {
  if (!c) return SONO_ERR_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SONO_OK;
}

int sono_stft_host_async(sono_ctx* c, const void* pcm, uint64_t L, void* out, uint64_t cap) {
  if (!c) return SONO_ERR_ARG;
  if (!out || (!pcm && L)) return fail(c, SONO_ERR_ARG, "NULL buffer");
  const uint64_t need = sono_out_bytes(c, L);
  if (cap < need) return fail(c, SONO_ERR_CAPACITY, "output needs %llu bytes, capacity is %llu", (unsigned long long)need,
                              (unsigned long long)cap);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure(c, &c->d_in, &c->d_in_bytes, in_bytes(c, L));
  if (rc != SONO_OK) return rc;
  rc = ensure(c, &c->d_out, &c->d_out_bytes, need);
  if (rc != SONO_OK) return rc;
  if (L) HIP_TRY(c, hipMemcpyAsync(c->d_in, pcm, in_bytes(c, L), hipMemcpyHostToDevice, c->stream));
  rc = sono_stft_async(c, c->d_in, L, c->d_out, c->d_out_bytes);
  if (rc != SONO_OK) return rc;
  if (need) HIP_TRY(c, hipMemcpyAsync(out, c->d_out, need, hipMemcpyDeviceToHost, c->stream));
  return SONO_OK;
}

int sono_stft(sono_ctx* c, const void* pcm, uint64_t L, void* out, uint64_t cap, int mem) {
  if (!c) return SONO_ERR_ARG;
  if (mem == SONO_MEM_DEVICE) {
    int rc = sono_stft_async(c, pcm, L, out, cap);
    return rc != SONO_OK ? rc : sono_wait(c);
  }
  if (mem != SONO_MEM_HOST) return fail(c, SONO_ERR_ARG, "mem must be SONO_MEM_HOST or SONO_MEM_DEVICE");
  int rc = sono_stft_host_async(c, pcm, L, out, cap);
  return rc != SONO_OK ? rc : sono_wait(c);
}

void* sono_host_malloc(uint64_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

int sono_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? SONO_OK : SONO_ERR_DEVICE; }

int sono_resample_cols(sono_ctx* c, const uint8_t* in, uint64_t F, uint32_t bins, uint64_t W, uint64_t col0, uint64_t ncols,
                       uint8_t* out, int mem) {
  if (!c) return SONO_ERR_ARG;
  if (!in || !out || bins == 0 || col0 > W || ncols > W - col0) return fail(c, SONO_ERR_ARG, "bad resample arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint8_t* din = in;
  uint8_t* dout = out;
  if (mem == SONO_MEM_HOST) {
    int rc = ensure(c, &c->d_aux, &c->d_aux_bytes, F * bins);
    if (rc != SONO_OK) return rc;
    rc = ensure(c, &c->d_out, &c->d_out_bytes, ncols * bins);
    if (rc != SONO_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_aux, in, F * bins, hipMemcpyHostToDevice, c->stream));
    din = (const uint8_t*)c->d_aux;
    dout = (uint8_t*)c->d_out;
  }
  if (F == 0 && ncols) HIP_TRY(c, hipMemsetAsync(dout, 0, ncols * bins, c->stream));
  int rc = launch_resample(din, F, bins, W, col0, ncols, dout, c->stream);
  if (rc != SONO_OK) return fail(c, rc, "resample launch failed");
  if (mem == SONO_MEM_HOST && ncols) HIP_TRY(c, hipMemcpyAsync(out, dout, ncols * bins, hipMemcpyDeviceToHost, c->stream));
  return sono_wait(c);
}

int sono_resample(sono_ctx* c, const uint8_t* in, uint64_t F, uint32_t bins, uint64_t W, uint8_t* out, int mem) {
  return sono_resample_cols(c, in, F, bins, W, 0, W, out, mem);
}

int sono_colorize(sono_ctx* c, const uint8_t* frames, uint64_t width, uint32_t bins, uint32_t row0, uint32_t rows,
                  uint8_t* rgba, int mem) {
  if (!c) return SONO_ERR_ARG;
  if (!frames || !rgba || bins == 0 || (uint64_t)row0 + rows > bins) return fail(c, SONO_ERR_ARG, "bad colorize arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint8_t* din = frames;
  uint8_t* dout = rgba;
  const uint64_t out_bytes = width * rows * 4;
  if (mem == SONO_MEM_HOST) {
    int rc = ensure(c, &c->d_aux, &c->d_aux_bytes, width * bins);
    if (rc != SONO_OK) return rc;
    rc = ensure(c, &c->d_out, &c->d_out_bytes, out_bytes);
    if (rc != SONO_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_aux, frames, width * bins, hipMemcpyHostToDevice, c->stream));
    din = (const uint8_t*)c->d_aux;
    dout = (uint8_t*)c->d_out;
  }
  int rc = launch_colorize(din, width, bins, row0, rows, c->have_lut ? c->d_lut : nullptr, dout, width, 0, c->stream);
  if (rc != SONO_OK) return fail(c, rc, "colorize launch failed");
  if (mem == SONO_MEM_HOST && out_bytes) HIP_TRY(c, hipMemcpyAsync(rgba, dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
  return sono_wait(c);
}

int sono_canvas_draw(sono_ctx* c, const uint8_t* frames, uint64_t width, uint32_t bins, int64_t sy, int64_t sh, double dy,
                     double dh, uint32_t height, const uint8_t* background_rgba, uint8_t* canvas, int mem) {
  if (!c) return SONO_ERR_ARG;
  if (!frames || !canvas || bins == 0 || dy != dy || dh != dh) return fail(c, SONO_ERR_ARG, "bad canvas_draw arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint8_t* din = frames;
  uint8_t* dout = canvas;
  const uint64_t out_bytes = width * (uint64_t)height * 4;
  if (mem == SONO_MEM_HOST) {
    int rc = ensure(c, &c->d_aux, &c->d_aux_bytes, width * bins);
    if (rc != SONO_OK) return rc;
    rc = ensure(c, &c->d_out, &c->d_out_bytes, out_bytes);
    if (rc != SONO_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_aux, frames, width * bins, hipMemcpyHostToDevice, c->stream));
    din = (const uint8_t*)c->d_aux;
    dout = (uint8_t*)c->d_out;
  }
  uint32_t bg = 0;
  if (background_rgba)
    bg = (uint32_t)background_rgba[0] | (uint32_t)background_rgba[1] << 8 | (uint32_t)background_rgba[2] << 16 |
         (uint32_t)background_rgba[3] << 24;
  int rc = launch_canvas_draw(din, width, bins, sy, sh, dy, dh, height, bg, c->have_lut ? c->d_lut : nullptr, dout, c->stream);
  if (rc != SONO_OK) return fail(c, rc, "canvas_draw launch failed");
  if (mem == SONO_MEM_HOST && out_bytes) HIP_TRY(c, hipMemcpyAsync(canvas, dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
  return sono_wait(c);
}

int sono_frame_peaks(sono_ctx* c, const uint8_t* idx, uint64_t frames, uint32_t bins, int32_t* peak_bin, uint8_t* peak_idx,
                     int mem) {
  if (!c) return SONO_ERR_ARG;
  if (!idx || !peak_bin || !peak_idx || bins == 0) return fail(c, SONO_ERR_ARG, "bad frame_peaks arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint8_t* din = idx;
  int32_t* dbin = peak_bin;
  uint8_t* dval = peak_idx;
  if (mem == SONO_MEM_HOST) {
    int rc = ensure(c, &c->d_aux, &c->d_aux_bytes, frames * bins);
    if (rc != SONO_OK) return rc;
    rc = ensure(c, &c->d_out, &c->d_out_bytes, frames * 8 + 64);
    if (rc != SONO_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_aux, idx, frames * bins, hipMemcpyHostToDevice, c->stream));
    din = (const uint8_t*)c->d_aux;
    dbin = (int32_t*)c->d_out;
    dval = (uint8_t*)c->d_out + frames * 4;
  }
  int rc = launch_frame_peaks(din, frames, bins, dbin, dval, c->stream);
  if (rc != SONO_OK) return fail(c, rc, "frame_peaks launch failed");
  if (mem == SONO_MEM_HOST && frames) {
    HIP_TRY(c, hipMemcpyAsync(peak_bin, dbin, frames * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(peak_idx, dval, frames, hipMemcpyDeviceToHost, c->stream));
  }
  return sono_wait(c);
}

void* sono_device_malloc(sono_ctx* c, uint64_t bytes) {
  if (!c) return nullptr;
  void* p = nullptr;
  if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
    fail(c, SONO_ERR_DEVICE, "hipMalloc(%llu) failed", (unsigned long long)bytes);
    return nullptr;
  }
  return p;
}

int sono_device_free(sono_ctx* c, void* p) {
  if (!c) return SONO_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipFree(p));
  return SONO_OK;
}

int sono_copy_to_device(sono_ctx* c, void* dst, const void* src, uint64_t bytes) {
  if (!c) return SONO_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  return sono_wait(c);
}

int sono_copy_to_host(sono_ctx* c, void* dst, const void* src, uint64_t bytes) {
  if (!c) return SONO_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  return sono_wait(c);
}

double sono_measure_valu_peak(sono_ctx* c) {
  if (!c) return -1.0;
  if (hipSetDevice(c->device) != hipSuccess) return -1.0;
  float* sink = nullptr;
  if (hipMalloc((void**)&sink, 256) != hipSuccess) return -1.0;
  double best = -1.0;
  for (int rep = 0; rep < 3; ++rep) {  // the first launch also loads the code
    if (hipEventRecord(c->ev_start, c->stream) != hipSuccess) break;
    const uint64_t n = launch_valu_peak(sink, 20000, c->stream);
    float ms = 0;
    if (hipEventRecord(c->ev_stop, c->stream) != hipSuccess || hipEventSynchronize(c->ev_stop) != hipSuccess ||
        hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) != hipSuccess || !(ms > 0))
      break;
    const double rate = (double)n / ((double)ms * 1e-3) / 1e9;
    if (rate > best) best = rate;
  }
  (void)hipFree(sink);
  return best;
}

int sono_timer_start(sono_ctx* c) {
  if (!c) return SONO_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
  return SONO_OK;
}

double sono_timer_stop_ms(sono_ctx* c) {
  if (!c) return -1.0;
  if (hipEventRecord(c->ev_stop, c->stream) != hipSuccess || hipEventSynchronize(c->ev_stop) != hipSuccess) {
    fail(c, SONO_ERR_DEVICE, "event record/synchronize failed");
    return -1.0;
  }
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) != hipSuccess) {
    fail(c, SONO_ERR_DEVICE, "hipEventElapsedTime failed");
    return -1.0;
  }
  return (double)ms;
}

}  // extern "C"
