// Native benchmark / profiling host for libsonospec.so: no Python, no PyTorch, just the C ABI.
//   sono_bench [seconds=600] [steps=5] [out=u8|rgba|feat] [fft=1024] [hop=256] [path=worker|plugin] [bin_lo bin_hi (0 0 = all)] [arith=fast|exact] [scale=mel|logarithmic|bark|erb]
// Environment: SONO_BENCH_WINDOW=<SONO_WIN_* id> selects the plugin path's window (default Hann).
// Generates a seeded synthetic recording on the host, copies it to the GPU once, runs the worker-path
// transform `steps` times on device-resident buffers and prints Msamples/s from HIP-event timing.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sonospec.h"

int main(int argc, char** argv) {
  const int seconds = argc > 1 ? atoi(argv[1]) : 600;
  const int steps = argc > 2 ? atoi(argv[2]) : 5;
  const bool rgba = argc > 3 && !strcmp(argv[3], "rgba");
  const bool feat = argc > 3 && !strcmp(argv[3], "feat");
  const bool plugin = argc > 6 && !strcmp(argv[6], "plugin");
  const uint32_t fft = argc > 4 ? (uint32_t)atoi(argv[4]) : 1024;
  const uint32_t hop = argc > 5 ? (uint32_t)atoi(argv[5]) : 256;
  const uint64_t L = (uint64_t)seconds * 384000ull;
  std::vector<float> x(L);
  const float pulse_amp = getenv("SONO_BENCH_PULSE") ? (float)atof(getenv("SONO_BENCH_PULSE")) : 0.3f;  // peak of the calls
  uint32_t s = 12345;
  for (uint64_t i = 0; i < L; i++) {
    s = (1103515245u * s + 12345u) & 0x7fffffffu;
    float v = 0.01f * (2.0f * (float)s / 2147483648.0f - 1.0f);
    const uint64_t p = i % 30720;  // 80 ms
    if (p < 1920) v += pulse_amp * 0.5f * (1.0f - cosf(6.2831853f * (float)p / 1920.0f)) * sinf(1.2f * (float)p - 1.6e-4f * (float)p * (float)p);
    x[i] = v;
  }
  sono_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.fft_n = fft;
  cfg.hop = hop;
  cfg.window_id = plugin ? SONO_WIN_HANN : SONO_WIN_WORKER_HANN;
  if (plugin && getenv("SONO_BENCH_WINDOW")) cfg.window_id = (uint32_t)atoi(getenv("SONO_BENCH_WINDOW"));  // a SONO_WIN_* id
  cfg.scale_mode = plugin ? SONO_SCALE_PLUGIN_DB : SONO_SCALE_WORKER_LOG5;
  cfg.frame_rule = plugin ? SONO_FRAMES_PLUGIN_STRICT : SONO_FRAMES_WORKER_COMPAT;
  cfg.twiddle_mode = plugin ? SONO_TW_PLUGIN_RECURRENCE : SONO_TW_ACCURATE;
  if (getenv("SONO_BENCH_TW")) cfg.twiddle_mode = (uint32_t)atoi(getenv("SONO_BENCH_TW"));  // a SONO_TW_* id (A/B of the table flavours)
  cfg.arith = (argc > 9 && !strcmp(argv[9], "exact")) ? SONO_ARITH_EXACT_F64 : SONO_ARITH_FAST_F32;
  cfg.out_kind = rgba ? SONO_OUT_RGBA8 : feat ? SONO_OUT_FEATURES : SONO_OUT_U8_INDEX;
  if (argc > 8 && atoi(argv[8]) > 0) {
    cfg.bin_lo = (uint32_t)atoi(argv[7]);
    cfg.bin_hi = (uint32_t)atoi(argv[8]);
  }
  cfg.in_kind = SONO_IN_F32;
  cfg.gain_db = 20;
  cfg.range_db = 80;
  sono_ctx* c = sono_create(0, &cfg);
  if (!c) {
    fprintf(stderr, "sono_create: %s\n", sono_create_error());
    return 1;
  }
  if (argc > 10) {  // a non-linear frequency scale: mel | logarithmic | bark | erb
    const char* names[] = {"linear", "mel", "logarithmic", "bark", "erb"};
    int which = -1;
    for (int i = 1; i < 5; i++)
      if (!strcmp(argv[10], names[i])) which = i;
    std::vector<int32_t> tap(fft / 2);
    std::vector<double> w0(fft / 2), w1(fft / 2);
    if (which < 0 || sono_filterbank_table((uint32_t)which, fft, 384000.0, fft / 2, tap.data(), w0.data(), w1.data()) != SONO_OK ||
        sono_set_filterbank(c, tap.data(), w0.data(), w1.data()) != SONO_OK) {
      fprintf(stderr, "filter bank '%s': %s\n", argv[10], sono_last_error(c));
      return 1;
    }
  }
  const uint64_t ob = sono_out_bytes(c, L);
  void* dx = sono_device_malloc(c, L * 4);
  void* dout = sono_device_malloc(c, ob);
  if (!dx || !dout || sono_copy_to_device(c, dx, x.data(), L * 4) != SONO_OK) {
    fprintf(stderr, "device setup: %s\n", sono_last_error(c));
    return 1;
  }
  if (sono_stft(c, dx, L, dout, ob, SONO_MEM_DEVICE) != SONO_OK) {
    fprintf(stderr, "stft: %s\n", sono_last_error(c));
    return 1;
  }
  sono_timer_start(c);
  for (int i = 0; i < steps; i++)
    if (sono_stft_async(c, dx, L, dout, ob) != SONO_OK) {
      fprintf(stderr, "stft: %s\n", sono_last_error(c));
      return 1;
    }
  const double ms = sono_timer_stop_ms(c) / steps;
  std::vector<uint8_t> head(4096);
  sono_copy_to_host(c, head.data(), dout, head.size());
  unsigned sum = 0;
  for (uint8_t b : head) sum += b;
  const double algo = (double)L * 4 + (double)ob;
  printf("{\"samples\": %llu, \"ms_per_step\": %.4f, \"msamples_per_s\": %.1f, \"algorithmic_gbps\": %.1f, \"hbm_frac\": %.4f, \"checksum\": %u, "
         "\"route\": \"%s\", \"recomputed\": %lld}\n",
         (unsigned long long)L, ms, (double)L / ms / 1e3, algo / ms / 1e6, algo / ms / 1e6 / 8000.0, sum, sono_kernel_route(c),
         (long long)sono_features_recomputed(c));
  sono_device_free(c, dx);
  sono_device_free(c, dout);
  sono_destroy(c);
  return 0;
}
