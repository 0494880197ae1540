// Issue-rate microbenchmarks for gfx950 (MI355X): how fast one wavefront, and 1 / 2 / 4 wavefronts per SIMD, issue
// vector-ALU instructions, and whether a vector-ALU phase and an LDS phase of different wavefronts overlap.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_issue.hip -o tools/microbench_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// 64 FMAs per block over ILP independent accumulators.
template <int ILP>
__global__ void valu_chain(float* sink, int iters) {
  float acc[ILP];
  float b = 1.0000001f, c = 1e-9f;
  asm volatile("" : "+v"(b), "+v"(c));
#pragma unroll
  for (int i = 0; i < ILP; ++i) acc[i] = (float)threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 64 / ILP; ++j)
#pragma unroll
      for (int i = 0; i < ILP; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(b), "v"(c));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += acc[i];
  if (s == 12345.678f) sink[0] = s;
}

// A wavefront alternates a vector-ALU phase of VALU_N FMAs (32 independent accumulators, like a register pass of a
// transform) and an LDS phase like a 32 x 32 transpose of one plane per half-wavefront: 32 4-byte stores, 8 16-byte
// loads, twice.  MODE: 0 = both, 1 = vector ALU only, 2 = LDS only.
template <int VALU_N, int MODE>
__global__ __launch_bounds__(1024) void phases(float* sink, int iters, int stagger) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, l = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  float* plane = lds + (wave * 2 + h) * (32 * 36);
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = (float)(threadIdx.x + i);
  float b = 1.0000001f, c = 1e-9f;
  asm volatile("" : "+v"(b), "+v"(c));
  // stagger >= 0: the wavefronts that share a SIMD (wave, wave + 4, ...) start `stagger` x 64 cycles apart
  if (stagger >= 0) {
    for (int d = 0; d < (wave >> 2); ++d)
      for (int e = 0; e < stagger; ++e) __builtin_amdgcn_s_sleep(1);
  } else if (stagger == -1) {  // two of the four wavefronts of a SIMD at a higher issue priority
    if ((wave >> 2) & 1) __builtin_amdgcn_s_setprio(3);
  } else if (stagger == -2) {  // four priority levels
    if ((wave >> 2) == 1) __builtin_amdgcn_s_setprio(1);
    if ((wave >> 2) == 2) __builtin_amdgcn_s_setprio(2);
    if ((wave >> 2) == 3) __builtin_amdgcn_s_setprio(3);
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE != 2) {
#pragma unroll
      for (int j = 0; j < VALU_N / 32; ++j)
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
    }
    if (MODE != 1) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
        for (int k = 0; k < 32; ++k) plane[k * 36 + l] = v[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4* row = reinterpret_cast<const float4*>(plane + l * 36);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 q = row[j];
          v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += v[i];
  if (s == 12345.678f) sink[0] = s;
}

__global__ void cvt_probe(const float* in, uint32_t* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}

template <typename F>
static double time_ms(F&& launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < 3; ++r) {
    CHECK(hipEventRecord(a, 0));
    launch();
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double mhz = prop.clockRate / 1000.0;
  printf("device %s, %d CUs, clockRate %.0f MHz\n", prop.name, cus, mhz);
  float* sink;
  CHECK(hipMalloc(&sink, 256));
  const int iters = 20000;
  printf("# valu_chain: waves/SIMD, ILP -> G wave-instr/s chip-wide, cycles per instr per SIMD at clockRate\n");
  const int blocks[4] = {64, 256, 512, 1024};
  for (int bi = 0; bi < 4; ++bi) {
    const int threads = blocks[bi];
    const double waves = threads / 64.0;
    auto report = [&](int ilp, double ms) {
      const double n = (double)cus * waves * 64.0 * iters;
      const double rate = n / (ms * 1e-3) / 1e9;
      const double simds = cus * (threads >= 256 ? 4.0 : 1.0);
      printf("threads/CU %4d  ilp %2d  %.3f ms  %8.1f G/s  %.2f cycles/instr/SIMD\n", threads, ilp, ms, rate,
             simds * mhz * 1e6 / (rate * 1e9));
    };
    report(1, time_ms([&] { hipLaunchKernelGGL(valu_chain<1>, dim3(cus), dim3(threads), 0, 0, sink, iters); }));
    report(2, time_ms([&] { hipLaunchKernelGGL(valu_chain<2>, dim3(cus), dim3(threads), 0, 0, sink, iters); }));
    report(4, time_ms([&] { hipLaunchKernelGGL(valu_chain<4>, dim3(cus), dim3(threads), 0, 0, sink, iters); }));
    report(8, time_ms([&] { hipLaunchKernelGGL(valu_chain<8>, dim3(cus), dim3(threads), 0, 0, sink, iters); }));
  }
  printf("# phases: 16 wavefronts per CU; per iteration VALU_N FMAs + two plane transposes through LDS\n");
  const int pit = 4000;
  const int ldsb = 16 * 2 * 32 * 36 * 4;
#define PH(N, M)                                                                                       \
  do {                                                                                                 \
    CHECK(hipFuncSetAttribute((const void*)phases<N, M>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); \
    for (int threads = 256; threads <= 1024; threads *= 2) {                                           \
      const double ms = time_ms([&] { hipLaunchKernelGGL((phases<N, M>), dim3(cus), dim3(threads), ldsb, 0, sink, pit, stag); }); \
      printf("stagger %3d valu %4d mode %d threads/CU %4d: %.3f ms = %.1f ns per iteration of the CU's wavefronts\n", stag, N, M, threads, ms, \
             ms * 1e6 / pit);                                                                          \
    }                                                                                                  \
  } while (0)
  for (int stag = -2; stag <= 20; stag += (stag < 0 ? 1 : 20)) {
    PH(640, 0);
    PH(640, 1);
    PH(640, 2);
    PH(1280, 0);
    PH(1280, 1);
    PH(320, 0);
    PH(320, 1);
  }
  {
    // v_cvt_pk_u8_f32: how does it round?
    float h_in[12] = {0.5f, 1.5f, 2.5f, 2.7f, 3.49f, 254.9f, 255.5f, 300.0f, -3.0f, -0.6f, __builtin_nanf(""), 0.999f};
    float* d_in; uint32_t* d_out; uint32_t h_out[12];
    CHECK(hipMalloc(&d_in, sizeof h_in)); CHECK(hipMalloc(&d_out, sizeof h_out));
    CHECK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, d_in, d_out, 12);
    CHECK(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
    for (int i = 0; i < 12; ++i) printf("v_cvt_pk_u8_f32(%g) = %u\n", h_in[i], h_out[i]);
  }
  CHECK(hipFree(sink));
  return 0;
}
