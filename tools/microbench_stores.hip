// Store-only microbenchmark for the image writers' access pattern: 256 persistent workgroups of 16 wavefronts, each
// owning a contiguous range of image columns; per round a workgroup writes RUN bytes (64-byte aligned) of each of the
// image's ROWS rows, 16 bytes per lane.  Prints TB/s per run length: what the memory system gives this pattern when
// nothing else runs.  Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_stores.hip -o scratch/microbench_stores
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// lanes: RUN >= 1024: one row per instruction, RUN/1024 instructions per row; RUN == 512: two rows per instruction;
// RUN == 256: four rows per instruction.
template <int RUN>
__global__ __launch_bounds__(1024) void store_kernel(uint8_t* out, uint64_t width_px, int rows, uint64_t cols_per_wg, int rounds, int order) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t pitch = width_px * 4;
  uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  for (int r = 0; r < rounds; ++r) {
    // order 0: a workgroup's rounds are consecutive column blocks (contiguous range); 1: rounds dealt round-robin over the grid
    const uint64_t blk = order == 0 ? (uint64_t)blockIdx.x * rounds + r : (uint64_t)r * gridDim.x + blockIdx.x;
    const uint64_t col_byte = blk * RUN;
    if (col_byte + RUN > pitch) continue;
    constexpr int kRowsPerInst = RUN >= 1024 ? 1 : 1024 / RUN;
    constexpr int kInstPerRow = RUN >= 1024 ? RUN / 1024 : 1;
    for (int row0 = wave * kRowsPerInst; row0 < rows; row0 += 16 * kRowsPerInst) {
      const int row = row0 + (kRowsPerInst > 1 ? lane / (64 / kRowsPerInst) : 0);
      const int lane_in_row = kRowsPerInst > 1 ? lane % (64 / kRowsPerInst) : lane;
      if (row >= rows) continue;
      uint64_t addr = (uint64_t)row * pitch + col_byte;
      addr &= ~(uint64_t)63;  // whole sectors, as the image writers arrange
#pragma unroll
      for (int i = 0; i < kInstPerRow; ++i)
        *reinterpret_cast<uint4*>(out + addr + (uint64_t)i * 1024 + lane_in_row * 16) = v;
    }
  }
}

int main(int argc, char** argv) {
  // argv: [width_px] [rows]; the product width x rows x 4 bytes is what gets written
  const uint64_t width = argc > 1 ? strtoull(argv[1], nullptr, 10) : 5399996ull;
  const int rows = argc > 2 ? atoi(argv[2]) : 512;
  printf("{\"width_px\": %llu, \"rows\": %d, \"row_pitch_bytes\": %llu}\n", (unsigned long long)width, rows, (unsigned long long)width * 4);
  const uint64_t bytes = width * 4 * rows;
  uint8_t* out = nullptr;
  CHECK(hipMalloc(&out, bytes + 4096));
  CHECK(hipMemset(out, 0, bytes + 4096));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int grid = 256;
  for (int order = 0; order < 2; ++order)
    for (int run : {256, 512, 1024, 2048, 4096}) {
      const uint64_t blocks = width * 4 / run;
      const int rounds = (int)((blocks + grid - 1) / grid);
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        switch (run) {
          case 256: hipLaunchKernelGGL(store_kernel<256>, dim3(grid), dim3(1024), 0, 0, out, width, rows, 0, rounds, order); break;
          case 512: hipLaunchKernelGGL(store_kernel<512>, dim3(grid), dim3(1024), 0, 0, out, width, rows, 0, rounds, order); break;
          case 1024: hipLaunchKernelGGL(store_kernel<1024>, dim3(grid), dim3(1024), 0, 0, out, width, rows, 0, rounds, order); break;
          case 2048: hipLaunchKernelGGL(store_kernel<2048>, dim3(grid), dim3(1024), 0, 0, out, width, rows, 0, rounds, order); break;
          default: hipLaunchKernelGGL(store_kernel<4096>, dim3(grid), dim3(1024), 0, 0, out, width, rows, 0, rounds, order); break;
        }
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("{\"order\": \"%s\", \"run_bytes\": %d, \"ms\": %.3f, \"tb_per_s\": %.3f}\n", order ? "interleaved" : "contiguous", run, best,
             (double)bytes / (best * 1e-3) / 1e12);
      fflush(stdout);
    }
  // reference points: a plain fill of the same bytes
  {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      CHECK(hipMemsetAsync(out, 1, bytes, 0));
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("{\"order\": \"memset\", \"ms\": %.3f, \"tb_per_s\": %.3f}\n", best, (double)bytes / (best * 1e-3) / 1e12);
  }
  return 0;
}
