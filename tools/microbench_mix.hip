// Memory-only model of the 1024-point image writer: 256 persistent workgroups of 16 wavefronts; per round of RF frames
// a workgroup reads its RF * 256 new float32 samples (linear) and writes RF * 4 bytes (whole 64-byte sectors) of each of
// the image's 512 rows, 16 bytes per lane.  Nothing is computed.  Variants: which rows a wavefront writes, how the
// rounds are dealt to the workgroups, reads on / off.  Prints ms and TB/s per variant: the floor the memory system sets
// for the kernel's access pattern.
// `microbench_mix floor [frames]` prints one line with the kernel's own pattern only (bench.py puts it beside the roofline).
// Built by spectrogram_amd/csrc/Makefile into tools/microbench_mix.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// rows_mode 0: wavefront w writes rows 2w + 32j (two rows per instruction, strided); 1: rows 32w .. 32w+31 (contiguous block)
// deal 0: contiguous range of rounds per workgroup; 1: round-robin over the grid; 2: round-robin inside an XCD (blocks with
// equal blockIdx % 8), each XCD a contiguous eighth of the image; 3: as 2, in chunks of four consecutive rounds per workgroup
// align 1: every row run starts on a 64-byte boundary; 0: runs start where the round starts in that row (rows are 48 bytes
// mod 64 apart, so three rows in four share their first and last sector with the neighbouring round)
template <int RF>
__global__ __launch_bounds__(1024) void mix_kernel(const float* __restrict__ pcm, uint8_t* __restrict__ out, float* sink, uint64_t width_px,
                                                   uint64_t nrounds, int rows_mode, int deal, int reads, int writes, int align,
                                                   int rows = 512, int hop = 256, uint4* slots = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t pitch = width_px * 4;
  constexpr int RUN = RF * 4;                 // bytes per row and round
  constexpr int kLanesPerRow = RUN / 16;      // 32 at RF = 128
  constexpr int kRowsPerInst = 64 / kLanesPerRow;
  const uint64_t per_wg = (nrounds + gridDim.x - 1) / gridDim.x;
  const uint32_t per = gridDim.x >> 3;
  const uint32_t vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);  // blocks of one XCD side by side
  float acc = 0.0f;
  uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  for (uint64_t r = 0; r < per_wg; ++r) {
    uint64_t round;
    if (deal == 0) round = (uint64_t)vb * per_wg + r;
    else if (deal == 1) round = r * gridDim.x + blockIdx.x;
    else if (deal == 2) round = (uint64_t)(blockIdx.x & 7) * per * per_wg + r * per + (blockIdx.x >> 3);
    else round = (uint64_t)(blockIdx.x & 7) * per * per_wg + ((r >> 2) * per + (blockIdx.x >> 3)) * 4 + (r & 3);
    if (round >= nrounds) continue;
    if (reads) {  // RF * 256 samples: 16 wavefronts x 64 lanes x 16 bytes = 16 KiB per pass
      const float4* src = reinterpret_cast<const float4*>(pcm + round * RF * hop);
      for (int i = threadIdx.x; i < RF * hop / 4; i += 1024) {
        const float4 t = src[i];
        acc += t.x + t.y + t.z + t.w;
      }
    }
    if (writes) {
      const uint64_t col_byte = round * RUN;
      if (slots) {  // the round's colour indices (rows x RF bytes) parked in this workgroup's slot and read back a round later
        uint4* slot = slots + (size_t)(2 * blockIdx.x + (r & 1)) * (rows * RF / 16);
        uint4* prev = slots + (size_t)(2 * blockIdx.x + ((r + 1) & 1)) * (rows * RF / 16);
        uint4 t = v;
        for (int i = threadIdx.x; i < rows * RF / 16; i += 1024) {
          if (r > 0) { const uint4 u = prev[i]; t.x ^= u.x; t.y += u.w; }
          slot[i] = v;
        }
        v.z = t.x + t.y;
      }
      for (int j = 0; j < rows / (16 * kRowsPerInst); ++j) {
        const int row = (rows_mode == 0 ? kRowsPerInst * wave + 16 * kRowsPerInst * j : 32 * wave + kRowsPerInst * j) + lane / kLanesPerRow;
        uint64_t addr = (uint64_t)row * pitch + col_byte;
        if (align) addr &= ~(uint64_t)63;
        *reinterpret_cast<uint4*>(out + addr + (lane % kLanesPerRow) * 16) = v;
      }
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const uint64_t width = argc > 2 ? strtoull(argv[2], nullptr, 10) : 5399996ull;  // frames; default 1 h at 384 kHz, hop 256
  if (width < 128 || width > (1ull << 26)) { fprintf(stderr, "frames out of range\n"); return 1; }
  // floor mode, optional: rows of the image (512; 1024 = the 2048-point writer; 256 = the 512-point writer), hop (256; 512;
  // 128), "slot" (or "-") = the round's colour indices go through a global scratch slot as in the 2048-point writer,
  // frames per round (128; 256)
  const int rows = argc > 3 ? atoi(argv[3]) : 512, hop = argc > 4 ? atoi(argv[4]) : 256;
  if ((rows != 256 && rows != 512 && rows != 1024) || (hop != 128 && hop != 256 && hop != 512)) { fprintf(stderr, "rows 256|512|1024, hop 128|256|512\n"); return 1; }
  const int rf = argc > 6 ? atoi(argv[6]) : 128;  // frames per round: 128; 256 = the 512-point writer
  if (rf != 128 && rf != 256) { fprintf(stderr, "round 128|256\n"); return 1; }
  const uint64_t img = width * 4 * rows, nsamp = width * hop + 4096;
  void* slots = nullptr;
  if (argc > 5 && std::string(argv[5]) == "slot") { CHECK(hipMalloc(&slots, (size_t)256 * 2 * rows * 128)); CHECK(hipMemset(slots, 0, (size_t)256 * 2 * rows * 128)); }
  uint8_t* out = nullptr;
  float *pcm = nullptr, *sink = nullptr;
  CHECK(hipMalloc(&out, img + 4096));
  CHECK(hipMalloc(&pcm, nsamp * 4 + 65536));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(out, 0, img + 4096));
  CHECK(hipMemset(pcm, 0, nsamp * 4 + 65536));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  if (argc > 1 && std::string(argv[1]) == "floor") {
    // one line for bench.py: the pattern the image writer uses (rounds of 128 frames, whole sectors, a contiguous range of
    // rounds per workgroup), writes alone, reads alone, both
    const uint64_t nrounds = (width + rf - 1) / rf;
    float ms3[3];
    for (int rw = 0; rw < 3; ++rw) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        if (rf == 128)
          hipLaunchKernelGGL(mix_kernel<128>, dim3(256), dim3(1024), 0, 0, pcm, out, sink, width, nrounds, 0, 0, rw != 0, rw != 1, 1, rows, hop,
                             (uint4*)(rw == 2 ? slots : nullptr));
        else
          hipLaunchKernelGGL(mix_kernel<256>, dim3(256), dim3(1024), 0, 0, pcm, out, sink, width, nrounds, 0, 0, rw != 0, rw != 1, 1, rows, hop,
                             (uint4*)(rw == 2 ? slots : nullptr));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      ms3[rw] = best;
    }
    printf("{\"frames\": %llu, \"rows\": %d, \"hop\": %d, \"round_frames\": %d, \"slot\": %s, \"image_bytes\": %llu, \"sample_bytes\": %llu, \"writes_ms\": %.4f, \"reads_ms\": %.4f, \"both_ms\": %.4f}\n",
           (unsigned long long)width, rows, hop, rf, slots ? "true" : "false", (unsigned long long)img, (unsigned long long)(nsamp * 4), ms3[0], ms3[1], ms3[2]);
    return 0;
  }
  printf("{\"image_gb\": %.3f, \"pcm_gb\": %.3f}\n", img / 1e9, nsamp * 4 / 1e9);
  for (int rf : {128, 256})
    for (int rw = 0; rw < 3; ++rw)             // 0: writes only, 1: reads only, 2: both
      for (int rows_mode = 0; rows_mode < 2; ++rows_mode)
        for (int deal = 0; deal < 4; ++deal)
         for (int align = 1; align >= 0; --align) {
          if (rw == 1 && (rows_mode != 0 || align == 0)) continue;
          if (rw != 1 && rows_mode != 0 && align == 0) continue;
          const uint64_t nrounds = (width + rf - 1) / rf;
          const int reads = rw != 0, writes = rw != 1;
          float best = 1e30f;
          for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            if (rf == 128) hipLaunchKernelGGL(mix_kernel<128>, dim3(256), dim3(1024), 0, 0, pcm, out, sink, width, nrounds, rows_mode, deal, reads, writes, align);
            else hipLaunchKernelGGL(mix_kernel<256>, dim3(256), dim3(1024), 0, 0, pcm, out, sink, width, nrounds, rows_mode, deal, reads, writes, align);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
          }
          const double bytes = (writes ? (double)img : 0.0) + (reads ? (double)nsamp * 4 : 0.0);
          printf("{\"round_frames\": %d, \"traffic\": \"%s\", \"rows\": \"%s\", \"deal\": \"%s\", \"runs\": \"%s\", \"ms\": %.3f, \"tb_per_s\": %.3f}\n", rf,
                 rw == 0 ? "writes" : rw == 1 ? "reads" : "reads+writes", rows_mode ? "block of 32 per wavefront" : "strided",
                 deal == 0 ? "contiguous range" : deal == 1 ? "round-robin over the grid" : deal == 2 ? "round-robin inside an XCD" : "chunks of 4 round-robin inside an XCD",
                 align ? "whole sectors" : "unaligned", best,
                 bytes / (best * 1e-3) / 1e12);
          fflush(stdout);
        }
  return 0;
}
