// The image side of the kernels that write RGBA themselves: a workgroup parks the colour indices of a round of RF
// (256 or 128) consecutive frames in a slot of a global scratch buffer that stays in cache, and writes the finished
// round out as pixel runs of 4*RF bytes per image row while it transforms the next one.
#ifndef SONO_RGBA_FLUSH_H
#define SONO_RGBA_FLUSH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sono_internal.h"

namespace sono {
namespace rgba_flush {

constexpr int kRoundFrames = 256;                  // frames per workgroup round (largest)
// slot: [reader unit of 32 bins][quarter of 8 bins][frame group RF/4][bin quad 2] x 16 bytes (four frames x four bins):
// what one wavefront reads for one quarter is one contiguous run (64 lanes x 16 bytes per 128 frames), so every cache
// line of the slot is fetched once -- with the bin quads of a unit side by side instead, a line served four quarters in
// four different iterations and came back from memory for each (2.1 x the slot's bytes in reads).  A workgroup owns two
// slots
constexpr int kRoundBytes = 128 * kRoundFrames * 4;
constexpr int kLutBytes = 1024;
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(4)));

// cell of the bin quad kq (bins 4kq .. 4kq+3) of frame group fg (frames 4fg .. 4fg+3 of the round), in uint4 units
template <int RF = 256>
__device__ __forceinline__ int slot_cell(int kq, int fg) {
  return (((kq >> 3) * 4 + ((kq & 7) >> 1)) * (RF / 4) + fg) * 2 + (kq & 1);
}

// Writes one quarter (`part` = 0..3) of a workgroup round (RF frames starting at frame `frame0`,
// indices in `slot`) to the RGBA image.  Wavefront w owns the 32 bins 32w .. 32w+31 (32 image rows) and
// writes the eight rows of bins 32w + 8*part .. +7 per quarter, complete: per step u lane j reads the
// bin quad 2*part + (j & 1) of the frame group 32u + (j >> 1), i.e. 4 bins x 4 frames (cell `lane` of the quarter's run), and stores four
// 16-byte pixel runs; 32 lanes form a 512-byte run of one row and the RF/128 steps complete its run of 4*RF bytes,
// so that sectors are whole when they leave L2 (up to the ends of the run, which the neighbouring
// workgroup completes).  Addresses are a wave-uniform base plus a 32-bit lane offset.
// first half of it: the two slot reads of lane `lane` for quarter `part`
template <int RF = 256>
__device__ __forceinline__ void flush_quarter_load(uint4 (&q)[2], const uint32_t* __restrict__ slot, int part, int wave, int lane) {
  asm volatile("" : "+s"(wave));
  asm volatile("" : "+v"(lane));
  const uint4* sp = reinterpret_cast<const uint4*>(slot) + (wave * 4 + part) * ((RF / 4) * 2) + lane;
#pragma unroll
  for (int u = 0; u < RF / 128; ++u) q[u] = sp[u * 64];
}
// second half: colours and stores
template <int RF = 256>
__device__ __forceinline__ void flush_quarter_store(const StftParams& p, const uint4 (&q)[2], const uint32_t* lut_lds,
                                                    uint64_t frame0, int part, int wave, int lane) {
  // Everything here is invariant in the caller's frame loop; computing it in place is cheap, while
  // hoisted copies would occupy registers that the FFT needs.
  asm volatile("" : "+s"(wave));
  asm volatile("" : "+v"(lane));
  const int qsel = lane & 1, fsel = lane >> 1;
  const int64_t width4 = (int64_t)p.width * 4;
  const uint32_t voff = (uint32_t)(4 - 4 * qsel) * (uint32_t)width4 + 16u * fsel;
  const bool whole = frame0 + RF <= p.nframes;  // workgroup-uniform
  const int bin0 = 32 * wave + 8 * part;                  // wave-uniform
  if (bin0 + 8 <= (int)p.bin_lo || bin0 >= (int)p.bin_hi) return;
  const int binq = bin0 + 4 * qsel;                       // this lane's rows are the bins binq .. binq + 3
#pragma unroll
  for (int u = 0; u < RF / 128; ++u) {
    const uint64_t fl = frame0 + 128 * u + 4 * fsel;  // this lane's first frame
    const int nvalid = fl >= p.nframes ? 0 : (p.nframes - fl >= 4 ? 4 : (int)(p.nframes - fl));
    // row of bin (bin0 + 4 + i), column of frame group 32u
    uint8_t* __restrict__ base = reinterpret_cast<uint8_t*>(p.out) + (int64_t)(p.col0 + frame0 + 128 * u) * 4 +
                                 (int64_t)(p.row_of_bin0 - bin0 - 4) * width4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4_unaligned px;
      px.x = lut_lds[(q[u].x >> (8 * i)) & 255u];
      px.y = lut_lds[(q[u].y >> (8 * i)) & 255u];
      px.z = lut_lds[(q[u].z >> (8 * i)) & 255u];
      px.w = lut_lds[(q[u].w >> (8 * i)) & 255u];
      uint8_t* dst = base - (int64_t)i * width4 + voff;
      if (binq + i < (int)p.bin_lo || binq + i >= (int)p.bin_hi) continue;  // a band may start or end inside a quad
      if (whole) {
        *reinterpret_cast<u32x4_unaligned*>(dst) = px;
      } else {  // the last round of the recording
        uint32_t* d = reinterpret_cast<uint32_t*>(dst);
        if (nvalid > 0) d[0] = px.x;
        if (nvalid > 1) d[1] = px.y;
        if (nvalid > 2) d[2] = px.z;
        if (nvalid > 3) d[3] = px.w;
      }
    }
  }
}

}  // namespace rgba_flush
}  // namespace sono
#endif
