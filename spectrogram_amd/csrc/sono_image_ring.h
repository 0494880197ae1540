// The image side of the transform kernels that write the RGBA image themselves with the workgroup's exchange of colour
// indices in LDS (kernels_fast_rgba.hip: 1024 points, kernels_fast512.hip: 512 points).
//
// A frame is one image column and image rows are megabytes apart, so a workgroup collects the colour indices of a round
// of RF consecutive frames and then writes every bin row of the round as one run of 4*RF bytes.  The image's row pitch
// (4 x width bytes) is no multiple of 64 bytes, so a run cut at a round boundary would straddle a 64-byte sector at
// either end, and the two halves of such a sector would arrive from different rounds at different times: the same image
// takes 4.2 ms to store that way and 2.9 ms with the runs cut at sector boundaries (1 h of audio at FFT 1024, stores
// alone).  So a workgroup owns a CONTIGUOUS range of frames, and the run of a row in a round starts e = 0..15 frames
// before the round, e chosen per row so that the run starts on a sector boundary; the last e indices of the round before
// are still in the slot.  The slot is one ring of RF + 16 frames per bin row (RF/4 + 4 dwords of four frames each,
// byte = frame): round k occupies the dwords (RF/4)*k .. +RF/4-1 (mod ring), which leaves the last 16 frames of round
// k-1 in place.  A writing lane reads two neighbouring dwords of its row and v_alignbyte_b32 cuts its four frames out
// of them.  Partial sectors remain only where the ranges of two workgroups meet (each masks the frames that are not its
// own) and at the ends of the recording.
#ifndef SONO_IMAGE_RING_H
#define SONO_IMAGE_RING_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sono_internal.h"

namespace sono {
namespace image_ring {

typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(4)));

template <int RF>
struct Ring {
  static constexpr int kDwords = RF / 4 + 4;   // per bin row
  static constexpr int kPitch = kDwords + 1;   // dwords per bin row (odd: the stores of a bin quad's four rows spread over banks)
  static constexpr int bytes(int bins) { return bins * kPitch * 4; }
};

// What the image writer needs to know about the workgroup's range of frames.
struct Range {
  uint64_t f_lo, f_hi;  // frames [f_lo, f_hi) are this workgroup's
  uint32_t align_px;    // (address of pixel (row 0, frame f_lo) / 4) mod 16
};

// Workgroup `vb` of a grid whose workgroups each take `rounds` consecutive rounds of RF frames.
template <int RF>
__device__ __forceinline__ Range make_range(const StftParams& p, uint32_t vb, uint64_t rounds) {
  Range rg;
  rg.f_lo = (uint64_t)vb * rounds * RF;
  rg.f_hi = rg.f_lo + rounds * RF < p.nframes ? rg.f_lo + rounds * RF : p.nframes;
  if (rg.f_hi < rg.f_lo) rg.f_hi = rg.f_lo;
  rg.align_px = (uint32_t)((reinterpret_cast<uintptr_t>(p.out) >> 2) + p.col0 + rg.f_lo) & 15u;
  return rg;
}

// The run of bin row `bin` starts e = 0..15 frames before a round, so that it starts on a 64-byte boundary.
__device__ __forceinline__ uint32_t row_lead(const StftParams& p, const Range& rg, int bin) {
  const uint32_t row = (uint32_t)(p.row_of_bin0 - bin);
  return (rg.align_px + (row & 15u) * ((uint32_t)p.width & 15u)) & 15u;
}

// The indices of four frames x four bins as an epilogue leaves them (word = frame, byte = bin) turned into word = bin,
// byte = frame: what the ring rows hold.
__device__ __forceinline__ void turn(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t (&r)[4]) {
  const uint32_t t0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), t1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
  const uint32_t t2 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), t3 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
  r[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
  r[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
  r[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
  r[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// Frame group fg (four frames, 0 .. RF/4-1) of round k (of this workgroup), bin quad kq: one dword into each of the
// quad's four rows.
template <int RF>
__device__ __forceinline__ void park(uint32_t* slot, const StftParams& p, const uint32_t (&r)[4], uint64_t k, int fg, int kq) {
  if (4 * kq + 3 < (int)p.bin_lo || 4 * kq >= (int)p.bin_hi) return;
  int d = (int)(((uint64_t)(RF / 4) * k) % Ring<RF>::kDwords) + fg;
  d -= d >= Ring<RF>::kDwords ? Ring<RF>::kDwords : 0;
  uint32_t* row = slot + 4 * kq * Ring<RF>::kPitch + d;
#pragma unroll
  for (int b = 0; b < 4; ++b) row[b * Ring<RF>::kPitch] = r[b];
}

// This lane's four frames of the rows of bins bin0 .. bin0+3 of round k: frames RF*k - e + 4*fsel .. +3 (byte t = frame + t),
// cut out of two neighbouring ring dwords.
template <int RF>
__device__ __forceinline__ void read4(const uint32_t* slot, const StftParams& p, const Range& rg, uint32_t (&d)[4], int bin0, int fsel,
                                      uint64_t k) {
  constexpr int kD = Ring<RF>::kDwords;
  const int ringpos = (int)(((uint64_t)(RF / 4) * k) % kD);  // first dword of round k
  const uint32_t wstep = (uint32_t)p.width & 15u;
  uint32_t e = row_lead(p, rg, bin0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int d0 = ringpos + fsel - (int)((e + 3) >> 2);  // -4 .. 2*kD - 6
    const uint32_t shift = (0u - e) & 3u;
    e = (e - wstep) & 15u;  // the next bin is one row up
    d0 += d0 < 0 ? kD : 0;
    d0 -= d0 >= kD ? kD : 0;
    int d1 = d0 + 1;
    d1 -= d1 >= kD ? kD : 0;
    const uint32_t* row = slot + (bin0 + i) * Ring<RF>::kPitch;
    d[i] = __builtin_amdgcn_alignbyte(row[d1], row[d0], shift);
  }
}

// Stores this lane's four pixels (16 bytes) of each of the rows of bins bin0 .. bin0+3 of the round that starts at frame
// `fround`: frames fround - e + 4*fsel .. +3; d[i] from read4.  `group_plain`: wave-uniform, every row this instruction
// group writes is in the band.  Where a round is inside the workgroup's range the stores are unconditional; at the ends
// of the range every lane masks the frames that are not the workgroup's.
template <int RF>
__device__ __forceinline__ void flush4(const StftParams& p, const Range& rg, const uint32_t (&d)[4], const uint32_t* lut_lds,
                                       uint64_t fround, int bin0, int fsel, bool group_plain) {
  const int64_t width4 = (int64_t)p.width * 4;
  uint8_t* __restrict__ base = reinterpret_cast<uint8_t*>(p.out) + (int64_t)(p.col0 + fround) * 4 +
                               (int64_t)(p.row_of_bin0 - bin0) * width4 + 16 * fsel;
  const uint32_t wstep = (uint32_t)p.width & 15u;  // e of consecutive rows differs by the width (mod 16)
  uint32_t e = row_lead(p, rg, bin0);
  if (group_plain && fround >= rg.f_lo + 16 && fround + RF <= rg.f_hi) {  // wave-uniform
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t v = d[i];
      u32x4_unaligned px;
      px.x = lut_lds[v & 255u];
      px.y = lut_lds[(v >> 8) & 255u];
      px.z = lut_lds[(v >> 16) & 255u];
      px.w = lut_lds[v >> 24];
      *reinterpret_cast<u32x4_unaligned*>(base - (int64_t)i * width4 - 4 * (int)e) = px;
      e = (e - wstep) & 15u;  // the next bin is one row up
    }
    return;
  }
  // (frames relative to the start of the range: an image is at most 2^24 columns wide)
  const int rel0 = (int)(fround - rg.f_lo) + 4 * fsel, span = (int)(rg.f_hi - rg.f_lo);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t ei = e;
    e = (e - wstep) & 15u;
    if (bin0 + i < (int)p.bin_lo || bin0 + i >= (int)p.bin_hi) continue;
    const uint32_t v = d[i];
    const uint32_t px[4] = {lut_lds[v & 255u], lut_lds[(v >> 8) & 255u], lut_lds[(v >> 16) & 255u], lut_lds[v >> 24]};
    uint32_t* o = reinterpret_cast<uint32_t*>(base - (int64_t)i * width4 - 4 * (int)ei);
    const int f = rel0 - (int)ei;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (f + t >= 0 && f + t < span) o[t] = px[t];
  }
}

}  // namespace image_ring
}  // namespace sono
#endif
