// 1024-point float32 STFT that writes the RGBA image itself, with the workgroup's exchange of colour indices
// kept in LDS.
//
// A frame is one image COLUMN and image rows are tens of megabytes apart, so frames cannot be stored as they are
// finished: a workgroup collects the colour indices of a round of 128 consecutive frames (8 per wavefront, two
// iterations of 4) and then every wavefront writes 32 image rows of the round as 512-byte runs.  That exchange is
// a transpose between "wavefront w holds frames 4w..4w+3 and 64+4w..64+4w+3, all bins" and "wavefront w writes bins
// 32w..32w+31, all 128 frames": 64 KiB per round.  Passing it through a global scratch buffer costs a quarter of everything the
// memory fabric moves for this kernel (the lines are evicted from L2 between their write and their read), so it
// lives in LDS here, and the rest of the kernel is arranged to leave room for it:
//   * the 32 x 32 transposes between the two register passes are split: register bit 4 <-> lane bit 4 by
//     v_permlane16_swap (no memory), the remaining 16 x 16 transposes of each 16-lane row through a 1 KiB block
//     (16-byte chunks XOR-swizzled by row: stores and row reads both conflict-free), one half of the registers after
//     the other: 4.25 KiB per wavefront instead of 9;
//   * the epilogue sends only the partners Z[N-k] through LDS, in two rounds of eight registers (2 KiB per transform);
//     Z[k] stays in its lane, and a lane ends up with one dword (four frames) per bin, which is what a ring row takes.
// LDS: 8 KiB twiddle matrix + 4.5 KiB window + 1 KiB colour map + 16 x 4.25 KiB + 74 KiB slot = 155.5 KiB.
//
// Whole 64-byte sectors only: the slot is the ring of sono_image_ring.h (a workgroup owns a contiguous range of frames,
// a row's run starts e = 0..15 frames before the round so that it starts on a sector boundary).
//
// One slot, two barriers per round, both between iterations.  Iteration 0 of round k+1: write out quarters 0, 1 of
// round k (read at the end of round k), transform, keep the epilogue's indices in registers, read quarters 2, 3 of
// round k.  Iteration 1: BARRIER (every wavefront has taken what it needs of round k out of the slot), write out
// quarters 2, 3, transform, store the kept indices and this epilogue's into the slot, BARRIER (slot complete), read
// quarters 0, 1 of round k+1.  After its last round a workgroup runs two more, empty ones: they write out the last
// round and the e frames per row that it left behind.
#include <hip/hip_runtime.h>

#include "sono_device.h"
#include "sono_fft32.h"
#include "sono_image_ring.h"
#include "sono_internal.h"
#include "sono_launch.h"
#include "sono_loud.h"

namespace sono {
namespace fast1024r {

using namespace fast1024;
using namespace image_ring;

constexpr int kWaves = 16;
constexpr int kRoundFrames = 128;
constexpr int kTileFrames = kRoundFrames / kWaves;  // frames per wavefront and round
constexpr int kTileIters = kTileFrames / 4;         // 2
constexpr int kTwBytes = 32 * 32 * 8;
constexpr int kWinPitch = 36;
constexpr int kWinBytes = 32 * kWinPitch * 4;
constexpr int kLutBytes = 1024;
constexpr int kWaveBytes = 4 * kBlkFloats * 4;       // 4352: four 16-lane rows
constexpr int kQuarterBytes = kWaveBytes / 2;        // epilogue, per transform: eight rows of 32 partner values (2 KiB)
static_assert(8 * 32 * 8 <= kQuarterBytes, "an epilogue round must fit half the wavefront's slice");
constexpr int kSlotBytes = Ring<kRoundFrames>::bytes(512);
constexpr int kOffWin = kTwBytes, kOffLut = kOffWin + kWinBytes, kOffWave = kOffLut + kLutBytes, kOffSlot = kOffWave + kWaves * kWaveBytes;
constexpr int kLdsBytes = kOffSlot + kSlotBytes;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

// Image rows of one quarter (`part` 0..3) of the round that starts at frame `fround`: wavefront `wave` owns bins
// 32*wave .. +31 and writes the eight rows of bins 32*wave + 8*part .. +7; lane j stores four pixels (16 bytes) of the rows
// 4*(j & 1) + i, i = 0..3, of the quarter: 32 lanes form the 512-byte run of a row.
__device__ __forceinline__ void flush_rows(const StftParams& p, const Range& rg, const uint32_t (&d)[4], const uint32_t* lut_lds,
                                           uint64_t fround, int part, int wave, int lane) {
  asm volatile("" : "+s"(wave));   // (everything below is loop-invariant; computed in place it costs a few instructions,
  asm volatile("" : "+v"(lane));   //  hoisted it would hold registers the transforms need)
  const int bin0 = 32 * wave + 8 * part;
  if (bin0 + 8 <= (int)p.bin_lo || bin0 >= (int)p.bin_hi) return;
  flush4<kRoundFrames>(p, rg, d, lut_lds, fround, bin0 + 4 * (lane & 1), lane >> 1, bin0 >= (int)p.bin_lo && bin0 + 8 <= (int)p.bin_hi);
}

// HOP32 > 0: hop == 32*HOP32 and one run of 32 + HOP32 loads feeds both frames of a pair; 0: any hop, two runs of 32.
// TABLE: twiddles are the context's per-stage table (the plugin's recurrence factors or a host table) instead of exact
// roots of unity; see stft1024_u8_kernel for how the two flavours read their factors.
template <bool PLUGIN, typename In, int HOP32, bool TABLE>
__global__ __launch_bounds__(kWaves * 64, 4) void stft1024_rgba_kernel(StftParams p, FastScale fs, const float2* __restrict__ tw_table) {
  constexpr int kThreads = kWaves * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw_lds = reinterpret_cast<float2*>(smem);
  float* win_lds = reinterpret_cast<float*>(smem + kOffWin);
  uint32_t* lut_lds = reinterpret_cast<uint32_t*>(smem + kOffLut);
  uint32_t* slot = reinterpret_cast<uint32_t*>(smem + kOffSlot);
  const float2* __restrict__ tw_flat = reinterpret_cast<const float2*>(p.twiddles);
  for (int i = threadIdx.x; i < 1024; i += kThreads) {
    if constexpr (TABLE) {
      const int r = i >> 5, lane_p = i & 31;  // row half - 1 + j of the second pass: stage T = 32*half, factor 32*j + lane
      if (r < 31) {
        const int half = 1 << (31 - __builtin_clz(r + 1)), j = r + 1 - half;
        tw_lds[i] = tw_flat[32 * half - 1 + 32 * j + lane_p];
      }  // (the 31 factors of the first pass are read by scalar loads)
    } else {
      tw_lds[i] = tw_table[i];
    }
    win_lds[(i & 31) * kWinPitch + (i >> 5)] = p.window[i];  // sample n = 32a + l at [l][a]
  }
  for (int i = threadIdx.x; i < 256; i += kThreads) lut_lds[i] = p.lut[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, l = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* wbuf = smem + kOffWave + wave * kWaveBytes;
  float* blk = reinterpret_cast<float*>(wbuf) + (lane >> 4) * kBlkFloats;
  const int ll = lane & 15;
  unsigned char* qbuf = wbuf + h * kQuarterBytes;
  const float4* win_row = reinterpret_cast<const float4*>(win_lds + l * kWinPitch);

  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  const uint64_t npairs = (p.nframes + 1) >> 1;
  // blocks with equal blockIdx % 8 share an XCD (and its L2): give them neighbouring frame ranges
  const uint32_t per = gridDim.x >> 3;
  const uint32_t vb = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  // every workgroup runs the same number of rounds and every wavefront goes through every round (with or without
  // frames of its own), because the rounds are fenced by workgroup barriers; a workgroup's rounds are consecutive frames
  const uint64_t nrounds = (p.nframes + kRoundFrames - 1) / kRoundFrames;
  const uint64_t rounds = (nrounds + gridDim.x - 1) / gridDim.x;
  const Range rg = make_range<kRoundFrames>(p, vb, rounds);

  struct Unit {
    uint64_t f0;
    bool active, st_a, st_b, shifted;
  };
  // iteration `it` of this wavefront: four consecutive frames (two per transform).  Frames (f0, f0+1) form the complex
  // input; an odd last frame is computed as the second member of the pair before it (shifted).
  auto locate = [&](uint64_t it) -> Unit {
    Unit u;
    // frame group (4 frames) 16 * sub + wave of the round: the 16 wavefronts of an iteration read one contiguous span
    // of samples, so the 3/4 overlap between neighbouring groups is re-read while it is still in cache
    const uint64_t group = ((uint64_t)vb * rounds * kTileIters + it) * kWaves + wave;
    const uint64_t pair0 = group * 2;
    u.active = it < rounds * kTileIters && pair0 < npairs;  // wave-uniform
    if (!u.active) {            // nothing left: point at frame 0 so that the sample loads stay in bounds
      u.f0 = 0;
      u.st_a = u.st_b = u.shifted = false;
      return u;
    }
    const uint64_t pair = pair0 + h;
    u.f0 = 2 * pair;
    u.st_a = u.st_b = pair < npairs;
    u.shifted = false;
    if (!u.st_a) {  // this half has no pair of its own: it repeats valid frames; its indices are never written out
      u.f0 = 2 * pair0 + 1 < p.nframes ? 2 * pair0 : 2 * pair0 - 1;
    } else if (u.f0 + 1 >= p.nframes) {
      u.f0 -= 1;
      u.st_a = false;
      u.shifted = true;
    }
    return u;
  };
  constexpr int kRaw = HOP32 > 0 ? 32 + HOP32 : 64;
  float raw[kRaw];
  auto fetch = [&](const Unit& u) {
    const In* xa = x + (p.frame0 + u.f0) * p.hop + l;
    if constexpr (HOP32 > 0) {
#pragma unroll
      for (int a = 0; a < 32 + HOP32; ++a) raw[a] = ld<In>(xa, 32 * a);
    } else {
      const In* xb = xa + p.hop;
#pragma unroll
      for (int a = 0; a < 32; ++a) {
        raw[a] = ld<In>(xa, 32 * a);
        raw[32 + a] = ld<In>(xb, 32 * a);
      }
    }
  };
  // this lane's four frames of each of its rows, for the two quarters the next iteration writes out of round k (of this
  // workgroup): row 4*(j & 1) + i of the quarter
  auto read_rows = [&](uint32_t (&fq)[2][4], int first_part, uint64_t k) {
    int w = wave, j = lane;
    asm volatile("" : "+s"(w));
    asm volatile("" : "+v"(j));
#pragma unroll
    for (int q = 0; q < 2; ++q) read4<kRoundFrames>(slot, p, rg, fq[q], 32 * w + 8 * (first_part + q) + 4 * (j & 1), j >> 1, k);
  };

  Unit cur = locate(0);
  fetch(cur);
  uint32_t fq[2][4] = {};     // indices of the two quarters the next iteration writes out
  uint32_t pend[8] = {};      // dwords of iteration 0's epilogue, stored into the slot in iteration 1
  // Frame group fg (four frames) of round k of this workgroup: one dword into each of the rows of the bins
  // bin0 + 64*m, m = 0..7 (whole groups of 64 rows outside the band are skipped; the rest of the slot is never read).
  auto park8 = [&](const uint32_t (&r)[8], uint64_t k, int fg, int bin0) {
    constexpr int kD = Ring<kRoundFrames>::kDwords, kP = Ring<kRoundFrames>::kPitch;
    int d = (int)(((uint64_t)(kRoundFrames / 4) * k) % kD) + fg;
    d -= d >= kD ? kD : 0;
    uint32_t* row = slot + bin0 * kP + d;
#pragma unroll
    for (int m = 0; m < 8; ++m)
      if (64 * m + 63 >= (int)p.bin_lo && 64 * m < (int)p.bin_hi) row[m * 64 * kP] = r[m];
  };
  auto park8_or = [&](const uint32_t (&r)[8], uint64_t k, int fg, int bin0) {  // ... OR-ed onto the dwords there
    constexpr int kD = Ring<kRoundFrames>::kDwords, kP = Ring<kRoundFrames>::kPitch;
    int d = (int)(((uint64_t)(kRoundFrames / 4) * k) % kD) + fg;
    d -= d >= kD ? kD : 0;
    uint32_t* row = slot + bin0 * kP + d;
#pragma unroll
    for (int m = 0; m < 8; ++m)
      if (64 * m + 63 >= (int)p.bin_lo && 64 * m < (int)p.bin_hi) row[m * 64 * kP] |= r[m];  // (this lane's own dwords)
  };
  bool pend_active = false;
  // Under the context's table a unit too loud for the pair separation (StftParams::loud_thr_raw) is transformed one frame
  // at a time: the iteration's transform runs twice, mode 1 = frame A alone, then mode 2 = frame B alone (whose bytes are
  // OR-ed onto frame A's).  Wave-uniform: the four frames of a wavefront go together (its halves run in lockstep anyway).
  // A NaN sample does not compare as loud and needs no second look (it poisons every bin either way).
  int mode = 0;
  for (uint64_t it = 0; it < (rounds + 2) * kTileIters;) {
    const uint64_t round = it / kTileIters;
    const int sub = (int)(it % kTileIters);
    const Unit nxt = locate(it + 1);
    if (!(TABLE && mode == 2)) {
      // iteration 1: every wavefront has read its quarters 2, 3 of the previous round (end of its iteration 0), so the
      // slot may be overwritten from here on
      if (sub == 1) __syncthreads();
      // (the empty asm is a use of the sample registers: the wait for their loads lands here, before the image stores
      // of this iteration, and not behind them)
#pragma unroll
      for (int a = 0; a < kRaw; ++a) asm volatile("" : "+v"(raw[a]));
      if (round > 0 && rg.f_lo < rg.f_hi) {
        const uint64_t fround = rg.f_lo + (round - 1) * kRoundFrames;
        flush_rows(p, rg, fq[0], lut_lds, fround, 2 * sub, wave, lane);
        flush_rows(p, rg, fq[1], lut_lds, fround, 2 * sub + 1, wave, lane);
      }
    }

    if (cur.active) {
      if constexpr (TABLE) {
        if (mode == 0 && p.loud_thr_raw > 0.0f && __builtin_amdgcn_ballot_w64(max_abs(raw) > p.loud_thr_raw) != 0) mode = 1;
      }
      const bool shifted = cur.shifted;
      int lq = l;
      asm volatile("" : "+v"(lq));
      uint32_t w[16];
      float re[32], im[32];
      {
        float win[32];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 v = win_row[j];
          win[4 * j] = v.x; win[4 * j + 1] = v.y; win[4 * j + 2] = v.z; win[4 * j + 3] = v.w;
        }
#pragma unroll
        for (int a = 0; a < 32; ++a) {
          re[brev5(a)] = raw[a] * win[a];
          im[brev5(a)] = raw[a + (HOP32 > 0 ? HOP32 : 32)] * win[a];
        }
        if (TABLE && mode != 0) {
#pragma unroll
          for (int a = 0; a < 32; ++a) im[a] = 0.0f;
        }
      }
      if constexpr (TABLE) {
        // stages T = 1..16: wave-uniform factors by scalar loads (see stft1024_u8_kernel)
        typedef const float __attribute__((address_space(4))) * const_table;
        const_table twc = (const_table)(uintptr_t)tw_flat;
        asm volatile("" : "+s"(twc));
        fft32_table<true>(re, im, [&](int half, int j) { return make_float2(twc[2 * (half - 1 + j)], twc[2 * (half - 1 + j) + 1]); });
      } else {
        fft32(re, im);  // Y_l[k1], then the twiddle W_1024^(l*k1)
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
          const float2 t = tw_lds[k1 * 32 + l];
          const float ur = __builtin_fmaf(re[k1], t.x, -im[k1] * t.y);
          const float ui = __builtin_fmaf(re[k1], t.y, im[k1] * t.x);
          re[k1] = ur;
          im[k1] = ui;
        }
      }
      transpose32_plane(re, blk, ll);
      transpose32_plane(im, blk, ll);
      if constexpr (TABLE) {
        int lt = l;
        asm volatile("" : "+v"(lt));  // (keeps 31 loop-invariant LDS addresses out of registers)
        fft32_table<false>(re, im, [&](int half, int j) { return tw_lds[(half - 1 + j) * 32 + lt]; });  // stages T = 32..512
      } else {
        fft32(re, im);  // X[l + 32*k2]
      }

      // Epilogue.  Bin k = l + 32*k2 (k2 < 16) stays in its lane; its partner Z[N - k] sits in lane 32 - l, register
      // 31 - k2 (lane 0: its own register 32 - k2).  Only the partners travel: two rounds of eight registers (16..23, then
      // 24..31) through the half-wavefront's 2-KiB buffer, register r0 + j of lane l at [j][l].  Lane 0 stores one row
      // down and puts register 24 (round 0) / 0 (round 1) into row 7, so that every lane reads [j][(32 - l) mod 32].
      // Byte 0 / 1 of w[k2]: the colour indices of the transform's frames A / B at that bin.
      const bool lane0 = lq == 0;
      unsigned char* wr = qbuf + 8 * lq - (lane0 ? 256 : 0);
      unsigned char* wr0 = lane0 ? qbuf + 7 * 256 : qbuf + 8 * lq;
      const unsigned char* rd = qbuf + 8 * ((32 - lq) & 31);
      if (TABLE && mode != 0) {
        // a frame alone: its spectrum is Z itself (no partner; the same factor 2 as the pair separation leaves); the
        // other frame's byte stays 0 in this pass
        if (mode == 1) {
#pragma unroll
          for (int k2 = 0; k2 < 16; ++k2) {
            const float ar = re[k2] + re[k2], ai = im[k2] + im[k2];
            w[k2] = put_index<PLUGIN, 0>(__builtin_fmaf(ar, ar, ai * ai), fs, 0u);
          }
        } else {
#pragma unroll
          for (int k2 = 0; k2 < 16; ++k2) {
            const float ar = re[k2] + re[k2], ai = im[k2] + im[k2];
            w[k2] = put_index<PLUGIN, 1>(__builtin_fmaf(ar, ar, ai * ai), fs, 0u);
          }
        }
      } else {
#pragma unroll
      for (int R = 0; R < 2; ++R) {
        const int r0 = 16 + 8 * R, rx = (24 + 8 * R) & 31;
        *reinterpret_cast<float2*>(wr0) = lane0 ? make_float2(re[rx], im[rx]) : make_float2(re[r0], im[r0]);
#pragma unroll
        for (int j = 1; j < 8; ++j) *reinterpret_cast<float2*>(wr + 256 * j) = make_float2(re[r0 + j], im[r0 + j]);
        wave_sync();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k2 = 15 - 8 * R - j;
          const float2 q = *reinterpret_cast<const float2*>(rd + 256 * j);
          const float zr = re[k2], zi = im[k2];
          const float ar = zr + q.x, ai = zi - q.y;
          const float br = zi + q.y, bi = q.x - zr;
          const uint32_t t = put_index<PLUGIN, 0>(__builtin_fmaf(ar, ar, ai * ai), fs, 0u);
          w[k2] = put_index<PLUGIN, 1>(__builtin_fmaf(br, br, bi * bi), fs, t);
        }
        wave_sync();  // the buffer is rewritten by the next round / the next iteration's transpose
      }
      }
      // an odd last frame was computed as B: it belongs in the A byte (one half-wavefront per launch)
      if (__builtin_amdgcn_ballot_w64(shifted) != 0) {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) w[k2] = shifted ? w[k2] >> 8 : w[k2];
      }
      // lanes of transform h hold frames (2h, 2h+1) of the iteration; after the swap lane (l, h) holds all four frames of
      // the bins l + 32*(2m + h), m = 0..7, as one dword per bin: what a ring row takes
      uint32_t r8[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const auto sw = __builtin_amdgcn_permlane32_swap(w[2 * m], w[2 * m + 1], false, false);
        r8[m] = sw[0] | (sw[1] << 16);
      }
      if (sub == 0) {
#pragma unroll
        for (int m = 0; m < 8; ++m) pend[m] = (TABLE && mode == 2) ? pend[m] | r8[m] : r8[m];
      } else if (TABLE && mode == 2) {
        park8_or(r8, round, kWaves + wave, lq + 32 * h);  // (onto what the pass for frame A left there)
      } else {
        park8(r8, round, kWaves + wave, lq + 32 * h);
      }
    }
    if constexpr (TABLE) {
      if (mode == 1) {
        // frame A alone is done: the same unit once more for frame B alone -- its samples (from the cache) and nothing
        // behind them, the frame after it may lie beyond the signal
        const In* xb1 = x + (p.frame0 + cur.f0 + 1) * p.hop + l;
        asm volatile("" : "+v"(xb1));  // (opaque: or the samples already loaded for the pair are kept in registers instead)
#pragma unroll
        for (int a = 0; a < 32; ++a) raw[a] = ld<In>(xb1, 32 * a);
#pragma unroll
        for (int a = 32; a < kRaw; ++a) raw[a] = 0.0f;
        mode = 2;
        continue;
      }
      mode = 0;
    }
    if (sub == 0) {
      pend_active = cur.active;
    } else if (pend_active) {  // frame group `wave` of the round: the dwords iteration 0 kept
      park8(pend, round, wave, l + 32 * h);
    }
    fetch(nxt);  // unconditional (an inactive unit points at frame 0): no stale values carried round the loop
    cur = nxt;

    if (sub == 0) {
      // the slot still holds the previous round: take the last two quarters of this wavefront's 32 rows
      if (round > 0) read_rows(fq, 2, round - 1);
    } else {
      // the round's slot is complete once every wavefront is here; then take the first two quarters (after the last
      // round: of the empty round that only has the frames the last one left behind)
      __syncthreads();
      if (round <= rounds) read_rows(fq, 0, round);
    }
    ++it;
  }
}

}  // namespace fast1024r

// The image writer above serves what fast_rgba_available admits (1024 points, any band, the flipped row mapping, a
// colour table); it needs no scratch buffer.
int launch_stft_fast_rgba(const StftParams& p, void* stream_v) {
  using namespace fast1024r;
  if (p.out_kind != SONO_OUT_RGBA8 || !fast_rgba_available(p)) return 1;
  if (reinterpret_cast<uintptr_t>(p.out) & 3) return 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  FastScale fs = make_fast_scale(p.scale_mode, p.n, p.gain_db, p.range_db);
  fs.b -= 2.0f * fs.a;  // the pair separation leaves 2*X: |2X|^2 = 4|X|^2, i.e. log2 is 2 too large
  const uint64_t need = (p.nframes + kRoundFrames - 1) / kRoundFrames;
  const int cus = device_cu_count();
  const uint32_t grid = (uint32_t)(need < (uint64_t)cus ? need : (uint64_t)cus);  // persistent: one workgroup per CU
  const float2* table = reinterpret_cast<const float2*>(p.fast_table);
  const bool tab = p.fast_mode == 2;
#define SONO_LAUNCH_T(PLUGIN, In, H, T)                                                         \
  do {                                                                                          \
    auto kern = stft1024_rgba_kernel<PLUGIN, In, H, T>;                                         \
    SONO_GRANT_LDS(kern, kLdsBytes);                                                            \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), kLdsBytes, stream, p, fs, table);   \
  } while (0)
#define SONO_LAUNCH_HOP(PLUGIN, In, H)                                                          \
  do {                                                                                          \
    if (tab) SONO_LAUNCH_T(PLUGIN, In, H, true); else SONO_LAUNCH_T(PLUGIN, In, H, false);      \
  } while (0)
#define SONO_LAUNCH(PLUGIN, In)                                                                 \
  do {                                                                                          \
    if (p.hop == 256) SONO_LAUNCH_HOP(PLUGIN, In, 8);                                           \
    else if (p.hop == 512) SONO_LAUNCH_HOP(PLUGIN, In, 16);                                     \
    else SONO_LAUNCH_HOP(PLUGIN, In, 0);                                                        \
  } while (0)
  const bool plugin = p.scale_mode == SONO_SCALE_PLUGIN_DB;
  if (p.in_kind == SONO_IN_I16) {
    if (plugin) SONO_LAUNCH(true, int16_t); else SONO_LAUNCH(false, int16_t);
  } else {
    if (plugin) SONO_LAUNCH(true, float); else SONO_LAUNCH(false, float);
  }
#undef SONO_LAUNCH
#undef SONO_LAUNCH_HOP
#undef SONO_LAUNCH_T
  note_kernel(tab ? (p.loud_thr_raw > 0.0f ? "stft1024_rgba_kernel<table,redo>" : "stft1024_rgba_kernel<table>") : "stft1024_rgba_kernel<roots>");
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

}  // namespace sono
