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
//   * the epilogue pairs Z[k] with Z[N-k] in quarters of 128 bins (2 KiB per transform).
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
constexpr int kQuarterBytes = kWaveBytes / 2;        // epilogue, per transform: 128 values Z[k], then 128 partners Z[N-k]
// 128 complex values laid out for the epilogue: value j (8 bytes) of a quarter sits in plane (j >> 1) & 1 at
// 16 * (j >> 2) + 8 * (j & 1), so that a lane's four consecutive values are one 16-byte read per plane at 16 * lane
// (conflict-free), and the second plane starts 16 banks after the first so that the 8-byte stores of 16 consecutive
// lanes cover 32 different banks.
constexpr int kPlane1 = 512 + 64;
constexpr int kMirrorOffset = kPlane1 + 512;
static_assert(2 * kMirrorOffset <= kQuarterBytes, "an epilogue quarter must fit half the wavefront's slice");
__device__ __forceinline__ int quarter_pos(int j) { return ((j >> 1) & 1) * kPlane1 + (j >> 2) * 16 + (j & 1) * 8; }
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
  uint32_t pend[2][4] = {};   // indices of iteration 0's epilogue, stored into the slot in iteration 1
  bool pend_active = false;
  for (uint64_t it = 0; it < (rounds + 2) * kTileIters; ++it) {
    const uint64_t round = it / kTileIters;
    const int sub = (int)(it % kTileIters);
    const Unit nxt = locate(it + 1);
    // iteration 1: every wavefront has read its quarters 2, 3 of the previous round (end of its iteration 0), so the slot
    // may be overwritten from here on
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

    if (cur.active) {
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

      // Epilogue in quarters of 128 bins.  For quarter Q the registers k2 = 4Q + r hold Z[k] for k = 128Q + 32r + l and
      // the registers 28 - 4Q + r hold the partners Z[N - k]; both go to LDS as value k - 128Q of their half of the
      // quarter buffer, so that a lane's four consecutive bins are two 16-byte reads each.  Lane 0's partner registers
      // sit one row off: its partner of value 0 is Z[N - 128Q], register (32 - 4Q) mod 32.
      const bool shifted = cur.shifted;
      int lq = l;
      asm volatile("" : "+v"(lq));
      unsigned char* zw = qbuf + quarter_pos(lq);                          // value 32r + l: + 128r
      unsigned char* pw = qbuf + kMirrorOffset + quarter_pos(32 - lq);     // partner 32(3-r) + 32 - l: + 128(3-r)
      // lane 0's partner of row 0 would be value 128, which nobody reads: it stores value 0 there instead
      unsigned char* pw0 = lq == 0 ? qbuf + kMirrorOffset : pw + 128 * 3;
      const unsigned char* zr = qbuf + 16 * lq;
      uint32_t wa2[2], wb2[2];
#pragma unroll
      for (int Q = 0; Q < 4; ++Q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kl = 4 * Q + r, km = 28 - 4 * Q + r, k0 = (32 - 4 * Q) & 31;
          *reinterpret_cast<float2*>(zw + 128 * r) = make_float2(re[kl], im[kl]);
          if (r == 0) *reinterpret_cast<float2*>(pw0) = lq == 0 ? make_float2(re[k0], im[k0]) : make_float2(re[km], im[km]);
          else *reinterpret_cast<float2*>(pw + 128 * (3 - r)) = make_float2(re[km], im[km]);
        }
        wave_sync();
        const float4 z01 = *reinterpret_cast<const float4*>(zr);
        const float4 z23 = *reinterpret_cast<const float4*>(zr + kPlane1);
        const float4 q01 = *reinterpret_cast<const float4*>(zr + kMirrorOffset);
        const float4 q23 = *reinterpret_cast<const float4*>(zr + kMirrorOffset + kPlane1);
        uint32_t wa = 0, wb = 0;
#define SONO_BIN(B, ZR, ZI, PR, PI)                    \
  {                                                    \
    const float ar = (ZR) + (PR), ai = (ZI) - (PI);    \
    const float br = (ZI) + (PI), bi = (PR) - (ZR);    \
    wa = put_index<PLUGIN, B>(__builtin_fmaf(ar, ar, ai * ai), fs, wa); \
    wb = put_index<PLUGIN, B>(__builtin_fmaf(br, br, bi * bi), fs, wb); \
  }
        SONO_BIN(0, z01.x, z01.y, q01.x, q01.y)
        SONO_BIN(1, z01.z, z01.w, q01.z, q01.w)
        SONO_BIN(2, z23.x, z23.y, q23.x, q23.y)
        SONO_BIN(3, z23.z, z23.w, q23.z, q23.w)
#undef SONO_BIN
        wa2[Q & 1] = shifted ? wb : wa;  // an odd last frame was computed as B: it belongs in the A slot
        wb2[Q & 1] = wb;
        wave_sync();  // the buffer is rewritten by the next quarter / the next iteration's transpose
        if (Q & 1) {
          // before: lanes of transform h hold frames (2h, 2h+1) of the iteration for the bin quads 64P + l (quarter 2P)
          // and 64P + 32 + l (quarter 2P + 1); after: lane (l, h) holds the quad 64P + l + 32h of all four frames
          const auto sa = __builtin_amdgcn_permlane32_swap(wa2[0], wa2[1], false, false);
          const auto sb = __builtin_amdgcn_permlane32_swap(wb2[0], wb2[1], false, false);
          if (sub == 0) {
            turn(sa[0], sb[0], sa[1], sb[1], pend[Q >> 1]);
          } else {
            uint32_t r4[4];
            turn(sa[0], sb[0], sa[1], sb[1], r4);
            park<kRoundFrames>(slot, p, r4, round, kWaves + wave, 32 * (Q - 1) + l + 32 * h);
          }
        }
      }
    }
    if (sub == 0) {
      pend_active = cur.active;
    } else if (pend_active) {  // frame group `wave` of the round: the cells iteration 0 kept
#pragma unroll
      for (int P = 0; P < 2; ++P) park<kRoundFrames>(slot, p, pend[P], round, wave, 64 * P + l + 32 * h);
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
  note_kernel(tab ? "stft1024_rgba_kernel<table>" : "stft1024_rgba_kernel<roots>");
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

}  // namespace sono
