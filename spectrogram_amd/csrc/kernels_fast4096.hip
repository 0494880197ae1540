// Float32 STFT kernel for 4096-point frames with table twiddles (the plugin's getFrequencies at the FFT size of the
// page's high-resolution views, spectrogram.esm.js:166-200,377-380,525-538).  Compile with -ffp-contract=fast.
//
// One wavefront transforms ONE real frame.  The reference's radix-2 network splits a 4096-point transform of x into the
// 2048-point transforms E of the even and O of the odd samples and a last stage X[k] = E[k] + w_k O[k]; E and O come
// out of one complex 2048-point transform of z[n] = x[2n] + i x[2n+1] run through the same network with the same
// factors: E[k] = (Z[k] + conj Z[2048-k]) / 2, O[k] = (Z[k] - conj Z[2048-k]) / 2i.
//   * lane (v, u) = v + 32u holds z[64a + 2v + u], a = 0..31: it loads sixteen 16-byte pieces x[128a' + 4v .. +3] for
//     a' = 16u .. 16u+15, multiplies them by the window as stored, and v_permlane32_swap hands the half it does not
//     need to the other lane of its row;
//   * stages T = 1..16 in registers (wave-uniform factors), a 32 x 32 transpose inside each half of the wavefront
//     (v_permlane16_swap + 16 x 16 LDS blocks, sono_fft32.h), stages T = 32..512 in registers with per-lane factors, stage
//     T = 1024 across the two lanes of a row: register m of lane k1 + 32 hb is Z[k1 + 32m + 1024hb];
//   * epilogue in eight passes of 256 bins: Z[k] and Z[2048 - k] go through LDS in ascending k, a lane reads four
//     consecutive bins of both and the four factors w_k, forms E + w O (twice the reference's value: exact), and fuses
//     |X|^2 -> log2 -> quantiser.  Passes outside the band are skipped.
// Outputs: [frame][bin] colour indices, or the RGBA image itself.  The image takes the workgroup's indices through a
// scratch slot in global memory: a round is 64 consecutive frames (four iterations of 16 wavefronts), a cell holds four
// bins x four frames, and during the next round every wavefront writes out its share of the finished round as
// 256-byte runs per image row.  Short recordings take rounds of 32 frames (128-byte runs): with a few rounds per
// workgroup the whole rounds that the last workgroups wait for are a tenth of the run time.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "sono_device.h"
#include "sono_fft32.h"
#include "sono_internal.h"
#include "sono_launch.h"
#include "sono_loud.h"

namespace sono {
namespace fast4096 {

using namespace fast1024;

constexpr int kWaves = 16;
#ifndef SONO_SYNC_ITERS_4096
#define SONO_SYNC_ITERS_4096 4
#endif
constexpr int kSyncIters = SONO_SYNC_ITERS_4096;
constexpr int kTwBytes = 32 * 64 * 8;   // factors of the stages T <= 1024, [63 rows + 1][32] float2 (see the fill loop)
constexpr int kLastBytes = 2048 * 8;    // factors w_k of the last stage
constexpr int kWinBytes = 4096 * 4;
constexpr int kLutBytes = 1024;
constexpr int kWaveBytes = 4 * kBlkFloats * 4;  // 4352: transpose blocks, then the epilogue's 256 + 257 cells of 8 bytes
constexpr int kPassBins = 256;
// An epilogue pass lays its 256 values (and, kRegion further on, their 256 partners) out in two planes: value j
// (8 bytes) sits in plane (j >> 1) & 1 at 16 (j >> 2) + 8 (j & 1), so that a lane's four consecutive values are one
// 16-byte read per plane at 16 * lane (conflict-free); the second plane starts 16 banks after the first so that the
// 8-byte stores of 16 consecutive lanes cover 32 different banks.  (Partner 256, which nobody reads, lands in the gap.)
constexpr int kPlane1 = 1024 + 64;
constexpr int kRegion = kPlane1 + 1024;
static_assert(2 * kRegion <= kWaveBytes, "an epilogue pass must fit the wavefront's buffer");
__device__ __forceinline__ int pass_pos(int j) { return ((j >> 1) & 1) * kPlane1 + (j >> 2) * 16 + (j & 1) * 8; }
constexpr int kOffLast = kTwBytes, kOffWin = kOffLast + kLastBytes, kOffLut = kOffWin + kWinBytes, kOffWave = kOffLut + kLutBytes;
constexpr int kLdsBytes = kOffWave + kWaves * kWaveBytes;
static_assert(kLdsBytes <= 160 * 1024, "LDS of one CU");

constexpr int kMaxRoundFrames = 64;              // frames per workgroup round (or 32)
constexpr int kSlotWords = (kMaxRoundFrames / 4) * 512 * 4;  // cell (frame group fg, bin quad kq) at 16-byte index fg * 512 + kq; word = frame & 3
constexpr int kFlushPasses = 2;                  // cells a lane writes out per iteration: iterations x 2 x 1024 lanes = 512 quads x groups

typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void swap32(float& a, float& b) {  // lanes 32..63 of a trade places with lanes 0..31 of b
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// four consecutive samples; `p` is 16-byte (float) / 8-byte (int16) aligned
template <typename In>
__device__ __forceinline__ float4 ld4(const In* p);
template <>
__device__ __forceinline__ float4 ld4<float>(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
template <>
__device__ __forceinline__ float4 ld4<int16_t>(const int16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  const float k = 1.0f / 32768.0f;
  return make_float4((float)(int16_t)(v.x & 0xffffu) * k, (float)(int16_t)(v.x >> 16) * k, (float)(int16_t)(v.y & 0xffffu) * k,
                     (float)(int16_t)(v.y >> 16) * k);
}

// The complex 2048-point transform of one wavefront with the factors of the reference's network.  In: lane (v, u),
// register brev5(a) = z[64a + 2v + u], i.e. the lane's registers are the positions 32 * (32u + brev5(v)) + r of the
// bit-reversed order.  Out: register m of lane k1 + 32 hb = Z[k1 + 32m + 1024hb].
__device__ __forceinline__ void transform2048(float (&re)[32], float (&im)[32], const float2* tw_lds, const float2* tw_flat, float* blk,
                                              int lane) {
  const int k1r = lane & 31, hb = lane >> 5;
  // stages T = 1..16: wave-uniform factors by scalar loads (the table is not written while the kernel runs; the opaque
  // copy keeps the scalars from being held in registers round the frame loop)
  typedef const float __attribute__((address_space(4))) * const_table;
  const_table twc = (const_table)(uintptr_t)tw_flat;
  asm volatile("" : "+s"(twc));
  fft32_table<true>(re, im, [&](int half, int j) { return make_float2(twc[2 * (half - 1 + j)], twc[2 * (half - 1 + j) + 1]); });
  // lane (v, u), register k1 -> lane (k1, u), register brev5(v): position 32 * (32u + m) + k1 in register m
  transpose32_plane(re, blk, lane & 15);
  transpose32_plane(im, blk, lane & 15);
  fft32_table<false>(re, im, [&](int half, int j) { return tw_lds[(half - 1 + j) * 32 + k1r]; });  // stages T = 32..512
  // stage T = 1024 pairs the two lanes of a row with the factor 32 mu + k1, mu = m + 16 hb: registers m and m + 16 trade
  // halves, each lane does 16 butterflies, and they trade back
  const float2* wrow = tw_lds + (31 + 16 * hb) * 32 + k1r;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    swap32(re[m], re[m + 16]);
    swap32(im[m], im[m + 16]);
    const float2 w = wrow[m * 32];
    const float ar = re[m], ai = im[m], br = re[m + 16], bi = im[m + 16];
    const float o1r = __builtin_fmaf(-w.y, bi, __builtin_fmaf(w.x, br, ar));
    const float o1i = __builtin_fmaf(w.y, br, __builtin_fmaf(w.x, bi, ai));
    re[m] = o1r; im[m] = o1i;
    re[m + 16] = __builtin_fmaf(2.0f, ar, -o1r);
    im[m + 16] = __builtin_fmaf(2.0f, ai, -o1i);
    swap32(re[m], re[m + 16]);
    swap32(im[m], im[m + 16]);
  }
}

// Epilogue pass P8: Z[256 P8 + j] as value j and the partners Z[2048 - 256 P8 - j] as partner j (j <= 256).  The bins
// are registers 8 (P8 & 3) + r of the lanes of half P8 >> 2 (j = k1 + 32r), the partners registers 8 (3 - (P8 & 3)) + r of
// the other half (j = 256 - k1 - 32r); the partner of j = 0 is one register off: Z[(2048 - 256 P8) mod 2048].
template <int P8>
__device__ __forceinline__ void epilogue_cells(const float (&re)[32], const float (&im)[32], unsigned char* buf, int le) {
  const int k1 = le & 31;
  const bool binlane = (le >> 5) == (P8 >> 2);
  constexpr uint64_t kBinLanes = (P8 >> 2) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
  unsigned char* c0 = buf + (binlane ? pass_pos(k1) : kRegion + pass_pos(32 - k1) + 128 * 7);
  const int step = binlane ? 128 : -128;  // 32 values further on = 8 quads = 128 bytes in the same plane
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int lo = 8 * (P8 & 3) + r, hi = 8 * (3 - (P8 & 3)) + r;
    *reinterpret_cast<float2*>(c0 + r * step) =
        make_float2(lane_select(re[hi], re[lo], kBinLanes), lane_select(im[hi], im[lo], kBinLanes));
  }
  constexpr int kQ = (8 - P8) & 7;  // Z[256 kQ]: lane k1 = 0 of half kQ >> 2, register 8 (kQ & 3)
  if (k1 == 0 && (le >> 5) == (kQ >> 2)) *reinterpret_cast<float2*>(buf + kRegion) = make_float2(re[8 * (kQ & 3)], im[8 * (kQ & 3)]);
}

// RGBA: the image [row = row_of_bin0 - bin][column = col0 + frame] of LUT colours (see the head of the file).
// RF: frames per image round, 0 = colour indices out.
template <bool PLUGIN, typename In, int RF>
__global__ __launch_bounds__(kWaves * 64) void stft4096_kernel(StftParams p, FastScale fs, uint8_t* __restrict__ tiles) {
  constexpr bool RGBA = RF != 0;
  constexpr int kRoundFrames = RGBA ? RF : kMaxRoundFrames;
  constexpr int kRoundIters = kRoundFrames / kWaves;
  constexpr int kGroups = kRoundFrames / 4;  // frame groups (four frames) per round
  constexpr int kThreads = kWaves * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw_lds = reinterpret_cast<float2*>(smem);
  float2* last_lds = reinterpret_cast<float2*>(smem + kOffLast);
  float* win_lds = reinterpret_cast<float*>(smem + kOffWin);
  uint32_t* lut_lds = reinterpret_cast<uint32_t*>(smem + kOffLut);
  const float2* __restrict__ tw_flat = reinterpret_cast<const float2*>(p.twiddles);  // stage with half size T at [T-1, 2T-1)
  for (int i = threadIdx.x; i < 2048; i += kThreads) {
    const int r = i >> 5, k = i & 31;  // rows 0..30: stage T = 32t, factor 32j + k (r = t-1+j); rows 31..62: T = 1024, 32m + k
    if (r < 31) {
      const int t = 1 << (31 - __builtin_clz(r + 1)), j = r + 1 - t;
      tw_lds[i] = tw_flat[32 * t - 1 + 32 * j + k];
    } else if (r < 63) {
      tw_lds[i] = tw_flat[1023 + 32 * (r - 31) + k];
    }  // (the 31 factors of the stages T = 1..16 are read by scalar loads)
    last_lds[(i >> 1 & 1) * 1024 + (i >> 2) * 2 + (i & 1)] = tw_flat[2047 + i];  // two planes, like an epilogue pass
    win_lds[i] = p.window[i];
    win_lds[2048 + i] = p.window[2048 + i];
  }
  if constexpr (RGBA) {
    for (int i = threadIdx.x; i < 256; i += kThreads) lut_lds[i] = p.lut[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* wbuf = smem + kOffWave + wave * kWaveBytes;
  float* blk = reinterpret_cast<float*>(wbuf) + (lane >> 4) * kBlkFloats;
  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  uint8_t* __restrict__ out = reinterpret_cast<uint8_t*>(p.out);
  // blocks with equal blockIdx % 8 share an XCD (and its L2): give them neighbouring frame ranges
  const uint32_t per = gridDim.x >> 3;
  const uint32_t vb = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  // every workgroup runs the same number of rounds and every wavefront goes through every round (the image rounds are
  // fenced by workgroup barriers); the image needs one more round to write out the last one
  const uint64_t nrounds = (p.nframes + kRoundFrames - 1) / kRoundFrames;
  const uint64_t rounds = (nrounds + gridDim.x - 1) / gridDim.x;
  uint32_t* slots = RGBA ? reinterpret_cast<uint32_t*>(tiles) + (size_t)blockIdx.x * 2 * kSlotWords : nullptr;
  const int qlo = (int)(p.bin_lo >> 2), nq = (int)((p.bin_hi + 3) >> 2) - qlo;  // bin quads of the band

  // this lane's cells of iteration `i`: bin quad qlo + tq, frame group lane & (kGroups - 1)
  auto flush_quad = [&](int i, int s) { return ((i * kFlushPasses + s) * kWaves + wave) * (64 / kGroups) + lane / kGroups; };
  uint4 fq[kFlushPasses] = {};
  // A frame's even and odd samples are transformed as one complex frame; under the context's table that split carries
  // the same kind of trace as a pair of frames does (StftParams::loud_thr_raw).  A frame too loud for it is transformed
  // twice more: mode 1 = its even samples alone, whose transform E is parked in this wavefront's 16 KiB of p.redo_scratch,
  // mode 2 = its odd samples alone, whose transform O meets E again in the epilogue.  Wave-uniform.
  float2* park = reinterpret_cast<float2*>(p.redo_scratch) + ((size_t)blockIdx.x * kWaves + wave) * 2048;
  int mode = 0;
  for (uint64_t it = 0; it < (rounds + (RGBA ? 1 : 0)) * kRoundIters;) {
    const uint64_t round = it / kRoundIters;
    const int i = (int)(it % kRoundIters);
    const uint64_t f = round < rounds ? (round * gridDim.x + vb) * kRoundFrames + kWaves * i + wave : p.nframes;
    const bool active = f < p.nframes;  // wave-uniform
    // (the index flavour has no barriers of its own: its wavefronts meet here every kSyncIters iterations, so that the
    // samples neighbouring frames share are still in the caches when the next wavefront asks for them)
    if constexpr (!RGBA) {
      if (mode == 0 && it != 0 && it % kSyncIters == 0) __syncthreads();
    }
    float4 raw[16];
    {
      const int v = lane & 31, u = lane >> 5;
      const In* x0 = x + (p.frame0 + (active ? f : 0)) * p.hop + 128 * 16 * u + 4 * v;
#pragma unroll
      for (int a = 0; a < 16; ++a) raw[a] = ld4<In>(x0 + 128 * a);  // (mode 2: the same samples again, from the cache)
    }
    if constexpr (RGBA) {
      if (round > 0 && mode != 2) {  // write out this lane's share of the round before, parked in fq at the end of the last iteration
        const uint64_t frame0 = ((round - 1) * gridDim.x + vb) * kRoundFrames;
        if (frame0 < p.nframes) {
          int w = wave, j = lane;
          asm volatile("" : "+s"(w));  // (loop-invariant addresses computed in place: hoisted, they would hold registers
          asm volatile("" : "+v"(j));  //  the transform needs)
          const int fg = j & (kGroups - 1);
          const int64_t width4 = (int64_t)p.width * 4;
          const bool whole = frame0 + kRoundFrames <= p.nframes;  // workgroup-uniform
          const uint64_t fl = frame0 + 4 * fg;
          const int nvalid = fl >= p.nframes ? 0 : (p.nframes - fl >= 4 ? 4 : (int)(p.nframes - fl));
#pragma unroll
          for (int s = 0; s < kFlushPasses; ++s) {
            const int tq = ((i * kFlushPasses + s) * kWaves + w) * (64 / kGroups) + j / kGroups;
            if (tq >= nq) continue;
            const int bin0 = 4 * (qlo + tq);
            uint8_t* __restrict__ base = out + (int64_t)(p.col0 + fl) * 4 + (int64_t)(p.row_of_bin0 - bin0) * width4;
            const uint4 q = fq[s];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              if (bin0 + b < (int)p.bin_lo || bin0 + b >= (int)p.bin_hi) continue;  // a band may start or end inside a quad
              u32x4_unaligned px;
              px.x = lut_lds[(q.x >> (8 * b)) & 255u];
              px.y = lut_lds[(q.y >> (8 * b)) & 255u];
              px.z = lut_lds[(q.z >> (8 * b)) & 255u];
              px.w = lut_lds[(q.w >> (8 * b)) & 255u];
              uint8_t* dst = base - (int64_t)b * width4;
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
      }
    }

    if (active) {
      if (mode == 0 && p.loud_thr_raw > 0.0f && p.redo_scratch) {
        // (a NaN sample does not compare as loud and needs no second look: it poisons every bin either way)
        float m0 = 0.0f, m1 = 0.0f;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
          m0 = max3_abs(raw[a].x, raw[a].y, m0);
          m1 = max3_abs(raw[a].z, raw[a].w, m1);
        }
        if (__builtin_amdgcn_ballot_w64(__builtin_fmaxf(m0, m1) > p.loud_thr_raw) != 0) mode = 1;
      }
      float re[32], im[32];
      {
        const float4* win4 = reinterpret_cast<const float4*>(win_lds) + 32 * 16 * (lane >> 5) + (lane & 31);
#pragma unroll
        for (int a = 0; a < 16; ++a) {
          const float4 w = win4[32 * a];
          float s0 = raw[a].x * w.x, s1 = raw[a].y * w.y, s2 = raw[a].z * w.z, s3 = raw[a].w * w.w;
          // lower lane (u = 0): keeps (s0, s1) = z[64a + 2v] and takes the upper lane's (s0, s1) = z[64(16 + a) + 2v];
          // upper lane: takes the lower lane's (s2, s3) = z[64a + 2v + 1] and keeps (s2, s3) = z[64(16 + a) + 2v + 1]
          swap32(s0, s2);
          swap32(s1, s3);
          re[brev5(a)] = s0; im[brev5(a)] = s1;
          re[brev5(16 + a)] = s2; im[brev5(16 + a)] = s3;
        }
      }
      if (mode == 1) {  // the even samples alone
#pragma unroll
        for (int a = 0; a < 32; ++a) im[a] = 0.0f;
      } else if (mode == 2) {  // the odd samples alone
#pragma unroll
        for (int a = 0; a < 32; ++a) {
          re[a] = im[a];
          im[a] = 0.0f;
        }
      }
      int lt = lane;
      asm volatile("" : "+v"(lt));  // (keeps the transform's loop-invariant LDS addresses out of registers it needs itself)
      transform2048(re, im, tw_lds, tw_flat, blk, lt);
      if (mode == 1) {
        // E[k1 + 32m + 1024hb] is register m of lane k1 + 32hb: parked for the epilogue of the next pass of this loop.
        // (The lanes that read a value back are not the ones that wrote it, but they are lanes of this wavefront: the
        // stores have left the compute unit's write-through vector cache once the fence of workgroup scope has passed,
        // and no other compute unit ever touches this wavefront's 16 KiB.)
        float2* e0 = park + (lt & 31) + 1024 * (lt >> 5);
#pragma unroll
        for (int m = 0; m < 32; ++m) e0[32 * m] = make_float2(re[m], im[m]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        mode = 2;
        continue;
      }

      int le = lane;
      asm volatile("" : "+v"(le));
      const uint64_t row = (p.col0 + f) * p.out_bins;
      uint32_t* slot = RGBA ? slots + (round & 1) * kSlotWords + ((kWaves * i + wave) >> 2) * (512 * 4) + (wave & 3) : nullptr;
      // ALONE (mode 2): the registers hold O, the transform of the odd samples alone; E comes back from where mode 1
      // parked it, and the last stage runs on 2 E and 2 O -- the same factor 2 as the split's separation leaves.
      auto pass = [&](auto P8c, auto alone_c) __attribute__((always_inline)) {
        constexpr int P8 = decltype(P8c)::value;
        constexpr bool ALONE = decltype(alone_c)::value;
        if (kPassBins * (P8 + 1) <= (int)p.bin_lo || kPassBins * P8 >= (int)p.bin_hi) return;  // wave-uniform
        const int k = kPassBins * P8 + 4 * le;
        float4 e01 = make_float4(0, 0, 0, 0), e23 = e01;
        if constexpr (ALONE) {  // (requested before the exchange through LDS, used after it)
          const float4* ep = reinterpret_cast<const float4*>(park + k);
          e01 = ep[0];
          e23 = ep[1];
        }
        epilogue_cells<P8>(re, im, wbuf, le);
        wave_sync();
        const unsigned char* zr = wbuf + 16 * le;
        float4 z01 = *reinterpret_cast<const float4*>(zr);
        float4 z23 = *reinterpret_cast<const float4*>(zr + kPlane1);
        float4 q01 = *reinterpret_cast<const float4*>(zr + kRegion);
        float4 q23 = *reinterpret_cast<const float4*>(zr + kRegion + kPlane1);
        if constexpr (ALONE) {
          // the values read back are O[k .. k+3]; E takes the place of Z and O the place of the partner below
          q01 = z01;
          q23 = z23;
          z01 = e01;
          z23 = e23;
        }
        const float4 w01 = reinterpret_cast<const float4*>(last_lds)[64 * P8 + le];
        const float4 w23 = reinterpret_cast<const float4*>(last_lds)[512 + 64 * P8 + le];
        uint32_t word = 0;
#define SONO_BIN(B, ZR, ZI, PR, PI, WR, WI)                                  \
  {                                                                          \
    /* ALONE: (ZR, ZI) is E[k] and (PR, PI) is O[k] themselves */            \
    const float ar = ALONE ? (ZR) + (ZR) : (ZR) + (PR), ai = ALONE ? (ZI) + (ZI) : (ZI) - (PI); /* 2 E[k] */ \
    const float br = ALONE ? (PR) + (PR) : (ZI) + (PI), bi = ALONE ? (PI) + (PI) : (PR) - (ZR); /* 2 O[k] */ \
    const float xr = __builtin_fmaf(-(WI), bi, __builtin_fmaf((WR), br, ar)); \
    const float xi = __builtin_fmaf((WI), br, __builtin_fmaf((WR), bi, ai)); \
    word = put_index<PLUGIN, B>(__builtin_fmaf(xr, xr, xi * xi), fs, word);  \
  }
        SONO_BIN(0, z01.x, z01.y, q01.x, q01.y, w01.x, w01.y)
        SONO_BIN(1, z01.z, z01.w, q01.z, q01.w, w01.z, w01.w)
        SONO_BIN(2, z23.x, z23.y, q23.x, q23.y, w23.x, w23.y)
        SONO_BIN(3, z23.z, z23.w, q23.z, q23.w, w23.z, w23.w)
#undef SONO_BIN
        if constexpr (RGBA) {
          if (k + 3 >= (int)p.bin_lo && k < (int)p.bin_hi) slot[(k >> 2) * 4] = word;
        } else {
          if (k >= (int)p.bin_lo && k < (int)p.bin_hi) *reinterpret_cast<uint32_t*>(out + row + (k - p.out_bin0)) = word;
        }
        wave_sync();  // the buffer is rewritten by the next pass / the next iteration's transpose
      };
      auto passes = [&](auto alone_c) __attribute__((always_inline)) {
        pass(std::integral_constant<int, 0>{}, alone_c);
        pass(std::integral_constant<int, 1>{}, alone_c);
        pass(std::integral_constant<int, 2>{}, alone_c);
        pass(std::integral_constant<int, 3>{}, alone_c);
        pass(std::integral_constant<int, 4>{}, alone_c);
        pass(std::integral_constant<int, 5>{}, alone_c);
        pass(std::integral_constant<int, 6>{}, alone_c);
        pass(std::integral_constant<int, 7>{}, alone_c);
      };
      if (mode == 2) passes(std::true_type{});
      else passes(std::false_type{});
      mode = 0;
    }

    if constexpr (RGBA) {
      if (i == kRoundIters - 1) __syncthreads();  // this round's slot is complete and everyone is done reading the other one
      const uint64_t nround = (it + 1) / kRoundIters;
      if (nround > 0 && nround <= rounds) {
        const int ni = (int)((it + 1) % kRoundIters);
        const uint4* prev = reinterpret_cast<const uint4*>(slots + ((nround - 1) & 1) * kSlotWords);
#pragma unroll
        for (int s = 0; s < kFlushPasses; ++s) {
          const int tq = flush_quad(ni, s);
          if (tq < nq) fq[s] = prev[(lane & (kGroups - 1)) * 512 + qlo + tq];
        }
      }
    }
    ++it;
  }
}

}  // namespace fast4096

static bool fast4096_common(const StftParams& p) {
  if (p.n != 4096 || p.nframes < 1 || p.fast_mode != 2 || !p.twiddles) return false;
  // the split's trace could show: the kernel transforms the loud frames' even and odd samples apart, given the scratch
  if (!p.split_ok && !(p.loud_thr_raw > 0.0f && p.redo_scratch)) return false;
  if (p.hop & 3) return false;  // frames start on 16-byte (float) / 8-byte (int16) boundaries
  const uintptr_t align = p.in_kind == SONO_IN_I16 ? 7 : 15;
  if (reinterpret_cast<uintptr_t>(p.pcm) & align) return false;
  return p.bin_lo < p.bin_hi && p.bin_hi <= 2048;
}

uint64_t redo_scratch_bytes() { return (uint64_t)device_cu_count() * fast4096::kWaves * 2048 * 8; }

// Colour indices [frame][bin]: whole groups of four bins.
bool fast4096_available(const StftParams& p) {
  if (!fast4096_common(p) || p.out_kind != SONO_OUT_U8_INDEX) return false;
  if ((p.bin_lo & 3) || (p.bin_hi & 3)) return false;
  return !(p.out_bins & 3) && !(p.out_bin0 & 3) && !(reinterpret_cast<uintptr_t>(p.out) & 3);
}

// The image output needs the flipped full-height row mapping, a colour table and the scratch buffer
// (fast_rgba_scratch_bytes()); any band.
bool fast4096_rgba_available(const StftParams& p) {
  if (!fast4096_common(p) || p.out_kind != SONO_OUT_RGBA8) return false;
  if (p.row_of_bin0 != (int32_t)p.bin_hi - 1 || (reinterpret_cast<uintptr_t>(p.out) & 3)) return false;
  return p.width != 0 && p.width <= ((uint64_t)1 << 24) && p.lut;
}

// Returns 1 if this launch is not served here.
int launch_stft_fast4096(const StftParams& p, void* stream_v) {
  using namespace fast4096;
  const bool rgba = p.out_kind == SONO_OUT_RGBA8;
  if (rgba ? !(fast4096_rgba_available(p) && p.tile_scratch) : !fast4096_available(p)) return 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  FastScale fs = make_fast_scale(p.scale_mode, p.n, p.gain_db, p.range_db);
  fs.b -= 2.0f * fs.a;  // the epilogue forms 2 E + w 2 O = 2X: |2X|^2 = 4|X|^2, i.e. log2 is 2 too large
  const int cus = device_cu_count();
  if (rgba && (uint64_t)cus * 2 * kSlotWords * 4 > fast_rgba_scratch_bytes()) return 1;
  // rounds of 32 frames when a workgroup would get fewer than 16 rounds of 64
  const int rf = rgba && (p.nframes + 63) / 64 < (uint64_t)16 * cus ? 32 : 64;
  const uint64_t need = (p.nframes + rf - 1) / rf;
  const uint32_t grid = (uint32_t)(need < (uint64_t)cus ? need : (uint64_t)cus);  // persistent: one workgroup per CU
  uint8_t* tiles = reinterpret_cast<uint8_t*>(p.tile_scratch);
#define SONO_L4096(PL, In, RG)                                                                   \
  do {                                                                                           \
    auto kern = stft4096_kernel<PL, In, RG>;                                                     \
    SONO_GRANT_LDS(kern, kLdsBytes);                                                             \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), kLdsBytes, stream, p, fs, tiles);    \
  } while (0)
#define SONO_L4096_I(PL, RG)                                                                     \
  do {                                                                                           \
    if (p.in_kind == SONO_IN_I16) SONO_L4096(PL, int16_t, RG); else SONO_L4096(PL, float, RG);   \
  } while (0)
  const bool plugin = p.scale_mode == SONO_SCALE_PLUGIN_DB;
  if (rgba && rf == 32) { if (plugin) SONO_L4096_I(true, 32); else SONO_L4096_I(false, 32); }
  else if (rgba) { if (plugin) SONO_L4096_I(true, 64); else SONO_L4096_I(false, 64); }
  else { if (plugin) SONO_L4096_I(true, 0); else SONO_L4096_I(false, 0); }
#undef SONO_L4096_I
#undef SONO_L4096
  // (redo: the kernel transforms the even and the odd samples of a loud frame apart)
  const bool redo = !p.split_ok;
  note_kernel(rgba ? (redo ? "stft4096_kernel<image,redo>" : "stft4096_kernel<image>") : redo ? "stft4096_kernel<redo>" : "stft4096_kernel");
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

}  // namespace sono
