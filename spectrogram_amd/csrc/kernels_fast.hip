// Fast float32 STFT kernel for 1024-point frames with accurate twiddles (the worker path and the
// plugin path in "accurate" twiddle mode).
//
// Two real frames per complex transform: z[n] = w[n]*(xA[n] + i*xB[n]) for the frame pair (A, B);
// after the 1024-point complex FFT the two spectra are separated with
//     XA[k] = (Z[k] + conj Z[N-k]) / 2        XB[k] = (Z[k] - conj Z[N-k]) / (2i)
// which costs four additions per bin pair and no twiddle.
//
// Mapping to CDNA4: one HALF wavefront (32 lanes) owns one complex FFT and keeps all 1024 points in
// registers, 32 per lane.  1024 = 32 x 32, so the transform is two in-register radix-32 passes
// (each a fully unrolled 5-stage radix-2 DIT with compile-time twiddles and the 6-FMA butterfly)
// with one transpose through LDS in between and one more LDS round trip that brings Z[k] and
// Z[N-k] together for the epilogue.  Each wavefront owns its slice of LDS, so the steady-state
// loop has no workgroup barrier.  The epilogue fuses |X|^2 -> log2 -> quantiser -> packed uint8.
//
// The kernel is bound by vector-ALU issue (about 1300 VALU instructions per iteration, which is close
// to the operation count of the algorithm) with the LDS pipe about half busy, so it lives on
// occupancy: the per-wavefront LDS slice is kept at 9 KiB (the transpose moves the real and the
// imaginary plane one after the other through the same 4.5 KiB buffer and the epilogue handles the
// bins in two halves) and registers at <= 128, which gives one 1024-thread workgroup (16 independent
// wavefronts = 4 per SIMD) per CU: 8 KiB twiddle matrix W_1024^(l*k1) + 4.5 KiB window + 32 x 4608 B.
// Packed FP32 (v_pk_fma_f32) was measured and does not pay here: it issues at 2/3 of the scalar rate
// on this chip and the operand pairing costs moves.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "sono_device.h"
#include "sono_fft32.h"
#include "sono_internal.h"
#include "sono_launch.h"
#include "sono_loud.h"
#include "sono_rgba_flush.h"

namespace sono {
namespace fast1024 {

constexpr int kNPoints = 1024;
constexpr int kMaxWaves = 16;                      // independent wavefronts per workgroup (one workgroup per CU)
constexpr int kPlaneStride = 36;                   // floats per transpose row: rows stay 16-B aligned and 16 consecutive
                                                   // rows start on 16 distinct 4-bank groups (ds_read_b128 conflict-free)
constexpr int kFftBufBytes = 32 * kPlaneStride * 4;  // 4608: one plane of the transpose, or one half of the epilogue
constexpr int kMirrorOffset = 256 * 8;             // epilogue: Z[k] in [0, 2048), Z[N-k] in [2048, 2048 + 257*8)
static_assert(kMirrorOffset + 257 * 8 <= kFftBufBytes, "epilogue half must fit the transpose buffer");
constexpr int kTwBytes = 32 * 32 * 8;
constexpr int kWinBytes = 32 * kPlaneStride * 4;   // window as [l][kPlaneStride]: lane l reads w[32a + l], a = 0..31, as 8 x 16 B
constexpr int lds_bytes(int waves) { return kTwBytes + kWinBytes + waves * 2 * kFftBufBytes; }  // 16 waves: 160256
// HOP32 > 0: hop == 32*HOP32, so frame B's sample 32a+l is frame A's sample 32(a+HOP32)+l and one
// run of 32+HOP32 loads feeds both frames.  HOP32 == 0: any hop, two runs of 32 loads.
// HOP32 < 0: ONE real frame per transform (imaginary part zero), for twiddle tables under which a frame would leave
// a visible trace in a partner (StftParams::pair_idx_ok) or float32 outputs would carry it (pair_exact): half the
// throughput of the paired flavour, which is still several times the general kernel's.  A wavefront then handles two
// frames per iteration, one per half.
//
// The image output is a kernel of its own (kernels_fast_rgba.hip).
// OUT: kOutU8 / kOutFeatures (per-frame spectral peak: strict-> first-wins argmax of |X|^2
// over the band and that bin's colour index; 8 bytes per frame).
// TABLE: twiddles are the context's per-stage table (flat layout, stage with half size T at [T-1, 2T-1))
// instead of exact roots of unity: the first pass reads its 31 factors as wave-uniform scalars, the
// second pass reads factor 32*(g mod t) + lane of stage T = 32t from a [31][32] copy in LDS, and there
// is no twiddle multiplication between the passes (it is part of those factors).
// kOutDb / kOutMag: float32 20*log10(m) / m per bin, [frame][bin] (what the tolerance tests read).
// kOutU8Bank: colour indices on a non-linear frequency scale: out[y] = m[tap[y]] * w0[y] + m[tap[y] + 1] * w1[y] over the
// frame's magnitudes (spectrogram.esm.js:479-488 with the two non-zero weights of a row), then the quantiser.
constexpr int kOutU8 = 0, kOutFeatures = 2, kOutDb = 3, kOutMag = 4, kOutU8Bank = 5;
#ifndef SONO_SYNC_ITERS
#define SONO_SYNC_ITERS 4
#endif
constexpr int kSyncIters = SONO_SYNC_ITERS;  // iterations between two barriers of the index kernel's wavefronts
template <bool PLUGIN, typename In, int HOP32, int OUT, bool TABLE>
__global__ __launch_bounds__(kMaxWaves * 64, 4) void stft1024_u8_kernel(StftParams p, FastScale fs,
                                                                        const float2* __restrict__ tw_table) {
  constexpr bool FEAT = OUT == kOutFeatures, FLT = OUT == kOutDb || OUT == kOutMag;
  constexpr bool SINGLE = HOP32 < 0, BANK = OUT == kOutU8Bank;
  // Flavours that pair frames under the context's table redo a unit that is too loud for the pair separation under that
  // table (StftParams::loud_thr_raw) themselves: two more passes of the loop body, frame A alone and frame B alone.
  constexpr bool REDO = TABLE && !SINGLE && (OUT == kOutU8 || BANK);
  static_assert(!(SINGLE && FEAT), "the feature records come from the paired flavour (its marks cover the cross-talk)");
  constexpr int kThreads = kMaxWaves * 64, kWaves = kMaxWaves;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw_lds = reinterpret_cast<float2*>(smem);
  float* win_lds = reinterpret_cast<float*>(smem + kTwBytes);
  const float2* __restrict__ tw_flat = reinterpret_cast<const float2*>(p.twiddles);
  for (int i = threadIdx.x; i < 1024; i += kThreads) {
    if constexpr (TABLE) {
      // row r = half - 1 + j of the second pass: stage T = 32*half, factor 32*j + lane
      const int r = i >> 5, lane_p = i & 31;
      if (r < 31) {
        const int half = 1 << (31 - __builtin_clz(r + 1)), j = r + 1 - half;
        tw_lds[i] = tw_flat[32 * half - 1 + 32 * j + lane_p];
      }  // (the 31 factors of the first pass, flat positions 0..30, are wave-uniform and read by scalar loads)
    } else {
      tw_lds[i] = tw_table[i];
    }
    win_lds[(i & 31) * kPlaneStride + (i >> 5)] = p.window[i];  // sample n = 32a + l at [l][a]
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, l = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // lets wave-uniform addresses live in SGPRs
  unsigned char* fbuf = smem + kTwBytes + kWinBytes + (wave * 2 + h) * kFftBufBytes;
  const float4* win_row = reinterpret_cast<const float4*>(win_lds + l * kPlaneStride);
  float* plane = reinterpret_cast<float*>(fbuf);                   // transpose view: [k1][kPlaneStride] floats
  float2* lbuf = reinterpret_cast<float2*>(fbuf);                  // epilogue view: Z[k], 256 entries
  float2* mbuf = reinterpret_cast<float2*>(fbuf + kMirrorOffset);  // epilogue view: Z[N-k], 257 entries

  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  uint8_t* __restrict__ out = reinterpret_cast<uint8_t*>(p.out);
  const uint64_t npairs = SINGLE ? p.nframes : (p.nframes + 1) >> 1;  // transforms
  // blocks with equal blockIdx % 8 share an XCD (and its L2): give them neighbouring frame ranges
  const uint32_t per = gridDim.x >> 3;
  const uint32_t vb = (gridDim.x & 7) ? blockIdx.x : (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  // A workgroup takes a CONTIGUOUS range of frames, `iters` iterations of its 16 wavefronts one after the other: the
  // N - hop samples that neighbouring blocks of 64 frames share are then re-read from the workgroup's own cache lines
  // an iteration later.  (Dealt out round-robin over the grid, every block's overlap with the next one was fetched
  // again by another CU: 1.3 x the input bytes at hop = N/4.)
  const uint64_t wg_iters_total = (npairs + 2 * kWaves - 1) / (2 * kWaves);
  const uint64_t iters = (wg_iters_total + gridDim.x - 1) / gridDim.x;

  // Iteration `it` of this wavefront: unit = four consecutive frames (two per half-wavefront).  Frames (f0, f0+1) form
  // the complex input; st_a / st_b say which of the two results are stored.  An odd last frame is computed as the
  // second member of the pair before it.  `active` is false when the wavefront has no frame left.
  struct Unit {
    uint64_t f0;
    bool active, st_a, st_b;
  };
  auto locate = [&](uint64_t it) -> Unit {
    Unit u;
    const uint64_t pair0 = (((uint64_t)vb * iters + it) * kWaves + wave) * 2;
    u.active = it < iters && pair0 < npairs;  // wave-uniform
    if (!u.active) {            // nothing left: point at frame 0 so that the sample loads stay in bounds
      u.f0 = 0;
      u.st_a = u.st_b = false;
      return u;
    }
    const uint64_t pair = pair0 + h;
    if constexpr (SINGLE) {  // transform `pair` is frame `pair`; a half without a frame repeats the other half's
      u.st_a = pair < npairs;
      u.st_b = false;
      u.f0 = u.st_a ? pair : pair0;
      return u;
    }
    u.f0 = 2 * pair;
    u.st_a = u.st_b = pair < npairs;
    if (!u.st_a) {  // this half has no pair of its own: it repeats valid frames and stores nothing
      u.f0 = 2 * pair0 + 1 < p.nframes ? 2 * pair0 : 2 * pair0 - 1;
    } else if (u.f0 + 1 >= p.nframes) {
      u.f0 -= 1;
      u.st_a = false;
    }
    return u;
  };
  // The samples of an iteration are loaded one iteration ahead (into raw), so that their latency overlaps with the
  // arithmetic.
  constexpr int kRaw = SINGLE ? 32 : HOP32 > 0 ? 32 + HOP32 : 64;
  float raw[kRaw];
  // (alone: frame u.f0 only -- the frame after it may lie beyond the signal)
  auto fetch = [&](const Unit& u, bool alone) {
    const In* xa = x + (p.frame0 + u.f0) * p.hop + l;
    if (SINGLE || alone) {
      // (opaque: or the samples already loaded for the pair are kept in registers instead of being read again)
      if constexpr (!SINGLE) asm volatile("" : "+v"(xa));
#pragma unroll
      for (int a = 0; a < 32; ++a) raw[a] = ld<In>(xa, 32 * a);
#pragma unroll
      for (int a = 32; a < kRaw; ++a) raw[a] = 0.0f;  // (defined, so that nothing is carried round the loop)
    } else if constexpr (HOP32 > 0) {
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
  uint64_t it = 0;
  Unit cur = locate(it);
  fetch(cur, false);
  int mode = 0;  // REDO: 0 = the pair (A, B) in one transform; 1 / 2 = frame A / frame B alone.  Wave-uniform.
  // Every wavefront of the workgroup goes through all `iters` iterations, with or without frames of its own, and they
  // meet at a barrier every kSyncIters of them: left alone they drift apart by more than an iteration, and the samples
  // that neighbouring wavefronts share (3/4 of a frame group at hop = N/4) are then fetched from memory again --
  // the L2 of an XCD turns over once per iteration (measured: 1.32 x the input bytes without the barrier).
  while (it < iters) {
    if (mode == 0 && it != 0 && it % kSyncIters == 0) __syncthreads();
    if (!cur.active) {  // (nothing left for this wavefront: every later unit of it is inactive as well)
      ++it;
      continue;
    }
    const uint64_t f0 = cur.f0;  // the frame whose row the first result goes to
    const bool st_a = cur.st_a;
    bool st_b = SINGLE ? false : cur.st_b;
    if constexpr (REDO) {
      // the four frames of a wavefront are redone together (its two halves run in lockstep anyway); a NaN sample does
      // not compare as loud and needs no second look (it poisons every bin either way)
      if (mode == 0 && p.loud_thr_raw > 0.0f && __builtin_amdgcn_ballot_w64(max_abs(raw) > p.loud_thr_raw) != 0) mode = 1;
    }
    const uint64_t it_next = (REDO && mode == 1) ? it : it + 1;
    Unit nxt = locate(it_next);
    it = it_next;
    if constexpr (REDO) {
      if (mode == 1) {  // frame A alone now; then this unit once more as "frame B alone" (its samples come from the cache)
        nxt.f0 = f0 + 1;
        nxt.st_a = st_b;
      }
      if (mode != 0) st_b = false;
    }

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
        im[brev5(a)] = SINGLE ? 0.0f : raw[SINGLE ? a : a + (HOP32 > 0 ? HOP32 : 32)] * win[a];
      }
      if (REDO && mode != 0) {
#pragma unroll
        for (int a = 0; a < 32; ++a) im[a] = 0.0f;
      }
    }
    // first pass; then (exact roots only) the twiddle W_1024^(l*k1); then transpose lane l, register
    // k1 -> lane k1, register l; the real and the imaginary plane go through the same LDS buffer one
    // after the other
    if constexpr (TABLE) {
      // stages T = 1..16: wave-uniform factors, read by scalar loads (the table is not written while the kernel runs;
      // the opaque copy keeps the 62 scalars from being held in registers round the frame loop)
      typedef const float __attribute__((address_space(4))) * const_table;
      const_table twc = (const_table)(uintptr_t)tw_flat;
      asm volatile("" : "+s"(twc));
      fft32_table<true>(re, im, [&](int half, int j) { return make_float2(twc[2 * (half - 1 + j)], twc[2 * (half - 1 + j) + 1]); });
    } else {
      fft32(re, im);  // Y_l[k1]
#pragma unroll
      for (int k1 = 1; k1 < 32; ++k1) {
        const float2 t = tw_lds[k1 * 32 + l];
        const float ur = __builtin_fmaf(re[k1], t.x, -im[k1] * t.y);
        const float ui = __builtin_fmaf(re[k1], t.y, im[k1] * t.x);
        re[k1] = ur;
        im[k1] = ui;
      }
    }
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) plane[k1 * kPlaneStride + l] = re[k1];
    wave_sync();
    {
      const float4* row = reinterpret_cast<const float4*>(plane + l * kPlaneStride);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v = row[j];
        re[brev5(4 * j)] = v.x; re[brev5(4 * j + 1)] = v.y; re[brev5(4 * j + 2)] = v.z; re[brev5(4 * j + 3)] = v.w;
      }
    }
    wave_sync();
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) plane[k1 * kPlaneStride + l] = im[k1];
    wave_sync();
    {
      const float4* row = reinterpret_cast<const float4*>(plane + l * kPlaneStride);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v = row[j];
        im[brev5(4 * j)] = v.x; im[brev5(4 * j + 1)] = v.y; im[brev5(4 * j + 2)] = v.z; im[brev5(4 * j + 3)] = v.w;
      }
    }
    wave_sync();
    if constexpr (TABLE) {
      int lt = l;
      asm volatile("" : "+v"(lt));  // (keeps 31 loop-invariant LDS addresses out of registers: hoisted, some were spilled)
      fft32_table<false>(re, im, [&](int half, int j) { return tw_lds[(half - 1 + j) * 32 + lt]; });  // stages T = 32..512
    } else {
      fft32(re, im);  // X[l + 32*k2]
    }

    // Epilogue in two halves of 256 bins.  For half P the registers k2 in [8P, 8P+8) hold Z[k] for
    // k = 256P .. 256P+255 and the registers k2 in [24-8P, 32-8P) hold the partners Z[N-k]; both are
    // laid out ascending in k (lbuf[k - 256P], mbuf[k - 256P]) so that four consecutive bins are two
    // aligned 16-byte reads each.  Lane 0's registers sit one row off (k1 = 0), so it also supplies
    // mbuf[0] (Z[N - 256P] = Z[0] or Z[768]).
    const uint64_t row_a = (p.col0 + f0) * p.out_bins, row_b = row_a + p.out_bins;
    // FLT: the pair separation leaves 2*X; the plugin's magnitude is 2/N * |X| (spectrogram.esm.js:195-198)
    // (one frame per transform: its spectrum is Z itself -- the reference's own transform of the frame, the table's
    // asymmetry included -- so there is no partner to read and no factor 2)
    const float mag_scale = SINGLE ? (PLUGIN ? 2.0f / (float)kNPoints : 1.0f) : PLUGIN ? 1.0f / (float)kNPoints : 0.5f;
    float best_a = -1.0f, best_b = -1.0f;  // FEAT: this lane's largest |X|^2 so far and its bin
    float next_a = -1.0f, next_b = -1.0f;  // FEAT: ... and the runner-up (what decides whether float32 can tell)
    int kbest_a = (int)p.bin_lo, kbest_b = (int)p.bin_lo;
    // FEAT, band narrower than the spectrum: the largest |X|^2 of any bin of either frame (what the transform's
    // rounding noise and the pair's cross-talk scale with; see feature_uncertain)
    const bool banded = p.bin_lo != 0 || p.bin_hi != kNPoints / 2;  // wave-uniform
    float seen_a = 0.0f, seen_b = 0.0f;
    float mg_a[BANK ? 16 : 1], mg_b[BANK ? 16 : 1];  // BANK: this lane's magnitudes of both frames, [8P + 4g + B]
    int lq = l;
    // (FEAT: an opaque copy keeps the loop-invariant bin numbers and band tests out of registers the transform needs;
    // hoisted, they were spilled and every reload waited on the memory counter)
    if constexpr (FEAT) asm volatile("" : "+v"(lq));
    // ALONE (REDO flavours, mode 1 / 2): the transform held one frame; its spectrum is Z itself, taken twice so that the
    // quantiser sees the same factor 2 as the pair separation leaves, and there is no second result.
    auto epilogue = [&](auto alone_c) __attribute__((always_inline)) {
    constexpr bool ALONE = decltype(alone_c)::value;
    constexpr bool S = SINGLE || ALONE;  // no partner to read
#pragma unroll
    for (int P = 0; P < 2; ++P) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int kl = 8 * P + r;        // Z[k1 + 32*kl]        -> lbuf[32*r + k1]
        const int km = 24 - 8 * P + r;   // Z[k1 + 32*km] = Z[N - j], j = 32*(7 - r) + 32 - k1 (+256P)
        lbuf[32 * r + l] = make_float2(re[kl], im[kl]);
        if constexpr (!S) mbuf[32 * (7 - r) + 32 - l] = make_float2(re[km], im[km]);
      }
      if constexpr (!S) {
        if (l == 0) mbuf[0] = make_float2(re[P ? 24 : 0], im[P ? 24 : 0]);
      }
      wave_sync();
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int idx = 4 * (lq + 32 * g);
        const int k = 256 * P + idx;
        const float4 z01 = *reinterpret_cast<const float4*>(lbuf + idx);
        const float4 z23 = *reinterpret_cast<const float4*>(lbuf + idx + 2);
        const float4 q01 = S ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4*>(mbuf + idx);
        const float4 q23 = S ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4*>(mbuf + idx + 2);
        uint32_t wa = 0, wb = 0;
        float fa[4] = {0, 0, 0, 0}, fb[4] = {0, 0, 0, 0};
        const bool in_band = k >= (int)p.bin_lo && k < (int)p.bin_hi;
#define SONO_BIN(B, ZR, ZI, PR, PI)                                          \
  {                                                                          \
    const float ar = ALONE ? (ZR) + (ZR) : SINGLE ? (ZR) : (ZR) + (PR);      \
    const float ai = ALONE ? (ZI) + (ZI) : SINGLE ? (ZI) : (ZI) - (PI);      \
    const float br = (ZI) + (PI), bi = (PR) - (ZR);                          \
    const float ma = __builtin_fmaf(ar, ar, ai * ai);                        \
    const float mb = __builtin_fmaf(br, br, bi * bi);                        \
    if constexpr (FEAT) {                                                    \
      fa[B] = ma;                                                            \
      fb[B] = mb;                                                            \
    } else if constexpr (BANK) {                                             \
      mg_a[8 * P + 4 * g + B] = __builtin_amdgcn_sqrtf(ma); /* (1 ulp: an index step is 0.3 dB) */ \
      mg_b[8 * P + 4 * g + B] = __builtin_amdgcn_sqrtf(mb);                  \
    } else if constexpr (FLT) {                                              \
      fa[B] = OUT == kOutDb ? fast_db(ma, fs) : __builtin_sqrtf(ma) * mag_scale; \
      fb[B] = OUT == kOutDb ? fast_db(mb, fs) : __builtin_sqrtf(mb) * mag_scale; \
    } else {                                                                 \
      wa = put_index<PLUGIN, B>(ma, fs, wa);                                 \
      wb = put_index<PLUGIN, B>(mb, fs, wb);                                 \
    }                                                                        \
  }
        SONO_BIN(0, z01.x, z01.y, q01.x, q01.y)
        SONO_BIN(1, z01.z, z01.w, q01.z, q01.w)
        SONO_BIN(2, z23.x, z23.y, q23.x, q23.y)
        SONO_BIN(3, z23.z, z23.w, q23.z, q23.w)
#undef SONO_BIN
        if constexpr (FEAT) {
          if (banded) {
            seen_a = fmaxf(fmaxf(fmaxf(seen_a, fa[0]), fa[1]), fmaxf(fa[2], fa[3]));
            seen_b = fmaxf(fmaxf(fmaxf(seen_b, fb[0]), fb[1]), fmaxf(fb[2], fb[3]));
          }
          if (in_band) {  // (one branch per bin quad; selects inside)
#pragma unroll
            for (int B = 0; B < 4; ++B) {  // a lane visits its bins in ascending order
              next_a = __builtin_amdgcn_fmed3f(fa[B], best_a, next_a);
              next_b = __builtin_amdgcn_fmed3f(fb[B], best_b, next_b);
              const bool ga = fa[B] > best_a, gb = fb[B] > best_b;
              best_a = ga ? fa[B] : best_a; kbest_a = ga ? k + B : kbest_a;
              best_b = gb ? fb[B] : best_b; kbest_b = gb ? k + B : kbest_b;
            }
          }
        }
        if constexpr (FLT) {
          if (in_band) {
            float* o = reinterpret_cast<float*>(p.out) + (k - p.out_bin0);
            if (st_a) *reinterpret_cast<float4*>(o + row_a) = make_float4(fa[0], fa[1], fa[2], fa[3]);
            if (st_b) *reinterpret_cast<float4*>(o + row_b) = make_float4(fb[0], fb[1], fb[2], fb[3]);
          }
        } else if (!FEAT && !BANK && in_band) {
          const uint32_t col = k - p.out_bin0;
          if (st_a) *reinterpret_cast<uint32_t*>(out + row_a + col) = wa;
          if (st_b) *reinterpret_cast<uint32_t*>(out + row_b + col) = wb;
        }
      }
      wave_sync();  // the buffer is rewritten by the next half / the next iteration's transpose
    }
    };  // epilogue
    if (REDO && mode != 0) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if constexpr (BANK) {
      // Both frames' magnitudes go to this half-wavefront's buffer (free between the epilogue and the next transpose),
      // each followed by a zero that a tap at the last bin reads as its neighbour; then every lane gathers the two taps
      // of its 16 output bins per frame.  The tap tables are read from memory (L1-resident, 6 KiB) every iteration.
      float* ma_lds = reinterpret_cast<float*>(fbuf);
      float* mb_lds = ma_lds + 516;
      static_assert((516 + 513) * 4 <= kFftBufBytes, "two frames of magnitudes must fit the transpose buffer");
      uint32_t top_a = 0, top_b = 0;  // largest bit pattern: magnitudes are >= +0, so infinities and NaNs sort highest
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int at = 256 * (j >> 1) + 4 * (l + 32 * (j & 1));
        *reinterpret_cast<float4*>(ma_lds + at) = make_float4(mg_a[4 * j], mg_a[4 * j + 1], mg_a[4 * j + 2], mg_a[4 * j + 3]);
        *reinterpret_cast<float4*>(mb_lds + at) = make_float4(mg_b[4 * j], mg_b[4 * j + 1], mg_b[4 * j + 2], mg_b[4 * j + 3]);
#pragma unroll
        for (int B = 0; B < 4; ++B) {
          top_a = max(top_a, __float_as_uint(mg_a[4 * j + B]));  // (a NaN with its sign bit set sorts higher still)
          top_b = max(top_b, __float_as_uint(mg_b[4 * j + B]));
        }
      }
      if (l == 0) ma_lds[512] = mb_lds[512] = 0.0f;
      wave_sync();
      // one infinite or NaN magnitude turns every output bin of its frame NaN: the reference's row sum visits every
      // bin of the spectrum, and 0 * inf and 0 * NaN are NaN (spectrogram.esm.js:479-488)
      const uint64_t ball_a = __ballot(top_a >= 0x7f800000u), ball_b = __ballot(top_b >= 0x7f800000u);
      const uint32_t keep_a = (uint32_t)(h ? ball_a >> 32 : ball_a) ? 0u : ~0u;  // (NaN quantises to index 0)
      const uint32_t keep_b = (uint32_t)(h ? ball_b >> 32 : ball_b) ? 0u : ~0u;
      int lf = l;
      asm volatile("" : "+v"(lf));  // (the table reads stay inside the loop: hoisted they would hold 48 registers)
      const char* bank = reinterpret_cast<const char*>(p.fb_packed) + 2048;  // tap at -2048, w0 at 0, w1 at +2048
      // the quantiser on a magnitude instead of its square: log2(v^2) = 2 log2(v); a negative sum (negative weights)
      // has no logarithm and quantises to index 0 like the reference's NaN
      FastScale fm = fs;
      fm.a = 2.0f * fs.a;
#pragma unroll 1
      for (int g = 0; g < 4; ++g) {
        const int quad = lf + 32 * g, y0 = 4 * quad;
        const char* at = bank + 16 * quad;
        const int4 tp = *reinterpret_cast<const int4*>(at - 2048);
        const float4 u0 = *reinterpret_cast<const float4*>(at), u1 = *reinterpret_cast<const float4*>(at + 2048);
        const int taps[4] = {tp.x, tp.y, tp.z, tp.w};
        const float c0[4] = {u0.x, u0.y, u0.z, u0.w}, c1[4] = {u1.x, u1.y, u1.z, u1.w};
        uint32_t wa = 0, wb = 0;
#pragma unroll
        for (int B = 0; B < 4; ++B) {
          const float* ta = ma_lds + taps[B];
          const float va = __builtin_fmaf(ta[1], c1[B], ta[0] * c0[B]);
          const float vb = __builtin_fmaf(ta[517], c1[B], ta[516] * c0[B]);
          if (B == 0) { wa = fast_plugin_byte<0>(va, fm, wa); wb = fast_plugin_byte<0>(vb, fm, wb); }
          if (B == 1) { wa = fast_plugin_byte<1>(va, fm, wa); wb = fast_plugin_byte<1>(vb, fm, wb); }
          if (B == 2) { wa = fast_plugin_byte<2>(va, fm, wa); wb = fast_plugin_byte<2>(vb, fm, wb); }
          if (B == 3) { wa = fast_plugin_byte<3>(va, fm, wa); wb = fast_plugin_byte<3>(vb, fm, wb); }
        }
        if (y0 >= (int)p.bin_lo && y0 < (int)p.bin_hi) {
          const uint32_t col = y0 - p.out_bin0;
          if (st_a) *reinterpret_cast<uint32_t*>(out + row_a + col) = wa & keep_a;
          if (st_b) *reinterpret_cast<uint32_t*>(out + row_b + col) = wb & keep_b;
        }
      }
      wave_sync();  // the buffer is rewritten by the next iteration's transpose
    }
    if constexpr (FEAT) {
      // reduce over the 32 lanes of the half-wavefront: larger |X|^2 wins, equal values go to the lower bin
      reduce_peak<32>(best_a, kbest_a, next_a);
      reduce_peak<32>(best_b, kbest_b, next_b);
      if (banded) {
        seen_a = reduce_max<32>(seen_a);
        seen_b = reduce_max<32>(seen_b);
      }
      const float top_a = fmaxf(best_a, seen_a), top_b = fmaxf(best_b, seen_b);  // the frames' largest |X|^2 anywhere
      if (l == 0) {
        sono_frame_feature* ft = reinterpret_cast<sono_frame_feature*>(p.out) + (p.col0 + f0);
        // reserved[0] = 1: float32 cannot tell (two top bins too close, or the peak too close to a step of the
        // quantiser); the exact-arithmetic pass that follows recomputes those records
        const float ta = __builtin_fmaf(__log2f(best_a), fs.a, fs.b), tb = __builtin_fmaf(__log2f(best_b), fs.a, fs.b);
        if (st_a) {
          sono_frame_feature r{};
          r.peak_bin = kbest_a;
          r.peak_idx = (uint8_t)(PLUGIN ? fast_plugin_peak_index(best_a, fs) : put_index<false, 0>(best_a, fs, 0));
          r.reserved[0] = (feature_uncertain(best_a, next_a, top_b, top_a, ta, fs) || peak_outside(kbest_a, p.feat_lo, p.feat_hi)) ? 1 : 0;
          ft[0] = r;
          if (r.reserved[0] && p.feat_queue) {  // (a few frames per thousand)
            const uint32_t at = atomicAdd(p.feat_queue, 1u);
            if (at < p.feat_queue_cap) p.feat_queue[4 + at] = (uint32_t)f0;
          }
        }
        if (st_b) {
          sono_frame_feature r{};
          r.peak_bin = kbest_b;
          r.peak_idx = (uint8_t)(PLUGIN ? fast_plugin_peak_index(best_b, fs) : put_index<false, 0>(best_b, fs, 0));
          r.reserved[0] = (feature_uncertain(best_b, next_b, top_a, top_b, tb, fs) || peak_outside(kbest_b, p.feat_lo, p.feat_hi)) ? 1 : 0;
          ft[1] = r;
          if (r.reserved[0] && p.feat_queue) {
            const uint32_t at = atomicAdd(p.feat_queue, 1u);
            if (at < p.feat_queue_cap) p.feat_queue[4 + at] = (uint32_t)f0 + 1;
          }
        }
      }
    }
    if constexpr (REDO) mode = mode == 1 ? 2 : 0;
    fetch(nxt, REDO && mode == 2);  // unconditional (an inactive unit points at frame 0): no stale values carried round the loop
    cur = nxt;
  }
}

}  // namespace fast1024

// W_1024^(l*k1) as [k1][l] float2 (cos, -sin), computed in double.
void build_fast_table_1024(std::vector<float>& t) {
  t.resize(2 * 1024);
  const double pi = 3.141592653589793;
  for (int k1 = 0; k1 < 32; ++k1)
    for (int l = 0; l < 32; ++l) {
      const double ang = -2.0 * pi * (double)(k1 * l) / 1024.0;
      t[2 * (k1 * 32 + l)] = (float)std::cos(ang);
      t[2 * (k1 * 32 + l) + 1] = (float)std::sin(ang);
    }
}

// fast_mode: 0 = no fast kernel for this context, 1 = exact roots of unity (fast_table holds the
// W_1024^(l*k1) matrix), 2 = the context's per-stage twiddle table (p.twiddles, float2).
static bool fast_base_available(const StftParams& p) {
  if (p.n != 1024 || p.fast_mode == 0) return false;
  if (p.fast_mode == 1 && !p.fast_table) return false;
  if (p.fast_mode == 2 && !p.twiddles) return false;
  if (p.nframes < 16) return false;  // a handful of frames (e.g. the worker's wrapped last frame): general kernel
  return true;
}

bool fast_u8_available(const StftParams& p) {
  if (!fast_base_available(p) || (!p.pair_idx_ok && p.fast_mode != 2)) return false;  // (table flavour: one frame per transform then)
  if ((p.bin_lo & 3) || (p.bin_hi & 3) || (p.out_bins & 3) || (p.out_bin0 & 3)) return false;
  return true;
}

bool fast_float_available(const StftParams& p) {
  return fast_base_available(p) && !(p.bin_lo & 3) && !(p.bin_hi & 3) && !(p.out_bins & 3) && !(p.out_bin0 & 3) &&
         !(reinterpret_cast<uintptr_t>(p.out) & 15);
}

bool fast_features_available(const StftParams& p) {
  return fast_base_available(p) && !(p.bin_lo & 3) && !(p.bin_hi & 3) && p.bin_lo < p.bin_hi;
}

// The fused image output needs the flipped full-height row mapping (any band: rows outside it are masked) and store
// offsets (a few image rows) that fit 32 bits.
bool fast_rgba_available(const StftParams& p) {
  if (!fast_base_available(p) || !p.pair_idx_ok || p.fb_packed) return false;
  if (p.bin_lo >= p.bin_hi) return false;
  if (p.row_of_bin0 != (int32_t)p.bin_hi - 1) return false;
  if (p.width == 0 || p.width > ((uint64_t)1 << 24) || !p.lut) return false;
  return true;
}

// Scratch of the image writers that exchange colour indices through global memory (2048 and 4096 points): two slots
// per workgroup, a slot of the 2048-point kernel being 1024 bins x 256 frames.
uint64_t fast_rgba_scratch_bytes() { return (uint64_t)device_cu_count() * 2 * 2 * rgba_flush::kRoundBytes; }

int launch_stft_fast(const StftParams& p, void* stream_v) {
  using namespace fast1024;
  const bool rgba = p.out_kind == SONO_OUT_RGBA8, feat = p.out_kind == SONO_OUT_FEATURES;
  const bool flt = p.out_kind == SONO_OUT_F32_DB || p.out_kind == SONO_OUT_F32_MAG;
  // one frame per transform where two would not do: colour indices under a table whose cross-talk could show
  // (StftParams::pair_idx_ok), float32 outputs under any asymmetric table (StftParams::pair_exact)
  const bool single = p.fast_mode == 2 && ((p.out_kind == SONO_OUT_U8_INDEX && !p.pair_idx_ok) || (flt && !p.pair_exact));
  // a non-linear frequency scale: colour indices only, plugin quantiser, paired frames
  const bool bank = p.fb_packed != nullptr;
  if (bank && (p.out_kind != SONO_OUT_U8_INDEX || p.scale_mode != SONO_SCALE_PLUGIN_DB)) return 1;
  if (rgba) {
    if (!fast_rgba_available(p)) return 1;
    return launch_stft_fast_rgba(p, stream_v);  // the image writer is a kernel of its own (kernels_fast_rgba.hip)
  } else if (flt) {
    if (!fast_float_available(p)) return 1;
  } else if (feat) {
    if (!fast_features_available(p)) return 1;
  } else {
    if (p.out_kind != SONO_OUT_U8_INDEX || !fast_u8_available(p)) return 1;
  }
  if (reinterpret_cast<uintptr_t>(p.out) & 3) return 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  FastScale fs = make_fast_scale(p.scale_mode, p.n, p.gain_db, p.range_db);
  // the pair separation leaves 2*X: |2X|^2 = 4|X|^2, i.e. log2 is 2 too large
  if (!single) {
    fs.b -= 2.0f * fs.a;
    fs.db_off -= 2.0f * 3.0102999566f;
  }
  fs.xtalk = (float)p.pair_asym;
  const uint64_t npairs = single ? p.nframes : (p.nframes + 1) / 2;  // transforms
  const uint64_t per_wg = (uint64_t)kMaxWaves * 2;  // one wavefront handles 2 transforms per iteration
  const uint64_t need = (npairs + per_wg - 1) / per_wg;
  const int cus = device_cu_count();
  const uint32_t grid = (uint32_t)(need < (uint64_t)cus ? need : (uint64_t)cus);  // persistent: one workgroup per CU
  const float2* table = reinterpret_cast<const float2*>(p.fast_table);
  const int lds_total = lds_bytes(kMaxWaves);
  const bool tab = p.fast_mode == 2;
#define SONO_LAUNCH_T(PLUGIN, In, H, O, T)                                                                  \
  do {                                                                                                      \
    auto kern = stft1024_u8_kernel<PLUGIN, In, H, O, T>;                                                    \
    SONO_GRANT_LDS(kern, lds_total);                                                                        \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kMaxWaves * 64), lds_total, stream, p, fs, table);            \
  } while (0)
#define SONO_LAUNCH_O(PLUGIN, In, H, O)                                                                     \
  do {                                                                                                      \
    if (tab) SONO_LAUNCH_T(PLUGIN, In, H, O, true); else SONO_LAUNCH_T(PLUGIN, In, H, O, false);            \
  } while (0)
#define SONO_LAUNCH_H(PLUGIN, In, H)                                                                        \
  do {                                                                                                      \
    if (feat) SONO_LAUNCH_O(PLUGIN, In, H, kOutFeatures);                                              \
    else if (p.out_kind == SONO_OUT_F32_DB) SONO_LAUNCH_O(PLUGIN, In, H, kOutDb);                           \
    else if (p.out_kind == SONO_OUT_F32_MAG) SONO_LAUNCH_O(PLUGIN, In, H, kOutMag);                         \
    else if (bank) SONO_LAUNCH_O(true, In, H, kOutU8Bank);                                                  \
    else SONO_LAUNCH_O(PLUGIN, In, H, kOutU8);                                                              \
  } while (0)
#define SONO_LAUNCH(PLUGIN, In)                                                                             \
  do {                                                                                                      \
    if (single) {                                                                                           \
      if (p.out_kind == SONO_OUT_F32_DB) SONO_LAUNCH_T(PLUGIN, In, -1, kOutDb, true);                       \
      else if (p.out_kind == SONO_OUT_F32_MAG) SONO_LAUNCH_T(PLUGIN, In, -1, kOutMag, true);                \
      else if (bank) SONO_LAUNCH_T(true, In, -1, kOutU8Bank, true);                                         \
      else SONO_LAUNCH_T(PLUGIN, In, -1, kOutU8, true);                                                     \
    } else if (p.hop == 256) SONO_LAUNCH_H(PLUGIN, In, 8);                                                  \
    else if (p.hop == 512) SONO_LAUNCH_H(PLUGIN, In, 16);                                                   \
    else SONO_LAUNCH_H(PLUGIN, In, 0);                                                                      \
  } while (0)
  const bool plugin = p.scale_mode == SONO_SCALE_PLUGIN_DB;
  if (p.in_kind == SONO_IN_I16) {
    if (plugin) SONO_LAUNCH(true, int16_t); else SONO_LAUNCH(false, int16_t);
  } else {
    if (plugin) SONO_LAUNCH(true, float); else SONO_LAUNCH(false, float);
  }
#undef SONO_LAUNCH
#undef SONO_LAUNCH_H
#undef SONO_LAUNCH_O
#undef SONO_LAUNCH_T
  // (redo: a flavour that pairs frames under the table and transforms its loud units one frame at a time)
  const bool redo = tab && !single && (bank || p.out_kind == SONO_OUT_U8_INDEX) && p.loud_thr_raw > 0.0f;
  note_kernel(single ? (bank ? "stft1024_u8_kernel<table,single,bank>" : "stft1024_u8_kernel<table,single>")
              : bank ? (tab ? (redo ? "stft1024_u8_kernel<table,bank,redo>" : "stft1024_u8_kernel<table,bank>") : "stft1024_u8_kernel<roots,bank>")
              : tab  ? (redo ? "stft1024_u8_kernel<table,redo>" : "stft1024_u8_kernel<table>")
                     : "stft1024_u8_kernel<roots>");
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

}  // namespace sono
