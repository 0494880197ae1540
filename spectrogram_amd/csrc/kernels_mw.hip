// Float32 STFT kernel for 512-, 2048- and 4096-point frames: a group of N/16 threads transforms one
// pair of real frames per iteration; a workgroup is one group (2048: 2 wavefronts, 4096: 4) or, at
// 512 points, eight groups of 32 threads.
//
// Dataflow = the reference's radix-2 decimation in time (spectrogramWorker.js:206-253,
// spectrogram.esm.js:166-200) with its per-stage twiddle TABLE (exact roots for the worker, the
// recurrence factors for the plugin, or a host table), regrouped into three register passes:
//   position k = [c | b | a]   (a: low 4 bits, b: next 4 bits, c: the top log2N - 8 bits)
//   pass A: stages T = 1..8      over a, thread = (c, b)      factors uniform in the workgroup
//   pass B: stages T = 16..128   over b, thread = (c, a)      factor 16*(b mod h) + a of stage 16h
//   pass C: stages T = 256..N/2  over c, thread = (b, a)      factor 256*(c mod h) + 16b + a of stage 256h
// so every butterfly multiplies by the very table entry the reference uses, and the data crosses LDS
// twice instead of log2N times.  Thread t loads samples t + (N/16)*a' (coalesced) into register
// a = rev4(a'), which is the bit-reversed load.  Two real frames share one complex transform
// (z = w*(xA + i*xB); XA = (Z[k] + conj Z[N-k])/2, XB = (Z[k] - conj Z[N-k])/(2i)); a third LDS round
// trip brings Z[k] and Z[N-k] together, and the epilogue fuses |X|^2 -> log2 -> quantiser -> packed uint8.
//
// LDS: two planes (re, im) of N floats, cell(k) = k + (k >> 5).  A 4-byte LDS access is served in two
// groups of 32 lanes over 32 banks; with one pad per 32 cells the stride-16 writes of pass A, the
// (a, c) reads of pass B (lanes 16-31 take c + 2, i.e. 528 = 16 mod 32 cells further), the unit-stride
// accesses of pass C and the stride-4 reads of the epilogue all fall on 32 distinct banks.  33 KiB at
// N = 4096: four workgroups per CU, each on its own barrier, so the CU always has wavefronts in a
// different phase.
#include <hip/hip_runtime.h>

#include "sono_device.h"
#include "sono_internal.h"
#include "sono_launch.h"

namespace sono {
namespace mw {

__host__ __device__ constexpr int rev4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); }
__device__ __forceinline__ int pad(int k) { return k + (k >> 5); }

// R-point decimation-in-time network on registers in position order.  tw(half, j): factor (cos, sin) of
// the butterfly (i, i + half), j = i mod half.  FIRST: j == 0 is the factor (1, 0) exactly.
template <int R, bool FIRST, class TwFn>
__device__ __forceinline__ void dit(float* re, float* im, TwFn tw) {
#pragma unroll
  for (int half = 1; half < R; half <<= 1) {
#pragma unroll
    for (int j = 0; j < half; ++j) {
      float2 w = make_float2(1.0f, 0.0f);
      if (!FIRST || j != 0) w = tw(half, j);
#pragma unroll
      for (int g = 0; g < R; g += 2 * half) {
        const int i0 = g + j, i1 = i0 + half;
        const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
        if (FIRST && j == 0) {
          re[i0] = ar + br; im[i0] = ai + bi;
          re[i1] = ar - br; im[i1] = ai - bi;
        } else {  // o1 = a + w*b, o2 = a - w*b = 2a - o1
          const float o1r = __builtin_fmaf(-w.y, bi, __builtin_fmaf(w.x, br, ar));
          const float o1i = __builtin_fmaf(w.y, br, __builtin_fmaf(w.x, bi, ai));
          re[i0] = o1r; im[i0] = o1i;
          re[i1] = __builtin_fmaf(2.0f, ar, -o1r);
          im[i1] = __builtin_fmaf(2.0f, ai, -o1i);
        }
      }
    }
  }
}

template <typename In>
__device__ __forceinline__ float ld(const In* p, uint64_t i);
template <>
__device__ __forceinline__ float ld<float>(const float* p, uint64_t i) { return p[i]; }
template <>
__device__ __forceinline__ float ld<int16_t>(const int16_t* p, uint64_t i) { return (float)p[i] * (1.0f / 32768.0f); }

template <bool PLUGIN, int BYTE>
__device__ __forceinline__ uint32_t put_index(float msq, const FastScale& fs, uint32_t acc) {
  if constexpr (PLUGIN) {
    return fast_plugin_byte<BYTE>(msq, fs, acc);
  } else {
    return fast_worker_byte<BYTE>(msq, fs, acc);
  }
}

// Synchronises the threads of one transform.  A transform of up to 64 threads lives inside one wavefront, whose
// LDS instructions execute in issue order: ordering them for the compiler is enough, and the wavefronts of the
// workgroup stay independent.  Larger transforms span wavefronts and need the workgroup barrier (G == 1 there).
template <int TN>
__device__ __forceinline__ void group_sync() {
  if constexpr (TN <= 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// M > 0: hop == M * N/16, so frame B's sample t + (N/16)*a' is frame A's sample t + (N/16)*(a' + M) and one
// run of 16 + M loads feeds both frames (75 %, 87.5 % and 93.75 % overlap are M = 4, 2, 1).  M == 0: any hop.
// M < 0: ONE real frame per transform (imaginary part zero; its spectrum is Z itself, i.e. the reference's own
// transform of the frame), for twiddle tables under which a frame would leave a visible trace in a partner
// (StftParams::pair_idx_ok) or float32 outputs would carry it (pair_exact).
// OUT: kOutU8 [frame][bin] colour indices; kOutDb / kOutMag [frame][bin] float32 20*log10(m) / m; kOutFeatures
// [frame] spectral peak (strict-> first-wins argmax of |X|^2 over the band and that bin's colour index).
// kOutU8Bank: colour indices on a non-linear frequency scale: out[y] = m[tap[y]] * w0[y] + m[tap[y] + 1] * w1[y] over the
// frame's magnitudes (spectrogram.esm.js:479-488 with the two non-zero weights of a row), then the quantiser.  The
// epilogue leaves the magnitudes in the planes (frame A in the real one, frame B in the imaginary one, each in the cells
// of the bins below N/2, which only their own thread has read) and a second sweep gathers the taps.
constexpr int kOutU8 = 0, kOutFeatures = 2, kOutDb = 3, kOutMag = 4, kOutU8Bank = 5;
template <int LOG2N, int G, int M, bool PLUGIN, typename In, int OUT>
__global__ __launch_bounds__((1 << LOG2N) / 16 * G, 4) void stft_mw_kernel(StftParams p, FastScale fs) {
  constexpr int N = 1 << LOG2N, TN = N / 16, CB = LOG2N - 8, NC = 1 << CB, SETS = 16 / NC;
  constexpr bool FEAT = OUT == kOutFeatures, FLT = OUT == kOutDb || OUT == kOutMag;
  constexpr bool SINGLE = M < 0, BANK = OUT == kOutU8Bank;
  __shared__ int s_bad[BANK ? G * 2 : 1];  // BANK: a frame of the group's pair has an infinite or NaN magnitude
  static_assert(!(SINGLE && FEAT), "the feature records come from the paired flavour (its marks cover the cross-talk)");
  constexpr int kWavesPerGroup = TN > 64 ? TN / 64 : 1;
  __shared__ float s_best[FEAT ? G * kWavesPerGroup * 2 : 1];   // FEAT, transforms wider than a wavefront:
  __shared__ int s_bestk[FEAT ? G * kWavesPerGroup * 2 : 1];    // per-wavefront peaks of the two frames
  __shared__ float s_next[FEAT ? G * kWavesPerGroup * 2 : 1];   // ... and their runners-up
  __shared__ float s_zmax[FEAT ? G * kWavesPerGroup : 1];       // ... and the largest component of the pair's transform
  static_assert(CB >= 1 && CB <= 4, "512 to 4096 points");
  constexpr int kPlane = N + N / 32;
  __shared__ float s_re_all[G][kPlane], s_im_all[G][kPlane];
  const int grp = threadIdx.x / TN, t = threadIdx.x % TN;  // group = one transform
  float* s_re = s_re_all[grp];
  float* s_im = s_im_all[grp];
  const float2* __restrict__ tw = reinterpret_cast<const float2*>(p.twiddles);
  const In* __restrict__ x = reinterpret_cast<const In*>(p.pcm);
  uint8_t* __restrict__ out = reinterpret_cast<uint8_t*>(p.out);

  // pass A identity: t = rev4(b) * NC + rev_CB(c)
  const int cA = (int)(__brev((unsigned)(t & (NC - 1))) >> (32 - CB)), bA = rev4(t >> CB);
  const int kA = (cA * 16 + bA) * 16;  // position of register a = 0
  // pass B identity: thread (c, a)
  const int aB = t & 15;
  const int cB = NC >= 4 ? ((t >> 4) & ~3) | (((t >> 4) & 1) << 1) | (((t >> 4) >> 1) & 1)  // bits 0, 1 of c swapped
                         : (t >> 4);
  // pass C identity: thread value w = 16b + a (two of them, w and w + 128, at N = 2048)
  float win[16];
#pragma unroll
  for (int a1 = 0; a1 < 16; ++a1) win[rev4(a1)] = p.window[t + TN * a1];

  const uint64_t npairs = SINGLE ? p.nframes : (p.nframes + 1) >> 1;  // transforms
  for (uint64_t base = (uint64_t)blockIdx.x * G; base < npairs; base += (uint64_t)gridDim.x * G) {
    // every group of the workgroup goes through every iteration (the larger sizes meet at barriers); one
    // without a pair of its own repeats the last pair and stores nothing
    const bool active = base + grp < npairs;
    const uint64_t pair = active ? base + grp : npairs - 1;
    uint64_t f0 = SINGLE ? pair : 2 * pair;
    bool st_a = active;
    if (!SINGLE && f0 + 1 >= p.nframes) {  // odd last frame: computed as the second member of the pair before it
      f0 -= 1;
      st_a = false;
    }
    const In* xa = x + (p.frame0 + f0) * p.hop + t;
    float re[16], im[16];
    if constexpr (SINGLE) {
#pragma unroll
      for (int a1 = 0; a1 < 16; ++a1) {
        re[rev4(a1)] = ld<In>(xa, (uint64_t)TN * a1);
        im[rev4(a1)] = 0.0f;
      }
    } else if constexpr (M > 0) {
      float raw[16 + M];
#pragma unroll
      for (int a1 = 0; a1 < 16 + M; ++a1) raw[a1] = ld<In>(xa, (uint64_t)TN * a1);
#pragma unroll
      for (int a1 = 0; a1 < 16; ++a1) {
        re[rev4(a1)] = raw[a1];
        im[rev4(a1)] = raw[a1 + M];
      }
    } else {
      const In* xb = xa + p.hop;
#pragma unroll
      for (int a1 = 0; a1 < 16; ++a1) {
        re[rev4(a1)] = ld<In>(xa, (uint64_t)TN * a1);
        im[rev4(a1)] = ld<In>(xb, (uint64_t)TN * a1);
      }
    }
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      re[a] *= win[a];
      im[a] *= win[a];
    }
    dit<16, true>(re, im, [&](int half, int j) { return tw[half - 1 + j]; });
    // LDS cells as one base per thread plus compile-time offsets
    const int cellA = kA + (kA >> 5);          // + a            (kA is a multiple of 16: a never carries into k >> 5)
    const int cellB = cB * 264 + aB;           // + 16 * b + (b >> 1)
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      s_re[cellA + a] = re[a];
      s_im[cellA + a] = im[a];
    }
    group_sync<TN>();
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      re[b] = s_re[cellB + 16 * b + (b >> 1)];
      im[b] = s_im[cellB + 16 * b + (b >> 1)];
    }
    // (the twiddle reads of passes B and C are invariant in this loop; the empty asm keeps them here, out
    // of 60 registers that hoisted copies would hold)
    int a_tw = aB;
    asm volatile("" : "+v"(a_tw));
    dit<16, false>(re, im, [&](int half, int j) { return tw[16 * half - 1 + 16 * j + a_tw]; });
#pragma unroll
    for (int b = 0; b < 16; ++b) {  // each thread rewrites exactly the cells it read
      s_re[cellB + 16 * b + (b >> 1)] = re[b];
      s_im[cellB + 16 * b + (b >> 1)] = im[b];
    }
    group_sync<TN>();
    float zmax = 0.0f;
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
      const int w = t + TN * s;
      const int cellC = w + (w >> 5);          // + 264 * c
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        re[s * NC + c] = s_re[cellC + 264 * c];
        im[s * NC + c] = s_im[cellC + 264 * c];
      }
      int w_tw = w;
      asm volatile("" : "+v"(w_tw));
      dit<NC, false>(re + s * NC, im + s * NC, [&](int half, int j) { return tw[256 * half - 1 + 256 * j + w_tw]; });
#pragma unroll
      for (int c = 0; c < NC; ++c) {  // X[c*256 + w], again in the thread's own cells
        s_re[cellC + 264 * c] = re[s * NC + c];
        s_im[cellC + 264 * c] = im[s * NC + c];
        // FEAT: the largest component of the transform (what its rounding noise scales with; see feature_uncertain)
        if constexpr (FEAT) zmax = fmaxf(zmax, fmaxf(fabsf(re[s * NC + c]), fabsf(im[s * NC + c])));
      }
    }
    if constexpr (BANK) {
      if (t == 0) s_bad[2 * grp] = s_bad[2 * grp + 1] = 0;
    }
    group_sync<TN>();
    // epilogue: thread handles the bin quads q = t, t + TN (N/8 quads per frame)
    const uint64_t row_a = (p.col0 + f0) * p.out_bins, row_b = row_a + p.out_bins;
    // the pair separation leaves 2*X; plugin: 2/N * |X|
    const float mag_scale = SINGLE ? (PLUGIN ? 2.0f / (float)N : 1.0f) : PLUGIN ? 1.0f / (float)N : 0.5f;
    float best_a = -1.0f, best_b = -1.0f;                     // FEAT: this thread's largest |X|^2 so far and its bin
    int kbest_a = (int)p.bin_lo, kbest_b = (int)p.bin_lo;
    float next_a = -1.0f, next_b = -1.0f;                     // FEAT: the runner-up (decides whether float32 can tell)
#pragma unroll
    for (int m = 0; m < (N / 8) / TN; ++m) {
      const int k = 4 * (t + TN * m);
      if (BANK || (k >= (int)p.bin_lo && k < (int)p.bin_hi)) {  // (BANK: a tap may lie outside the band)
        uint32_t wa = 0, wb = 0;
        float fa[4] = {0, 0, 0, 0}, fb[4] = {0, 0, 0, 0};
        const int cellZ = k + (k >> 5);  // + B (k is a multiple of 4)
#define SONO_BIN(B)                                                            \
  {                                                                            \
    const int km = (N - k - B) & (N - 1);                                      \
    const float zr = s_re[cellZ + B], zi = s_im[cellZ + B];                    \
    const float pr = SINGLE ? 0.0f : s_re[pad(km)], pi = SINGLE ? 0.0f : s_im[pad(km)]; \
    const float ar = zr + pr, ai = zi - pi;                                    \
    const float br = zi + pi, bi = pr - zr;                                    \
    const float ma = __builtin_fmaf(ar, ar, ai * ai);                          \
    const float mb = __builtin_fmaf(br, br, bi * bi);                          \
    if constexpr (FEAT) { /* a thread visits its bins in ascending order */    \
      next_a = __builtin_amdgcn_fmed3f(ma, best_a, next_a);                    \
      next_b = __builtin_amdgcn_fmed3f(mb, best_b, next_b);                    \
      if (ma > best_a) { best_a = ma; kbest_a = k + B; }                       \
      if (mb > best_b) { best_b = mb; kbest_b = k + B; }                       \
    } else if constexpr (BANK) {                                               \
      fa[B] = __builtin_amdgcn_sqrtf(ma); /* (1 ulp: an index step is 0.3 dB) */ \
      fb[B] = __builtin_amdgcn_sqrtf(mb);                                      \
    } else if constexpr (FLT) {                                                \
      fa[B] = OUT == kOutDb ? fast_db(ma, fs) : __builtin_sqrtf(ma) * mag_scale; \
      fb[B] = OUT == kOutDb ? fast_db(mb, fs) : __builtin_sqrtf(mb) * mag_scale; \
    } else {                                                                   \
      wa = put_index<PLUGIN, B>(ma, fs, wa);                                   \
      wb = put_index<PLUGIN, B>(mb, fs, wb);                                   \
    }                                                                          \
  }
        SONO_BIN(0) SONO_BIN(1) SONO_BIN(2) SONO_BIN(3)
#undef SONO_BIN
        const uint32_t col = k - p.out_bin0;
        if constexpr (BANK) {
          // magnitudes are >= +0, so as bit patterns infinities and NaNs (of either sign) sort highest
          uint32_t top_a = 0, top_b = 0;
#pragma unroll
          for (int B = 0; B < 4; ++B) {
            s_re[cellZ + B] = fa[B];
            if (!SINGLE) s_im[cellZ + B] = fb[B];
            top_a = max(top_a, __float_as_uint(fa[B]));
            top_b = max(top_b, __float_as_uint(fb[B]));
          }
          if (top_a >= 0x7f800000u) s_bad[2 * grp] = 1;
          if (!SINGLE && top_b >= 0x7f800000u) s_bad[2 * grp + 1] = 1;
        } else
        if constexpr (FLT) {
          float* o = reinterpret_cast<float*>(p.out) + col;
          if (st_a) *reinterpret_cast<float4*>(o + row_a) = make_float4(fa[0], fa[1], fa[2], fa[3]);
          if (!SINGLE && active) *reinterpret_cast<float4*>(o + row_b) = make_float4(fb[0], fb[1], fb[2], fb[3]);
        } else if constexpr (!FEAT) {
          if (st_a) *reinterpret_cast<uint32_t*>(out + row_a + col) = wa;
          if (!SINGLE && active) *reinterpret_cast<uint32_t*>(out + row_b + col) = wb;
        }
      }
    }
    if constexpr (BANK) {
      if (t == 0) s_re[pad(N / 2)] = s_im[pad(N / 2)] = 0.0f;  // what a tap at the last bin reads as its neighbour
      group_sync<TN>();
      // one infinite or NaN magnitude turns every output bin of its frame NaN (index 0): the reference's row sum visits
      // every bin of the spectrum, and 0 * inf and 0 * NaN are NaN
      const uint32_t keep_a = s_bad[2 * grp] ? 0u : ~0u, keep_b = s_bad[2 * grp + 1] ? 0u : ~0u;
      FastScale fm = fs;
      fm.a = 2.0f * fs.a;  // the quantiser on a magnitude instead of its square: log2(v^2) = 2 log2(v)
      const char* bank = reinterpret_cast<const char*>(p.fb_packed);
#pragma unroll 1
      for (int m = 0; m < (N / 8) / TN; ++m) {
        int quad = t + TN * m;
        asm volatile("" : "+v"(quad));  // (the table reads stay inside the frame loop)
        const int y0 = 4 * quad;
        const int4 tp = *reinterpret_cast<const int4*>(bank + 16 * quad);
        const float4 u0 = *reinterpret_cast<const float4*>(bank + 2 * N + 16 * quad);
        const float4 u1 = *reinterpret_cast<const float4*>(bank + 4 * N + 16 * quad);
        const int taps[4] = {tp.x, tp.y, tp.z, tp.w};
        const float c0[4] = {u0.x, u0.y, u0.z, u0.w}, c1[4] = {u1.x, u1.y, u1.z, u1.w};
        uint32_t wa = 0, wb = 0;
#pragma unroll
        for (int B = 0; B < 4; ++B) {
          const int q0 = pad(taps[B]), q1 = pad(taps[B] + 1);
          const float va = __builtin_fmaf(s_re[q1], c1[B], s_re[q0] * c0[B]);
          const float vb = SINGLE ? 0.0f : __builtin_fmaf(s_im[q1], c1[B], s_im[q0] * c0[B]);
          if (B == 0) { wa = fast_plugin_byte<0>(va, fm, wa); wb = fast_plugin_byte<0>(vb, fm, wb); }
          if (B == 1) { wa = fast_plugin_byte<1>(va, fm, wa); wb = fast_plugin_byte<1>(vb, fm, wb); }
          if (B == 2) { wa = fast_plugin_byte<2>(va, fm, wa); wb = fast_plugin_byte<2>(vb, fm, wb); }
          if (B == 3) { wa = fast_plugin_byte<3>(va, fm, wa); wb = fast_plugin_byte<3>(vb, fm, wb); }
        }
        if (y0 >= (int)p.bin_lo && y0 < (int)p.bin_hi) {
          const uint32_t col = y0 - p.out_bin0;
          if (st_a) *reinterpret_cast<uint32_t*>(out + row_a + col) = wa & keep_a;
          if (!SINGLE && active) *reinterpret_cast<uint32_t*>(out + row_b + col) = wb & keep_b;
        }
      }
    }
    if constexpr (FEAT) {
      // larger |X|^2 wins, equal values go to the lower bin: first across the lanes of the transform's wavefront
      // (or its 32-lane half at 512 points), then across its wavefronts through LDS
      auto take = [](float& bm, int& bk, float& nx, float om, int ok, float on) {
        nx = __builtin_fmaxf(__builtin_fminf(bm, om), __builtin_fmaxf(nx, on));
        if (om > bm || (om == bm && ok < bk)) { bm = om; bk = ok; }
      };
#pragma unroll
      for (int msk = 1; msk < (TN < 64 ? TN : 64); msk <<= 1) {
        take(best_a, kbest_a, next_a, __shfl_xor(best_a, msk, 64), __shfl_xor(kbest_a, msk, 64), __shfl_xor(next_a, msk, 64));
        take(best_b, kbest_b, next_b, __shfl_xor(best_b, msk, 64), __shfl_xor(kbest_b, msk, 64), __shfl_xor(next_b, msk, 64));
        zmax = fmaxf(zmax, __shfl_xor(zmax, msk, 64));
      }
      if constexpr (kWavesPerGroup > 1) {
        const int wv = t >> 6;
        if ((t & 63) == 0) {
          s_zmax[grp * kWavesPerGroup + wv] = zmax;
          s_best[(grp * kWavesPerGroup + wv) * 2] = best_a;     s_bestk[(grp * kWavesPerGroup + wv) * 2] = kbest_a;
          s_best[(grp * kWavesPerGroup + wv) * 2 + 1] = best_b; s_bestk[(grp * kWavesPerGroup + wv) * 2 + 1] = kbest_b;
          s_next[(grp * kWavesPerGroup + wv) * 2] = next_a;     s_next[(grp * kWavesPerGroup + wv) * 2 + 1] = next_b;
        }
        __syncthreads();
        if (t == 0) {
#pragma unroll
          for (int v = 1; v < kWavesPerGroup; ++v) {
            take(best_a, kbest_a, next_a, s_best[(grp * kWavesPerGroup + v) * 2], s_bestk[(grp * kWavesPerGroup + v) * 2],
                 s_next[(grp * kWavesPerGroup + v) * 2]);
            take(best_b, kbest_b, next_b, s_best[(grp * kWavesPerGroup + v) * 2 + 1], s_bestk[(grp * kWavesPerGroup + v) * 2 + 1],
                 s_next[(grp * kWavesPerGroup + v) * 2 + 1]);
            zmax = fmaxf(zmax, s_zmax[grp * kWavesPerGroup + v]);
          }
        }
      }
      if (t == 0) {
        // reserved[0] = 1: float32 cannot tell; the exact-arithmetic pass that follows recomputes the record
        sono_frame_feature* ft = reinterpret_cast<sono_frame_feature*>(p.out) + (p.col0 + f0);
        const float ta = __builtin_fmaf(__log2f(best_a), fs.a, fs.b), tb = __builtin_fmaf(__log2f(best_b), fs.a, fs.b);
        // either frame's largest |2X|^2 anywhere is at most |2Z|^2 <= 4 (re^2 + im^2) <= 8 zmax^2
        const float top = fmaxf(fmaxf(best_a, best_b), 8.0f * zmax * zmax);
        if (st_a) {
          sono_frame_feature r{};
          r.peak_bin = kbest_a;
          r.peak_idx = (uint8_t)(PLUGIN ? fast_plugin_peak_index(best_a, fs) : put_index<false, 0>(best_a, fs, 0));
          r.reserved[0] = (feature_uncertain(best_a, next_a, top, top, ta, fs) || peak_outside(kbest_a, p.feat_lo, p.feat_hi)) ? 1 : 0;
          ft[0] = r;
        }
        if (active) {
          sono_frame_feature r{};
          r.peak_bin = kbest_b;
          r.peak_idx = (uint8_t)(PLUGIN ? fast_plugin_peak_index(best_b, fs) : put_index<false, 0>(best_b, fs, 0));
          r.reserved[0] = (feature_uncertain(best_b, next_b, top, top, tb, fs) || peak_outside(kbest_b, p.feat_lo, p.feat_hi)) ? 1 : 0;
          ft[1] = r;
        }
      }
    }
    group_sync<TN>();  // the planes are rewritten by the next pair
  }
}

}  // namespace mw

bool mw_u8_available(const StftParams& p) {
  if ((p.n != 512 && p.n != 2048 && p.n != 4096) || !p.mw_ok || !p.twiddles) return false;  // (pairs that would leak: one frame per transform)
  if ((p.bin_lo & 3) || (p.bin_hi & 3) || (p.out_bins & 3) || (p.out_bin0 & 3)) return false;
  if (p.nframes < 2) return false;
  return true;
}

int launch_stft_mw(const StftParams& p, void* stream_v) {
  using namespace mw;
  const bool u8 = p.out_kind == SONO_OUT_U8_INDEX, feat = p.out_kind == SONO_OUT_FEATURES;
  const bool flt = p.out_kind == SONO_OUT_F32_DB || p.out_kind == SONO_OUT_F32_MAG;
  if (!(u8 || feat || flt)) return 1;
  // one frame per transform where two would not do: float32 outputs under an asymmetric table (StftParams::pair_exact),
  // colour indices under a table whose cross-talk could show (StftParams::pair_idx_ok)
  // (this kernel's pairing flavour does not redo loud pairs: where the caller relies on that, StftParams::loud_thr_raw, it
  // transforms one frame at a time)
  const bool single = (flt && !p.pair_exact) || (u8 && (!p.pair_idx_ok || p.loud_thr_raw > 0.0f));
  // a non-linear frequency scale: colour indices only, plugin quantiser
  const bool bank = p.fb_packed != nullptr;
  if (bank && (!u8 || p.scale_mode != SONO_SCALE_PLUGIN_DB)) return 1;
  if (feat) {  // features do not use the output row layout
    if ((p.n != 512 && p.n != 2048 && p.n != 4096) || !p.mw_ok || !p.twiddles || (p.bin_lo & 3) || (p.bin_hi & 3) ||
        p.bin_lo >= p.bin_hi || p.nframes < 2)
      return 1;
  } else if (!mw_u8_available(p)) {
    return 1;
  }
  if (reinterpret_cast<uintptr_t>(p.out) & (flt ? 15 : 3)) return 1;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  FastScale fs = make_fast_scale(p.scale_mode, p.n, p.gain_db, p.range_db);
  if (!single) {
    fs.b -= 2.0f * fs.a;  // the pair separation leaves 2*X
    fs.db_off -= 2.0f * 3.0102999566f;
  }
  fs.xtalk = (float)p.pair_asym;
  const uint64_t npairs = single ? p.nframes : (p.nframes + 1) / 2;  // transforms
  const int cus = device_cu_count();
  const int groups = p.n == 512 ? 8 : 1;  // transforms per workgroup
  const uint64_t need = (npairs + groups - 1) / groups;
  const uint64_t cap = (uint64_t)cus * (p.n == 2048 ? 8 : 4);
  const uint32_t grid = (uint32_t)(need < cap ? need : cap);
  const bool plugin = p.scale_mode == SONO_SCALE_PLUGIN_DB;
  const uint32_t tn = p.n / 16;
  // shared sample loads (hop a small multiple of N/16) exist for the colour-index output only
  const int m = (u8 && p.hop % tn == 0 && (p.hop / tn == 1 || p.hop / tn == 2 || p.hop / tn == 4)) ? (int)(p.hop / tn) : 0;
#define SONO_MW(L2, GR, MM, PL, In, O) \
  hipLaunchKernelGGL((stft_mw_kernel<L2, GR, MM, PL, In, O>), dim3(grid), dim3((1 << L2) / 16 * GR), 0, stream, p, fs)
#define SONO_MW_M(L2, GR, PL, In)                                                                          \
  do {                                                                                                     \
    if (feat) SONO_MW(L2, GR, 0, PL, In, kOutFeatures);                                                    \
    else if (bank && single) SONO_MW(L2, GR, -1, true, In, kOutU8Bank);                                    \
    else if (bank) SONO_MW(L2, GR, 0, true, In, kOutU8Bank);                                               \
    else if (single && p.out_kind == SONO_OUT_F32_DB) SONO_MW(L2, GR, -1, PL, In, kOutDb);                 \
    else if (single && p.out_kind == SONO_OUT_F32_MAG) SONO_MW(L2, GR, -1, PL, In, kOutMag);               \
    else if (single) SONO_MW(L2, GR, -1, PL, In, kOutU8);                                                  \
    else if (p.out_kind == SONO_OUT_F32_DB) SONO_MW(L2, GR, 0, PL, In, kOutDb);                            \
    else if (p.out_kind == SONO_OUT_F32_MAG) SONO_MW(L2, GR, 0, PL, In, kOutMag);                          \
    else if (m == 1) SONO_MW(L2, GR, 1, PL, In, kOutU8); else if (m == 2) SONO_MW(L2, GR, 2, PL, In, kOutU8); \
    else if (m == 4) SONO_MW(L2, GR, 4, PL, In, kOutU8); else SONO_MW(L2, GR, 0, PL, In, kOutU8);          \
  } while (0)
#define SONO_MW_IN(L2, GR, PL)                                                                   \
  do {                                                                                           \
    if (p.in_kind == SONO_IN_I16) SONO_MW_M(L2, GR, PL, int16_t); else SONO_MW_M(L2, GR, PL, float); \
  } while (0)
  if (p.n == 4096) {
    if (plugin) SONO_MW_IN(12, 1, true); else SONO_MW_IN(12, 1, false);
  } else if (p.n == 2048) {
    if (plugin) SONO_MW_IN(11, 1, true); else SONO_MW_IN(11, 1, false);
  } else {
    if (plugin) SONO_MW_IN(9, 8, true); else SONO_MW_IN(9, 8, false);
  }
#undef SONO_MW_IN
#undef SONO_MW_M
#undef SONO_MW
  if (bank)
    note_kernel(single ? (p.n == 4096 ? "stft_mw_kernel<4096,single,bank>" : p.n == 2048 ? "stft_mw_kernel<2048,single,bank>" : "stft_mw_kernel<512,single,bank>")
                       : (p.n == 4096 ? "stft_mw_kernel<4096,bank>" : p.n == 2048 ? "stft_mw_kernel<2048,bank>" : "stft_mw_kernel<512,bank>"));
  else
  note_kernel(single ? (p.n == 4096 ? "stft_mw_kernel<4096,single>" : p.n == 2048 ? "stft_mw_kernel<2048,single>" : "stft_mw_kernel<512,single>")
                     : (p.n == 4096 ? "stft_mw_kernel<4096>" : p.n == 2048 ? "stft_mw_kernel<2048>" : "stft_mw_kernel<512>"));
  return hipGetLastError() == hipSuccess ? SONO_OK : SONO_ERR_DEVICE;
}

}  // namespace sono
