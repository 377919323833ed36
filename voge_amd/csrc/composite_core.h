// Shared pieces of the depth-ordered compositing (VoGE/Aggregation.py:30-107): constants, the erfc / Gaussian
// evaluators, the LDS row layout, and the forward row pass of the lane-owns-NS-slots form.  Used by composite.hip
// (voge_composite_fwd / _bwd) and by the sweep's fused epilogue in trace_fwd.hip: ONE implementation, so fragments
// composited inside the sweep are bit-identical to the stand-alone kernel's.
#pragma once
#include "voge_common.h"

namespace voge {

typedef float v2f __attribute__((ext_vector_type(2)));

#ifdef VOGE_COMP_PRECISE
#define FAST_EXP(x) expf(x)
#define FAST_SQRT(x) sqrtf(x)
#else
// hardware exp2 / sqrt: ~1-2 ulp, far inside the 1e-4 parity tolerance; the libm versions cost
// ~35 VALU instructions per lane in a VALU-bound kernel
#define FAST_EXP(x) __builtin_amdgcn_exp2f((x) * 1.4426950408889634f)
#define FAST_SQRT(x) __builtin_amdgcn_sqrtf(x)
#endif

#ifndef VOGE_CF_ABL
#define VOGE_CF_ABL 0      // timing experiments on the forward composite (bit 0: no window walks, bit 1: no own block)
#endif
#ifndef VOGE_CF_COLWALK
#define VOGE_CF_COLWALK 1  // forward composite, sorted pixels: walks over the rows around a lane's own COLUMNS (each column's own window) instead of
#endif                     // over the columns around its rows (the pixel's widest window); see compn_fwd_rows
#ifndef VOGE_COMP_MAXT
#define VOGE_COMP_MAXT 256
#endif
constexpr int kCompThreads = VOGE_COMP_MAXT;
constexpr float kInvNorm = 1.6487212707001282f;  // 1 / exp(-0.5), Aggregation.py:79
constexpr float kRsqrtPi = 0.5641895835477563f;
#ifndef VOGE_KSAT
#define VOGE_KSAT 3.5f
#endif
constexpr float kSat = VOGE_KSAT;                // erfc(3.5)/2 = 3.7e-7, below the fp32 rounding of S ~ O(1..K)
constexpr float kCs = 1.2011224087864498f;       // sqrt(log2 e): x' = x * kCs, exp(-x^2) = 2^(-x'^2)
constexpr float kXcap = 5.0f * kCs;              // the fit's range; h(5) = 7.7e-13
constexpr float kBig = 3.0e38f;

// log2(erfc(x)/2) as a polynomial in x' = x sqrt(log2 e) on [0, 5 sqrt(log2 e)], weighted minimax
// on the absolute error of 2^Q (tools/fit_erfc.py).  Degree 6: |err| <= 1.5e-7 (the accuracy of
// Abramowitz-Stegun 7.1.26); VOGE_ERFC_DEG=8 gives 5.2e-8 and h(0) = 1/2 exactly for two more
// packed FMAs per pair of entries.
#ifndef VOGE_ERFC_DEG
#define VOGE_ERFC_DEG 6
#endif
#if VOGE_ERFC_DEG == 8
constexpr float kQ0 = -1.000000000e+00f, kQ1 = -1.355323434e+00f, kQ2 = -6.365932822e-01f,
                kQ3 = -8.570024371e-02f, kQ4 = 1.359716244e-02f, kQ5 = -3.297536168e-04f,
                kQ6 = -4.863584472e-04f, kQ7 = 1.211055496e-04f, kQ8 = -1.022832203e-05f;
#else
constexpr float kQ0 = -9.999997020e-01f, kQ1 = -1.355341077e+00f, kQ2 = -6.364040971e-01f,
                kQ3 = -8.642258495e-02f, kQ4 = 1.487037074e-02f, kQ5 = -1.475012978e-03f,
                kQ6 = 4.851150516e-05f;
#endif

__device__ __forceinline__ v2f pk_fma(const v2f a, const v2f b, const v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(const float x) { return (v2f){x, x}; }

// h(x') = erfc(x'/kCs)/2 for x' >= 0 (capped at kXcap), two at a time
__device__ __forceinline__ v2f h_pair(v2f xp) {
  xp.x = fminf(xp.x, kXcap);
  xp.y = fminf(xp.y, kXcap);
#if VOGE_ERFC_DEG == 8
  v2f q = pk_fma(splat(kQ8), xp, splat(kQ7));
  q = pk_fma(q, xp, splat(kQ6));
  q = pk_fma(q, xp, splat(kQ5));
#else
  v2f q = pk_fma(splat(kQ6), xp, splat(kQ5));
#endif
  q = pk_fma(q, xp, splat(kQ4));
  q = pk_fma(q, xp, splat(kQ3));
  q = pk_fma(q, xp, splat(kQ2));
  q = pk_fma(q, xp, splat(kQ1));
  q = pk_fma(q, xp, splat(kQ0));
  return (v2f){__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
}
// 2^(-x'^2) = exp(-x^2), two at a time
__device__ __forceinline__ v2f gauss_pair(const v2f xp) {
  const v2f q = -(xp * xp);
  return (v2f){__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
}
__device__ __forceinline__ float h_one(const float xp) { return h_pair(splat(xp)).x; }
// entries (e, e+1), e even: one 8-byte LDS read
__device__ __forceinline__ v2f ld2(const float *a, const int e) { return *reinterpret_cast<const v2f *>(a + e); }
__device__ __forceinline__ v2f abs2(const v2f v) { return (v2f){fabsf(v.x), fabsf(v.y)}; }

// Element at a BYTE offset from a base pointer.  With a 32-bit unsigned offset the compiler emits the scalar-base form
// (global_load v, voff, s[base:base+1]) and keeps the address arithmetic off the vector unit; kernels take the offset
// type as a template parameter and the host picks uint32_t when every byte offset of the launch fits.
template <typename T, typename OffT>
__device__ __forceinline__ const T &at_bytes(const void *base, const OffT byte_off) {
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <typename T, typename OffT>
__device__ __forceinline__ T &at_bytes_w(void *base, const OffT byte_off) {
  return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off);
}

// Row stride of the padded per-pixel arrays: two sentinels, K entries, two or three sentinels.
// PAD (two sentinel entries) and the stride are even: an entry's parity is its slot's parity and the
// pairs (2t, 2t+1) of a row are 8-byte aligned (ds_read_b64: 2 LDS cycles per wave; ds_read2_b32 of an odd pair costs 4).
__host__ __device__ constexpr int comp_pad(const int K) { return 2; }
__host__ __device__ constexpr int comp_row_stride(const int K) { return ((K + 1) & ~1) + 2 * comp_pad(K); }

__host__ __device__ inline int compn_lanes(const int K, const int NS) { return (K + NS - 1) / NS; }
__host__ __device__ inline int compn_stride(const int K, const int NS) { return compn_lanes(K, NS) * NS + 2 * comp_pad(K); }
// pixels a workgroup of `threads` lanes holds: workgroup form = threads / LP, wave form = (threads / 64) * (64 / LP)
__host__ __device__ inline int compn_pixels(const int K, const int NS, const int threads, const bool wave) {
  const int lp = compn_lanes(K, NS);
  return wave ? (threads / 64) * (64 / lp) : threads / lp;
}
__host__ __device__ inline int compn_rows(const int K, const int NS, const int threads, const bool wave) {
  return (compn_pixels(K, NS, threads, wave) * compn_stride(K, NS) + 3) & ~3;
}
// Forward row pass, wave form: the pixel's lanes (LP = ceil(K / NS) of them, lane q owns slots [NS q, NS q + NS))
// sit inside ONE wave, the pixel's padded (len, s', E) rows are in LDS (row origin of this lane's group: d0, an even
// index) and visible to the wave.  lm / sm / em: the lane's own len, s = sqrt(dsd + 1e-10), E = exp(-act) (E = 0 for
// an empty slot).  `sorted`: the pixel's list is depth ordered (the windowed walk); otherwise every column is visited.
// LP, q, seg_lo, d0, k0, K are PER-LANE values (pixels of different lane counts may share a wave: the packed kernels);
// LPmax is a wave-uniform upper bound of LP (the trip count of the scans).
// Returns the weights of the lane's slots.  The association of the scans is a function of the lane's index in the
// pixel only, so a pixel's result does not depend on where it sits (row bands == whole frame, fused == stand-alone).
template <int NS>
__device__ __forceinline__ void compn_fwd_rows(const float (&lm)[NS], const float (&sm)[NS], const float (&em)[NS],
                                               const float *Llen, const float *Lsp, const float *LE, const int d0,
                                               const int k0, const int K, const int q, const int LP, const int LPmax,
                                               const bool in_wg, const bool active, const bool sorted, const int seg_lo,
                                               const float occ, float (&w)[NS], unsigned *Lcell = nullptr) {
  float sp[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) sp[a] = sm[a] * kCs;
  float mx = 0.0f, esum = 0.0f;
#pragma unroll
  for (int a = 0; a < NS; ++a) {
    mx = fmaxf(mx, (em[a] != 0.0f) ? kSat * __builtin_amdgcn_rcpf(sm[a]) : 0.0f);
    esum += em[a];
  }
  // Exclusive prefix (over the lanes of the pixel) of the per-lane sums of E, Hillis-Steele on wave shuffles with the
  // window radius riding along
#if VOGE_CF_COLWALK
  float ex;      // (no pixel-wide window radius: a column walk goes as far as its own columns reach)
  {
    const float y = __shfl_up(esum, 1, 64);
    float x = (q > 0 && in_wg) ? y : 0.0f;
    for (int o = 1; o < LPmax; o <<= 1) {
      const float z = __shfl_up(x, o, 64);
      if (q >= o && in_wg) x += z;
    }
    ex = x;
  }
#else
  float ex, wave_rmax;
  if (Lcell != nullptr) {
    // the window radius through the pixel's LDS cell (compn_bwd_wave), the prefix sum alone through the shuffles
    if (in_wg && q == 0) *Lcell = 0u;
    wave_lds_sync();
    if (in_wg && mx > 0.0f) atomicMax(Lcell, __float_as_uint(mx));
    const float y = __shfl_up(esum, 1, 64);
    float x = (q > 0 && in_wg) ? y : 0.0f;
    for (int o = 1; o < LPmax; o <<= 1) {
      const float z = __shfl_up(x, o, 64);
      if (q >= o && in_wg) x += z;
    }
    ex = x;
    wave_lds_sync();
    wave_rmax = in_wg ? __uint_as_float(*lds_volatile(Lcell)) : 0.0f;
  } else {
    v2f x = {esum, mx};
    const v2f y = (v2f){__shfl_up(x.x, 1, 64), __shfl_up(x.y, 1, 64)};
    x = (q > 0 && in_wg) ? (v2f){y.x, fmaxf(mx, y.y)} : (v2f){0.0f, mx};
    for (int o = 1; o < LPmax; o <<= 1) {
      const v2f z = (v2f){__shfl_up(x.x, o, 64), __shfl_up(x.y, o, 64)};
      if (q >= o && in_wg) {
        x.x += z.x;
        x.y = fmaxf(x.y, z.y);
      }
    }
    ex = x.x;
    wave_rmax = __shfl(x.y, min(63, seg_lo + LP - 1), 64);      // the pixel's last lane holds the maximum
  }
#endif
#if !VOGE_CF_COLWALK
  const float rwin = sorted ? (in_wg ? wave_rmax : 0.0f) : 0.0f;
#endif
  bool any_e = false;
#pragma unroll
  for (int a = 0; a < NS; ++a) any_e = any_e || (em[a] != 0.0f);
  const float h0 = __builtin_amdgcn_exp2f(kQ0);      // h(0), exactly what h_pair(0) returns
  float S[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) S[a] = 0.0f;
  if (any_e && sorted) {
    v2f accF[NS], accB[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) { accF[a] = splat(0.0f); accB[a] = splat(0.0f); }
    // diagonal block (registers): for rows a < b, column b is behind row a and column a in front of row b
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      accF[a].x = em[a] * h0;         // self
#pragma unroll
      for (int b2 = a + 1; b2 < ((VOGE_CF_ABL & 2) ? 0 : NS); ++b2) {
        const float gap = lm[b2] - lm[a];
        const v2f xp = (v2f){gap * sp[b2], gap * sp[a]};                 // (row a, col b), (row b, col a)
        const v2f h = h_pair(xp);
        accB[a].y = fmaf(em[b2], h.x, accB[a].y);
        accF[b2].y = fmaf(em[a], h.y, accF[b2].y);
      }
    }
#if VOGE_CF_COLWALK
    // Column walks (round 6): the lane walks the ROWS around its own columns, each as far as that column's own window
    // kSat / s_j reaches, and adds -E_j h (row behind: the column is in front of it) / +E_j h (row in front) to the row's cell
    // of LR -- the pixel's s' row, which a sorted pixel's walks no longer read: a lane's own s' and E are in its registers.
    // The row walks below went as far as the pixel's WIDEST window for every column.  Read - add - write as in
    // compn_bwd_wave<NS, true>: in one iteration the lanes of a wave address different row pairs.
    float *const LR = const_cast<float *>(Lsp);
    float reachB = -kBig, reachF = kBig, lmS[NS], spS[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      const bool live = em[a] != 0.0f;
      const float r = kSat * __builtin_amdgcn_rcpf(sm[a]);
      reachB = live ? fmaxf(reachB, lm[a] + r) : reachB;
      reachF = live ? fminf(reachF, lm[a] - r) : reachF;
      lmS[a] = live ? lm[a] : 0.0f; spS[a] = live ? sp[a] : 0.0f;      // (an empty column: x = 0, E = 0 -- nothing, and nothing non-finite)
    }
#pragma unroll
    for (int a = 0; a < NS; a += 2) *reinterpret_cast<v2f *>(LR + d0 + a) = splat(0.0f);
    wave_lds_sync();
    for (int e = d0 + NS;; e += 2) {     // row pairs behind the own columns
      const v2f l2 = ld2(Llen, e);
      if (!(l2.x < reachB)) break;
      v2f racc = ld2(LR, e);
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const v2f xa = (l2 - splat(lmS[a])) * splat(spS[a]);
        racc = pk_fma(splat(-em[a]), h_pair(xa), racc);
      }
      *reinterpret_cast<v2f *>(LR + e) = racc;
      wave_lds_sync();
    }
    for (int e = d0 - 2;; e -= 2) {      // row pairs in front of the own columns
      const v2f l2 = ld2(Llen, e);
      if (!(l2.y > reachF)) break;
      v2f racc = ld2(LR, e);
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const v2f xa = (splat(lmS[a]) - l2) * splat(spS[a]);
        racc = pk_fma(splat(em[a]), h_pair(xa), racc);
      }
      *reinterpret_cast<v2f *>(LR + e) = racc;
      wave_lds_sync();
    }
    wave_lds_sync();      // (every column walk of the pixel has passed: its lanes share this wave)
    float pre = ex;
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      pre += em[a];                                                     // inclusive prefix sum of E
      S[a] = ((pre - (accF[a].x + accF[a].y)) + (accB[a].x + accB[a].y)) + *lds_volatile(LR + d0 + a);
    }
#else
    float lmB = lm[0];                 // the last live row decides how far back to walk
#pragma unroll
    for (int a = 1; a < NS; ++a) lmB = (em[a] != 0.0f) ? lm[a] : lmB;
#if VOGE_CF_ABL & 1      // (timing experiment: no window walks)
    if (lmB == -7.0f)
#endif
    for (int e = d0 - 2;; e -= 2) {      // column pairs in front of every own row; row 0 is the nearest
      const v2f l2 = ld2(Llen, e), s2 = ld2(Lsp, e), E2 = ld2(LE, e);
      if (!(lm[0] - l2.y < rwin)) break;
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const v2f xa = (splat(lm[a]) - l2) * s2;
        accF[a] = pk_fma(E2, h_pair(xa), accF[a]);
      }
    }
#if VOGE_CF_ABL & 1
    if (lmB == -7.0f)
#endif
    for (int e = d0 + NS;; e += 2) {     // column pairs behind every own row
      const v2f l2 = ld2(Llen, e), s2 = ld2(Lsp, e), E2 = ld2(LE, e);
      if (!(l2.x - lmB < rwin)) break;
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const v2f xa = (l2 - splat(lm[a])) * s2;
        accB[a] = pk_fma(E2, h_pair(xa), accB[a]);
      }
    }
    float pre = ex;
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      pre += em[a];                                                     // inclusive prefix sum of E
      S[a] = (pre - (accF[a].x + accF[a].y)) + (accB[a].x + accB[a].y);
    }
#endif
  } else if (any_e && active) {          // unsorted list: every column, signs from the data
    const int r0 = d0 - k0;
    for (int j = 0; j < K; ++j) {
      const float Ej = LE[r0 + j];
      if (Ej == 0.0f) continue;
      const float lj = Llen[r0 + j], sj = Lsp[r0 + j];
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const float xp = (lm[a] - lj) * sj;
        const float h = h_one(fabsf(xp));
        S[a] = fmaf(Ej, xp >= 0.0f ? 1.0f - h : h, S[a]);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < NS; ++a) w[a] = (em[a] != 0.0f) ? FAST_EXP(-occ * S[a]) * em[a] * kInvNorm : 0.0f;
}

// -DVOGE_FB_TIMES builds: lane utilisation of the window loops (tools/fb_sections.py).  Per workgroup (= wave) row:
// [0] row-loop iterations x 64, [1] lanes active in them, [2] column-loop iterations x 64, [3] lanes active in them.
#ifdef VOGE_FB_TIMES
__device__ unsigned long long g_cw_stats[1 << 16][4];
#define CW_COUNT(i) do { const unsigned long long m_ = __ballot(true); if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(m_) && blockIdx.x < (1 << 16)) { \
      g_cw_stats[blockIdx.x][i] += 64; g_cw_stats[blockIdx.x][(i) + 1] += (unsigned long long)__popcll(m_); } } while (0)
#else
#define CW_COUNT(i) do {} while (0)
#endif

// Backward, wave form, with the forward's weights given (u_m = g_m w_m comes from the caller): the closed-form
// gradients of the lane's own slots (header of composite.hip).  The pixel's padded rows len / s' / E s' are in LDS
// (the sentinel pads' u entries are zero); this routine stores u into Lu itself.  Same conventions as compn_fwd_rows.
// RCOL (round 5): the row sums r_m = sum_j E_j s_j phi(x_mj) are not walked for at all.  phi(x_mj), x_mj = (len_m - len_j) s_j,
// is a value the COLUMN walk of j's owner evaluates anyway (for u_m phi(x_mj)); that lane adds E_j s_j phi to row m's cell of
// LR (an LDS row array laid out like Lu, zeroed by the caller) and the row's owner reads the sum afterwards.  The add is a plain
// read - add - write: in one iteration of a walk the lanes of a wave address DIFFERENT row pairs (a pixel's lanes start NS rows
// apart and step by two together; other pixels have other rows), and a wave's LDS operations complete in order, so the next
// iteration's read sees this one's write -- the wave_barrier only keeps the compiler from moving the read above the write.
// (ds_add_f32 instead costs ~60 cycles per wave instruction: the kernel 129 -> 251 us.)  The column walk reaches every row inside column j's OWN window 3.5 / s_j -- where phi >= 4.8e-6 --
// while the row walk went as far as the pixel's widest window: the terms dropped are below that, like the column sums' own.
// One of the kernel's two window walks and its divergence are gone for one packed multiply and two LDS adds per column pair.
// (RCOL = true needs LR: a row array of the caller's LDS laid out like Lu, zeroed together with it.)
template <int NS, bool RCOL = false>
__device__ __forceinline__ void compn_bwd_wave(const float (&lm)[NS], const float (&sm)[NS], const float (&em)[NS],
                                               const float (&um)[NS], const float *Llen, const float *Lsp, const float *LE,
                                               float *Lu, const int d0, const int k0, const int K, const int q, const int LP,
                                               const int LPmax, const bool in_wg, const bool active, const bool sorted,
                                               const int seg_lo, const float occ, float (&ga)[NS], float (&gl)[NS],
                                               float (&gd)[NS], unsigned *Lcell = nullptr, float *LR = nullptr) {
  constexpr int NP = NS / 2;
  const int lane = threadIdx.x & 63;
  float sp[NS], Es[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) { sp[a] = sm[a] * kCs; Es[a] = em[a] * sp[a]; }
  float mx = 0.0f;
#pragma unroll
  for (int a = 0; a < NS; ++a) mx = fmaxf(mx, (em[a] != 0.0f) ? kSat * __builtin_amdgcn_rcpf(sm[a]) : 0.0f);
  float wave_rmax = 0.0f;
  if (RCOL) {
    // (the column walks go as far as each column's OWN window: nobody reads the pixel-wide radius -- until round 6 this form still
    //  paid for it: an LDS store, an LDS atomic max and a read-back, three dependent round trips per round)
  } else if (Lcell != nullptr) {
    // the pixel's window radius through ONE LDS cell (the pixel's own; radii are >= 0, so their bit patterns order
    // like unsigned integers): zero, max, read -- three LDS operations in the wave's in-order LDS queue instead of a
    // chain of seven dependent cross-lane shuffles
    if (in_wg && q == 0) *Lcell = 0u;
    wave_lds_sync();
    if (in_wg && mx > 0.0f) atomicMax(Lcell, __float_as_uint(mx));
    wave_lds_sync();
    wave_rmax = in_wg ? __uint_as_float(*lds_volatile(Lcell)) : 0.0f;
  } else {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float y = __shfl_down(mx, o, 64);
      if (lane + o < 64 && q + o < LP) mx = fmaxf(mx, y);
    }
    wave_rmax = __shfl(mx, seg_lo, 64);                // the pixel's first lane holds the maximum
  }
  const float rwin = sorted ? (in_wg ? wave_rmax : 0.0f) : 0.0f;
  bool any_e = false;
#pragma unroll
  for (int a = 0; a < NS; ++a) any_e = any_e || (em[a] != 0.0f);
  const float h0 = __builtin_amdgcn_exp2f(kQ0);      // h(0), exactly what h_pair(0) returns
  // ---- rows = the lane's own slots: r_m = sum over the window of E_j s_j phi_mj ----
  float rterm[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) rterm[a] = 0.0f;
  if (any_e && sorted) {
    v2f accR[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) accR[a] = splat(0.0f);
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      accR[a].x = Es[a];         // self
#pragma unroll
      for (int b2 = a + 1; b2 < NS; ++b2) {
        const float gap = lm[b2] - lm[a];
        const v2f xp = (v2f){gap * sp[b2], gap * sp[a]};                 // (row a, col b), (row b, col a)
        const v2f g = gauss_pair(xp);
        accR[a].y = fmaf(Es[b2], g.x, accR[a].y);
        accR[b2].y = fmaf(Es[a], g.y, accR[b2].y);
      }
    }
    if (!RCOL) {
    float lmB = lm[0];                 // the last live row decides how far back to walk
#pragma unroll
    for (int a = 1; a < NS; ++a) lmB = (em[a] != 0.0f) ? lm[a] : lmB;
    for (int e = d0 - 2;; e -= 2) {      // column pairs in front of every own row; row 0 is the nearest
      const v2f l2 = ld2(Llen, e), s2 = ld2(Lsp, e), E2 = ld2(LE, e);
      if (!(lm[0] - l2.y < rwin)) break;
      CW_COUNT(0);
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const v2f xa = (splat(lm[a]) - l2) * s2;
        accR[a] = pk_fma(E2, gauss_pair(xa), accR[a]);
      }
    }
    for (int e = d0 + NS;; e += 2) {     // column pairs behind every own row
      const v2f l2 = ld2(Llen, e), s2 = ld2(Lsp, e), E2 = ld2(LE, e);
      if (!(l2.x - lmB < rwin)) break;
      CW_COUNT(0);
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const v2f xa = (l2 - splat(lm[a])) * s2;
        accR[a] = pk_fma(E2, gauss_pair(xa), accR[a]);
      }
    }
    }
    // (RCOL: the own block only; the other lanes' columns arrive through LR during their column walks below)
#pragma unroll
    for (int a = 0; a < NS; ++a) rterm[a] = (accR[a].x + accR[a].y) * (RCOL ? 1.0f : kRsqrtPi / kCs);
  } else if (any_e && active) {          // unsorted list: every column
    const int r0 = d0 - k0;
    for (int j = 0; j < K; ++j) {
      const float Ej = LE[r0 + j];
      if (Ej == 0.0f) continue;
      const float lj = Llen[r0 + j], sj = Lsp[r0 + j];
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const float xp = (lm[a] - lj) * sj;
        const float xc = fminf(fabsf(xp), 16.0f);
        rterm[a] = fmaf(Ej * (kRsqrtPi / kCs), __builtin_amdgcn_exp2f(-xc * xc), rterm[a]);
      }
    }
  }
  // ---- u, its suffix sums over the pixel, then the lane's own columns ----
  float usum = 0.0f;
#pragma unroll
  for (int a = 0; a < NS; ++a) usum += um[a];
  if (in_wg) {
#pragma unroll
    for (int h2 = 0; h2 < NP; ++h2) *reinterpret_cast<v2f *>(Lu + d0 + 2 * h2) = (v2f){um[2 * h2], um[2 * h2 + 1]};
  }
  float sx;   // sum of u over the slots behind this lane's group
  {
    // (Round 6 tried this ladder two other ways -- LDS cells: a store, then LPmax - 1 reads per lane, + 5 us; a linear scan on DPP
    //  wave_shl:1 shifts, LPmax - 1 dependent VALU operations and no LDS trip, + 0.7 us -- the 1 + log2(LP) shuffles stay.)
    const float y = __shfl_down(usum, 1, 64);
    float x = (q + 1 < LP && in_wg) ? y : 0.0f;
    for (int o = 1; o < LPmax; o <<= 1) {
      const float z = __shfl_down(x, o, 64);
      if (q + o < LP && in_wg) x += z;
    }
    sx = x;
    wave_lds_sync();
  }
#pragma unroll
  for (int a = 0; a < NS; ++a) { ga[a] = 0.0f; gl[a] = 0.0f; gd[a] = 0.0f; }
  if (any_e && active) {
    float cPhi[NS], cphi[NS], cphil[NS];
    if (sorted) {
      float rj[NS];
      v2f aH[NS], aP[NS], aL[NS], bH[NS], bP[NS], bL[NS];   // per own column: rows behind (a*) / in front (b*)
#pragma unroll
      for (int b2 = 0; b2 < NS; ++b2) {
        rj[b2] = (em[b2] != 0.0f) ? kSat * __builtin_amdgcn_rcpf(sm[b2]) : 0.0f;   // an empty column needs no rows
        aH[b2] = (v2f){um[b2] * h0, 0.0f}; aP[b2] = (v2f){um[b2], 0.0f}; aL[b2] = splat(0.0f);   // self
        bH[b2] = splat(0.0f); bP[b2] = splat(0.0f); bL[b2] = splat(0.0f);
      }
      // diagonal block: for own slots a < b, row b is behind column a and row a in front of column b
#pragma unroll
      for (int a = 0; a < NS; ++a) {
#pragma unroll
        for (int b2 = a + 1; b2 < NS; ++b2) {
          const float gap = lm[b2] - lm[a];
          const v2f xp = (v2f){gap * sp[a], gap * sp[b2]};      // (row b, col a), (row a, col b)
          const v2f g = gauss_pair(xp), h = h_pair(xp);
          aH[a].y = fmaf(um[b2], h.x, aH[a].y); aP[a].y = fmaf(um[b2], g.x, aP[a].y); aL[a].y = fmaf(um[b2] * g.x, gap, aL[a].y);
          bH[b2].y = fmaf(um[a], h.y, bH[b2].y); bP[b2].y = fmaf(um[a], g.y, bP[b2].y); bL[b2].y = fmaf(um[a] * g.y, gap, bL[b2].y);
        }
      }
      for (int e = d0 + NS;; e += 2) {     // row pairs behind every own column
        const v2f l2 = ld2(Llen, e), u2 = ld2(Lu, e);
        v2f racc = RCOL ? ld2(LR, e) : splat(0.0f);
        bool need = false;
#pragma unroll
        for (int b2 = 0; b2 < NS; ++b2) need = need || (l2.x - lm[b2] < rj[b2]);
        if (!need) break;
        CW_COUNT(2);
#pragma unroll
        for (int b2 = 0; b2 < NS; ++b2) {
          const v2f d = l2 - splat(lm[b2]);
          const v2f xp = d * splat(sp[b2]);
          const v2f g = gauss_pair(xp);
          const v2f y = u2 * g;
          aH[b2] = pk_fma(u2, h_pair(xp), aH[b2]); aP[b2] = aP[b2] + y; aL[b2] = pk_fma(y, d, aL[b2]);
          if (RCOL) racc = pk_fma(splat(Es[b2]), g, racc);
        }
        if (RCOL) { *reinterpret_cast<v2f *>(LR + e) = racc; wave_lds_sync(); }
      }
      for (int e = d0 - 2;; e -= 2) {      // row pairs in front of every own column
        const v2f l2 = ld2(Llen, e), u2 = ld2(Lu, e);
        v2f racc = RCOL ? ld2(LR, e) : splat(0.0f);
        bool need = false;
#pragma unroll
        for (int b2 = 0; b2 < NS; ++b2) need = need || (lm[b2] - l2.y < rj[b2]);
        if (!need) break;
        CW_COUNT(2);
#pragma unroll
        for (int b2 = 0; b2 < NS; ++b2) {
          const v2f d = splat(lm[b2]) - l2;
          const v2f xp = d * splat(sp[b2]);
          const v2f g = gauss_pair(xp);
          const v2f y = u2 * g;
          bH[b2] = pk_fma(u2, h_pair(xp), bH[b2]); bP[b2] = bP[b2] + y; bL[b2] = pk_fma(y, d, bL[b2]);
          if (RCOL) racc = pk_fma(splat(Es[b2]), g, racc);
        }
        if (RCOL) { *reinterpret_cast<v2f *>(LR + e) = racc; wave_lds_sync(); }
      }
      if (RCOL) {      // every column walk of the pixel has passed (its lanes share this wave): the row sums are complete
        wave_lds_sync();
#pragma unroll
        for (int a = 0; a < NS; ++a) rterm[a] = (rterm[a] + *lds_volatile(LR + d0 + a)) * (kRsqrtPi / kCs);
      }
      float suf = sx;
#pragma unroll
      for (int b2 = NS - 1; b2 >= 0; --b2) {
        suf += um[b2];                                                   // inclusive suffix sum of u
        cPhi[b2] = (suf - (aH[b2].x + aH[b2].y)) + (bH[b2].x + bH[b2].y);
        cphi[b2] = ((aP[b2].x + aP[b2].y) + (bP[b2].x + bP[b2].y)) * kRsqrtPi;
        cphil[b2] = ((aL[b2].x + aL[b2].y) - (bL[b2].x + bL[b2].y)) * kRsqrtPi;
      }
    } else {
#pragma unroll
      for (int b2 = 0; b2 < NS; ++b2) { cPhi[b2] = 0.0f; cphi[b2] = 0.0f; cphil[b2] = 0.0f; }
      const int r0 = d0 - k0;
      for (int m = 0; m < K; ++m) {
        const float ur = Lu[r0 + m];
        if (ur == 0.0f) continue;
        const float lr = Llen[r0 + m];
#pragma unroll
        for (int b2 = 0; b2 < NS; ++b2) {
          const float dl = lr - lm[b2];
          const float xp = dl * sp[b2];
          const float h = h_one(fabsf(xp));
          const float xc = fminf(fabsf(xp), 16.0f);
          const float ph = ur * (__builtin_amdgcn_exp2f(-xc * xc) * kRsqrtPi);
          cPhi[b2] = fmaf(ur, xp >= 0.0f ? 1.0f - h : h, cPhi[b2]);
          cphi[b2] += ph;
          cphil[b2] = fmaf(ph, dl, cphil[b2]);
        }
      }
    }
#pragma unroll
    for (int b2 = 0; b2 < NS; ++b2) {
      if (em[b2] != 0.0f) {
        ga[b2] = fmaf(occ * em[b2], cPhi[b2], -um[b2]);
        gl[b2] = -occ * (um[b2] * rterm[b2] - em[b2] * sm[b2] * cphi[b2]);
        gd[b2] = -occ * em[b2] * (0.5f * __builtin_amdgcn_rcpf(sm[b2])) * cphil[b2];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Lane packing by hit count.  With the trace's per-pixel hit count at hand a pixel needs ceil(count / NS) lanes, not
// ceil(K / NS): at cfg3 the lit pixels hold 18 of K = 40 slots on average and 40 % of the pixels none, so a wave laid
// out for K idles more than half of its lanes through every scan, window walk and table update.  The packed kernels
// give a wave a GROUP of G <= 64 pixels (lane g < G holds pixel g's lane need) and walk it in rounds; a round takes
// the longest run of consecutive pixels whose lanes fit the wave.  A pixel's lanes stay adjacent and in slot order,
// so everything per pixel (scan association, window walks) is what the K-strided layout computes: bit-identical.
struct PackLane {
  int p;      // the lane's pixel inside the group (-1: idle lane)
  int s0;     // first lane of that pixel
  int np;     // its lanes
  int ord;    // its ordinal among the round's non-empty pixels (row origins in LDS)
};
// inclusive prefix sum over the wave
__device__ __forceinline__ int wave_incl_scan(int v, const int lane) {
  // DPP steps (VALU operands) instead of six dependent LDS-crossbar shuffles: shifts by 1, 2, 4, 8 inside every row of
  // 16 lanes (zeros shifted in), then lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
  (void)lane;
  return v;
}
// One round: pixels [pc, pe) of the group (pe returned; uniform), `off` = lanes consumed by earlier rounds.
// need / incl: lane g's pixel need and its inclusive prefix (lanes >= G: need 0).  np_max: the longest pixel of the round.
__device__ __forceinline__ int pack_round(const int need, const int incl, const int G, const int lane, const int pc,
                                          const int off, PackLane &pl, int &np_max) {
  const unsigned long long fit = __ballot(lane >= pc && lane < G && incl - off <= 64);
  const int pe = pc + __popcll(fit);      // (incl is monotone: the fitting pixels are a run; need <= 64: never empty)
  pl.p = -1; pl.s0 = 0; pl.np = 0; pl.ord = 0;
  int c = 0, nm = 0;
  const int excl = incl - need;
  for (int p = pc; p < pe; ++p) {         // uniform
    const int n = __builtin_amdgcn_readlane(need, p);
    if (n == 0) continue;
    const int s = __builtin_amdgcn_readlane(excl, p) - off;
    if (lane >= s) { pl.p = p; pl.s0 = s; pl.np = n; pl.ord = c; }
    ++c;
    nm = max(nm, n);
  }
  if (lane >= pl.s0 + pl.np) pl.p = -1;   // lanes behind the round's last pixel
  np_max = nm;
  return pe;
}

}  // namespace voge
