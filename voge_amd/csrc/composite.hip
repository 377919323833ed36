// Depth-ordered volumetric compositing (the reference's "aggregation") for gfx950.
//
// Reference behaviour being reproduced: VoGE/Aggregation.py:30-107 (get_cross_activation,
// assign2weight, aggregation).  The reference materialises ~6 [npix,K,K] fp32 temporaries and
// lets autograd replay them.  Here ONE LANE OWNS ONE (pixel, slot) PAIR: a 256-thread
// workgroup covers floor(256/K) whole pixels, every HBM access is a perfectly coalesced
// stream over the flat [npix*K] arrays, the pixel's K (len, s, E) triples sit in LDS.
// Nothing of size K x K ever exists.
//
//   fwd : row m        S_m = sum_k E_k Phi((len_m - len_k) s_k),  w_m = exp(-occ S_m) E_m e^{1/2}
//   bwd : row m        u_m = g_m w_m,  r_m = sum_k E_k s_k phi_mk
//         column j     dL/dact_j = -u_j + occ E_j sum_m u_m Phi_mj
//                      dL/dlen_j = -occ (u_j r_j - E_j s_j sum_m u_m phi_mj)
//                      dL/ddsd_j = -occ E_j / (2 s_j) sum_m u_m phi_mj (len_m - len_j)
//   with E = exp(-act), s = sqrt(dsd + 1e-10), Phi = (erf + 1)/2, phi = exp(-x^2)/sqrt(pi).
//
// The kernel is VALU-issue bound (rocprofv3: SQ_INSTS_VALU x 4 cycles = kernel time), so the
// inner loops are written for instruction count:
//   * The pixel's list is depth sorted (the trace emits it that way), so the sign of
//     x = (len_m - len_k) s_k is known from the side of the diagonal: both loops evaluate only
//     h(|x|) = erfc(|x|)/2, and  Phi = 1 - h in front, h behind.  The "1" parts are prefix /
//     suffix sums, the h parts die out beyond |len_m - len_k| >= 4 / s (h < 8e-9), so every row
//     walks a WINDOW away from the diagonal and stops.
//   * h(x) = 2^Q(x'), x' = x sqrt(log2 e), Q a degree-6 polynomial (|err| <= 1.5e-7 absolute on
//     [0, 5]): one transcendental per evaluation, no reciprocal, no select.
//   * Two list entries per trip: (len, s, E) live in separate LDS arrays, ds_read2_b32 returns
//     neighbours as a register pair and the arithmetic is packed fp32 (v_pk_fma_f32 ...).
//   * Pads of sentinel entries (E = 0, len = -/+ 3e38) on both sides of every pixel's row: no
//     index clamps or bounds tests in the loops.
// An unsorted list (possible through the public API) takes a plain full K x K scan instead.
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

// Row stride of the padded per-pixel arrays: two sentinels, K entries, two or three sentinels.
// PAD (two sentinel entries) and the stride are even: an entry's parity is its slot's parity and the
// pairs (2t, 2t+1) of a row are 8-byte aligned (ds_read_b64: 2 LDS cycles per wave; ds_read2_b32 of an odd pair costs 4).
__host__ __device__ constexpr int comp_pad(const int K) { return 2; }
__host__ __device__ constexpr int comp_row_stride(const int K) { return ((K + 1) & ~1) + 2 * comp_pad(K); }

struct CompLds {
  v2f scan[2][kCompThreads];   // (running sum, running max) / (running sum, -) of the scans
  int rmaxi[kCompThreads];      // per pixel: bits of the largest 3.5/s
  int cnt[kCompThreads];        // per pixel: number of assigned slots (forward)
  int unsorted[kCompThreads];
};
// dynamic LDS: CompLds, then the padded per-pixel arrays len / sp / E (/ u in the backward),
// `rows` floats each
__host__ __device__ inline int comp_rows(const int K) { return ((kCompThreads / K) * comp_row_stride(K) + 3) & ~3; }   // sized for the largest workgroup
__host__ __device__ inline size_t comp_lds_bytes(const int K, const bool bwd) {
  return sizeof(CompLds) + sizeof(float) * (size_t)comp_rows(K) * (bwd ? 4 : 3);
}

// MODE 0: forward.  1: backward, weights recomputed (S_m again).  2: backward with the forward's
// weights given: the row pass only needs r_m, i.e. exp(-x^2) but no erfc.
template <int MODE>
__global__ void __launch_bounds__(kCompThreads)
composite_kernel(const int32_t *__restrict__ idx, const float *__restrict__ act,
                 const float *__restrict__ len, const float *__restrict__ dsd,
                 const float *__restrict__ w_in, const float *__restrict__ g_weight, const int32_t *__restrict__ cnt_in,
                 const float occ, const long npix, const int K,
                 const int ppw, float *__restrict__ out0 /* weight | g_act */,
                 float *__restrict__ out1 /* g_len */, float *__restrict__ out2 /* g_dsd */,
                 int64_t *__restrict__ valid_num) {
  constexpr bool BWD = MODE != 0;
  constexpr bool HAVE_W = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char comp_smem[];
  CompLds &L = *reinterpret_cast<CompLds *>(comp_smem);
  const int rows = comp_rows(K);
  float *const Llen = reinterpret_cast<float *>(comp_smem + sizeof(CompLds));
  float *const Lsp = Llen + rows;
  float *const LE = Lsp + rows;
  float *const Lu = LE + rows;   // backward only
  const int tid = threadIdx.x;
  // tid / K without the integer-division sequence (exact for tid < 2^20)
  const int p = __float2int_rz(((float)tid + 0.5f) * __builtin_amdgcn_rcpf((float)K)), k = tid - p * K;
  const long pix = (long)blockIdx.x * ppw + p;
  const bool in_wg = p < ppw;
  const bool active = in_wg && (pix < npix);
  const long f = pix * K + k;
  const int RS = comp_row_stride(K);
  const int PAD = comp_pad(K);
  const int bi = (in_wg ? p : 0) * RS + PAD + (in_wg ? k : 0);   // this slot's entry in the padded arrays
  // cnt_in (optional) = the trace's per-pixel hit count: slots k >= cnt hold the trace's sentinels
  // (len = act = 1e10, dsd = 0, idx = -1), i.e. E = 0.  Those slots are not even loaded, and a
  // workgroup whose pixels are all empty writes its zeros and leaves before touching anything else
  // (a sparse frame like cfg3 has ~40 % empty pixels and ~25 % empty slots in the others).
  int lead = K;
  if (cnt_in != nullptr) {
    lead = active ? min(K, max(0, cnt_in[pix])) : 0;
    if (!__syncthreads_or(lead > 0)) {
      if (active) {
        out0[f] = 0.0f;
        if (BWD) { out1[f] = 0.0f; out2[f] = 0.0f; }
        else if (k == 0) valid_num[pix] = 0;
      }
      return;
    }
  }
  if (tid < ppw) { L.unsorted[tid] = 0; L.rmaxi[tid] = 0; L.cnt[tid] = 0; }
  float lm = VOGE_SENT_LEN, sm = 1e-5f, em = 0.f, gw = 0.f, wgiven = 0.f;   // what a sentinel slot evaluates to
  int id = -1;
  if (active && k < lead) {
    em = FAST_EXP(-act[f]);
    lm = len[f];
    sm = FAST_SQRT(dsd[f] + 1e-10f);
    if (BWD) gw = g_weight[f]; else if (cnt_in == nullptr) id = idx[f];
    if (HAVE_W) wgiven = w_in[f];
  }
  if (in_wg) {
    Llen[bi] = lm; Lsp[bi] = sm * kCs; LE[bi] = HAVE_W ? em * (sm * kCs) : em;
    if (BWD) Lu[bi] = 0.0f;
  }
  // sentinels: one aligned pair in front of the row, one (K odd: three entries) behind it.  Every
  // window loop stops at the first sentinel it meets (len = -/+ 3e38, and the tests are written so
  // that a NaN also stops them), so nothing beyond is ever read.
  if (in_wg && k < 3) {
    const int r0 = p * RS;
    for (int q = k; q < 3; q += K) {     // (K < 3: a thread writes more than one)
      if (q < PAD) { Llen[r0 + q] = -kBig; Lsp[r0 + q] = 1.0f; LE[r0 + q] = 0.0f; if (BWD) Lu[r0 + q] = 0.0f; }
      const int eb = r0 + PAD + K + q;
      if (eb < r0 + RS) { Llen[eb] = kBig; Lsp[eb] = 1.0f; LE[eb] = 0.0f; if (BWD) Lu[eb] = 0.0f; }
    }
  }
  __syncthreads();
  if (active && k > 0 && !(Llen[bi - 1] <= lm)) L.unsorted[p] = 1;
  // Per-pixel reductions.  The workgroup scan below is LDS-issue bound, so as little as possible
  // rides on it: the assigned-slot count is a ballot + popcount per wave merged by one LDS atomic
  // per (wave, pixel) run; the window radius (largest 3.5/s) shares the scan's 64-bit elements
  // when there is a scan (an 8-byte LDS access costs the same issue slots as a 4-byte one), and is
  // a segmented wave max + one atomic per run when there is none (backward with given weights).
  // Max and integer add are order independent: no determinism is lost.
  const int lane = tid & 63;
  const bool head = in_wg && (lane == 0 || k == 0);
  if (!BWD && cnt_in == nullptr) {
    const unsigned long long m = __ballot(id >= 0);
    const int lo = max(0, lane - k), hi = min(63, lane + (K - 1 - k));     // this pixel's lanes in the wave
    const unsigned long long seg = ((hi - lo == 63) ? ~0ull : ((1ull << (hi - lo + 1)) - 1ull) << lo);
    if (head) atomicAdd(&L.cnt[p], __popcll(m & seg));
  }
  float mx = (em != 0.0f) ? kSat * __builtin_amdgcn_rcpf(sm) : 0.0f;
  // Inclusive prefix sum of E within each pixel: ping-pong Hillis-Steele scan over the workgroup.
  // Its association is a function of the slot index only, so a pixel's result does not depend on
  // where it sits in the workgroup (row bands reproduce the whole frame bit for bit).
  float pre_incl = em;
  if (!HAVE_W) {
    v2f x = {em, mx};
    int par = 0;
    for (int o = 1; o < K; o <<= 1) {
      L.scan[par][tid] = x;
      __syncthreads();
      if (k >= o && in_wg) {
        const v2f y = L.scan[par][tid - o];
        x.x += y.x;
        x.y = fmaxf(x.y, y.y);
      }
      par ^= 1;
    }
    pre_incl = x.x;
    if (in_wg && k == K - 1) L.rmaxi[p] = __float_as_int(x.y);
  } else {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {     // lane + o holds slot k + o of the same pixel iff k + o < K
      const float y = __shfl_down(mx, o, 64);
      if (lane + o < 64 && k + o < K) mx = fmaxf(mx, y);
    }
    if (head) atomicMax(&L.rmaxi[p], __float_as_int(mx));     // mx >= 0: float order == int order
  }
  __syncthreads();
  const float rwin_all = in_wg ? __int_as_float(L.rmaxi[p]) : 0.0f;
  const int cnt_all = (!BWD && in_wg) ? (cnt_in != nullptr ? (active ? cnt_in[pix] : 0) : L.cnt[p]) : 0;
  const bool sorted = active && (L.unsorted[p] == 0);
  const float rwin = sorted ? rwin_all : 0.0f;   // 0: the windowed loops do nothing

  // ---- row m (sorted list): S_m = prefix_E[m] - sum_front E_j h_mj + sum_back E_j h_mj ;
  //      r_m = sum_window E_j s_j phi_mj.  Pairs walk away from the diagonal until the nearer
  //      entry of the pair leaves the window.
  float sum = 0.0f, rterm = 0.0f;
  if (em != 0.0f && sorted) {
    v2f accF = splat(0.0f), accB = splat(0.0f), accR = splat(0.0f);
    const v2f lm2 = splat(lm);
    const int d0 = bi & ~1;              // the aligned pair that holds the diagonal
    const bool odd = (bi & 1) != 0;
    {   // diagonal pair: entry d0 is in front of (or is) this slot; d0+1 is this slot (odd) or behind
      const v2f s2 = ld2(Lsp, d0), E2 = ld2(LE, d0);
      const v2f xp = abs2(lm2 - ld2(Llen, d0)) * s2;
      if (!HAVE_W) {
        const v2f eh = E2 * h_pair(xp);
        accF = (v2f){eh.x, odd ? eh.y : 0.0f};
        accB = (v2f){0.0f, odd ? 0.0f : eh.y};
      }
      if (BWD) accR = (HAVE_W ? E2 : E2 * s2) * gauss_pair(xp);
    }
    for (int e = d0 - 2;; e -= 2) {      // pairs in front, nearest first
      const v2f l2 = ld2(Llen, e), s2 = ld2(Lsp, e), E2 = ld2(LE, e);
      const v2f d = lm2 - l2;
      if (!(d.y < rwin)) break;
      const v2f xp = d * s2;
      if (!HAVE_W) accF = pk_fma(E2, h_pair(xp), accF);
      if (BWD) accR = pk_fma(HAVE_W ? E2 : E2 * s2, gauss_pair(xp), accR);
    }
    for (int e = d0 + 2;; e += 2) {      // pairs behind
      const v2f l2 = ld2(Llen, e), s2 = ld2(Lsp, e), E2 = ld2(LE, e);
      const v2f d = l2 - lm2;
      if (!(d.x < rwin)) break;
      const v2f xp = d * s2;
      if (!HAVE_W) accB = pk_fma(E2, h_pair(xp), accB);
      if (BWD) accR = pk_fma(HAVE_W ? E2 : E2 * s2, gauss_pair(xp), accR);
    }
    sum = (pre_incl - (accF.x + accF.y)) + (accB.x + accB.y);
    rterm = (accR.x + accR.y) * (kRsqrtPi / kCs);
  } else if (em != 0.0f && active) {     // unsorted list: every column, signs from the data
    for (int j = 0; j < K; ++j) {
      const int e = bi - k + j;
      const float Ej = LE[e];
      if (Ej == 0.0f) continue;
      const float xp = (lm - Llen[e]) * Lsp[e];
      if (!HAVE_W) {
        const float h = h_one(fabsf(xp));
        sum = fmaf(Ej, xp >= 0.0f ? 1.0f - h : h, sum);
      }
      if (BWD) {
        const float xc = fminf(fabsf(xp), 16.0f);
        rterm = fmaf(HAVE_W ? Ej : Ej * Lsp[e], __builtin_amdgcn_exp2f(-xc * xc) * (kRsqrtPi / kCs), rterm);
      }
    }
  }
  const float w = HAVE_W ? wgiven : ((em != 0.0f) ? FAST_EXP(-occ * sum) * em * kInvNorm : 0.0f);
  if (!BWD) {
    if (active) {
      out0[f] = w;
      if (k == 0) valid_num[pix] = cnt_all;
    }
    return;
  }
  const float um = gw * w;
  if (in_wg) Lu[bi] = um;
  float suf_incl;
  {   // inclusive suffix sum of u within each pixel (same scan, mirrored)
    float x = um;
    int par = 0;
    float(*sf)[kCompThreads] = reinterpret_cast<float(*)[kCompThreads]>(L.scan);   // 4-byte elements here
    for (int o = 1; o < K; o <<= 1) {
      sf[par][tid] = x;
      __syncthreads();
      if (k + o < K && in_wg) x += sf[par][tid + o];
      par ^= 1;
    }
    suf_incl = x;
  }
  __syncthreads();
  // ---- column j (= this lane): Phi_mj = 1 - h for rows behind (suffix sum of u minus the h
  // part), h for rows in front; phi terms live in the window |len_m - len_j| < 4 / s_j only.
  float ga = 0.0f, gl = 0.0f, gd = 0.0f;
  if (em != 0.0f && active) {
    float cPhi, cphi, cphil;
    const float sp = sm * kCs;
    if (sorted) {
      const float rj = kSat * __builtin_amdgcn_rcpf(sm);
      const v2f lm2 = splat(lm), sp2 = splat(sp);
      v2f aH, aP, aL;   // behind (and self): sum u h, sum u y, sum u y |dl|
      v2f bH, bP, bL;   // in front
      const int d0 = bi & ~1;
      const bool odd = (bi & 1) != 0;
      {   // diagonal pair: entry d0 is this slot (even) or the row just in front (odd); d0+1 is behind / this slot
        const v2f u2 = ld2(Lu, d0);
        const v2f d = abs2(ld2(Llen, d0) - lm2);
        const v2f xp = d * sp2;
        const v2f uy = u2 * gauss_pair(xp), uh = u2 * h_pair(xp), ul = uy * d;
        aH = (v2f){odd ? 0.0f : uh.x, uh.y}; bH = (v2f){odd ? uh.x : 0.0f, 0.0f};
        aP = (v2f){odd ? 0.0f : uy.x, uy.y}; bP = (v2f){odd ? uy.x : 0.0f, 0.0f};
        aL = (v2f){odd ? 0.0f : ul.x, ul.y}; bL = (v2f){odd ? ul.x : 0.0f, 0.0f};
      }
      for (int e = d0 + 2;; e += 2) {      // rows behind
        const v2f l2 = ld2(Llen, e), u2 = ld2(Lu, e);
        const v2f d = l2 - lm2;
        if (!(d.x < rj)) break;
        const v2f xp = d * sp2;
        const v2f uy = u2 * gauss_pair(xp);
        aH = pk_fma(u2, h_pair(xp), aH);
        aP = aP + uy;
        aL = pk_fma(uy, d, aL);
      }
      for (int e = d0 - 2;; e -= 2) {      // rows in front
        const v2f l2 = ld2(Llen, e), u2 = ld2(Lu, e);
        const v2f d = lm2 - l2;
        if (!(d.y < rj)) break;
        const v2f xp = d * sp2;
        const v2f uy = u2 * gauss_pair(xp);
        bH = pk_fma(u2, h_pair(xp), bH);
        bP = bP + uy;
        bL = pk_fma(uy, d, bL);
      }
      cPhi = (suf_incl - (aH.x + aH.y)) + (bH.x + bH.y);
      cphi = ((aP.x + aP.y) + (bP.x + bP.y)) * kRsqrtPi;
      cphil = ((aL.x + aL.y) - (bL.x + bL.y)) * kRsqrtPi;
    } else {
      cPhi = 0.0f; cphi = 0.0f; cphil = 0.0f;
      for (int m = 0; m < K; ++m) {
        const int e = bi - k + m;
        const float ur = Lu[e];
        if (ur == 0.0f) continue;
        const float dl = Llen[e] - lm;
        const float xp = dl * sp;
        const float h = h_one(fabsf(xp));
        const float xc = fminf(fabsf(xp), 16.0f);
        const float ph = ur * (__builtin_amdgcn_exp2f(-xc * xc) * kRsqrtPi);
        cPhi = fmaf(ur, xp >= 0.0f ? 1.0f - h : h, cPhi);
        cphi += ph;
        cphil = fmaf(ph, dl, cphil);
      }
    }
    ga = fmaf(occ * em, cPhi, -um);
    gl = -occ * (um * rterm - em * sm * cphi);
    gd = -occ * em / (2.0f * sm) * cphil;
  }
  if (active) {
    out0[f] = ga;
    out1[f] = gl;
    out2[f] = gd;
  }
}


// ------------------------------------------------------------------------------------------
// Two slots per lane (forward, and backward with the forward's weights; sorted lists).
// A lane owns the ALIGNED PAIR of slots (2q, 2q+1) of its pixel, so every pair of list entries
// it reads from LDS serves four (row, column) evaluations instead of two, the lane's own pair is
// the diagonal block (no LDS read at all), global loads / stores are 8 bytes wide and the
// per-pixel scans run over half as many lanes.  The kernel above remains the reference form: it
// handles the backward without given weights and is what VOGE_COMP_ONE_SLOT=1 builds select.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline int comp2_lanes(const int K) { return (K + 1) >> 1; }
__host__ __device__ inline int comp2_rows(const int K) { return ((kCompThreads / comp2_lanes(K)) * comp_row_stride(K) + 3) & ~3; }
__host__ __device__ inline size_t comp2_lds_bytes(const int K, const bool bwd) {
  return sizeof(CompLds) + sizeof(float) * (size_t)comp2_rows(K) * (bwd ? 4 : 3);
}

template <int MODE>   // 0: forward, 2: backward with weights
__global__ void __launch_bounds__(kCompThreads)
composite2_kernel(const int32_t *__restrict__ idx, const float *__restrict__ act,
                  const float *__restrict__ len, const float *__restrict__ dsd,
                  const float *__restrict__ w_in, const float *__restrict__ g_weight,
                  const int32_t *__restrict__ cnt_in, const float occ, const long npix, const int K, const int ppw,
                  float *__restrict__ out0 /* weight | g_act */, float *__restrict__ out1 /* g_len */,
                  float *__restrict__ out2 /* g_dsd */, int64_t *__restrict__ valid_num) {
  constexpr bool BWD = MODE != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char comp_smem[];
  CompLds &L = *reinterpret_cast<CompLds *>(comp_smem);
  const int rows = comp2_rows(K);
  float *const Llen = reinterpret_cast<float *>(comp_smem + sizeof(CompLds));
  float *const Lsp = Llen + rows;
  float *const LE = Lsp + rows;     // E (forward) or E * s' (backward)
  float *const Lu = LE + rows;      // backward only
  const int tid = threadIdx.x, lane = tid & 63;
  const int LP = comp2_lanes(K);
  const int p = __float2int_rz(((float)tid + 0.5f) * __builtin_amdgcn_rcpf((float)LP)), q = tid - p * LP;
  const long pix = (long)blockIdx.x * ppw + p;
  const bool in_wg = p < ppw;
  const bool active = in_wg && (pix < npix);
  const int k0 = 2 * q;
  const bool has1 = k0 + 1 < K;
  const long f = pix * K + k0;
  const int RS = comp_row_stride(K);
  const int PAD = comp_pad(K);
  const int d0 = (in_wg ? p : 0) * RS + PAD + (in_wg ? k0 : 0);   // own aligned pair in the padded arrays
  const bool vec = (K & 1) == 0;                                    // pix*K + 2q even: 8-byte accesses
  int lead = K;
  if (cnt_in != nullptr) {
    lead = active ? min(K, max(0, cnt_in[pix])) : 0;
    if (!__syncthreads_or(lead > 0)) {      // every pixel of the workgroup is empty
      if (active) {
        if (vec) {
          *reinterpret_cast<v2f *>(out0 + f) = splat(0.0f);
          if (BWD) { *reinterpret_cast<v2f *>(out1 + f) = splat(0.0f); *reinterpret_cast<v2f *>(out2 + f) = splat(0.0f); }
        } else {
          out0[f] = 0.0f; if (has1) out0[f + 1] = 0.0f;
          if (BWD) { out1[f] = 0.0f; out2[f] = 0.0f; if (has1) { out1[f + 1] = 0.0f; out2[f + 1] = 0.0f; } }
        }
        if (!BWD && q == 0) valid_num[pix] = 0;
      }
      return;
    }
  }
  if (tid < ppw) { L.unsorted[tid] = 0; L.rmaxi[tid] = 0; L.cnt[tid] = 0; }
  // what a sentinel slot evaluates to: E = 0, len = 1e10, s = 1e-5
  v2f lm = splat(VOGE_SENT_LEN), sm = splat(1e-5f), em = splat(0.0f), gw = splat(0.0f), wg = splat(0.0f);
  int id0 = -1, id1 = -1;
  if (active) {
    const bool ld0 = k0 < lead, ld1 = has1 && (k0 + 1 < lead);
    if (vec && ld1) {
      const v2f a2 = *reinterpret_cast<const v2f *>(act + f), l2 = *reinterpret_cast<const v2f *>(len + f),
                d2 = *reinterpret_cast<const v2f *>(dsd + f);
      em = (v2f){FAST_EXP(-a2.x), FAST_EXP(-a2.y)};
      lm = l2;
      sm = (v2f){FAST_SQRT(d2.x + 1e-10f), FAST_SQRT(d2.y + 1e-10f)};
      if (BWD) { gw = *reinterpret_cast<const v2f *>(g_weight + f); wg = *reinterpret_cast<const v2f *>(w_in + f); }
      else if (cnt_in == nullptr) { id0 = idx[f]; id1 = idx[f + 1]; }
    } else {
      if (ld0) {
        em.x = FAST_EXP(-act[f]); lm.x = len[f]; sm.x = FAST_SQRT(dsd[f] + 1e-10f);
        if (BWD) { gw.x = g_weight[f]; wg.x = w_in[f]; } else if (cnt_in == nullptr) id0 = idx[f];
      }
      if (ld1) {
        em.y = FAST_EXP(-act[f + 1]); lm.y = len[f + 1]; sm.y = FAST_SQRT(dsd[f + 1] + 1e-10f);
        if (BWD) { gw.y = g_weight[f + 1]; wg.y = w_in[f + 1]; } else if (cnt_in == nullptr) id1 = idx[f + 1];
      }
    }
  }
  const v2f sp = sm * splat(kCs);
  if (in_wg) {
    // (K odd: the last lane's second entry is the first back sentinel; E = 0 there either way)
    *reinterpret_cast<v2f *>(Llen + d0) = has1 ? lm : (v2f){lm.x, kBig};
    *reinterpret_cast<v2f *>(Lsp + d0) = has1 ? sp : (v2f){sp.x, 1.0f};
    *reinterpret_cast<v2f *>(LE + d0) = BWD ? em * sp : em;
    if (BWD) *reinterpret_cast<v2f *>(Lu + d0) = splat(0.0f);
    if (q < 2) {      // sentinels: the aligned pair in front, and the one (or 1.5) behind
      const int r0 = p * RS;
      for (int t = q; t < 3; t += LP) {
        if (t < PAD) { Llen[r0 + t] = -kBig; Lsp[r0 + t] = 1.0f; LE[r0 + t] = 0.0f; if (BWD) Lu[r0 + t] = 0.0f; }
        const int eb = r0 + PAD + ((K + 1) & ~1) + t;
        if (eb < r0 + RS) { Llen[eb] = kBig; Lsp[eb] = 1.0f; LE[eb] = 0.0f; if (BWD) Lu[eb] = 0.0f; }
      }
    }
  }
  __syncthreads();
  // sortedness: own pair, and the seam to the previous pair
  if (active && (!(lm.x <= lm.y) && has1)) L.unsorted[p] = 1;
  if (active && q > 0 && !(Llen[d0 - 1] <= lm.x)) L.unsorted[p] = 1;
  const bool head = in_wg && (lane == 0 || q == 0);
  if (!BWD && cnt_in == nullptr) {   // assigned-slot count: two ballots, one LDS atomic per (wave, pixel) run
    const unsigned long long m0 = __ballot(id0 >= 0), m1 = __ballot(id1 >= 0);
    const int lo = max(0, lane - q), hi = min(63, lane + (LP - 1 - q));
    const unsigned long long seg = ((hi - lo == 63) ? ~0ull : ((1ull << (hi - lo + 1)) - 1ull) << lo);
    if (head) atomicAdd(&L.cnt[p], __popcll(m0 & seg) + __popcll(m1 & seg));
  }
  float mx = fmaxf((em.x != 0.0f) ? kSat * __builtin_amdgcn_rcpf(sm.x) : 0.0f,
                   (em.y != 0.0f) ? kSat * __builtin_amdgcn_rcpf(sm.y) : 0.0f);
  // Exclusive prefix (over lanes of the pixel) of the per-lane sums E0 + E1, Hillis-Steele with the
  // window radius riding along; association = function of the lane's index in the pixel only.
  float ex = 0.0f;   // sum of E over the slots in front of this lane's pair
  if (!BWD) {
    v2f x = {em.x + em.y, mx};
    L.scan[0][tid] = x;
    __syncthreads();
    v2f y = (q > 0 && in_wg) ? L.scan[0][tid - 1] : splat(0.0f);
    x = (v2f){y.x, fmaxf(mx, y.y)};      // exclusive sum so far, inclusive max so far
    int par = 1;
    for (int o = 1; o < LP; o <<= 1) {
      L.scan[par][tid] = x;
      __syncthreads();
      if (q > o && in_wg) {               // element q-1-o of the shifted sequence exists
        const v2f z = L.scan[par][tid - o];
        x.x += z.x;
        x.y = fmaxf(x.y, z.y);
      } else if (q == o && in_wg) {
        x.y = fmaxf(x.y, L.scan[par][tid - o].y);
      }
      par ^= 1;
    }
    ex = x.x;
    if (in_wg && q == LP - 1) L.rmaxi[p] = __float_as_int(x.y);
  } else {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float y = __shfl_down(mx, o, 64);
      if (lane + o < 64 && q + o < LP) mx = fmaxf(mx, y);
    }
    if (head) atomicMax(&L.rmaxi[p], __float_as_int(mx));
  }
  __syncthreads();
  const float rwin_all = in_wg ? __int_as_float(L.rmaxi[p]) : 0.0f;
  const bool sorted = active && (L.unsorted[in_wg ? p : 0] == 0);
  const float rwin = sorted ? rwin_all : 0.0f;
  const bool any_e = (em.x != 0.0f) || (em.y != 0.0f);
  const float h0 = __builtin_amdgcn_exp2f(kQ0);      // h(0), exactly what h_pair(0) returns
  const float gap = lm.y - lm.x;                     // >= 0 in a sorted list

  // ---- rows r0 = 2q, r1 = 2q+1 ----------------------------------------------------------------
  v2f S = splat(0.0f), rterm = splat(0.0f);
  if (any_e && sorted) {
    v2f accF0 = splat(0.0f), accF1 = splat(0.0f), accB0 = splat(0.0f), accB1 = splat(0.0f);
    v2f accR0 = splat(0.0f), accR1 = splat(0.0f);
    const v2f Es = BWD ? em * sp : em;               // what LE holds
    {   // diagonal block: column 1 is behind row 0, column 0 in front of row 1
      const v2f xp = (v2f){gap * sp.y, gap * sp.x};
      if (!BWD) {
        const v2f h = h_pair(xp);
        accF0.x = em.x * h0;            // (r0, c0) self
        accB0.x = em.y * h.x;           // (r0, c1)
        accF1.x = em.x * h.y;           // (r1, c0)
        accF1.y = em.y * h0;            // (r1, c1) self
      } else {
        const v2f g = gauss_pair(xp);
        accR0 = (v2f){Es.x, Es.y * g.x};
        accR1 = (v2f){Es.x * g.y, Es.y};
      }
    }
    const v2f lm0 = splat(lm.x), lm1 = splat(lm.y);
    const float lmB = (em.y != 0.0f) ? lm.y : lm.x;     // the row that decides how far back to walk
    for (int e = d0 - 2;; e -= 2) {      // column pairs in front of both rows; row 0 is the nearer one
      const v2f l2 = ld2(Llen, e), s2 = ld2(Lsp, e), E2 = ld2(LE, e);
      const v2f dA = lm0 - l2, dB = lm1 - l2;
      if (!(dA.y < rwin)) break;
      const v2f xa = dA * s2, xb = dB * s2;
      if (!BWD) { accF0 = pk_fma(E2, h_pair(xa), accF0); accF1 = pk_fma(E2, h_pair(xb), accF1); }
      else { accR0 = pk_fma(E2, gauss_pair(xa), accR0); accR1 = pk_fma(E2, gauss_pair(xb), accR1); }
    }
    for (int e = d0 + 2;; e += 2) {      // column pairs behind both rows; row 1 is the nearer one
      const v2f l2 = ld2(Llen, e), s2 = ld2(Lsp, e), E2 = ld2(LE, e);
      const v2f dA = l2 - lm0, dB = l2 - lm1;
      if (!(l2.x - lmB < rwin)) break;
      const v2f xa = dA * s2, xb = dB * s2;
      if (!BWD) { accB0 = pk_fma(E2, h_pair(xa), accB0); accB1 = pk_fma(E2, h_pair(xb), accB1); }
      else { accR0 = pk_fma(E2, gauss_pair(xa), accR0); accR1 = pk_fma(E2, gauss_pair(xb), accR1); }
    }
    if (!BWD) {
      const float pre0 = ex + em.x, pre1 = pre0 + em.y;     // inclusive prefix sums of E
      S.x = (pre0 - (accF0.x + accF0.y)) + (accB0.x + accB0.y);
      S.y = (pre1 - (accF1.x + accF1.y)) + (accB1.x + accB1.y);
    } else {
      rterm = (v2f){(accR0.x + accR0.y), (accR1.x + accR1.y)} * splat(kRsqrtPi / kCs);
    }
  } else if (any_e && active) {          // unsorted list: every column, signs from the data
    const int r0 = d0 - k0;
    for (int j = 0; j < K; ++j) {
      const float Ej = LE[r0 + j];
      if (Ej == 0.0f) continue;
      const float lj = Llen[r0 + j], sj = Lsp[r0 + j];
      const v2f xp = (lm - splat(lj)) * splat(sj);
      if (!BWD) {
        const v2f h = h_pair(abs2(xp));
        S.x = fmaf(Ej, xp.x >= 0.0f ? 1.0f - h.x : h.x, S.x);
        S.y = fmaf(Ej, xp.y >= 0.0f ? 1.0f - h.y : h.y, S.y);
      } else {
        const v2f xc = (v2f){fminf(fabsf(xp.x), 16.0f), fminf(fabsf(xp.y), 16.0f)};
        rterm = pk_fma(splat(Ej * (kRsqrtPi / kCs)), gauss_pair(xc), rterm);
      }
    }
  }
  if (!BWD) {
    v2f w;
    w.x = (em.x != 0.0f) ? FAST_EXP(-occ * S.x) * em.x * kInvNorm : 0.0f;
    w.y = (em.y != 0.0f) ? FAST_EXP(-occ * S.y) * em.y * kInvNorm : 0.0f;
    if (active) {
      if (vec) *reinterpret_cast<v2f *>(out0 + f) = w;
      else { out0[f] = w.x; if (has1) out0[f + 1] = w.y; }
      if (q == 0) valid_num[pix] = (cnt_in != nullptr) ? (int64_t)cnt_in[pix] : (int64_t)L.cnt[p];
    }
    return;
  }
  // ---- backward: u = g_w * w, suffix sums, then the lane's two columns --------------------------
  const v2f um = gw * wg;
  if (in_wg) *reinterpret_cast<v2f *>(Lu + d0) = um;
  float sx = 0.0f;   // sum of u over the slots behind this lane's pair
  {
    float(*sf)[kCompThreads] = reinterpret_cast<float(*)[kCompThreads]>(L.scan);
    sf[0][tid] = um.x + um.y;
    __syncthreads();
    float x = (q + 1 < LP && in_wg) ? sf[0][tid + 1] : 0.0f;
    int par = 1;
    for (int o = 1; o < LP; o <<= 1) {
      sf[par][tid] = x;
      __syncthreads();
      if (q + 1 + o < LP && in_wg) x += sf[par][tid + o];
      par ^= 1;
    }
    sx = x;
  }
  __syncthreads();
  v2f ga = splat(0.0f), gl = splat(0.0f), gd = splat(0.0f);
  if (any_e && active) {
    v2f cPhi, cphi, cphil;     // per column (x: column 2q, y: column 2q+1)
    if (sorted) {
      const v2f rj = (v2f){em.x != 0.0f ? kSat * __builtin_amdgcn_rcpf(sm.x) : 0.0f,
                           em.y != 0.0f ? kSat * __builtin_amdgcn_rcpf(sm.y) : 0.0f};   // an empty column needs no rows
      // accumulators over the two rows of a pair: aX* rows behind (incl. self), bX* rows in front
      v2f aH0, aP0, aL0, aH1, aP1, aL1, bH0 = splat(0.0f), bP0 = splat(0.0f), bL0 = splat(0.0f), bH1, bP1, bL1;
      {   // diagonal block: row 1 is behind column 0, row 0 in front of column 1
        const v2f xp = (v2f){gap * sp.x, gap * sp.y};      // (row1, col0), (row0, col1)
        const v2f g = gauss_pair(xp), h = h_pair(xp);
        aH0 = (v2f){um.x * h0, um.y * h.x}; aP0 = (v2f){um.x, um.y * g.x}; aL0 = (v2f){0.0f, um.y * g.x * gap};
        aH1 = (v2f){um.y * h0, 0.0f};       aP1 = (v2f){um.y, 0.0f};       aL1 = splat(0.0f);
        bH1 = (v2f){um.x * h.y, 0.0f};      bP1 = (v2f){um.x * g.y, 0.0f}; bL1 = (v2f){um.x * g.y * gap, 0.0f};
      }
      const v2f lm0 = splat(lm.x), lm1 = splat(lm.y), sp0 = splat(sp.x), sp1 = splat(sp.y);
      for (int e = d0 + 2;; e += 2) {      // row pairs behind both columns; column 1 is the nearer one
        const v2f l2 = ld2(Llen, e), u2 = ld2(Lu, e);
        const v2f dA = l2 - lm0, dB = l2 - lm1;
        if (!(dA.x < rj.x) && !(dB.x < rj.y)) break;
        const v2f xa = dA * sp0, xb = dB * sp1;
        const v2f ya = u2 * gauss_pair(xa), yb = u2 * gauss_pair(xb);
        aH0 = pk_fma(u2, h_pair(xa), aH0); aP0 = aP0 + ya; aL0 = pk_fma(ya, dA, aL0);
        aH1 = pk_fma(u2, h_pair(xb), aH1); aP1 = aP1 + yb; aL1 = pk_fma(yb, dB, aL1);
      }
      for (int e = d0 - 2;; e -= 2) {      // row pairs in front of both columns; column 0 is the nearer one
        const v2f l2 = ld2(Llen, e), u2 = ld2(Lu, e);
        const v2f dA = lm0 - l2, dB = lm1 - l2;
        if (!(dA.y < rj.x) && !(dB.y < rj.y)) break;
        const v2f xa = dA * sp0, xb = dB * sp1;
        const v2f ya = u2 * gauss_pair(xa), yb = u2 * gauss_pair(xb);
        bH0 = pk_fma(u2, h_pair(xa), bH0); bP0 = bP0 + ya; bL0 = pk_fma(ya, dA, bL0);
        bH1 = pk_fma(u2, h_pair(xb), bH1); bP1 = bP1 + yb; bL1 = pk_fma(yb, dB, bL1);
      }
      const float suf1 = sx + um.y, suf0 = suf1 + um.x;     // inclusive suffix sums of u
      cPhi = (v2f){(suf0 - (aH0.x + aH0.y)) + (bH0.x + bH0.y), (suf1 - (aH1.x + aH1.y)) + (bH1.x + bH1.y)};
      cphi = (v2f){(aP0.x + aP0.y) + (bP0.x + bP0.y), (aP1.x + aP1.y) + (bP1.x + bP1.y)} * splat(kRsqrtPi);
      cphil = (v2f){(aL0.x + aL0.y) - (bL0.x + bL0.y), (aL1.x + aL1.y) - (bL1.x + bL1.y)} * splat(kRsqrtPi);
    } else {
      cPhi = splat(0.0f); cphi = splat(0.0f); cphil = splat(0.0f);
      const int r0 = d0 - k0;
      for (int m = 0; m < K; ++m) {
        const float ur = Lu[r0 + m];
        if (ur == 0.0f) continue;
        const v2f dl = splat(Llen[r0 + m]) - lm;
        const v2f xp = dl * sp;
        const v2f h = h_pair(abs2(xp));
        const v2f xc = (v2f){fminf(fabsf(xp.x), 16.0f), fminf(fabsf(xp.y), 16.0f)};
        const v2f ph = splat(ur * kRsqrtPi) * gauss_pair(xc);
        cPhi.x = fmaf(ur, xp.x >= 0.0f ? 1.0f - h.x : h.x, cPhi.x);
        cPhi.y = fmaf(ur, xp.y >= 0.0f ? 1.0f - h.y : h.y, cPhi.y);
        cphi = cphi + ph;
        cphil = pk_fma(ph, dl, cphil);
      }
    }
    ga = pk_fma(splat(occ) * em, cPhi, -um);
    gl = splat(-occ) * (um * rterm - em * sm * cphi);
    gd = splat(-occ) * em * cphil * (v2f){0.5f * __builtin_amdgcn_rcpf(sm.x), 0.5f * __builtin_amdgcn_rcpf(sm.y)};
    if (em.x == 0.0f) { ga.x = 0.0f; gl.x = 0.0f; gd.x = 0.0f; }
    if (em.y == 0.0f) { ga.y = 0.0f; gl.y = 0.0f; gd.y = 0.0f; }
  }
  if (active) {
    if (vec) {
      *reinterpret_cast<v2f *>(out0 + f) = ga; *reinterpret_cast<v2f *>(out1 + f) = gl; *reinterpret_cast<v2f *>(out2 + f) = gd;
    } else {
      out0[f] = ga.x; out1[f] = gl.x; out2[f] = gd.x;
      if (has1) { out0[f + 1] = ga.y; out1[f + 1] = gl.y; out2[f + 1] = gd.y; }
    }
  }
}

}  // namespace voge

using namespace voge;

static int launch_composite(int mode, const int32_t *idx, const float *act, const float *len, const float *dsd,
                            const float *w_in, const float *g_weight, const int32_t *cnt_in, float occ, long npix, int K, float *o0, float *o1,
                            float *o2, int64_t *valid_num, voge_stream_t stream) {
#ifndef VOGE_COMP_T
#define VOGE_COMP_T 256
#endif
  // workgroup size: the smallest multiple of 64 that is >= VOGE_COMP_T and holds one pixel
  const int threads = (K <= VOGE_COMP_T) ? VOGE_COMP_T : kCompThreads;
  hipStream_t st = (hipStream_t)stream;
#ifndef VOGE_COMP_ONE_SLOT
  if (mode != 1) {   // two slots per lane
    const int ppw2 = kCompThreads / comp2_lanes(K);
    const dim3 grid2((unsigned)((npix + ppw2 - 1) / ppw2)), block2(kCompThreads);
    const size_t lds2 = comp2_lds_bytes(K, mode != 0);
    if (mode == 2)
      hipLaunchKernelGGL(composite2_kernel<2>, grid2, block2, lds2, st, idx, act, len, dsd, w_in, g_weight, cnt_in, occ, npix,
                         K, ppw2, o0, o1, o2, valid_num);
    else
      hipLaunchKernelGGL(composite2_kernel<0>, grid2, block2, lds2, st, idx, act, len, dsd, w_in, g_weight, cnt_in, occ, npix,
                         K, ppw2, o0, o1, o2, valid_num);
    return launch_status();
  }
#endif
  const int ppw = threads / K;
  const dim3 grid((unsigned)((npix + ppw - 1) / ppw)), block(threads);
  const size_t lds = comp_lds_bytes(K, mode != 0);
  if (mode == 2)
    hipLaunchKernelGGL(composite_kernel<2>, grid, block, lds, st, idx, act, len, dsd, w_in, g_weight, cnt_in, occ, npix, K, ppw, o0,
                       o1, o2, valid_num);
  else if (mode == 1)
    hipLaunchKernelGGL(composite_kernel<1>, grid, block, lds, st, idx, act, len, dsd, w_in, g_weight, cnt_in, occ, npix, K, ppw, o0,
                       o1, o2, valid_num);
  else
    hipLaunchKernelGGL(composite_kernel<0>, grid, block, lds, st, idx, act, len, dsd, w_in, g_weight, cnt_in, occ, npix, K, ppw, o0,
                       o1, o2, valid_num);
  return launch_status();
}

extern "C" int voge_composite_fwd(const int32_t *idx, const int32_t *cnt, const float *act, const float *len,
                                  const float *dsd, float occ, long npix, int K, float *weight,
                                  int64_t *valid_num, voge_stream_t stream) {
  if (npix < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K || K > kCompThreads) return VOGE_ERR_K_TOO_LARGE;
  if (npix == 0) return 0;
  if ((!idx && !cnt) || !act || !len || !dsd || !weight || !valid_num) return VOGE_ERR_BAD_ARG;
  return launch_composite(0, idx, act, len, dsd, nullptr, nullptr, cnt, occ, npix, K, weight, nullptr, nullptr, valid_num, stream);
}

extern "C" int voge_composite_bwd(const float *act, const float *len, const float *dsd, const float *weight,
                                  const int32_t *cnt, const float *g_weight, float occ, long npix, int K, float *g_act,
                                  float *g_len, float *g_dsd, voge_stream_t stream) {
  if (npix < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K || K > kCompThreads) return VOGE_ERR_K_TOO_LARGE;
  if (npix == 0) return 0;
  if (!act || !len || !dsd || !g_weight || !g_act || !g_len || !g_dsd) return VOGE_ERR_BAD_ARG;
  return launch_composite(weight ? 2 : 1, nullptr, act, len, dsd, weight, g_weight, cnt, occ, npix, K, g_act, g_len, g_dsd,
                          nullptr, stream);
}
