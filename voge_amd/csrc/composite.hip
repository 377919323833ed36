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
  const int ppw = threads / K;
  const dim3 grid((unsigned)((npix + ppw - 1) / ppw)), block(threads);
  const size_t lds = comp_lds_bytes(K, mode != 0);
  hipStream_t st = (hipStream_t)stream;
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
