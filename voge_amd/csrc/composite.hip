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
#include "composite_core.h"

namespace voge {

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

// (the wave form keeps no per-workgroup state: no CompLds block in front of the arrays)
__host__ __device__ inline size_t compn_lds_bytes(const int K, const int NS, const bool bwd, const int threads, const bool wave) {
  // (+ one cell per pixel of a wave: the window radius, composite_core.h)
  return (wave ? 0 : sizeof(CompLds)) + sizeof(float) * (size_t)compn_rows(K, NS, threads, wave) * (bwd ? 5 : 3) + (wave ? 64 * sizeof(unsigned) : 0);      // (backward: + u, + the row sums of compn_bwd_wave<NS, true>)
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
// NS (2 or 4) consecutive slots per lane (forward, and backward with the forward's weights).
// A lane owns the aligned group of slots [NS q, NS q + NS) of its pixel, so every pair of list
// entries it reads from LDS serves 2 NS (row, column) evaluations, the lane's own group is the
// diagonal block (registers only), global loads / stores are 4 NS bytes wide and the per-pixel
// scans run over K / NS lanes.  The one-slot kernel above remains the reference form: it handles
// the backward without given weights and is what VOGE_COMP_ONE_SLOT=1 builds select.
// ------------------------------------------------------------------------------------------
#ifndef VOGE_COMP_WPE
#define VOGE_COMP_WPE 1
#endif
#ifndef VOGE_CS_LDS
#define VOGE_CS_LDS 1      // the shade sums' cross-lane reduction through LDS columns instead of a shuffle tree (see the shade block)
#endif
#ifndef VOGE_COMP_LDS_RMAX
#define VOGE_COMP_LDS_RMAX 1
#endif
// WAVE: every pixel's lanes sit inside ONE wave (64 / LP pixels per wave, the remaining lanes idle), so the
// per-pixel scans, flags and reductions are wave shuffles / ballots and the kernel has no workgroup barrier at
// all: each wave runs from its loads to its stores on its own.  (Needs LP <= 64.)
// SC > 0 (forward from the records, wave form): the SHADE stage rides in the same pass -- merge_final + get_silhouette +
// to_colored_background (VoGE/Aggregation.py:111-141, VoGE/Renderer.py:157-171) for SC colour channels: the lane gathers
// the colours of its own slots (their indices are in registers), the pixel's lanes sum w * colour and w, lane 0 writes
// rgb / wsum / img, and empty slots of the index list are rewritten -1 -> 0 as merge_final does (:131).  What a separate
// shade kernel would read again (idx and weight: 8 bytes per slot) is never read.
struct CompShade {
  const float *colors, *bg;
  float thr;
  long Nattr;
  float *rgb, *img, *wsum;
  int32_t *idx_fix;
  // (round 6) zero_n4 float4s this launch sets to zero on its way: the accumulator of the frame's backward
  // (voge_frame_shade_fwd_iso's bwd_acc) -- no fill launch in front of that backward.  The first ceil(zero_n4 / 64) workgroups
  // store one float4 per thread; for everybody else it is one scalar compare (the kernel is VALU-bound: as a grid-stride loop
  // in every thread the zeroing cost 1.3 us)
  float4 *zero_p = nullptr;
  long zero_n4 = 0;
  float *sil = nullptr;      // (round 6) NULL | [npix]: get_silhouette = min(sum_k w_k, 1) (Renderer.py:157-159), written with the sum
};
// GEN (forward from the records): the records are the general path's packed (mu, A), three float4 per Gaussian
// (voge_trace_lean_fwd); act / dsd come from make_eval + pair_eval, the operations of the sweep's own epilogue.
// GEN 2 (round 6): the compact per-axis records (mu, a0, a1, a2, 0, 0), two float4 per Gaussian (voge_frame_trace_fwd_gen, kind 1);
// act / dsd from pair_eval_diag -- the general chain's bits without its zero coefficients.
template <int MODE, int NS, bool WAVE, typename OffT, int SC = 0, int GEN = 0>   // MODE 0: forward, 2: backward with weights; OffT: composite_core.h at_bytes
__global__ void __launch_bounds__(kCompThreads) __attribute__((amdgpu_waves_per_eu(VOGE_COMP_WPE)))
compositen_kernel(const int32_t *__restrict__ idx, const float *__restrict__ act,
                  const float *__restrict__ len, const float *__restrict__ dsd,
                  const float *__restrict__ w_in, const float *__restrict__ g_weight,
                  const int32_t *__restrict__ cnt_in, const float occ, const long npix, const int K, const int ppw,
                  float *__restrict__ out0 /* weight | g_act */, float *__restrict__ out1 /* g_len */,
                  float *__restrict__ out2 /* g_dsd */, int64_t *__restrict__ valid_num,
                  const float4 *__restrict__ rec /* forward with act == NULL: [P] (mu, a) */, const float *__restrict__ rays,
                  const CompShade sh = CompShade{}) {
  static_assert(SC == 0 || (MODE == 0 && WAVE && SC <= 4), "the shade stage rides in the wave-form forward only");
  constexpr bool BWD = MODE != 0;
  if (SC > 0 && (long)blockIdx.x * blockDim.x < sh.zero_n4) {      // (uniform; the launch covers zero_n4: composite_shade_fwd_impl)
    const long zi = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (zi < sh.zero_n4) sh.zero_p[zi] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  constexpr int NP = NS / 2;       // own aligned pairs
  extern __shared__ __attribute__((aligned(16))) unsigned char comp_smem[];
  CompLds &L = *reinterpret_cast<CompLds *>(comp_smem);      // (workgroup form only)
  const int rows = compn_rows(K, NS, (int)blockDim.x, WAVE);
  float *const Llen = reinterpret_cast<float *>(comp_smem + (WAVE ? 0 : sizeof(CompLds)));
  float *const Lsp = Llen + rows;
  float *const LE = Lsp + rows;     // E (forward) or E * s' (backward)
  float *const Lu = LE + rows;      // backward only
  float *const LR = Lu + rows;      // backward, wave form: the row sums the column walks accumulate (composite_core.h)
  const int tid = threadIdx.x, lane = tid & 63;
  const int LP = compn_lanes(K, NS);
  int p, q;
  bool in_wg;
  if (WAVE) {
    const int pw = 64 / LP;                       // pixels per wave
    const int pl = __float2int_rz(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)LP));
    q = lane - pl * LP;
    in_wg = pl < pw;
    p = (tid >> 6) * pw + (in_wg ? pl : 0);
  } else {
    p = __float2int_rz(((float)tid + 0.5f) * __builtin_amdgcn_rcpf((float)LP));
    q = tid - p * LP;
    in_wg = p < ppw;
  }
  const long pix = (long)blockIdx.x * ppw + p;
  // lanes of this lane's pixel inside the wave (WAVE: the whole pixel)
  const int seg_lo = max(0, lane - q), seg_hi = min(63, lane + (LP - 1 - q));
  const unsigned long long seg = ((seg_hi - seg_lo == 63) ? ~0ull : ((1ull << (seg_hi - seg_lo + 1)) - 1ull) << seg_lo);
  const bool active = in_wg && (pix < npix);
  const int k0 = NS * q;
  const long f = pix * K + k0;
  const OffT fb = (OffT)f * (OffT)4;      // byte offset of the lane's group in the [npix][K] arrays
  const int RS = compn_stride(K, NS);
  const int PAD = comp_pad(K);
  const int d0 = (in_wg ? p : 0) * RS + PAD + (in_wg ? k0 : 0);   // own group in the padded arrays (even index)
  const bool vec = (K % NS) == 0;                                    // whole groups, 4 NS-byte aligned accesses
  bool has[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) has[a] = k0 + a < K;
  int lead = K;
  if (cnt_in != nullptr) {
    lead = active ? min(K, max(0, cnt_in[pix])) : 0;
    if (WAVE ? !__any(lead > 0) : !__syncthreads_or(lead > 0)) {      // every pixel of the workgroup (wave) is empty
      if (active) {
#pragma unroll
        for (int a = 0; a < NS; ++a)
          if (has[a]) { out0[f + a] = 0.0f; if (BWD) { out1[f + a] = 0.0f; out2[f + a] = 0.0f; } if (SC > 0) sh.idx_fix[f + a] = 0; }
        if (!BWD && q == 0) valid_num[pix] = 0;
        if (SC > 0 && q == 0) {      // nothing was hit: the background
          sh.wsum[pix] = 0.0f;
          if (sh.sil != nullptr) sh.sil[pix] = 0.0f;
#pragma unroll
          for (int c = 0; c < SC; ++c) {
            sh.rgb[pix * SC + c] = 0.0f;
            if (sh.img != nullptr) sh.img[pix * SC + c] = fminf(sh.bg[c], 1.0f);
          }
        }
      }
      return;
    }
  }
  if (!WAVE && tid < ppw) { L.unsorted[tid] = 0; L.rmaxi[tid] = 0; L.cnt[tid] = 0; }
  // what a sentinel slot evaluates to: E = 0, len = 1e10, s = 1e-5
  float lm[NS], sm[NS], em[NS], gw[NS], wg[NS];
  int id[NS];
  int ivk[NS];      // (SC > 0) the lane's indices, kept for the colour gathers
#pragma unroll
  for (int a = 0; a < NS; ++a) { lm[a] = VOGE_SENT_LEN; sm[a] = 1e-5f; em[a] = 0.0f; gw[a] = 0.0f; wg[a] = 0.0f; id[a] = -1; ivk[a] = -1; }
  if (!BWD && act == nullptr) {
    // Fragments without act / dsd in memory (voge_fragments_fwd_iso*): the sweep wrote index and len only; act and dsd
    // of A = a I are re-derived here from the SAME records with the SAME operations its epilogue would have used
    // (pair_eval_iso_at), so the weights are bit-identical to the two-call form.  (host: counts given)
    if (active && k0 < lead) {
      const float3 d = at_bytes<float3>(rays, (OffT)pix * (OffT)12);
      const float dn2 = (d.x * d.x + d.y * d.y) + d.z * d.z;
      int iv[NS];
      float lv[NS];
      if (!vec) {                 // K not a multiple of NS (odd K): the group is read slot by slot
#pragma unroll
        for (int a = 0; a < NS; ++a) {
          iv[a] = 0; lv[a] = 0.0f;
          if (k0 + a < lead) { iv[a] = idx[f + a]; lv[a] = len[f + a]; }
        }
      } else if (NS == 4) {
        const int4 i4 = at_bytes<int4>(idx, fb);
        const float4 l4 = at_bytes<float4>(len, fb);
        iv[0] = i4.x; iv[1] = i4.y; iv[NS - 2] = i4.z; iv[NS - 1] = i4.w;
        lv[0] = l4.x; lv[1] = l4.y; lv[NS - 2] = l4.z; lv[NS - 1] = l4.w;
      } else {
        const int2 i2 = at_bytes<int2>(idx, fb);
        const v2f l2 = at_bytes<v2f>(len, fb);
        iv[0] = i2.x; iv[1] = i2.y; lv[0] = l2.x; lv[1] = l2.y;
      }
      constexpr int NR = GEN == 1 ? 3 : (GEN == 2 ? 2 : 1);      // float4s per record
      float4 rc[NS][NR];
#pragma unroll
      for (int a = 0; a < NS; ++a)
#pragma unroll
        for (int r = 0; r < NR; ++r)
          rc[a][r] = at_bytes<float4>(rec, (uint32_t)max(iv[a], 0) * (16u * NR) + 16u * r);      // (a slot beyond the count: record 0, selected out below -- no exec-mask region per slot)
      if (SC > 0) {
#pragma unroll
        for (int a = 0; a < NS; ++a) ivk[a] = (k0 + a < lead) ? iv[a] : -1;
      }
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        if (GEN && !(k0 + a < lead)) continue;
        {
          PairOut o;
          if (GEN == 2) {
            const float4 r0 = rc[a][0], r1 = rc[a][NR - 1];
            o = pair_eval_diag(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, d.x, d.y, d.z);
            if (out1 != nullptr) { out1[f + a] = o.act; out2[f + a] = o.dsd; }
          } else if (GEN) {
            const float4 r0 = rc[a][0], r1 = rc[a][GEN == 1 ? 1 : 0], r2 = rc[a][GEN == 1 ? 2 : 0];
            const float A[9] = {r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
            o = pair_eval(r0.x, r0.y, r0.z, make_eval(r0.x, r0.y, r0.z, A), d.x, d.y, d.z, d.x * d.x, d.y * d.y, d.z * d.z,
                          d.x * d.y, d.x * d.z, d.y * d.z);
            // (optional: act / dsd kept for the fused backward, which is VALU-bound and would rather read 8 bytes per slot
            // than gather 48 and evaluate again: out1 / out2 are free in the forward)
            if (out1 != nullptr) { out1[f + a] = o.act; out2[f + a] = o.dsd; }
          } else {
            o = pair_eval_iso_at(rc[a][0].x, rc[a][0].y, rc[a][0].z, rc[a][0].w, lv[a], d.x, d.y, d.z, dn2);
          }
          const bool lv_on = GEN || (k0 + a < lead);      // (scalar sigmas: evaluated for every slot, kept for the live ones)
          em[a] = lv_on ? FAST_EXP(-o.act) : 0.0f; lm[a] = lv_on ? lv[a] : VOGE_SENT_LEN; sm[a] = lv_on ? FAST_SQRT(o.dsd + 1e-10f) : 1e-5f;
        }
      }
    }
  } else if (active) {
    if (vec && k0 + NS <= lead) {        // the whole group is live: one wide access per array
      float av[NS], lv[NS], dv[NS];
      if (NS == 4) {
        const float4 a4 = at_bytes<float4>(act, fb), l4 = at_bytes<float4>(len, fb),
                     d4 = at_bytes<float4>(dsd, fb);
        av[0] = a4.x; av[1] = a4.y; av[NS - 2] = a4.z; av[NS - 1] = a4.w;
        lv[0] = l4.x; lv[1] = l4.y; lv[NS - 2] = l4.z; lv[NS - 1] = l4.w;
        dv[0] = d4.x; dv[1] = d4.y; dv[NS - 2] = d4.z; dv[NS - 1] = d4.w;
        if (BWD) {
          const float4 g4 = at_bytes<float4>(g_weight, fb), w4 = at_bytes<float4>(w_in, fb);
          gw[0] = g4.x; gw[1] = g4.y; gw[NS - 2] = g4.z; gw[NS - 1] = g4.w;
          wg[0] = w4.x; wg[1] = w4.y; wg[NS - 2] = w4.z; wg[NS - 1] = w4.w;
        }
      } else {
        const v2f a2 = at_bytes<v2f>(act, fb), l2 = at_bytes<v2f>(len, fb),
                  d2 = at_bytes<v2f>(dsd, fb);
        av[0] = a2.x; av[1] = a2.y; lv[0] = l2.x; lv[1] = l2.y; dv[0] = d2.x; dv[1] = d2.y;
        if (BWD) {
          const v2f g2 = at_bytes<v2f>(g_weight, fb), w2 = at_bytes<v2f>(w_in, fb);
          gw[0] = g2.x; gw[1] = g2.y; wg[0] = w2.x; wg[1] = w2.y;
        }
      }
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        em[a] = FAST_EXP(-av[a]); lm[a] = lv[a]; sm[a] = FAST_SQRT(dv[a] + 1e-10f);
        if (!BWD && cnt_in == nullptr) id[a] = idx[f + a];
      }
    } else {
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        if (has[a] && k0 + a < lead) {
          em[a] = FAST_EXP(-act[f + a]); lm[a] = len[f + a]; sm[a] = FAST_SQRT(dsd[f + a] + 1e-10f);
          if (BWD) { gw[a] = g_weight[f + a]; wg[a] = w_in[f + a]; } else if (cnt_in == nullptr) id[a] = idx[f + a];
        }
      }
    }
  }
  float sp[NS], Es[NS];           // s' = s sqrt(log2 e); what LE holds
#pragma unroll
  for (int a = 0; a < NS; ++a) { sp[a] = sm[a] * kCs; Es[a] = BWD ? em[a] * sp[a] : em[a]; }
  if (in_wg) {
    // (slots of the last group beyond K are back sentinels: len = +big, E = 0)
#pragma unroll
    for (int h2 = 0; h2 < NP; ++h2) {
      const int a = 2 * h2;
      *reinterpret_cast<v2f *>(Llen + d0 + a) = (v2f){has[a] ? lm[a] : kBig, has[a + 1] ? lm[a + 1] : kBig};
      *reinterpret_cast<v2f *>(Lsp + d0 + a) = (v2f){has[a] ? sp[a] : 1.0f, has[a + 1] ? sp[a + 1] : 1.0f};
      *reinterpret_cast<v2f *>(LE + d0 + a) = (v2f){Es[a], Es[a + 1]};
      if (BWD) { *reinterpret_cast<v2f *>(Lu + d0 + a) = splat(0.0f); if (WAVE) *reinterpret_cast<v2f *>(LR + d0 + a) = splat(0.0f); }
    }
    if (q < 2) {      // the sentinel pair in front of the row and the one behind it
      const int r0 = p * RS;
      for (int t = q; t < 2; t += LP) {
        Llen[r0 + t] = -kBig; Lsp[r0 + t] = 1.0f; LE[r0 + t] = 0.0f; if (BWD) { Lu[r0 + t] = 0.0f; if (WAVE) LR[r0 + t] = 0.0f; }
        const int eb = r0 + RS - 2 + t;
        Llen[eb] = kBig; Lsp[eb] = 1.0f; LE[eb] = 0.0f; if (BWD) { Lu[eb] = 0.0f; if (WAVE) LR[eb] = 0.0f; }
      }
    }
  }
  if (WAVE) wave_lds_sync(); else __syncthreads();
  // sortedness: inside the own group, and the seam to the previous group
  bool uns = false;
  if (active) {
    uns = (q > 0) && !(Llen[d0 - 1] <= lm[0]);
#pragma unroll
    for (int a = 1; a < NS; ++a) uns = uns || (has[a] && !(lm[a - 1] <= lm[a]));
    if (!WAVE && uns) L.unsorted[p] = 1;
  }
  const bool wave_unsorted = WAVE && ((__ballot(uns) & seg) != 0ull);
  const bool head = in_wg && (lane == 0 || q == 0);
  int wave_cnt = 0;
  if (!BWD && cnt_in == nullptr) {   // assigned-slot count: NS ballots, one LDS atomic per (wave, pixel) run
    int c = 0;
#pragma unroll
    for (int a = 0; a < NS; ++a) c += __popcll(__ballot(id[a] >= 0) & seg);
    wave_cnt = c;
    if (!WAVE && head) atomicAdd(&L.cnt[p], c);
  }
  if (BWD && WAVE) {
    // backward, wave form: shared with the fused fragment backward (composite_core.h, fragment_bwd.hip)
    float um[NS], ga[NS], gl[NS], gd[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) um[a] = gw[a] * wg[a];
    compn_bwd_wave<NS, true>(lm, sm, em, um, Llen, Lsp, LE, Lu, d0, k0, K, q, LP, LP, in_wg, active, active && !wave_unsorted, seg_lo,
                             occ, ga, gl, gd, nullptr, LR);
    if (active) {
      if (vec && NS == 4) {
        at_bytes_w<float4>(out0, fb) = make_float4(ga[0], ga[1], ga[NS - 2], ga[NS - 1]);
        at_bytes_w<float4>(out1, fb) = make_float4(gl[0], gl[1], gl[NS - 2], gl[NS - 1]);
        at_bytes_w<float4>(out2, fb) = make_float4(gd[0], gd[1], gd[NS - 2], gd[NS - 1]);
      } else if (vec) {
        at_bytes_w<v2f>(out0, fb) = (v2f){ga[0], ga[1]};
        at_bytes_w<v2f>(out1, fb) = (v2f){gl[0], gl[1]};
        at_bytes_w<v2f>(out2, fb) = (v2f){gd[0], gd[1]};
      } else {
#pragma unroll
        for (int a = 0; a < NS; ++a) if (has[a]) { out0[f + a] = ga[a]; out1[f + a] = gl[a]; out2[f + a] = gd[a]; }
      }
    }
    return;
  }
  if (!BWD && WAVE) {
    // forward, wave form: the row pass shared with the sweep's fused epilogue (composite_core.h)
    float w[NS];
    unsigned *const Lcells = reinterpret_cast<unsigned *>(LE + rows);      // (forward: three row arrays, then the cells)
    compn_fwd_rows<NS>(lm, sm, em, Llen, Lsp, LE, d0, k0, K, q, LP, LP, in_wg, active, active && !wave_unsorted, seg_lo, occ, w,
                       (VOGE_COMP_LDS_RMAX && blockDim.x == 64) ? Lcells + (in_wg ? p : 0) : nullptr);      // (one-wave workgroups)
    if (active) {
      if (vec && NS == 4) at_bytes_w<float4>(out0, fb) = make_float4(w[0], w[1], w[NS - 2], w[NS - 1]);
      else if (vec) at_bytes_w<v2f>(out0, fb) = (v2f){w[0], w[1]};
      else {
#pragma unroll
        for (int a = 0; a < NS; ++a) if (has[a]) out0[f + a] = w[a];
      }
      if (q == 0) valid_num[pix] = (cnt_in != nullptr) ? (int64_t)cnt_in[pix] : (int64_t)wave_cnt;
    }
    if (SC > 0) {
      // ---- shade: sum_k w_k colour[idx_k] and sum_k w_k over the pixel's lanes; blend over the background ----
      float part[SC + 1];
#pragma unroll
      for (int c = 0; c <= SC; ++c) part[c] = 0.0f;
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const int p2 = ivk[a];
        if (SC == 3 || SC == 4) {
          // (no exec-mask region per slot: a slot that contributes nothing reads colour row 0 and multiplies a zero; the
          //  row is selected to zero first, so a non-finite colour there cannot leak)
          const bool on = p2 >= 0 && p2 < sh.Nattr && w[a] != 0.0f;
          const uint32_t o = (uint32_t)(on ? p2 : 0) * (uint32_t)(4 * SC);      // (bytes; Nattr * SC < 2^30: host)
          if (SC == 3) {
            const float3 v = at_bytes<float3>(sh.colors, o);
            part[0] = fmaf(w[a], on ? v.x : 0.0f, part[0]); part[1] = fmaf(w[a], on ? v.y : 0.0f, part[1]);
            part[2] = fmaf(w[a], on ? v.z : 0.0f, part[2]);
          } else {
            const float4 v = at_bytes<float4>(sh.colors, o);
            part[0] = fmaf(w[a], on ? v.x : 0.0f, part[0]); part[1] = fmaf(w[a], on ? v.y : 0.0f, part[1]);
            part[2] = fmaf(w[a], on ? v.z : 0.0f, part[2]); part[SC - 1] = fmaf(w[a], on ? v.w : 0.0f, part[SC - 1]);
          }
        } else if (p2 >= 0 && p2 < sh.Nattr && w[a] != 0.0f) {
          const uint32_t o = (uint32_t)p2 * (uint32_t)(4 * SC);
          {
#pragma unroll
            for (int c = 0; c < SC; ++c) part[c] = fmaf(w[a], at_bytes<float>(sh.colors, o + 4u * c), part[c]);
          }
        }
        part[SC] += w[a];
      }
#if VOGE_CS_LDS
      if (SC == 3 && LP >= 3 && blockDim.x == 64) {      // (uniform; one-wave workgroups: the scratch below is the wave's OWN rows -- in a
                                                          //  workgroup of several waves another wave may still be walking them)
        // Three colour sums and sum w over the pixel's lanes.  Round 6: through LDS -- every lane leaves its four partial sums in the
        // (now idle) row arrays, the pixel's lanes 0 .. 2 each add up ONE colour's column and the weights' column in lane order:
        // two dependent LDS round trips instead of the four of a shuffle tree (which were 5 us of this kernel, -DVOGE_CS_ABL; a
        // tree with fewer shuffles but five dependent steps gained nothing).  A fixed association per pixel, as before.
        float *const A0 = const_cast<float *>(Llen), *const A1 = const_cast<float *>(Lsp), *const A2 = const_cast<float *>(LE);
        wave_lds_sync();      // (the walks' reads of these rows are over)
        constexpr int T_ = 64;      // (rows >= 128 for every K with LP >= 3: compn_rows)
        A0[tid] = part[0]; A1[tid] = part[1]; A2[tid] = part[2]; A0[T_ + tid] = part[3];
        wave_lds_sync();
        float xs = 0.0f, ws = 0.0f;
        if (in_wg && q < 3) {
          const float *const col = (q == 0 ? A0 : (q == 1 ? A1 : A2)) + (tid - q), *const wc = A0 + T_ + (tid - q);
          for (int j = 0; j < LP; ++j) { xs += col[j]; ws += wc[j]; }
        }
        if (active) {
          // merge_final rewrites empty slots of the index list in place, -1 -> 0 (Aggregation.py:131)
          if (k0 + NS > lead) {
#pragma unroll
            for (int a = 0; a < NS; ++a) if (has[a] && k0 + a >= lead) sh.idx_fix[f + a] = 0;
          }
          if (q < 3) {
            float sil = fminf(ws, 1.0f);
            if (sh.thr > 0.0f) sil = sil > sh.thr ? 1.0f : 0.0f;
            sh.rgb[pix * SC + q] = xs;
            if (sh.img != nullptr) sh.img[pix * SC + q] = fminf(fmaf(1.0f - sil, sh.bg[q], xs), 1.0f);
            if (q == 0) {
              sh.wsum[pix] = ws;
              if (sh.sil != nullptr) sh.sil[pix] = fminf(ws, 1.0f);
            }
          }
        }
        return;
      }
#endif
      // the pixel's lanes are consecutive in the wave: sum towards its first lane (a fixed association per pixel)
#pragma unroll
      for (int c = 0; c <= SC; ++c) {
        float x = part[c];
#ifdef VOGE_CS_ABL      // (timing experiment: no cross-lane reduction of the shade sums)
        for (int o = 64; o < LP; o <<= 1) {
#else
        for (int o = 1; o < LP; o <<= 1) {
#endif
          const float y = __shfl_down(x, o, 64);
          if (q + o < LP && in_wg) x += y;
        }
        part[c] = x;
      }
      if (active) {
        // merge_final rewrites empty slots of the index list in place, -1 -> 0 (Aggregation.py:131)
        if (k0 + NS > lead) {
#pragma unroll
          for (int a = 0; a < NS; ++a) if (has[a] && k0 + a >= lead) sh.idx_fix[f + a] = 0;
        }
        if (q == 0) {
          const float ws = part[SC];
          float sil = fminf(ws, 1.0f);
          if (sh.thr > 0.0f) sil = sil > sh.thr ? 1.0f : 0.0f;
          sh.wsum[pix] = ws;
          if (sh.sil != nullptr) sh.sil[pix] = fminf(ws, 1.0f);
#pragma unroll
          for (int c = 0; c < SC; ++c) {
            sh.rgb[pix * SC + c] = part[c];
            if (sh.img != nullptr) sh.img[pix * SC + c] = fminf(fmaf(1.0f - sil, sh.bg[c], part[c]), 1.0f);
          }
        }
      }
    }
    return;
  }
  float mx = 0.0f, esum = 0.0f;
#pragma unroll
  for (int a = 0; a < NS; ++a) {
    mx = fmaxf(mx, (em[a] != 0.0f) ? kSat * __builtin_amdgcn_rcpf(sm[a]) : 0.0f);
    esum += em[a];
  }
  // Exclusive prefix (over the lanes of the pixel) of the per-lane sums of E, Hillis-Steele with the
  // window radius riding along; the association is a function of the lane's index in the pixel only.
  float ex = 0.0f;   // sum of E over the slots in front of this lane's group
  float wave_rmax = 0.0f;
  if (!BWD && WAVE) {            // the same Hillis-Steele steps on wave shuffles (same association)
    v2f x = {esum, mx};
    const v2f y = (v2f){__shfl_up(x.x, 1, 64), __shfl_up(x.y, 1, 64)};
    x = (q > 0 && in_wg) ? (v2f){y.x, fmaxf(mx, y.y)} : (v2f){0.0f, mx};
    for (int o = 1; o < LP; o <<= 1) {
      const v2f z = (v2f){__shfl_up(x.x, o, 64), __shfl_up(x.y, o, 64)};
      if (q >= o && in_wg) {
        x.x += z.x;
        x.y = fmaxf(x.y, z.y);
      }
    }
    ex = x.x;
    wave_rmax = __shfl(x.y, min(63, seg_lo + LP - 1), 64);      // the pixel's last lane holds the maximum
  } else if (!BWD) {
    v2f x = {esum, mx};
    L.scan[0][tid] = x;
    __syncthreads();
    const v2f y = (q > 0 && in_wg) ? L.scan[0][tid - 1] : splat(0.0f);
    x = (v2f){y.x, fmaxf(mx, y.y)};      // exclusive sum so far, inclusive max so far
    int par = 1;
    for (int o = 1; o < LP; o <<= 1) {
      L.scan[par][tid] = x;
      __syncthreads();
      if (q >= o && in_wg) {
        const v2f z = L.scan[par][tid - o];
        x.x += z.x;                        // (z.x = 0 for the pixel's first lane)
        x.y = fmaxf(x.y, z.y);
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
    if (WAVE) wave_rmax = __shfl(mx, seg_lo, 64);                // the pixel's first lane holds the maximum
    else if (head) atomicMax(&L.rmaxi[p], __float_as_int(mx));
  }
  if (!WAVE) __syncthreads();
  const float rwin_all = WAVE ? (in_wg ? wave_rmax : 0.0f) : (in_wg ? __int_as_float(L.rmaxi[p]) : 0.0f);
  const bool sorted = active && (WAVE ? !wave_unsorted : (L.unsorted[in_wg ? p : 0] == 0));
  const float rwin = sorted ? rwin_all : 0.0f;
  bool any_e = false;
#pragma unroll
  for (int a = 0; a < NS; ++a) any_e = any_e || (em[a] != 0.0f);
  const float h0 = __builtin_amdgcn_exp2f(kQ0);      // h(0), exactly what h_pair(0) returns

  // ---- rows = the lane's own slots -------------------------------------------------------------
  float S[NS], rterm[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) { S[a] = 0.0f; rterm[a] = 0.0f; }
  if (any_e && sorted) {
    v2f accF[NS], accB[NS], accR[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) { accF[a] = splat(0.0f); accB[a] = splat(0.0f); accR[a] = splat(0.0f); }
    // diagonal block (registers): for rows a < b, column b is behind row a and column a in front of row b
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      if (!BWD) accF[a].x = em[a] * h0; else accR[a].x = Es[a];         // self
#pragma unroll
      for (int b2 = a + 1; b2 < NS; ++b2) {
        const float gap = lm[b2] - lm[a];
        const v2f xp = (v2f){gap * sp[b2], gap * sp[a]};                 // (row a, col b), (row b, col a)
        if (!BWD) {
          const v2f h = h_pair(xp);
          accB[a].y = fmaf(em[b2], h.x, accB[a].y);
          accF[b2].y = fmaf(em[a], h.y, accF[b2].y);
        } else {
          const v2f g = gauss_pair(xp);
          accR[a].y = fmaf(Es[b2], g.x, accR[a].y);
          accR[b2].y = fmaf(Es[a], g.y, accR[b2].y);
        }
      }
    }
    float lmB = lm[0];                 // the last live row decides how far back to walk
#pragma unroll
    for (int a = 1; a < NS; ++a) lmB = (em[a] != 0.0f) ? lm[a] : lmB;
    for (int e = d0 - 2;; e -= 2) {      // column pairs in front of every own row; row 0 is the nearest
      const v2f l2 = ld2(Llen, e), s2 = ld2(Lsp, e), E2 = ld2(LE, e);
      if (!(lm[0] - l2.y < rwin)) break;
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const v2f xa = (splat(lm[a]) - l2) * s2;
        if (!BWD) accF[a] = pk_fma(E2, h_pair(xa), accF[a]); else accR[a] = pk_fma(E2, gauss_pair(xa), accR[a]);
      }
    }
    for (int e = d0 + NS;; e += 2) {     // column pairs behind every own row
      const v2f l2 = ld2(Llen, e), s2 = ld2(Lsp, e), E2 = ld2(LE, e);
      if (!(l2.x - lmB < rwin)) break;
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const v2f xa = (l2 - splat(lm[a])) * s2;
        if (!BWD) accB[a] = pk_fma(E2, h_pair(xa), accB[a]); else accR[a] = pk_fma(E2, gauss_pair(xa), accR[a]);
      }
    }
    float pre = ex;
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      pre += em[a];                                                     // inclusive prefix sum of E
      if (!BWD) S[a] = (pre - (accF[a].x + accF[a].y)) + (accB[a].x + accB[a].y);
      else rterm[a] = (accR[a].x + accR[a].y) * (kRsqrtPi / kCs);
    }
  } else if (any_e && active) {          // unsorted list: every column, signs from the data
    const int r0 = d0 - k0;
    for (int j = 0; j < K; ++j) {
      const float Ej = LE[r0 + j];
      if (Ej == 0.0f) continue;
      const float lj = Llen[r0 + j], sj = Lsp[r0 + j];
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const float xp = (lm[a] - lj) * sj;
        if (!BWD) {
          const float h = h_one(fabsf(xp));
          S[a] = fmaf(Ej, xp >= 0.0f ? 1.0f - h : h, S[a]);
        } else {
          const float xc = fminf(fabsf(xp), 16.0f);
          rterm[a] = fmaf(Ej * (kRsqrtPi / kCs), __builtin_amdgcn_exp2f(-xc * xc), rterm[a]);
        }
      }
    }
  }
  if (!BWD) {
    float w[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) w[a] = (em[a] != 0.0f) ? FAST_EXP(-occ * S[a]) * em[a] * kInvNorm : 0.0f;
    if (active) {
      if (vec && NS == 4) at_bytes_w<float4>(out0, fb) = make_float4(w[0], w[1], w[NS - 2], w[NS - 1]);
      else if (vec) at_bytes_w<v2f>(out0, fb) = (v2f){w[0], w[1]};
      else {
#pragma unroll
        for (int a = 0; a < NS; ++a) if (has[a]) out0[f + a] = w[a];
      }
      if (q == 0) valid_num[pix] = (cnt_in != nullptr) ? (int64_t)cnt_in[pix] : (int64_t)(WAVE ? wave_cnt : L.cnt[p]);
    }
    return;
  }
  // ---- backward: u = g_w * w, suffix sums, then the lane's own columns --------------------------
  float um[NS], usum = 0.0f;
#pragma unroll
  for (int a = 0; a < NS; ++a) { um[a] = gw[a] * wg[a]; usum += um[a]; }
  if (in_wg) {
#pragma unroll
    for (int h2 = 0; h2 < NP; ++h2) *reinterpret_cast<v2f *>(Lu + d0 + 2 * h2) = (v2f){um[2 * h2], um[2 * h2 + 1]};
  }
  float sx = 0.0f;   // sum of u over the slots behind this lane's group
  if (WAVE) {
    const float y = __shfl_down(usum, 1, 64);
    float x = (q + 1 < LP && in_wg) ? y : 0.0f;
    for (int o = 1; o < LP; o <<= 1) {
      const float z = __shfl_down(x, o, 64);
      if (q + o < LP && in_wg) x += z;
    }
    sx = x;
    wave_lds_sync();
  } else {
    float(*sf)[kCompThreads] = reinterpret_cast<float(*)[kCompThreads]>(L.scan);
    sf[0][tid] = usum;
    __syncthreads();
    float x = (q + 1 < LP && in_wg) ? sf[0][tid + 1] : 0.0f;
    int par = 1;
    for (int o = 1; o < LP; o <<= 1) {
      sf[par][tid] = x;
      __syncthreads();
      if (q + o < LP && in_wg) x += sf[par][tid + o];      // (0 for the pixel's last lane)
      par ^= 1;
    }
    sx = x;
  }
  if (!WAVE) __syncthreads();
  float ga[NS], gl[NS], gd[NS];
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
        bool need = false;
#pragma unroll
        for (int b2 = 0; b2 < NS; ++b2) need = need || (l2.x - lm[b2] < rj[b2]);
        if (!need) break;
#pragma unroll
        for (int b2 = 0; b2 < NS; ++b2) {
          const v2f d = l2 - splat(lm[b2]);
          const v2f xp = d * splat(sp[b2]);
          const v2f y = u2 * gauss_pair(xp);
          aH[b2] = pk_fma(u2, h_pair(xp), aH[b2]); aP[b2] = aP[b2] + y; aL[b2] = pk_fma(y, d, aL[b2]);
        }
      }
      for (int e = d0 - 2;; e -= 2) {      // row pairs in front of every own column
        const v2f l2 = ld2(Llen, e), u2 = ld2(Lu, e);
        bool need = false;
#pragma unroll
        for (int b2 = 0; b2 < NS; ++b2) need = need || (lm[b2] - l2.y < rj[b2]);
        if (!need) break;
#pragma unroll
        for (int b2 = 0; b2 < NS; ++b2) {
          const v2f d = splat(lm[b2]) - l2;
          const v2f xp = d * splat(sp[b2]);
          const v2f y = u2 * gauss_pair(xp);
          bH[b2] = pk_fma(u2, h_pair(xp), bH[b2]); bP[b2] = bP[b2] + y; bL[b2] = pk_fma(y, d, bL[b2]);
        }
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
  if (active) {
    if (vec && NS == 4) {
      at_bytes_w<float4>(out0, fb) = make_float4(ga[0], ga[1], ga[NS - 2], ga[NS - 1]);
      at_bytes_w<float4>(out1, fb) = make_float4(gl[0], gl[1], gl[NS - 2], gl[NS - 1]);
      at_bytes_w<float4>(out2, fb) = make_float4(gd[0], gd[1], gd[NS - 2], gd[NS - 1]);
    } else if (vec) {
      at_bytes_w<v2f>(out0, fb) = (v2f){ga[0], ga[1]};
      at_bytes_w<v2f>(out1, fb) = (v2f){gl[0], gl[1]};
      at_bytes_w<v2f>(out2, fb) = (v2f){gd[0], gd[1]};
    } else {
#pragma unroll
      for (int a = 0; a < NS; ++a) if (has[a]) { out0[f + a] = ga[a]; out1[f + a] = gl[a]; out2[f + a] = gd[a]; }
    }
  }
}

}  // namespace voge

using namespace voge;

static int launch_composite(int mode, const int32_t *idx, const float *act, const float *len, const float *dsd,
                            const float *w_in, const float *g_weight, const int32_t *cnt_in, float occ, long npix, int K, float *o0, float *o1,
                            float *o2, int64_t *valid_num, voge_stream_t stream, const float *rec = nullptr,
                            const float *rays = nullptr) {
#ifndef VOGE_COMP_T
#define VOGE_COMP_T 256
#endif
  // workgroup size: the smallest multiple of 64 that is >= VOGE_COMP_T and holds one pixel
  const int threads = (K <= VOGE_COMP_T) ? VOGE_COMP_T : kCompThreads;
  hipStream_t st = (hipStream_t)stream;
#ifndef VOGE_COMP_ONE_SLOT
#ifndef VOGE_COMP_NS
#define VOGE_COMP_NS 4
#endif
  if (mode != 1) {   // NS slots per lane: 4 when the lists are whole groups of 4, else 2
#ifndef VOGE_COMP_NS_BWD
#define VOGE_COMP_NS_BWD 2     // the backward is evaluation-bound, not LDS-bound: 4 slots only cost registers
#endif
    const int NS = ((mode == 0 ? VOGE_COMP_NS : VOGE_COMP_NS_BWD) == 4 && (K & 3) == 0) ? 4 : 2;
    const int LPn = compn_lanes(K, NS);
#ifndef VOGE_COMP_WAVE          // bit 0: forward, bit 1: backward take the barrier-free one-wave-per-pixel form
#define VOGE_COMP_WAVE 3        // measured (cfg3, one-wave workgroups): backward 101 -> 87 us, forward 52 -> 50 us (the
#endif                          // forward lost 7 us in this form while it still ran as 256-thread workgroups)
    const bool wavem = ((VOGE_COMP_WAVE >> (mode == 0 ? 0 : 1)) & 1) && LPn <= 64;   // needs a pixel's lanes inside one wave
#ifndef VOGE_COMP_WAVE_T
#define VOGE_COMP_WAVE_T 64     // waves of the barrier-free form never talk to each other: one-wave workgroups schedule finest
                                // (cfg3 backward: 256 / 128 / 64 threads -> 92.2 / 89.1 / 86.7 us)
#endif
    const int tn = wavem ? VOGE_COMP_WAVE_T : kCompThreads;
    const int ppwn = compn_pixels(K, NS, tn, wavem);
    const dim3 gridn((unsigned)((npix + ppwn - 1) / ppwn)), blockn(tn);
    const size_t ldsn = compn_lds_bytes(K, NS, mode != 0, tn, wavem);
    const bool small = (double)npix * K < (double)(1l << 30);      // every byte offset fits 32 bits
#define VOGE_LAUNCH_COMPN(M, N, WV)                                                                                     \
    do {                                                                                                                \
      if (small)                                                                                                        \
        hipLaunchKernelGGL((compositen_kernel<M, N, WV, uint32_t>), gridn, blockn, ldsn, st, idx, act, len, dsd, w_in, g_weight,    \
                           cnt_in, occ, npix, K, ppwn, o0, o1, o2, valid_num, reinterpret_cast<const float4 *>(rec), rays); \
      else                                                                                                              \
        hipLaunchKernelGGL((compositen_kernel<M, N, WV, size_t>), gridn, blockn, ldsn, st, idx, act, len, dsd, w_in, g_weight,      \
                           cnt_in, occ, npix, K, ppwn, o0, o1, o2, valid_num, reinterpret_cast<const float4 *>(rec), rays); \
    } while (0)
    if (wavem) {
      if (mode == 2) { if (NS == 4) VOGE_LAUNCH_COMPN(2, 4, true); else VOGE_LAUNCH_COMPN(2, 2, true); }
      else { if (NS == 4) VOGE_LAUNCH_COMPN(0, 4, true); else VOGE_LAUNCH_COMPN(0, 2, true); }
    } else {
      if (mode == 2) { if (NS == 4) VOGE_LAUNCH_COMPN(2, 4, false); else VOGE_LAUNCH_COMPN(2, 2, false); }
      else { if (NS == 4) VOGE_LAUNCH_COMPN(0, 4, false); else VOGE_LAUNCH_COMPN(0, 2, false); }
    }
#undef VOGE_LAUNCH_COMPN
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

// Composite forward for A = a I straight from the sweep's keys: act / dsd are derived from the (mu, a) records instead of
// being read (see compositen_kernel).  cnt required.
extern "C" int voge_composite_fwd_iso(const int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                                      const float *rays, float occ, long npix, int K, float *weight, int64_t *valid_num,
                                      voge_stream_t stream) {
  if (npix < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K || K > kCompThreads) return VOGE_ERR_K_TOO_LARGE;
  if (npix == 0) return 0;
  if (!idx || !cnt || !len || !records || !rays || !weight || !valid_num) return VOGE_ERR_BAD_ARG;
  return launch_composite(0, idx, nullptr, len, nullptr, nullptr, nullptr, cnt, occ, npix, K, weight, nullptr, nullptr, valid_num,
                          stream, records, rays);
}

// Composite forward (from the records) with the shade stage in the same pass: weights, valid_num AND the image.
// gen: the records are the general path's packed (mu, A) (voge_trace_lean_fwd) instead of (mu, a).  C = 0: no shade stage.
static int composite_shade_fwd_impl(const int gen /* 0: (mu, a) records; 1: packed (mu, A); 2: compact per-axis */, int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                                    const float *rays, float occ, const float *colors, const float *bg, float thr,
                                    long npix, int K, int C, long Nattr, float *weight, int64_t *valid_num,
                                    float *rgb, float *img, float *wsum, voge_stream_t stream, float *act_out = nullptr,
                                    float *dsd_out = nullptr, void *zero_p = nullptr, size_t zero_bytes = 0, float *sil = nullptr) {
  if ((act_out == nullptr) != (dsd_out == nullptr)) return VOGE_ERR_BAD_ARG;
  if (zero_bytes > 0 && (!zero_p || C == 0 || (zero_bytes & 15) || (reinterpret_cast<uintptr_t>(zero_p) & 15))) return VOGE_ERR_BAD_ARG;
  if (npix < 0 || K <= 0 || Nattr < 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K || (C != 0 && C != 3 && C != 4)) return VOGE_ERR_K_TOO_LARGE;      // four slots per lane (any K: a last group may be short); RGB / RGBA
  if (npix == 0) return 0;
  // (img == NULL: merge_final + the weight sum only -- interpolate_attr and get_silhouette, no background; bg unused)
  if (!idx || !cnt || !len || !records || !rays || !weight || !valid_num) return VOGE_ERR_BAD_ARG;
  if (C > 0 && (!rgb || !wsum || (img && !bg) || (Nattr > 0 && !colors))) return VOGE_ERR_BAD_ARG;
  // (the shade stage's colour gather is branch-free: a slot that contributes nothing still LOADS colour row 0 and drops
  //  the value.  An empty table (Nattr == 0: colors may be NULL) therefore reads its discarded row from the rgb output
  //  instead -- C readable floats that exist whenever there is a shade stage -- so a direct C-ABI caller in that state
  //  dereferences nothing of its own)
  if (C > 0 && Nattr == 0) colors = rgb;
  if (Nattr * (C > 0 ? C : 1) >= (1l << 30)) return VOGE_ERR_BAD_ARG;      // 32-bit byte offsets of the colour gathers
  constexpr int NS = 4;
  const int tn = VOGE_COMP_WAVE_T;
  const int ppwn = compn_pixels(K, NS, tn, true);
  const dim3 gridn((unsigned)((npix + ppwn - 1) / ppwn)), blockn(tn);
  const size_t ldsn = compn_lds_bytes(K, NS, false, tn, true);
  const bool small = (double)npix * K < (double)(1l << 30);
  CompShade sh{colors, bg, thr, Nattr, rgb, img, wsum, idx};
  sh.sil = C > 0 ? sil : nullptr;
  if (zero_bytes > 0) {
    sh.zero_p = reinterpret_cast<float4 *>(zero_p); sh.zero_n4 = (long)(zero_bytes / 16);
    if ((long)gridn.x * tn < sh.zero_n4) {      // (more to zero than the launch has threads: a fill of its own, once in a blue moon)
      const hipError_t e = voge_fill_async(zero_p, 0, zero_bytes, (hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
      sh.zero_p = nullptr; sh.zero_n4 = 0;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const float4 *rec = reinterpret_cast<const float4 *>(records);
#define VOGE_LAUNCH_CS(OT, CC, GG)                                                                                           \
  hipLaunchKernelGGL((compositen_kernel<0, NS, true, OT, CC, GG>), gridn, blockn, ldsn, st, idx, nullptr, len, nullptr, nullptr, nullptr, \
                     cnt, occ, npix, K, ppwn, weight, act_out, dsd_out, valid_num, rec, rays, sh)
#define VOGE_LAUNCH_CS2(CC, GG) do { if (small) VOGE_LAUNCH_CS(uint32_t, CC, GG); else VOGE_LAUNCH_CS(size_t, CC, GG); } while (0)
  if (gen == 2) {
    if (C == 3) VOGE_LAUNCH_CS2(3, 2); else if (C == 4) VOGE_LAUNCH_CS2(4, 2); else VOGE_LAUNCH_CS2(0, 2);
  } else if (gen) {
    if (C == 3) VOGE_LAUNCH_CS2(3, 1); else if (C == 4) VOGE_LAUNCH_CS2(4, 1); else VOGE_LAUNCH_CS2(0, 1);
  } else {
    if (C == 3) VOGE_LAUNCH_CS2(3, 0); else if (C == 4) VOGE_LAUNCH_CS2(4, 0); else VOGE_LAUNCH_CS2(0, 0);
  }
#undef VOGE_LAUNCH_CS2
#undef VOGE_LAUNCH_CS
  return launch_status();
}

extern "C" int voge_composite_shade_fwd_iso(int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                                            const float *rays, float occ, const float *colors, const float *bg, float thr,
                                            long npix, int K, int C, long Nattr, float *weight, int64_t *valid_num,
                                            float *rgb, float *img, float *wsum, voge_stream_t stream) {
  if (C != 3 && C != 4) return VOGE_ERR_K_TOO_LARGE;
  return composite_shade_fwd_impl(0, idx, cnt, len, records, rays, occ, colors, bg, thr, npix, K, C, Nattr, weight, valid_num,
                                  rgb, img, wsum, stream);
}

// Round 6, the frame's forward behind the sweep: voge_composite_shade_fwd_iso, which on its way also zeroes bwd_acc -- the
// accumulator voge_frame_shade_bwd_iso / voge_frame_merge_bwd_iso add into (voge_frame_bwd_acc_bytes), so that backward has no
// fill launch in front of it.  bwd_acc == NULL: nothing to zero (a frame nobody differentiates).
extern "C" int voge_frame_shade_fwd_iso(int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                                        const float *rays, float occ, const float *colors, const float *bg, float thr,
                                        long npix, int K, int C, long Nattr, float *weight, int64_t *valid_num,
                                        float *rgb, float *img, float *wsum, float *sil, void *bwd_acc, size_t bwd_acc_bytes,
                                        voge_stream_t stream) {
  if (C != 3 && C != 4) return VOGE_ERR_K_TOO_LARGE;
  if (bwd_acc == nullptr) bwd_acc_bytes = 0;
  if (npix == 0) return bwd_acc_bytes ? (int)voge_fill_async(bwd_acc, 0, bwd_acc_bytes, (hipStream_t)stream) : 0;      // (no launch to ride on)
  return composite_shade_fwd_impl(0, idx, cnt, len, records, rays, occ, colors, bg, thr, npix, K, C, Nattr, weight, valid_num,
                                  rgb, img, wsum, stream, nullptr, nullptr, bwd_acc, bwd_acc_bytes, sil);
}

// ... and for the general forms (records = the packed (mu, A) of voge_frame_trace_fwd_gen / voge_trace_lean_fwd; act / dsd kept
// for the backward when given): bwd_acc = voge_frame_bwd_gen_acc_bytes(B * N) bytes.
extern "C" int voge_frame_shade_fwd_rec(int kind, int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                                        const float *rays, float occ, const float *colors, const float *bg, float thr,
                                        long npix, int K, int C, long Nattr, float *weight, int64_t *valid_num,
                                        float *rgb, float *img, float *wsum, float *sil, float *act, float *dsd, void *bwd_acc,
                                        size_t bwd_acc_bytes, voge_stream_t stream) {
  if (C != 0 && C != 3 && C != 4) return VOGE_ERR_K_TOO_LARGE;      // (C = 0: the weights alone -- voge_composite_fwd_rec's form)
  if (kind != 1 && kind != 2) return VOGE_ERR_BAD_ARG;
  if (bwd_acc == nullptr || C == 0) bwd_acc_bytes = 0;
  if (npix == 0) return bwd_acc_bytes ? (int)voge_fill_async(bwd_acc, 0, bwd_acc_bytes, (hipStream_t)stream) : 0;
  return composite_shade_fwd_impl(kind == 1 ? 2 : 1, idx, cnt, len, records, rays, occ, colors, bg, thr, npix, K, C, Nattr, weight, valid_num,
                                  rgb, img, wsum, stream, act, dsd, bwd_acc, bwd_acc_bytes, sil);
}

// The same two for the general path: records = the packed (mu, A) [B*N][12] of voge_trace_lean_fwd.
extern "C" int voge_composite_shade_fwd_rec(int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                                            const float *rays, float occ, const float *colors, const float *bg, float thr,
                                            long npix, int K, int C, long Nattr, float *weight, int64_t *valid_num,
                                            float *rgb, float *img, float *wsum, float *act, float *dsd, voge_stream_t stream) {
  if (C != 3 && C != 4) return VOGE_ERR_K_TOO_LARGE;
  return composite_shade_fwd_impl(1, idx, cnt, len, records, rays, occ, colors, bg, thr, npix, K, C, Nattr, weight, valid_num,
                                  rgb, img, wsum, stream, act, dsd);
}
extern "C" int voge_composite_fwd_rec(int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                                      const float *rays, float occ, long npix, int K, float *weight, int64_t *valid_num,
                                      float *act, float *dsd, voge_stream_t stream) {
  return composite_shade_fwd_impl(1, idx, cnt, len, records, rays, occ, nullptr, nullptr, -1.0f, npix, K, 0, 0, weight,
                                  valid_num, nullptr, nullptr, nullptr, stream, act, dsd);
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
