// Depth-ordered volumetric compositing (the reference's "aggregation") for gfx950.
//
// Reference behaviour being reproduced: VoGE/Aggregation.py:30-107 (get_cross_activation,
// assign2weight, aggregation).  The reference materialises ~6 [npix,K,K] fp32 temporaries and
// lets autograd replay them.  Here ONE LANE OWNS ONE (pixel, slot) PAIR: a 256-thread
// workgroup covers floor(256/K) whole pixels, every HBM access is a perfectly coalesced
// stream over the flat [npix*K] arrays, the pixel's K (len, s, E) triples sit in 4 KB of LDS
// and are broadcast-read in the K-long inner loop.  Nothing of size K x K ever exists.
//
//   fwd : row m        S_m = sum_k E_k Phi((len_m - len_k) s_k),  w_m = exp(-occ S_m) E_m e^{1/2}
//   bwd : row m        u_m = g_m w_m,  r_m = sum_k E_k s_k phi_mk
//         column j     dL/dact_j = -u_j + occ E_j sum_m u_m Phi_mj
//                      dL/dlen_j = -occ (u_j r_j - E_j s_j sum_m u_m phi_mj)
//                      dL/ddsd_j = -occ E_j / (2 s_j) sum_m u_m phi_mj (len_m - len_j)
//   with E = exp(-act), s = sqrt(dsd + 1e-10), Phi = (erf + 1)/2, phi = exp(-x^2)/sqrt(pi).
#include "voge_common.h"

namespace voge {

constexpr int kCompThreads = 256;
constexpr float kInvNorm = 1.6487212707001282f;  // 1 / exp(-0.5), Aggregation.py:79
constexpr float kRsqrtPi = 0.5641895835477563f;
constexpr float kSat = 4.0f;  // erf(4) = 1 - 1.5e-8

// Phi(x) = (erf(x)+1)/2 and y = exp(-x^2) in one go.  erf by Abramowitz-Stegun 7.1.26
// (|err| <= 1.5e-7 absolute), branch-free; the same exponential feeds phi in the backward.
__device__ __forceinline__ float phi_cdf(const float x, float &y) {
  const float ax = fabsf(x);
  y = __expf(-ax * ax);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float h = 0.5f * (p * t) * y;  // (1 - erf(|x|)) / 2
  return x >= 0.0f ? 1.0f - h : h;
}

struct CompLds {
  float4 rec[kCompThreads];  // (len, s, E, u) of slot tid
  float pre[kCompThreads];   // exclusive prefix sum of E within the pixel
  float suf[kCompThreads];   // inclusive suffix sum of u within the pixel (backward)
  float scan[2][kCompThreads];
  int cnt[kCompThreads];     // per local pixel: #(idx >= 0)
  int hi[kCompThreads];      // per local pixel: 1 + last slot with E != 0
  int rmax[kCompThreads];    // per local pixel: bits of max_k 4/s_k (window radius in len)
  int unsorted[kCompThreads];
};

// The pixel's list is depth sorted (the trace emits it that way).  Slot j influences row m only
// through Phi((len_m - len_j) s_j), which saturates to 0 / 1 beyond |len_m - len_j| >= 4 / s_j
// (|Phi - step| < 8e-9), so every row scans a WINDOW around itself and takes the far front
// slots (Phi = 1) from a prefix sum of E; far back slots contribute nothing.  An unsorted list
// (possible through the public API) simply gets an infinite window, i.e. the full K x K scan.
template <bool BWD>
__global__ void __launch_bounds__(kCompThreads)
composite_kernel(const int32_t *__restrict__ idx, const float *__restrict__ act,
                 const float *__restrict__ len, const float *__restrict__ dsd,
                 const float *__restrict__ g_weight, const float occ, const long npix, const int K,
                 const int ppw, float *__restrict__ out0 /* weight | g_act */,
                 float *__restrict__ out1 /* g_len */, float *__restrict__ out2 /* g_dsd */,
                 int64_t *__restrict__ valid_num) {
  __shared__ CompLds L;
  const int tid = threadIdx.x;
  const int p = tid / K, k = tid - p * K;
  const long pix = (long)blockIdx.x * ppw + p;
  const bool active = (p < ppw) && (pix < npix);
  const long f = pix * K + k;
  if (tid < ppw) { L.cnt[tid] = 0; L.hi[tid] = 0; L.rmax[tid] = 0; L.unsorted[tid] = 0; }
  float lm = 0.f, sm = 1.f, em = 0.f, gw = 0.f;
  int id = -1;
  if (active) {
    em = expf(-act[f]);
    lm = len[f];
    sm = sqrtf(dsd[f] + 1e-10f);
    if (BWD) gw = g_weight[f]; else id = idx[f];
  }
  L.rec[tid] = make_float4(lm, sm, em, 0.0f);
  __syncthreads();
  if (active) {
    if (!BWD && id >= 0) atomicAdd(&L.cnt[p], 1);
    if (em != 0.0f) {
      atomicMax(&L.hi[p], k + 1);
      atomicMax(&L.rmax[p], __float_as_int(kSat / sm));   // positive floats order like ints
    }
    if (k > 0 && !(L.rec[tid - 1].x <= lm)) L.unsorted[p] = 1;
  }
  __syncthreads();
  const int hi = active ? L.hi[p] : 0;
  const float4 *row = L.rec + (active ? p * K : 0);
  const float *pre = L.pre + (active ? p * K : 0);
  const bool sorted = active && (L.unsorted[p] == 0);
  const float rwin = sorted ? __int_as_float(L.rmax[p]) : INFINITY;
  {   // exclusive prefix sum of E within each pixel: log2(K) ping-pong steps over the workgroup
    float x = em;
    int par = 0;
    for (int o = 1; o < K; o <<= 1) {
      L.scan[par][tid] = x;
      __syncthreads();
      if (k >= o && p < ppw) x += L.scan[par][tid - o];
      par ^= 1;
    }
    L.pre[tid] = x - em;
  }
  __syncthreads();

  // ---- row m: S_m = sum_{far front} E_j + sum_{window} E_j Phi_mj ; r_m = sum_{window} E_j s_j phi_mj
  float sum = 0.0f, rterm = 0.0f;
  if (em != 0.0f) {
    int j = min(k, hi - 1);
    for (; j >= 0; --j) {             // self and towards the camera
      const float4 r = row[j];
      const float d = lm - r.x;
      if (d >= rwin) break;
      if (r.z != 0.0f) {
        float y;
        sum = fmaf(r.z, phi_cdf(d * r.y, y), sum);
        if (BWD) rterm = fmaf(r.z * r.y, y * kRsqrtPi, rterm);
      }
    }
    if (j >= 0) sum += pre[j + 1];    // slots 0..j are fully in front: Phi = 1
    for (j = k + 1; j < hi; ++j) {    // away from the camera
      const float4 r = row[j];
      const float d = lm - r.x;
      if (-d >= rwin) break;
      if (r.z != 0.0f) {
        float y;
        sum = fmaf(r.z, phi_cdf(d * r.y, y), sum);
        if (BWD) rterm = fmaf(r.z * r.y, y * kRsqrtPi, rterm);
      }
    }
  }
  const float w = (em != 0.0f) ? expf(-occ * sum) * em * kInvNorm : 0.0f;
  if (!BWD) {
    if (active) {
      out0[f] = w;
      if (k == 0) valid_num[pix] = L.cnt[p];
    }
    return;
  }
  const float um = gw * w;
  L.rec[tid].w = um;
  __syncthreads();
  {   // inclusive suffix sum of u within each pixel
    float x = um;
    int par = 0;
    for (int o = 1; o < K; o <<= 1) {
      L.scan[par][tid] = x;
      __syncthreads();
      if (k + o < K && p < ppw) x += L.scan[par][tid + o];
      par ^= 1;
    }
    L.suf[tid] = x;
  }
  __syncthreads();
  const float *suf = L.suf + (active ? p * K : 0);
  // ---- column j (= this lane): rows far behind see Phi_mj = 1 (suffix sum of u), rows far in
  // front see 0; phi terms live in the window |len_m - len_j| < 4 / s_j only.
  float ga = 0.0f, gl = 0.0f, gd = 0.0f;
  if (em != 0.0f) {
    const float rj = sorted ? kSat / sm : INFINITY;
    float cPhi = 0.0f, cphi = 0.0f, cphil = 0.0f;
    int m = min(k, hi - 1);
    for (; m < hi; ++m) {             // self and rows behind
      const float4 r = row[m];
      const float dl = r.x - lm;
      if (dl >= rj) break;
      if (r.w != 0.0f) {
        float y;
        const float Phi = phi_cdf(dl * sm, y);
        const float ph = r.w * (y * kRsqrtPi);
        cPhi = fmaf(r.w, Phi, cPhi);
        cphi += ph;
        cphil = fmaf(ph, dl, cphil);
      }
    }
    if (m < hi) cPhi += suf[m];       // rows m..hi-1 are fully behind: Phi = 1
    for (m = k - 1; m >= 0; --m) {    // rows in front
      const float4 r = row[m];
      const float dl = r.x - lm;
      if (-dl >= rj) break;
      if (r.w != 0.0f) {
        float y;
        const float Phi = phi_cdf(dl * sm, y);
        const float ph = r.w * (y * kRsqrtPi);
        cPhi = fmaf(r.w, Phi, cPhi);
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

static int launch_composite(bool bwd, const int32_t *idx, const float *act, const float *len, const float *dsd,
                            const float *g_weight, float occ, long npix, int K, float *o0, float *o1, float *o2,
                            int64_t *valid_num, voge_stream_t stream) {
  const int ppw = kCompThreads / K;
  const long blocks = (npix + ppw - 1) / ppw;
  if (bwd)
    hipLaunchKernelGGL(composite_kernel<true>, dim3((unsigned)blocks), dim3(kCompThreads), 0, (hipStream_t)stream, idx,
                       act, len, dsd, g_weight, occ, npix, K, ppw, o0, o1, o2, valid_num);
  else
    hipLaunchKernelGGL(composite_kernel<false>, dim3((unsigned)blocks), dim3(kCompThreads), 0, (hipStream_t)stream, idx,
                       act, len, dsd, g_weight, occ, npix, K, ppw, o0, o1, o2, valid_num);
  return launch_status();
}

extern "C" int voge_composite_fwd(const int32_t *idx, const float *act, const float *len,
                                  const float *dsd, float occ, long npix, int K, float *weight,
                                  int64_t *valid_num, voge_stream_t stream) {
  if (npix < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K || K > kCompThreads) return VOGE_ERR_K_TOO_LARGE;
  if (npix == 0) return 0;
  if (!idx || !act || !len || !dsd || !weight || !valid_num) return VOGE_ERR_BAD_ARG;
  return launch_composite(false, idx, act, len, dsd, nullptr, occ, npix, K, weight, nullptr, nullptr, valid_num, stream);
}

extern "C" int voge_composite_bwd(const float *act, const float *len, const float *dsd,
                                  const float *g_weight, float occ, long npix, int K, float *g_act,
                                  float *g_len, float *g_dsd, voge_stream_t stream) {
  if (npix < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K || K > kCompThreads) return VOGE_ERR_K_TOO_LARGE;
  if (npix == 0) return 0;
  if (!act || !len || !dsd || !g_weight || !g_act || !g_len || !g_dsd) return VOGE_ERR_BAD_ARG;
  return launch_composite(true, nullptr, act, len, dsd, g_weight, occ, npix, K, g_act, g_len, g_dsd, nullptr, stream);
}
