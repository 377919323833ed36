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
  int cnt[kCompThreads];     // per local pixel: #(idx >= 0)
  int hi[kCompThreads];      // per local pixel: 1 + last slot with E != 0
};

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
  if (tid < ppw) { L.cnt[tid] = 0; L.hi[tid] = 0; }
  float lm = 0.f, sm = 1.f, em = 0.f, gw = 0.f;
  int id = -1;
  if (active) {
    em = expf(-act[f]);
    lm = len[f];
    sm = sqrtf(dsd[f] + 1e-10f);
    if (BWD) gw = g_weight[f]; else id = idx[f];
  }
  __syncthreads();
  if (active) {
    if (!BWD && id >= 0) atomicAdd(&L.cnt[p], 1);
    if (em != 0.0f) atomicMax(&L.hi[p], k + 1);
  }
  L.rec[tid] = make_float4(lm, sm, em, 0.0f);
  __syncthreads();
  const int hi = active ? L.hi[p] : 0;
  const float4 *row = L.rec + (active ? p * K : 0);

  float sum = 0.0f, rterm = 0.0f;
  if (em != 0.0f) {
    for (int j = 0; j < hi; ++j) {
      const float4 r = row[j];
      // |ca| >= 4: erf saturates (|Phi - step| < 8e-9, phi < 7e-8) -> the slot either occludes
      // fully or not at all.  Lists are depth sorted, so the lanes of a wave (consecutive m) fall
      // outside slot j's window together and the wave skips the transcendental path uniformly.
      const float ca = (lm - r.x) * r.y;
      const bool in_win = (r.z != 0.0f) && (fabsf(ca) < kSat);
      if (__any(in_win)) {
        if (r.z != 0.0f) {
          float y;
          sum = fmaf(r.z, phi_cdf(ca, y), sum);
          if (BWD) rterm = fmaf(r.z * r.y, y * kRsqrtPi, rterm);
        }
      } else {
        sum += (ca > 0.0f) ? r.z : 0.0f;
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
  float ga = 0.0f, gl = 0.0f, gd = 0.0f;
  if (em != 0.0f) {
    float cPhi = 0.0f, cphi = 0.0f, cphil = 0.0f;
    for (int m = 0; m < hi; ++m) {
      const float4 r = row[m];
      const float dl = r.x - lm;
      const float ca = dl * sm;
      const bool in_win = (r.w != 0.0f) && (fabsf(ca) < kSat);
      if (__any(in_win)) {
        if (r.w != 0.0f) {
          float y;
          const float Phi = phi_cdf(ca, y);
          const float ph = r.w * (y * kRsqrtPi);
          cPhi = fmaf(r.w, Phi, cPhi);
          cphi += ph;
          cphil = fmaf(ph, dl, cphil);
        }
      } else {
        cPhi += (ca > 0.0f) ? r.w : 0.0f;
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
