// Depth-ordered volumetric compositing (the reference's "aggregation") for gfx950.
//
// Reference behaviour being reproduced: VoGE/Aggregation.py:30-107 (get_cross_activation,
// assign2weight, aggregation).  The reference materialises ~6 [npix,K,K] fp32 temporaries and
// lets autograd replay them; here the K x K occlusion integral of one pixel lives in one
// lane's registers, the per-pixel K-vectors are staged through LDS with coalesced HBM
// traffic, and the backward uses the closed form (DESIGN.md §Composite backward).
#include "voge_common.h"

namespace voge {

constexpr int kCompPix = 64;            // lanes per workgroup; pixels per workgroup (ppw) <= this, LDS permitting
constexpr float kInvNorm = 1.6487212707001282f;   // 1 / exp(-0.5), Aggregation.py:79
constexpr float kRsqrtPi = 0.5641895835477563f;

// Cooperative, coalesced copy of a [npl, K] row block from HBM into LDS rows of stride Kp
// (Kp odd -> lane-per-row access is bank-conflict free).
template <typename TI>
__device__ __forceinline__ void load_rows(const TI *__restrict__ src, TI *dst, const int npl,
                                          const int K, const int Kp) {
  const int n = npl * K;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int r = i / K, c = i - r * K;
    dst[r * Kp + c] = src[i];
  }
}
__device__ __forceinline__ void store_rows(float *__restrict__ dst, const float *src, const int npl,
                                           const int K, const int Kp) {
  const int n = npl * K;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int r = i / K, c = i - r * K;
    dst[i] = src[r * Kp + c];
  }
}

__global__ void __launch_bounds__(kCompPix)
composite_fwd_kernel(const int32_t *__restrict__ idx, const float *__restrict__ act,
                     const float *__restrict__ len, const float *__restrict__ dsd, const float occ,
                     const long npix, const int K, const int ppw, float *__restrict__ weight,
                     int64_t *__restrict__ valid_num) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Kp = K | 1;
  float *E = sm;                     // exp(-act)      [64][Kp]
  float *Ln = E + ppw * Kp;     // len
  float *S = Ln + ppw * Kp;     // sqrt(dsd+1e-10)
  float *Wt = S + ppw * Kp;     // weights (output staging)
  const long pix0 = (long)blockIdx.x * ppw;
  const int npl = (int)min((long)ppw, npix - pix0);
  load_rows(act + pix0 * K, E, npl, K, Kp);
  load_rows(len + pix0 * K, Ln, npl, K, Kp);
  load_rows(dsd + pix0 * K, S, npl, K, Kp);
  __syncthreads();
  const int lane = threadIdx.x;
  if (lane < npl) {
    float *e = E + lane * Kp, *l = Ln + lane * Kp, *s = S + lane * Kp;
    for (int k = 0; k < K; ++k) {
      e[k] = expf(-e[k]);
      s[k] = sqrtf(s[k] + 1e-10f);
    }
    int nv = 0;
    const int32_t *irow = idx + (pix0 + lane) * K;
    for (int k = 0; k < K; ++k) nv += irow[k] >= 0;
    valid_num[pix0 + lane] = nv;
    // w_m needs every (len_k, s_k, E_k), so the weights go to a fourth LDS buffer
    float *wrow = Wt + lane * Kp;
    for (int m = 0; m < K; ++m) {
      const float em = e[m];
      float w = 0.0f;
      if (em != 0.0f) {
        const float lm = l[m];
        float sum = 0.0f;
        for (int k = 0; k < K; ++k) {
          const float ek = e[k];
          if (ek != 0.0f) sum = fmaf(ek, 0.5f * (erff((lm - l[k]) * s[k]) + 1.0f), sum);
        }
        w = expf(-occ * sum) * em * kInvNorm;
      }
      wrow[m] = w;
    }
  }
  __syncthreads();
  store_rows(weight + pix0 * K, Wt, npl, K, Kp);
}

__global__ void __launch_bounds__(kCompPix)
composite_bwd_kernel(const float *__restrict__ act, const float *__restrict__ len,
                     const float *__restrict__ dsd, const float *__restrict__ g_weight,
                     const float occ, const long npix, const int K, const int ppw,
                     float *__restrict__ g_act, float *__restrict__ g_len, float *__restrict__ g_dsd) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Kp = K | 1;
  float *E = sm;                    // exp(-act)           -> g_act
  float *Ln = E + ppw * Kp;    // len
  float *S = Ln + ppw * Kp;    // sqrt(dsd+1e-10)
  float *U = S + ppw * Kp;     // g_w, then u = g_w * w -> g_len
  float *R = U + ppw * Kp;     // row term sum_k E_k phi_mk s_k -> g_dsd
  const long pix0 = (long)blockIdx.x * ppw;
  const int npl = (int)min((long)ppw, npix - pix0);
  load_rows(act + pix0 * K, E, npl, K, Kp);
  load_rows(len + pix0 * K, Ln, npl, K, Kp);
  load_rows(dsd + pix0 * K, S, npl, K, Kp);
  load_rows(g_weight + pix0 * K, U, npl, K, Kp);
  __syncthreads();
  const int lane = threadIdx.x;
  if (lane < npl) {
    float *e = E + lane * Kp, *l = Ln + lane * Kp, *s = S + lane * Kp, *u = U + lane * Kp, *r = R + lane * Kp;
    for (int k = 0; k < K; ++k) {
      e[k] = expf(-e[k]);
      s[k] = sqrtf(s[k] + 1e-10f);
    }
    // pass 1: forward weight of row m, u_m = g_m * w_m, and the row term of d/dlen_m
    for (int m = 0; m < K; ++m) {
      const float em = e[m];
      float um = 0.0f, rm = 0.0f;
      if (em != 0.0f) {
        const float lm = l[m];
        float sum = 0.0f;
        for (int k = 0; k < K; ++k) {
          const float ek = e[k];
          if (ek != 0.0f) {
            const float ca = (lm - l[k]) * s[k];
            sum = fmaf(ek, 0.5f * (erff(ca) + 1.0f), sum);
            rm = fmaf(ek * s[k], expf(-ca * ca) * kRsqrtPi, rm);
          }
        }
        um = u[m] * (expf(-occ * sum) * em * kInvNorm);
      }
      u[m] = um;
      r[m] = rm;
    }
    // pass 2: column sums over m for every j.  Column j reads e[j], s[j], r[j], l[], u[] only,
    // so its three results can overwrite e[j], s[j], r[j] in place.
    for (int j = 0; j < K; ++j) {
      const float ej = e[j], sj = s[j], lj = l[j], uj = u[j];
      float ga = 0.0f, gl = 0.0f, gd = 0.0f;
      if (ej != 0.0f) {
        float cPhi = 0.0f, cphi = 0.0f, cphil = 0.0f;
        for (int m = 0; m < K; ++m) {
          const float um = u[m];
          if (um != 0.0f) {
            const float dl = l[m] - lj;
            const float ca = dl * sj;
            const float ph = um * (expf(-ca * ca) * kRsqrtPi);
            cPhi = fmaf(um, 0.5f * (erff(ca) + 1.0f), cPhi);
            cphi += ph;
            cphil = fmaf(ph, dl, cphil);
          }
        }
        ga = fmaf(occ * ej, cPhi, -uj);
        gl = -occ * (uj * r[j] - ej * sj * cphi);
        gd = -occ * ej / (2.0f * sj) * cphil;
      }
      e[j] = ga;
      s[j] = gl;
      r[j] = gd;
    }
  }
  __syncthreads();
  store_rows(g_act + pix0 * K, E, npl, K, Kp);
  store_rows(g_len + pix0 * K, S, npl, K, Kp);
  store_rows(g_dsd + pix0 * K, R, npl, K, Kp);
}

}  // namespace voge

using namespace voge;

// pixels per workgroup: 64 when `nbuf` staging rows of K floats per pixel fit in 48 KB (3
// workgroups per CU), fewer for large K (one workgroup may use up to 144 KB).
static int pixels_per_wg(int nbuf, int K) {
  const size_t row = sizeof(float) * nbuf * (size_t)(K | 1);
  if (row * 64 <= 48 * 1024) return 64;
  size_t p = (144 * 1024) / row;
  if (p > 64) p = 64;
  if (p < 1) p = 1;
  return (int)p;
}

extern "C" int voge_composite_fwd(const int32_t *idx, const float *act, const float *len,
                                  const float *dsd, float occ, long npix, int K, float *weight,
                                  int64_t *valid_num, voge_stream_t stream) {
  if (npix < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K) return VOGE_ERR_K_TOO_LARGE;
  if (npix == 0) return 0;
  if (!idx || !act || !len || !dsd || !weight || !valid_num) return VOGE_ERR_BAD_ARG;
  const int ppw = pixels_per_wg(4, K);
  const size_t lds = sizeof(float) * 4 * ppw * (size_t)(K | 1);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(composite_fwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const long blocks = (npix + ppw - 1) / ppw;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((unsigned)blocks), dim3(kCompPix), lds, (hipStream_t)stream,
                     idx, act, len, dsd, occ, npix, K, ppw, weight, valid_num);
  return launch_status();
}

extern "C" int voge_composite_bwd(const float *act, const float *len, const float *dsd,
                                  const float *g_weight, float occ, long npix, int K, float *g_act,
                                  float *g_len, float *g_dsd, voge_stream_t stream) {
  if (npix < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K) return VOGE_ERR_K_TOO_LARGE;
  if (npix == 0) return 0;
  if (!act || !len || !dsd || !g_weight || !g_act || !g_len || !g_dsd) return VOGE_ERR_BAD_ARG;
  const int ppw = pixels_per_wg(5, K);
  const size_t lds = sizeof(float) * 5 * ppw * (size_t)(K | 1);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(composite_bwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const long blocks = (npix + ppw - 1) / ppw;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)blocks), dim3(kCompPix), lds, (hipStream_t)stream,
                     act, len, dsd, g_weight, occ, npix, K, ppw, g_act, g_len, g_dsd);
  return launch_status();
}
