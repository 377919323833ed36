// Fine ray trace backward for gfx950.
//
// Reference behaviour being reproduced: RayTraceFineVogeBackwardKernel
// (VoGE/csrc/ray_trace_voge/ray_trace_voge.cu:283-332) with Innerdot3dBackward (:41-91):
// per valid (pixel, slot) the chain rule of :324-326, then three outer-product scatters.
//
// The three scatters are algebraically merged per target so that each target receives ONE
// well-conditioned term per slot (the reference adds large terms of opposite sign through
// separate float atomics).  With t = len, v = mu - t d, c1 = g_len / dsd:
//   g_mu  = c1 A d + g_act [ (A + A^T) v + t (A^T - A) d ]
//   g_A   = g_act [ v v^T + t (d v^T - v d^T) ] + g_dsd d d^T + c1 v d^T      (not symmetrised)
//   g_ray = g_dsd (A + A^T) d + c1 (A^T v - t A d) + g_act t ( -2 A^T v + t (A - A^T) d )
// which expand to exactly g_ksk/g_msk/g_msm of ray_trace_voge.cu:324-326 pushed through
// Innerdot3dBackward (checked against the embedded known answer, :381-448).
#include "voge_common.h"

namespace voge {

constexpr int kBwdPix = 64;  // pixels per workgroup

__global__ void __launch_bounds__(256)
trace_bwd_kernel(const float *__restrict__ mus, const float *__restrict__ isg,
                 const float *__restrict__ rays, const int32_t *__restrict__ idx,
                 const float *__restrict__ g_len, const float *__restrict__ g_act,
                 const float *__restrict__ g_dsd, const int P, const long npix, const int K,
                 float *__restrict__ g_ray, float *__restrict__ g_mus, float *__restrict__ g_isg) {
  __shared__ float ray_acc[kBwdPix * 3];
  const long pix0 = (long)blockIdx.x * kBwdPix;
  const int npl = (int)min((long)kBwdPix, npix - pix0);
  for (int i = threadIdx.x; i < kBwdPix * 3; i += blockDim.x) ray_acc[i] = 0.0f;
  __syncthreads();
  const int items = npl * K;
  for (int it = threadIdx.x; it < items; it += blockDim.x) {
    const long pid = pix0 * K + it;
    const int p = idx[pid];
    if (p < 0 || p >= P) continue;
    const float gl = g_len[pid], ga = g_act[pid], gd = g_dsd[pid];
    if (gl == 0.0f && ga == 0.0f && gd == 0.0f) continue;
    const int lp = it / K;
    const float *ry = rays + (pix0 + lp) * 3;
    const float dx = ry[0], dy = ry[1], dz = ry[2];
    const float mx = mus[3 * (size_t)p], my = mus[3 * (size_t)p + 1], mz = mus[3 * (size_t)p + 2];
    float A[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = isg[9 * (size_t)p + i];
    // A d, A^T d
    const float adx = fmaf(A[2], dz, fmaf(A[1], dy, A[0] * dx));
    const float ady = fmaf(A[5], dz, fmaf(A[4], dy, A[3] * dx));
    const float adz = fmaf(A[8], dz, fmaf(A[7], dy, A[6] * dx));
    const float tdx = fmaf(A[6], dz, fmaf(A[3], dy, A[0] * dx));
    const float tdy = fmaf(A[7], dz, fmaf(A[4], dy, A[1] * dx));
    const float tdz = fmaf(A[8], dz, fmaf(A[5], dy, A[2] * dx));
    const float ksk = fmaf(dz, adz, fmaf(dy, ady, dx * adx));
    const float msk = fmaf(mz, adz, fmaf(my, ady, mx * adx));
    const float t = msk / ksk;
    const float vx = fmaf(-t, dx, mx), vy = fmaf(-t, dy, my), vz = fmaf(-t, dz, mz);
    const float avx = fmaf(A[2], vz, fmaf(A[1], vy, A[0] * vx));
    const float avy = fmaf(A[5], vz, fmaf(A[4], vy, A[3] * vx));
    const float avz = fmaf(A[8], vz, fmaf(A[7], vy, A[6] * vx));
    const float tvx = fmaf(A[6], vz, fmaf(A[3], vy, A[0] * vx));
    const float tvy = fmaf(A[7], vz, fmaf(A[4], vy, A[1] * vx));
    const float tvz = fmaf(A[8], vz, fmaf(A[5], vy, A[2] * vx));
    const float c1 = gl / ksk;
    const float gat = ga * t;

    // g_mu
    float *gm = g_mus + 3 * (size_t)p;
    unsafeAtomicAdd(gm + 0, fmaf(c1, adx, ga * (avx + tvx + t * (tdx - adx))));
    unsafeAtomicAdd(gm + 1, fmaf(c1, ady, ga * (avy + tvy + t * (tdy - ady))));
    unsafeAtomicAdd(gm + 2, fmaf(c1, adz, ga * (avz + tvz + t * (tdz - adz))));

    // g_A[i][j] = ga*(v_i v_j + t (d_i v_j - v_i d_j)) + gd d_i d_j + c1 v_i d_j
    float *gA = g_isg + 9 * (size_t)p;
    const float d[3] = {dx, dy, dz}, v[3] = {vx, vy, vz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float val = fmaf(ga, fmaf(v[i], v[j], t * (d[i] * v[j] - v[i] * d[j])),
                               fmaf(gd, d[i] * d[j], c1 * (v[i] * d[j])));
        unsafeAtomicAdd(gA + 3 * i + j, val);
      }
    }

    // g_ray (pixel-owned: LDS accumulation, one global store per pixel)
    const float rx = fmaf(gd, adx + tdx, fmaf(c1, fmaf(-t, adx, tvx), gat * fmaf(t, adx - tdx, -2.0f * tvx)));
    const float ryv = fmaf(gd, ady + tdy, fmaf(c1, fmaf(-t, ady, tvy), gat * fmaf(t, ady - tdy, -2.0f * tvy)));
    const float rz = fmaf(gd, adz + tdz, fmaf(c1, fmaf(-t, adz, tvz), gat * fmaf(t, adz - tdz, -2.0f * tvz)));
    atomicAdd(&ray_acc[lp * 3 + 0], rx);
    atomicAdd(&ray_acc[lp * 3 + 1], ryv);
    atomicAdd(&ray_acc[lp * 3 + 2], rz);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < npl * 3; i += blockDim.x) g_ray[pix0 * 3 + i] = ray_acc[i];
}

}  // namespace voge

using namespace voge;

extern "C" int voge_trace_bwd(const float *mus, const float *isigmas, const float *rays,
                              const int32_t *idx, const float *g_len, const float *g_act,
                              const float *g_dsd, int P, long npix, int K, float *g_ray,
                              float *g_mus, float *g_isg, voge_stream_t stream) {
  if (P < 0 || npix < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (P > 0) {
    if (!g_mus || !g_isg) return VOGE_ERR_BAD_ARG;
    hipError_t e = hipMemsetAsync(g_mus, 0, sizeof(float) * 3 * (size_t)P, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(g_isg, 0, sizeof(float) * 9 * (size_t)P, st);
    if (e != hipSuccess) return (int)e;
  }
  if (npix == 0) return 0;
  if (!rays || !idx || !g_len || !g_act || !g_dsd || !g_ray) return VOGE_ERR_BAD_ARG;
  if (P > 0 && (!mus || !isigmas)) return VOGE_ERR_BAD_ARG;
  const long blocks = (npix + kBwdPix - 1) / kBwdPix;
  hipLaunchKernelGGL(trace_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mus, isigmas, rays, idx,
                     g_len, g_act, g_dsd, P, npix, K, g_ray, g_mus, g_isg);
  return launch_status();
}
