// Fine ray trace backward for gfx950.
//
// Reference behaviour being reproduced: RayTraceFineVogeBackwardKernel
// (VoGE/csrc/ray_trace_voge/ray_trace_voge.cu:283-332) with Innerdot3dBackward (:41-91):
// per valid (pixel, slot) the chain rule of :324-326, then three outer-product scatters
// (45 global float atomics per slot in the reference).
//
// 1. The three scatters are algebraically merged per target so that each target receives ONE
//    well-conditioned term per slot.  With t = len, v = mu - t d, c1 = g_len / dsd:
//      g_mu  = c1 A d + g_act [ (A + A^T) v + t (A^T - A) d ]
//      g_A   = g_act [ v v^T + t (d v^T - v d^T) ] + g_dsd d d^T + c1 v d^T   (not symmetrised)
//      g_ray = g_dsd (A + A^T) d + c1 (A^T v - t A d) + g_act t ( -2 A^T v + t (A - A^T) d )
//    which expand to exactly g_ksk/g_msk/g_msm of ray_trace_voge.cu:324-326 pushed through
//    Innerdot3dBackward (checked against the embedded known answer, :381-448).
// 2. A workgroup owns a 16x16 pixel tile (256*K slots, streamed as coalesced runs of the flat
//    [npix*K] arrays).  The 12 per-Gaussian sums are accumulated in an LDS hash table keyed by
//    Gaussian index (LDS float atomics; a pixel never lists a Gaussian twice, so the lanes of a
//    wave rarely collide) and flushed with ONE global atomic per (Gaussian, component) per tile:
//    ~10x fewer HBM/L2 atomics than one per slot.  Table overflow falls back to direct atomics.
//    g_ray is pixel-owned: LDS accumulation, plain stores.
#include "voge_common.h"

namespace voge {

constexpr int kBT = 16;        // tile edge (pixels)
constexpr int kHS = 1024;      // hash slots per workgroup
constexpr int kNV = 12;        // values per Gaussian: g_mu (3) + g_A (9)
constexpr int kProbe = 24;

struct BwdLds {
  int keys[kHS];
  float vals[kHS * kNV];
  float ray[kBT * kBT * 3];
};

__global__ void __launch_bounds__(256)
trace_bwd_kernel(const float4 *__restrict__ rec, const float *__restrict__ rays,
                 const int32_t *__restrict__ idx, const float *__restrict__ g_len,
                 const float *__restrict__ g_act, const float *__restrict__ g_dsd, const int P,
                 const long nrows, const int W, const int K, float *__restrict__ g_ray,
                 float *__restrict__ acc /* [P][16]: g_mu (3), g_A (9), pad (4) */) {
  __shared__ BwdLds L;
  const int tid = threadIdx.x;
  const int tiles_x = (W + kBT - 1) / kBT;
  const int tx = blockIdx.x % tiles_x;
  const long ty = blockIdx.x / tiles_x;
  const int x0 = tx * kBT;
  const long y0 = ty * kBT;
  const int tw = min(kBT, W - x0);                 // tile width in pixels
  const int th = (int)min((long)kBT, nrows - y0);  // tile height
  for (int i = tid; i < kHS; i += 256) L.keys[i] = -1;
  for (int i = tid; i < kHS * kNV; i += 256) L.vals[i] = 0.0f;
  for (int i = tid; i < kBT * kBT * 3; i += 256) L.ray[i] = 0.0f;
  __syncthreads();

  const int row_items = tw * K;  // contiguous floats per tile row
  for (int r = 0; r < th; ++r) {
    const long row_base = ((y0 + r) * W + x0) * (long)K;
    for (int it = tid; it < row_items; it += 256) {
      const long pid = row_base + it;
      const int p = idx[pid];
      if (p < 0 || p >= P) continue;
      const float gl = g_len[pid], ga = g_act[pid], gd = g_dsd[pid];
      if (gl == 0.0f && ga == 0.0f && gd == 0.0f) continue;
      const int lx = it / K;
      const long pix = (y0 + r) * W + x0 + lx;
      const float dx = rays[3 * pix + 0], dy = rays[3 * pix + 1], dz = rays[3 * pix + 2];
      const float4 r0 = rec[3 * (size_t)p], r1 = rec[3 * (size_t)p + 1], r2 = rec[3 * (size_t)p + 2];
      const float mx = r0.x, my = r0.y, mz = r0.z;
      const float A[9] = {r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
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

      float val[kNV];
      val[0] = fmaf(c1, adx, ga * (avx + tvx + t * (tdx - adx)));
      val[1] = fmaf(c1, ady, ga * (avy + tvy + t * (tdy - ady)));
      val[2] = fmaf(c1, adz, ga * (avz + tvz + t * (tdz - adz)));
      const float d[3] = {dx, dy, dz}, v[3] = {vx, vy, vz};
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          val[3 + 3 * i + j] = fmaf(ga, fmaf(v[i], v[j], t * (d[i] * v[j] - v[i] * d[j])),
                                    fmaf(gd, d[i] * d[j], c1 * (v[i] * d[j])));

      // LDS hash: find / claim the slot of Gaussian p
      unsigned h = ((unsigned)p * 2654435761u) >> 22;  // 10 bits
      int slot = -1;
#pragma unroll 1
      for (int pr = 0; pr < kProbe; ++pr) {
        const int old = atomicCAS(&L.keys[h], -1, p);
        if (old == -1 || old == p) { slot = (int)h; break; }
        h = (h + 1) & (kHS - 1);
      }
      if (slot >= 0) {
        float *dst = L.vals + slot * kNV;
#pragma unroll
        for (int i = 0; i < kNV; ++i) atomicAdd(dst + i, val[i]);
      } else {
#pragma unroll
        for (int i = 0; i < kNV; ++i) unsafeAtomicAdd(acc + 16 * (size_t)p + i, val[i]);
      }

      if (g_ray != nullptr) {
        const float rx = fmaf(gd, adx + tdx, fmaf(c1, fmaf(-t, adx, tvx), gat * fmaf(t, adx - tdx, -2.0f * tvx)));
        const float ryv = fmaf(gd, ady + tdy, fmaf(c1, fmaf(-t, ady, tvy), gat * fmaf(t, ady - tdy, -2.0f * tvy)));
        const float rz = fmaf(gd, adz + tdz, fmaf(c1, fmaf(-t, adz, tvz), gat * fmaf(t, adz - tdz, -2.0f * tvz)));
        float *ra = L.ray + (r * kBT + lx) * 3;
        atomicAdd(ra + 0, rx);
        atomicAdd(ra + 1, ryv);
        atomicAdd(ra + 2, rz);
      }
    }
  }
  __syncthreads();
  // flush: 16 adjacent lanes per table entry add 12 adjacent floats of ONE 64-byte line of
  // acc[p][16] -- lane-coalesced atomics run ~15x faster than 64 scattered ones
  // (tools/atomic_bench.hip: 330 vs 21 Gatomic/s).
  {
    const int c = tid & 15;
    for (int s = tid >> 4; s < kHS; s += 16) {
      const int p = L.keys[s];
      if (p >= 0 && c < kNV) unsafeAtomicAdd(acc + 16 * (size_t)p + c, L.vals[s * kNV + c]);
    }
  }
  if (g_ray != nullptr) {
    for (int i = tid; i < th * tw * 3; i += 256) {
      const int r = i / (tw * 3), c = i - r * (tw * 3);
      g_ray[((y0 + r) * W + x0) * 3 + c] = L.ray[r * kBT * 3 + c];
    }
  }
}

// mus [P,3] + isigmas [P,9] -> 3 x float4 per Gaussian, so the sweep gathers with dwordx4 loads
__global__ void __launch_bounds__(256)
bwd_pack_kernel(const float *__restrict__ mus, const float *__restrict__ isg, const int P,
                float4 *__restrict__ rec) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P) return;
  const float *m = mus + 3 * (size_t)g, *A = isg + 9 * (size_t)g;
  rec[3 * (size_t)g + 0] = make_float4(m[0], m[1], m[2], A[0]);
  rec[3 * (size_t)g + 1] = make_float4(A[1], A[2], A[3], A[4]);
  rec[3 * (size_t)g + 2] = make_float4(A[5], A[6], A[7], A[8]);
}

__global__ void __launch_bounds__(256)
bwd_unpack_kernel(const float *__restrict__ acc, const int P, float *__restrict__ g_mus,
                  float *__restrict__ g_isg) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = t >> 4, c = t & 15;
  if (g >= P || c >= kNV) return;
  const float v = acc[t];
  if (c < 3) g_mus[3 * (size_t)g + c] = v; else g_isg[9 * (size_t)g + (c - 3)] = v;
}

}  // namespace voge

using namespace voge;

extern "C" size_t voge_trace_bwd_workspace_bytes(int P) {
  return P <= 0 ? 0 : (size_t)P * (48 + 64);
}

extern "C" int voge_trace_bwd(const float *mus, const float *isigmas, const float *rays,
                              const int32_t *idx, const float *g_len, const float *g_act,
                              const float *g_dsd, int P, long nrows, int W, int K, void *workspace,
                              size_t workspace_bytes, float *g_ray, float *g_mus, float *g_isg,
                              voge_stream_t stream) {
  if (P < 0 || nrows < 0 || W < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0) {
    if (g_ray && nrows * W > 0) return (int)hipMemsetAsync(g_ray, 0, sizeof(float) * 3 * (size_t)(nrows * W), st);
    return 0;
  }
  if (!g_mus || !g_isg || !mus || !isigmas || !workspace) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < voge_trace_bwd_workspace_bytes(P)) return VOGE_ERR_WORKSPACE;
  float *acc = reinterpret_cast<float *>(workspace);
  float4 *rec = reinterpret_cast<float4 *>(reinterpret_cast<char *>(workspace) + (size_t)P * 64);
  hipError_t e = hipMemsetAsync(acc, 0, (size_t)P * 64, st);
  if (e != hipSuccess) return (int)e;
  if (nrows * W > 0) {
    if (!rays || !idx || !g_len || !g_act || !g_dsd) return VOGE_ERR_BAD_ARG;
    hipLaunchKernelGGL(bwd_pack_kernel, dim3((P + 255) / 256), dim3(256), 0, st, mus, isigmas, P, rec);
    const long tiles = (long)((W + kBT - 1) / kBT) * ((nrows + kBT - 1) / kBT);
    hipLaunchKernelGGL(trace_bwd_kernel, dim3((unsigned)tiles), dim3(256), 0, st, rec, rays, idx, g_len, g_act,
                       g_dsd, P, nrows, W, K, g_ray, acc);
  }
  hipLaunchKernelGGL(bwd_unpack_kernel, dim3((unsigned)(((size_t)P * 16 + 255) / 256)), dim3(256), 0, st, acc, P,
                     g_mus, g_isg);
  return launch_status();
}
