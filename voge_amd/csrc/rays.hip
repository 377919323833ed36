// Pixel-ray generation for gfx950.
//
// Reference behaviour being reproduced: VoGE/Renderer.py:124-130 --
// NDCMultinomialRaysampler(image_width, image_height, unit_directions=True, ...)(cameras)
// for a screen-space (in_ndc=False) PyTorch3D PerspectiveCameras: unit world-space directions of
// the pixel centres and the camera centre as common origin.  Conventions (row vectors):
//   X_view = X_world @ R + T,  view axes +X left / +Y up / +Z forward,
//   d_view(i, j) = [(px - j - 0.5)/fx, (py - i - 0.5)/fy, 1],
//   d_world = normalise(d_view @ R^-1),  C = -T @ R^-1.
// The reference runs ~25 small torch kernels for this (inverse via LU, two GEMMs, norms ...);
// here it is one kernel forward and two backward (gradients reach R, T, focal, principal point,
// which the pose-estimation demos optimise).
#include "voge_common.h"

namespace voge {

__global__ void __launch_bounds__(256)
rays_fwd_kernel(const float *__restrict__ R, const float *__restrict__ T, const float *__restrict__ focal,
                const float *__restrict__ pp, const int row0, const int h, const int W, const int stripe_h, const int pitch,
                float *__restrict__ rays, float *__restrict__ origin, const int ray_blocks,
                ConeRec *__restrict__ cones /* NULL | [B][nsty][nstx] */) {
  // row i of the output is image row irow(i): a contiguous band (stripe_h >= h), or every `pitch`-th stripe of
  // `stripe_h` rows starting at row0 (a rank's share of a frame dealt in stripes: voge_amd/distributed.py)
  const int b = blockIdx.y;
  const CamView cam{R, T, focal, pp, row0, stripe_h, pitch, h, W, 0, nullptr, nullptr};
  const CamK ck = cam_load(cam, b);
  if ((int)blockIdx.x >= ray_blocks) {
    // The workgroups behind the ray blocks: one per 32x32-pixel super-tile, its bounding cone for the trace's
    // binning (voge_trace_topk_fwd*'s `cones`).  The directions are recomputed with the arithmetic below (no loads),
    // so the cones cost the frame nothing: the trace otherwise reads every ray once more to derive them.
    __shared__ ConeHierLds Lc;
    const int nstx = (W + kST - 1) / kST;
    const int st = (int)blockIdx.x - ray_blocks;
    const int x0 = (st % nstx) * kST, y0 = (st / nstx) * kST;
    int ly, lx0;
    cone_thread_rays(threadIdx.x, lx0, ly);
    float cx[4], cy[4], cz[4];
    unsigned has = 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = x0 + lx0 + u, i = y0 + ly;
      has |= (j < W && i < h) ? (1u << u) : 0u;
      cam_ray(ck, cam_irow(cam, min(i, max(h - 1, 0))), j, cx[u], cy[u], cz[u]);
    }
    // (the super-tile's cone and, since round 5, its quads' and tiles': voge_common.h, block_cones_hier256)
    const size_t nst = gridDim.x - ray_blocks;
    block_cones_hier256(cx, cy, cz, has, cones + cone_super_at(b, nst, st), cones + cone_quad_at(b, nst, st, 0),
                        cones + cone_tile_at(b, nst, st, 0), Lc);
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) cam_origin(ck, T + 3 * b, origin[3 * b + 0], origin[3 * b + 1], origin[3 * b + 2]);
  const int n = h * W;
  const float iW = 1.0f / (float)W;
  float *ob = rays + (size_t)b * n * 3;
  // four consecutive pixels per thread: 48 contiguous bytes leave as three 16-byte stores
  const int n4 = (n + 3) >> 2;
  const bool aligned = (reinterpret_cast<uintptr_t>(ob) & 15) == 0;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += ray_blocks * blockDim.x) {
    float o[12];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = min(4 * q + u, n - 1);
      int i = __float2int_rz(((float)p + 0.5f) * iW);      // p / W, corrected below (exact for any size)
      int j = p - i * W;
      if (j < 0) { --i; j += W; } else if (j >= W) { ++i; j -= W; }
      cam_ray(ck, cam_irow(cam, i), j, o[3 * u], o[3 * u + 1], o[3 * u + 2]);
    }
    if (4 * q + 3 < n && aligned) {
      float4 *dst = reinterpret_cast<float4 *>(ob + (size_t)q * 12);
      dst[0] = make_float4(o[0], o[1], o[2], o[3]);
      dst[1] = make_float4(o[4], o[5], o[6], o[7]);
      dst[2] = make_float4(o[8], o[9], o[10], o[11]);
    } else {
      for (int u = 0; u < 4 && 4 * q + u < n; ++u) { ob[(size_t)(4 * q + u) * 3] = o[3 * u]; ob[(size_t)(4 * q + u) * 3 + 1] = o[3 * u + 1]; ob[(size_t)(4 * q + u) * 3 + 2] = o[3 * u + 2]; }
    }
  }
}

// The same cones from a ray tensor the caller built any other way (the trace's fallback when it is given none).
__global__ void __launch_bounds__(256)
cones_kernel(const float *__restrict__ rays, const int H, const int W, ConeRec *__restrict__ cones) {
  __shared__ ConeHierLds Lc;
  const int b = blockIdx.y, nstx = (W + kST - 1) / kST;
  const int x0 = ((int)blockIdx.x % nstx) * kST, y0 = ((int)blockIdx.x / nstx) * kST;
  int ly, lx0;
  cone_thread_rays(threadIdx.x, lx0, ly);
  float cx[4], cy[4], cz[4];
  unsigned has = 0u;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = x0 + lx0 + u, i = y0 + ly;
    has |= (j < W && i < H) ? (1u << u) : 0u;
    const float *r = rays + (((size_t)b * H + min(i, H - 1)) * W + min(j, W - 1)) * 3;
    cx[u] = r[0]; cy[u] = r[1]; cz[u] = r[2];
  }
  const size_t nst = gridDim.x;
  block_cones_hier256(cx, cy, cz, has, cones + cone_super_at(b, nst, blockIdx.x), cones + cone_quad_at(b, nst, blockIdx.x, 0),
                      cones + cone_tile_at(b, nst, blockIdx.x, 0), Lc);
}

// Backward, stage 1: per-batch sums over pixels.  part[b][0..8] = d_view^T g_dw (gradient of
// R^-1 from the directions), [9] = sum g_vx * d vx/d fx, [10] = same for fy, [11],[12] = g_px, g_py.
__global__ void __launch_bounds__(256)
rays_bwd_reduce_kernel(const float *__restrict__ R, const float *__restrict__ focal,
                       const float *__restrict__ pp, const float *__restrict__ g_rays, const int row0,
                       const int h, const int W, const int stripe_h, const int pitch, float *__restrict__ part /* [B][16], zeroed */) {
  __shared__ float red[4][13];
  const int b = blockIdx.y;
  const Mat3 Ri = inv3(R + 9 * b);
  const float fx = focal[2 * b], fy = focal[2 * b + 1], px = pp[2 * b], py = pp[2 * b + 1];
  float acc[13];
#pragma unroll
  for (int q = 0; q < 13; ++q) acc[q] = 0.0f;
  const int n = h * W;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const int i = p / W, j = p - i * W;
    const int sk = i / stripe_h;
    const float ax = px - ((float)j + 0.5f), ay = py - ((float)(row0 + sk * pitch + (i - sk * stripe_h)) + 0.5f);
    const float vx = ax / fx, vy = ay / fy;
    const float wx = vx * Ri.m[0] + vy * Ri.m[3] + Ri.m[6];
    const float wy = vx * Ri.m[1] + vy * Ri.m[4] + Ri.m[7];
    const float wz = vx * Ri.m[2] + vy * Ri.m[5] + Ri.m[8];
    const float inv = 1.0f / sqrtf(wx * wx + wy * wy + wz * wz);
    const float ux = wx * inv, uy = wy * inv, uz = wz * inv;
    const float *g = g_rays + ((size_t)b * n + p) * 3;
    const float gu = g[0] * ux + g[1] * uy + g[2] * uz;
    const float gx = (g[0] - ux * gu) * inv, gy = (g[1] - uy * gu) * inv, gz = (g[2] - uz * gu) * inv;  // g_dw
    acc[0] += vx * gx; acc[1] += vx * gy; acc[2] += vx * gz;
    acc[3] += vy * gx; acc[4] += vy * gy; acc[5] += vy * gz;
    acc[6] += gx;      acc[7] += gy;      acc[8] += gz;
    const float gvx = gx * Ri.m[0] + gy * Ri.m[1] + gz * Ri.m[2];   // g_dview = g_dw @ Rinv^T
    const float gvy = gx * Ri.m[3] + gy * Ri.m[4] + gz * Ri.m[5];
    acc[9] += -gvx * vx / fx;
    acc[10] += -gvy * vy / fy;
    acc[11] += gvx / fx;
    acc[12] += gvy / fy;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 13; ++q) {
    const float s = wave_sum(acc[q]);
    if (lane == 0) red[wave][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < 13) {
    const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    unsafeAtomicAdd(part + 16 * b + threadIdx.x, s);
  }
}

// Backward, stage 2: one thread per batch element turns the sums (+ the origin gradient) into
// g_R, g_T, g_focal, g_pp.   d(R^-1) = -R^-1 dR R^-1  ->  g_R = -R^-T g_Rinv R^-T.
__global__ void rays_bwd_finish_kernel(const float *__restrict__ R, const float *__restrict__ T,
                                       const float *__restrict__ part, const float *__restrict__ g_origin,
                                       const int B, float *__restrict__ g_R, float *__restrict__ g_T,
                                       float *__restrict__ g_focal, float *__restrict__ g_pp) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const Mat3 Ri = inv3(R + 9 * b);
  float G[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) G[q] = part[16 * b + q];
  const float *t = T + 3 * b;
  float go[3] = {0.f, 0.f, 0.f};
  if (g_origin != nullptr) { go[0] = g_origin[3 * b]; go[1] = g_origin[3 * b + 1]; go[2] = g_origin[3 * b + 2]; }
  // origin = -T @ Rinv :  g_Rinv += -T^T g_o ,  g_T = -g_o @ Rinv^T
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) G[3 * r + c] -= t[r] * go[c];
  if (g_T != nullptr)
    for (int r = 0; r < 3; ++r) g_T[3 * b + r] = -(go[0] * Ri.m[3 * r] + go[1] * Ri.m[3 * r + 1] + go[2] * Ri.m[3 * r + 2]);
  if (g_R != nullptr) {
    float M[9];  // Rinv^T @ G
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M[3 * r + c] = Ri.m[r] * G[c] + Ri.m[3 + r] * G[3 + c] + Ri.m[6 + r] * G[6 + c];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)  // (M @ Rinv^T)[r][c] = sum_k M[r][k] * Rinv[c][k]
        g_R[9 * b + 3 * r + c] = -(M[3 * r] * Ri.m[3 * c] + M[3 * r + 1] * Ri.m[3 * c + 1] + M[3 * r + 2] * Ri.m[3 * c + 2]);
  }
  if (g_focal != nullptr) { g_focal[2 * b] = part[16 * b + 9]; g_focal[2 * b + 1] = part[16 * b + 10]; }
  if (g_pp != nullptr) { g_pp[2 * b] = part[16 * b + 11]; g_pp[2 * b + 1] = part[16 * b + 12]; }
}

}  // namespace voge

using namespace voge;

extern "C" int voge_rays_striped_fwd(const float *R, const float *T, const float *focal, const float *pp, int B,
                                     int row0, int h, int stripe_h, int pitch, int W, float *rays, float *origin, float *cones,
                                     voge_stream_t stream);
extern "C" int voge_rays_striped_bwd(const float *R, const float *T, const float *focal, const float *pp,
                                     const float *g_rays, const float *g_origin, int B, int row0, int h, int stripe_h, int pitch,
                                     int W, float *scratch, float *g_R, float *g_T, float *g_focal, float *g_pp, voge_stream_t stream);

extern "C" int voge_rays_fwd(const float *R, const float *T, const float *focal, const float *pp, int B,
                             int row0, int h, int W, float *rays, float *origin, float *cones, voge_stream_t stream) {
  return voge_rays_striped_fwd(R, T, focal, pp, B, row0, h, h > 0 ? h : 1, 0, W, rays, origin, cones, stream);
}

extern "C" int voge_rays_striped_fwd(const float *R, const float *T, const float *focal, const float *pp, int B,
                                     int row0, int h, int stripe_h, int pitch, int W, float *rays, float *origin, float *cones,
                                     voge_stream_t stream) {
  if (B < 0 || h < 0 || W < 0 || stripe_h <= 0 || pitch < 0) return VOGE_ERR_BAD_ARG;
  if (B == 0) return 0;
  if (!R || !T || !focal || !pp || !origin || ((size_t)h * W > 0 && !rays)) return VOGE_ERR_BAD_ARG;
  const int n = h * W;
  int blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  const int nst = (cones != nullptr && n > 0) ? ((W + kST - 1) / kST) * ((h + kST - 1) / kST) : 0;
  hipLaunchKernelGGL(rays_fwd_kernel, dim3(blocks + nst, B), dim3(256), 0, (hipStream_t)stream, R, T, focal, pp, row0, h, W,
                     stripe_h, pitch, rays, origin, blocks, reinterpret_cast<ConeRec *>(cones));
  return launch_status();
}

extern "C" size_t voge_cones_floats(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  // (per super-tile: its own record, four quad records, sixteen tile records -- voge_common.h, block_cones_hier256)
  return cone_records((size_t)B, (size_t)((W + kST - 1) / kST) * ((H + kST - 1) / kST)) * (sizeof(ConeRec) / sizeof(float));
}

extern "C" int voge_ray_cones(const float *rays, int B, int H, int W, float *cones, voge_stream_t stream) {
  if (B < 0 || H < 0 || W < 0) return VOGE_ERR_BAD_ARG;
  if ((size_t)B * H * W == 0) return 0;
  if (!rays || !cones) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(cones_kernel, dim3(((W + kST - 1) / kST) * ((H + kST - 1) / kST), B), dim3(256), 0, (hipStream_t)stream,
                     rays, H, W, reinterpret_cast<ConeRec *>(cones));
  return launch_status();
}

extern "C" int voge_rays_bwd(const float *R, const float *T, const float *focal, const float *pp,
                             const float *g_rays, const float *g_origin, int B, int row0, int h, int W,
                             float *scratch /* [B][16] floats */, float *g_R, float *g_T, float *g_focal,
                             float *g_pp, voge_stream_t stream) {
  return voge_rays_striped_bwd(R, T, focal, pp, g_rays, g_origin, B, row0, h, h > 0 ? h : 1, 0, W, scratch, g_R, g_T, g_focal, g_pp, stream);
}

extern "C" int voge_rays_striped_bwd(const float *R, const float *T, const float *focal, const float *pp,
                                     const float *g_rays, const float *g_origin, int B, int row0, int h, int stripe_h, int pitch,
                                     int W, float *scratch /* [B][16] floats */, float *g_R, float *g_T, float *g_focal,
                                     float *g_pp, voge_stream_t stream) {
  if (B < 0 || h < 0 || W < 0 || stripe_h <= 0 || pitch < 0) return VOGE_ERR_BAD_ARG;
  if (B == 0) return 0;
  if (!R || !T || !focal || !pp || !scratch) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = voge_fill_async(scratch, 0, sizeof(float) * 16 * (size_t)B, st);
  if (e != hipSuccess) return (int)e;
  const int n = h * W;
  if (n > 0 && g_rays != nullptr) {
    int blocks = (n + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(rays_bwd_reduce_kernel, dim3(blocks, B), dim3(256), 0, st, R, focal, pp, g_rays, row0, h, W,
                       stripe_h, pitch, scratch);
  }
  hipLaunchKernelGGL(rays_bwd_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, st, R, T, scratch, g_origin, B, g_R,
                     g_T, g_focal, g_pp);
  return launch_status();
}

// ------------------------------------------------------------------------------------------
// The renderer's elementwise preamble for (N,3) and (N,3,3) sigmas (VoGE/Renderer.py:130-137 with
// VoGE/Aggregation.py:144-175: centred = verts - origin[b]; isigma = 2 * expend_sigma(sigmas)) as ONE launch each way
// instead of four to five torch kernels: the general path's frames are short enough for those launches to show
// (8 % of the (N,3) frame at the cfg3 size).  Same fp32 operations, so the values are the torch chain's.
//   kind 1: sigmas [.., N, 3] (per-axis values: A = 2 diag(s)), kind 2: [.., N, 3, 3] (A = 2 S).
//   shared_verts / shared_sigmas: one [N, ...] set seen by every view, or [B, N, ...].
// ------------------------------------------------------------------------------------------
namespace voge {
__global__ void __launch_bounds__(256)
general_preamble_fwd_kernel(const float *__restrict__ verts, const float *__restrict__ sigmas, const float *__restrict__ origin,
                            const int B, const int N, const int shared_verts, const int shared_sigmas, const int kind,
                            float *__restrict__ mus, float *__restrict__ isg) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long)B * N) return;
  const int b = (int)(g / N), n = (int)(g - (long)b * N);
  const float *v = verts + 3 * (shared_verts ? (long)n : g), *o = origin + 3 * b;
  mus[3 * g + 0] = v[0] - o[0]; mus[3 * g + 1] = v[1] - o[1]; mus[3 * g + 2] = v[2] - o[2];
  float *A = isg + 9 * g;
  if (kind == 1) {
    const float *s = sigmas + 3 * (shared_sigmas ? (long)n : g);
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = (i % 4 == 0) ? 2.0f * s[i / 4] : 0.0f;
  } else {
    const float *s = sigmas + 9 * (shared_sigmas ? (long)n : g);
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = 2.0f * s[i];
  }
}
// thread <-> one Gaussian of the parameter set; a shared set sums its views in a fixed order (deterministic)
__global__ void __launch_bounds__(256)
general_preamble_bwd_kernel(const float *__restrict__ g_mus, const float *__restrict__ g_isg, const int B, const int N,
                            const int shared_verts, const int shared_sigmas, const int kind, float *__restrict__ g_verts,
                            float *__restrict__ g_sigmas) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g_verts != nullptr && g_mus != nullptr) {
    const long nv = shared_verts ? (long)N : (long)B * N;
    if (t < nv) {
      float x = 0.f, y = 0.f, z = 0.f;
      for (int b = 0; b < (shared_verts ? B : 1); ++b) {
        const float *gm = g_mus + 3 * (shared_verts ? (long)b * N + t : t);
        x += gm[0]; y += gm[1]; z += gm[2];
      }
      g_verts[3 * t + 0] = x; g_verts[3 * t + 1] = y; g_verts[3 * t + 2] = z;
    }
  }
  if (g_sigmas != nullptr && g_isg != nullptr) {
    const long ns = shared_sigmas ? (long)N : (long)B * N;
    if (t < ns) {
      float a[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) a[i] = 0.f;
      for (int b = 0; b < (shared_sigmas ? B : 1); ++b) {
        const float *ga = g_isg + 9 * (shared_sigmas ? (long)b * N + t : t);
#pragma unroll
        for (int i = 0; i < 9; ++i) a[i] += ga[i];
      }
      if (kind == 1) {
        g_sigmas[3 * t + 0] = 2.0f * a[0]; g_sigmas[3 * t + 1] = 2.0f * a[4]; g_sigmas[3 * t + 2] = 2.0f * a[8];
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) g_sigmas[9 * t + i] = 2.0f * a[i];
      }
    }
  }
}
}  // namespace voge

extern "C" int voge_general_preamble_fwd(const float *verts, const float *sigmas, const float *origin, int B, int N,
                                         int shared_verts, int shared_sigmas, int kind, float *mus, float *isigmas,
                                         voge_stream_t stream) {
  if (B < 0 || N < 0 || (kind != 1 && kind != 2)) return VOGE_ERR_BAD_ARG;
  if ((long)B * N == 0) return 0;
  if (!verts || !sigmas || !origin || !mus || !isigmas) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(general_preamble_fwd_kernel, dim3((unsigned)(((long)B * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     verts, sigmas, origin, B, N, shared_verts, shared_sigmas, kind, mus, isigmas);
  return launch_status();
}

extern "C" int voge_general_preamble_bwd(const float *g_mus, const float *g_isigmas, int B, int N, int shared_verts,
                                         int shared_sigmas, int kind, float *g_verts, float *g_sigmas,
                                         voge_stream_t stream) {
  if (B < 0 || N < 0 || (kind != 1 && kind != 2)) return VOGE_ERR_BAD_ARG;
  if ((long)B * N == 0) return 0;
  if ((g_verts && !g_mus) || (g_sigmas && !g_isigmas)) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(general_preamble_bwd_kernel, dim3((unsigned)(((long)B * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     g_mus, g_isigmas, B, N, shared_verts, shared_sigmas, kind, g_verts, g_sigmas);
  return launch_status();
}
