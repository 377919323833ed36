// Attribute merge and background blend for gfx950.
//
// Reference behaviour being reproduced: merge_final (VoGE/Aggregation.py:111-141, reached via
// interpolate_attr, VoGE/Renderer.py:153) and get_silhouette / to_colored_background
// (VoGE/Renderer.py:157-171).  The reference gathers a [B,H,W,K,C] temporary; here each
// thread owns one pixel and a group of up to 4 channels and accumulates in registers.
#include "voge_common.h"

namespace voge {

__global__ void __launch_bounds__(256)
merge_fwd_kernel(const float *__restrict__ attr, int32_t *__restrict__ idx,
                 const float *__restrict__ weight, const int64_t *__restrict__ valid_num,
                 const long npix, const int K, const int C, const long Nattr, const int fix_idx,
                 float *__restrict__ out) {
  const int groups = (C + 3) / 4;
  const long total = npix * groups;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long pix = t / groups;
    const int c0 = (int)(t - pix * groups) * 4;
    const int nc = min(4, C - c0);
    const int nv = (int)min((int64_t)K, max((int64_t)0, valid_num[pix]));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nv; ++k) {
      int p = idx[pix * K + k];
      p += (p < 0);
      if (p < 0 || p >= Nattr) continue;
      const float w = weight[pix * K + k];
      const float *a = attr + (size_t)p * C + c0;
      for (int c = 0; c < nc; ++c) acc[c] = fmaf(a[c], w, acc[c]);
    }
    for (int c = 0; c < nc; ++c) out[pix * C + c0 + c] = acc[c];
  }
}

// The reference mutates the index tensor in place over ALL slots (Aggregation.py:131).
__global__ void __launch_bounds__(256)
fix_idx_kernel(int32_t *__restrict__ idx, const long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int p = idx[i];
    if (p < 0) idx[i] = p + 1;
  }
}

__global__ void __launch_bounds__(256)
merge_bwd_kernel(const float *__restrict__ attr, const int32_t *__restrict__ idx,
                 const float *__restrict__ weight, const int64_t *__restrict__ valid_num,
                 const float *__restrict__ g_out, const long npix, const int K, const int C,
                 const long Nattr, float *__restrict__ g_attr, float *__restrict__ g_weight) {
  const long total = npix * K;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long pix = t / K;
    const int k = (int)(t - pix * K);
    float gw = 0.0f;
    if (k < valid_num[pix]) {
      int p = idx[t];
      p += (p < 0);
      if (p >= 0 && p < Nattr) {
        const float w = weight[t];
        const float *a = attr + (size_t)p * C;
        const float *g = g_out + pix * C;
        for (int c = 0; c < C; ++c) {
          const float gc = g[c];
          gw = fmaf(gc, a[c], gw);
          if (g_attr != nullptr && w != 0.0f && gc != 0.0f) unsafeAtomicAdd(g_attr + (size_t)p * C + c, w * gc);
        }
      }
    }
    if (g_weight != nullptr) g_weight[t] = gw;
  }
}

// C <= 4 (colours): a workgroup owns a 16x16 pixel tile and accumulates the per-Gaussian
// attribute gradient in an LDS hash table, flushed with one global atomic per (Gaussian,
// channel) per tile (same scheme as trace_bwd_kernel).
constexpr int kMT = 16, kMHS = 1024, kMProbe = 24;
struct MergeBwdLds {
  int keys[kMHS];
  float vals[kMHS * 4];
};

__global__ void __launch_bounds__(256)
merge_bwd_tile_kernel(const float *__restrict__ attr, const int32_t *__restrict__ idx,
                      const float *__restrict__ weight, const int64_t *__restrict__ valid_num,
                      const float *__restrict__ g_out, const long nrows, const int W, const int K,
                      const int C, const long Nattr, float *__restrict__ g_attr,
                      float *__restrict__ g_weight) {
  __shared__ MergeBwdLds L;
  const int tid = threadIdx.x;
  const int tiles_x = (W + kMT - 1) / kMT;
  const int x0 = (blockIdx.x % tiles_x) * kMT;
  const long y0 = (long)(blockIdx.x / tiles_x) * kMT;
  const int tw = min(kMT, W - x0);
  const int th = (int)min((long)kMT, nrows - y0);
  for (int i = tid; i < kMHS; i += 256) L.keys[i] = -1;
  for (int i = tid; i < kMHS * 4; i += 256) L.vals[i] = 0.0f;
  __syncthreads();
  const int row_items = tw * K;
  for (int r = 0; r < th; ++r) {
    const long row_base = ((y0 + r) * W + x0) * (long)K;
    for (int it = tid; it < row_items; it += 256) {
      const long t = row_base + it;
      const int lx = it / K, k = it - lx * K;
      const long pix = (y0 + r) * W + x0 + lx;
      float gw = 0.0f;
      if (k < valid_num[pix]) {
        int p = idx[t];
        p += (p < 0);
        if (p >= 0 && p < Nattr) {
          const float w = weight[t];
          float ga[4] = {0.f, 0.f, 0.f, 0.f};
          bool any = false;
          for (int c = 0; c < C; ++c) {
            const float gc = g_out[pix * C + c];
            gw = fmaf(gc, attr[(size_t)p * C + c], gw);
            ga[c] = w * gc;
            any = any || (ga[c] != 0.0f);
          }
          if (g_attr != nullptr && any) {
            unsigned h = ((unsigned)p * 2654435761u) >> 22;
            int slot = -1;
#pragma unroll 1
            for (int pr = 0; pr < kMProbe; ++pr) {
              const int old = atomicCAS(&L.keys[h], -1, p);
              if (old == -1 || old == p) { slot = (int)h; break; }
              h = (h + 1) & (kMHS - 1);
            }
            if (slot >= 0) {
              for (int c = 0; c < C; ++c) atomicAdd(&L.vals[slot * 4 + c], ga[c]);
            } else {
              for (int c = 0; c < C; ++c) unsafeAtomicAdd(g_attr + (size_t)p * C + c, ga[c]);
            }
          }
        }
      }
      if (g_weight != nullptr) g_weight[t] = gw;
    }
  }
  __syncthreads();
  if (g_attr != nullptr) {
    // 4 adjacent lanes per table entry -> adjacent floats of g_attr[p]: lane-coalesced atomics
    const int c = tid & 3;
    for (int s = tid >> 2; s < kMHS; s += 64) {
      const int p = L.keys[s];
      if (p >= 0 && c < C) unsafeAtomicAdd(g_attr + (size_t)p * C + c, L.vals[s * 4 + c]);
    }
  }
}

// min(x, 1) passes the gradient where x < 1 and half of it at the tie, like torch.min.
__device__ __forceinline__ float clamp1_pass(float x) { return x < 1.0f ? 1.0f : (x == 1.0f ? 0.5f : 0.0f); }

__global__ void __launch_bounds__(256)
blend_fwd_kernel(const float *__restrict__ rgb, const float *__restrict__ weight,
                 const float *__restrict__ bg, const float thr, const long npix, const int K,
                 const int C, float *__restrict__ out, float *__restrict__ sil_out) {
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += weight[pix * K + k];
    float sil = fminf(s, 1.0f);
    if (sil_out != nullptr) sil_out[pix] = sil;
    if (thr > 0.0f) sil = sil > thr ? 1.0f : 0.0f;
    for (int c = 0; c < C; ++c) out[pix * C + c] = fminf(fmaf(1.0f - sil, bg[c], rgb[pix * C + c]), 1.0f);
  }
}

__global__ void __launch_bounds__(256)
blend_bwd_kernel(const float *__restrict__ rgb, const float *__restrict__ weight,
                 const float *__restrict__ bg, const float thr, const float *__restrict__ g_out,
                 const long npix, const int K, const int C, float *__restrict__ g_rgb,
                 float *__restrict__ g_weight_add) {
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += weight[pix * K + k];
    float sil = fminf(s, 1.0f);
    const float pass_s = (thr > 0.0f) ? 0.0f : clamp1_pass(s);
    if (thr > 0.0f) sil = sil > thr ? 1.0f : 0.0f;
    float g_mask = 0.0f;
    for (int c = 0; c < C; ++c) {
      const float x = fmaf(1.0f - sil, bg[c], rgb[pix * C + c]);
      const float g = g_out[pix * C + c] * clamp1_pass(x);
      if (g_rgb != nullptr) g_rgb[pix * C + c] = g;
      g_mask = fmaf(-g, bg[c], g_mask);
    }
    if (g_weight_add != nullptr) {
      const float gs = g_mask * pass_s;
      for (int k = 0; k < K; ++k) g_weight_add[pix * K + k] = gs;
    }
  }
}

static inline unsigned grid_for(long items) {
  long b = (items + 255) / 256;
  if (b < 1) b = 1;
  if (b > 256L * 32) b = 256L * 32;
  return (unsigned)b;
}

}  // namespace voge

using namespace voge;

extern "C" int voge_merge_fwd(const float *attr, int32_t *idx, const float *weight,
                              const int64_t *valid_num, long npix, int K, int C, long Nattr,
                              int fix_negative_idx, float *out, voge_stream_t stream) {
  if (npix < 0 || K <= 0 || C <= 0 || Nattr < 0) return VOGE_ERR_BAD_ARG;
  if (npix == 0) return 0;
  if (!idx || !weight || !valid_num || !out || (Nattr > 0 && !attr)) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(merge_fwd_kernel, dim3(grid_for(npix * ((C + 3) / 4))), dim3(256), 0, st, attr, idx, weight,
                     valid_num, npix, K, C, Nattr, fix_negative_idx, out);
  int rc = launch_status();
  if (rc || !fix_negative_idx) return rc;
  hipLaunchKernelGGL(fix_idx_kernel, dim3(grid_for(npix * K)), dim3(256), 0, st, idx, npix * K);
  return launch_status();
}

extern "C" int voge_merge_bwd(const float *attr, const int32_t *idx, const float *weight,
                              const int64_t *valid_num, const float *g_out, long nrows, int W, int K,
                              int C, long Nattr, float *g_attr, float *g_weight, voge_stream_t stream) {
  if (nrows < 0 || W < 0 || K <= 0 || C <= 0 || Nattr < 0) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (g_attr != nullptr && Nattr > 0) {
    hipError_t e = hipMemsetAsync(g_attr, 0, sizeof(float) * (size_t)Nattr * C, st);
    if (e != hipSuccess) return (int)e;
  }
  const long npix = nrows * W;
  if (npix == 0) return 0;
  if (!idx || !weight || !valid_num || !g_out || (Nattr > 0 && !attr)) return VOGE_ERR_BAD_ARG;
  if (C <= 4) {
    const long tiles = (long)((W + kMT - 1) / kMT) * ((nrows + kMT - 1) / kMT);
    hipLaunchKernelGGL(merge_bwd_tile_kernel, dim3((unsigned)tiles), dim3(256), 0, st, attr, idx, weight, valid_num,
                       g_out, nrows, W, K, C, Nattr, g_attr, g_weight);
  } else {
    hipLaunchKernelGGL(merge_bwd_kernel, dim3(grid_for(npix * K)), dim3(256), 0, st, attr, idx, weight, valid_num,
                       g_out, npix, K, C, Nattr, g_attr, g_weight);
  }
  return launch_status();
}

extern "C" int voge_blend_fwd(const float *rgb, const float *weight, const float *bg, float thr,
                              long npix, int K, int C, float *out, float *sil_out,
                              voge_stream_t stream) {
  if (npix < 0 || K <= 0 || C <= 0) return VOGE_ERR_BAD_ARG;
  if (npix == 0) return 0;
  if (!rgb || !weight || !bg || !out) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(blend_fwd_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, rgb, weight, bg,
                     thr, npix, K, C, out, sil_out);
  return launch_status();
}

extern "C" int voge_blend_bwd(const float *rgb, const float *weight, const float *bg, float thr,
                              const float *g_out, long npix, int K, int C, float *g_rgb,
                              float *g_weight_add, voge_stream_t stream) {
  if (npix < 0 || K <= 0 || C <= 0) return VOGE_ERR_BAD_ARG;
  if (npix == 0) return 0;
  if (!rgb || !weight || !bg || !g_out) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, rgb, weight, bg,
                     thr, g_out, npix, K, C, g_rgb, g_weight_add);
  return launch_status();
}

extern "C" int voge_abi_version(void) { return VOGE_ABI_VERSION; }

extern "C" const char *voge_error_string(int code) {
  switch (code) {
    case 0: return "success";
    case VOGE_ERR_BAD_ARG: return "voge: bad argument (null pointer or non-positive size)";
    case VOGE_ERR_WORKSPACE: return "voge: workspace smaller than voge_trace_workspace_bytes()";
    case VOGE_ERR_K_TOO_LARGE: return "voge: K exceeds VOGE_MAX_K (top-K lists are kept in LDS)";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "voge: unknown error code";
}
