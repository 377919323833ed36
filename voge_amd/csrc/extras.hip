// Rows "next" of SURVEY.md §8f that reuse the path's arithmetic:
//   coarse stage     VoGE/csrc/rasterize_coarse/rasterize_coarse.cu:20-188 (EllipseBoundingBoxKernel +
//                    RasterizeCoarseCudaKernel): the reference's own candidate lists, for callers that want them
//   dense ray API    VoGE/csrc/voge_ray_tracing_ray/voge_ray_tracing_ray.cu:114-239
//                    (RayTraceVogeRayKernel, RayTraceVogeRayBackwardKernel, FindNearestKKernel)
//   scatter_max      VoGE/csrc/sample_voge/sample_voge.cu:69-92 (ScatterMaxKernel)
// (sample_voge itself is the transpose of merge_final and is served by voge_merge_fwd/bwd.)
#include "voge_common.h"

namespace voge {

// ---- dense forward: every (ray n, Gaussian m) pair -> len, act, dsd [N,M] --------------------
__global__ void __launch_bounds__(256)
ray_dense_fwd_kernel(const float *__restrict__ mus, const float *__restrict__ isg,
                     const float *__restrict__ rays, const int M, const long N, float *__restrict__ len,
                     float *__restrict__ act, float *__restrict__ dsd) {
  const long total = N * M;
  for (long pid = (long)blockIdx.x * blockDim.x + threadIdx.x; pid < total; pid += (long)gridDim.x * blockDim.x) {
    const long n = pid / M;
    const int m = (int)(pid - n * M);
    float A[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = isg[9 * (size_t)m + i];
    const float mx = mus[3 * (size_t)m], my = mus[3 * (size_t)m + 1], mz = mus[3 * (size_t)m + 2];
    const float dx = rays[3 * n], dy = rays[3 * n + 1], dz = rays[3 * n + 2];
    const EvalRec e = make_eval(mx, my, mz, A);
    const PairOut o = pair_eval(mx, my, mz, e, dx, dy, dz, dx * dx, dy * dy, dz * dz, dx * dy, dx * dz, dy * dz);
    len[pid] = o.len;
    act[pid] = o.act;
    dsd[pid] = o.dsd;
  }
}

// ---- dense backward.  Lane = Gaussian (64 consecutive m per wave), loop over a chunk of rays:
// the 12 per-Gaussian sums stay in registers across the chunk (one run of atomics per lane at the
// end), the per-ray sums are wave reductions (one atomic per ray and wave).  Same merged,
// cancellation-free terms as trace_bwd.hip.
constexpr int kDenseRays = 64;

__global__ void __launch_bounds__(64)
ray_dense_bwd_kernel(const float *__restrict__ mus, const float *__restrict__ isg,
                     const float *__restrict__ rays, const float *__restrict__ g_len,
                     const float *__restrict__ g_act, const float *__restrict__ g_dsd, const int M,
                     const long N, float *__restrict__ g_ray, float *__restrict__ g_mus,
                     float *__restrict__ g_isg) {
  const int lane = threadIdx.x;
  const int m = blockIdx.x * 64 + lane;
  const long n0 = (long)blockIdx.y * kDenseRays;
  const bool has = m < M;
  float A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, mx = 0, my = 0, mz = 0;
  if (has) {
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = isg[9 * (size_t)m + i];
    mx = mus[3 * (size_t)m]; my = mus[3 * (size_t)m + 1]; mz = mus[3 * (size_t)m + 2];
  }
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
  for (long n = n0; n < min(N, n0 + kDenseRays); ++n) {
    const float dx = rays[3 * n], dy = rays[3 * n + 1], dz = rays[3 * n + 2];
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (has) {
      const float gl = g_len[n * M + m], ga = g_act[n * M + m], gd = g_dsd[n * M + m];
      const float adx = fmaf(A[2], dz, fmaf(A[1], dy, A[0] * dx));
      const float ady = fmaf(A[5], dz, fmaf(A[4], dy, A[3] * dx));
      const float adz = fmaf(A[8], dz, fmaf(A[7], dy, A[6] * dx));
      const float tdx = fmaf(A[6], dz, fmaf(A[3], dy, A[0] * dx));
      const float tdy = fmaf(A[7], dz, fmaf(A[4], dy, A[1] * dx));
      const float tdz = fmaf(A[8], dz, fmaf(A[5], dy, A[2] * dx));
      const float ksk = fmaf(dz, adz, fmaf(dy, ady, dx * adx));
      const float msk = fmaf(mz, adz, fmaf(my, ady, mx * adx));
      const float ik = __builtin_amdgcn_rcpf(ksk);
      const float t = msk * ik;
      const float vx = fmaf(-t, dx, mx), vy = fmaf(-t, dy, my), vz = fmaf(-t, dz, mz);
      const float avx = fmaf(A[2], vz, fmaf(A[1], vy, A[0] * vx));
      const float avy = fmaf(A[5], vz, fmaf(A[4], vy, A[3] * vx));
      const float avz = fmaf(A[8], vz, fmaf(A[7], vy, A[6] * vx));
      const float tvx = fmaf(A[6], vz, fmaf(A[3], vy, A[0] * vx));
      const float tvy = fmaf(A[7], vz, fmaf(A[4], vy, A[1] * vx));
      const float tvz = fmaf(A[8], vz, fmaf(A[5], vy, A[2] * vx));
      const float c1 = gl * ik, gat = ga * t;
      acc[0] += fmaf(c1, adx, ga * (avx + tvx + t * (tdx - adx)));
      acc[1] += fmaf(c1, ady, ga * (avy + tvy + t * (tdy - ady)));
      acc[2] += fmaf(c1, adz, ga * (avz + tvz + t * (tdz - adz)));
      const float d[3] = {dx, dy, dz}, v[3] = {vx, vy, vz};
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          acc[3 + 3 * i + c] += fmaf(ga, fmaf(v[i], v[c], t * (d[i] * v[c] - v[i] * d[c])), fmaf(gd, d[i] * d[c], c1 * (v[i] * d[c])));
      rx = fmaf(gd, adx + tdx, fmaf(c1, fmaf(-t, adx, tvx), gat * fmaf(t, adx - tdx, -2.0f * tvx)));
      ry = fmaf(gd, ady + tdy, fmaf(c1, fmaf(-t, ady, tvy), gat * fmaf(t, ady - tdy, -2.0f * tvy)));
      rz = fmaf(gd, adz + tdz, fmaf(c1, fmaf(-t, adz, tvz), gat * fmaf(t, adz - tdz, -2.0f * tvz)));
    }
    rx = wave_sum(rx); ry = wave_sum(ry); rz = wave_sum(rz);
    if (lane < 3) unsafeAtomicAdd(g_ray + 3 * n + lane, lane == 0 ? rx : (lane == 1 ? ry : rz));
  }
  if (has) {
#pragma unroll
    for (int i = 0; i < 3; ++i) unsafeAtomicAdd(g_mus + 3 * (size_t)m + i, acc[i]);
#pragma unroll
    for (int i = 0; i < 9; ++i) unsafeAtomicAdd(g_isg + 9 * (size_t)m + i, acc[3 + i]);
  }
}

// ---- top-K over dense rows: one ray per lane, sorted key list in LDS (as the list sweep) -------
__global__ void __launch_bounds__(64)
nearest_k_kernel(const float *__restrict__ len_in, const float *__restrict__ act_in,
                 const float *__restrict__ dsd_in, const float thr_act, const int M, const int K,
                 const long N, int32_t *__restrict__ out_idx, float *__restrict__ out_len,
                 float *__restrict__ out_act, float *__restrict__ out_dsd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t *keys = reinterpret_cast<uint64_t *>(smem_raw);
  const int lane = threadIdx.x;
  const long n = (long)blockIdx.x * 64 + lane;
  if (n >= N) return;
  uint64_t *mykeys = keys + lane;
  int cnt = 0;
  uint64_t worst = ~0ull, tail = 0ull;
  for (int m = 0; m < M; ++m) {
    const float a = act_in[n * M + m];
    const float l = len_in[n * M + m] + 0.0f;
    if (a < thr_act && l < VOGE_SENT_LEN) {
      const uint64_t key = ((uint64_t)f2ord(l) << 32) | (uint32_t)m;
      if (key < worst) topk_insert(mykeys, 64, K, cnt, worst, tail, key);
    }
  }
  for (int s = 0; s < K; ++s) {
    int32_t oi = -1;
    float ol = VOGE_SENT_LEN, oa = 0.0f, od = 0.0f;   // FindNearestK fills act with 0 (voge_ray_tracing_ray.cu:344-347)
    if (s < cnt) {
      oi = (int32_t)(uint32_t)mykeys[(size_t)s * 64];
      ol = len_in[n * M + oi];
      oa = act_in[n * M + oi];
      od = dsd_in[n * M + oi];
    }
    out_idx[n * K + s] = oi;
    out_len[n * K + s] = ol;
    out_act[n * K + s] = oa;
    out_dsd[n * K + s] = od;
  }
}

// gradient of the selection: g_in[n, idx[n,k]] = g_out[n,k] for idx >= 0 (inputs pre-zeroed)
__global__ void __launch_bounds__(256)
nearest_k_bwd_kernel(const int32_t *__restrict__ idx, const float *__restrict__ g_len,
                     const float *__restrict__ g_act, const float *__restrict__ g_dsd, const int M,
                     const int K, const long N, float *__restrict__ gi_len, float *__restrict__ gi_act,
                     float *__restrict__ gi_dsd) {
  const long total = N * K;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int m = idx[t];
    if (m < 0 || m >= M) continue;
    const long n = t / K;
    gi_len[n * M + m] = g_len[t];
    gi_act[n * M + m] = g_act[t];
    gi_dsd[n * M + m] = g_dsd[t];
  }
}

// ---- scatter_max: out[idx] = max over slots of weight (weights are >= 0: int-ordered) ----------
__global__ void __launch_bounds__(256)
scatter_max_kernel(const float *__restrict__ weight, const int32_t *__restrict__ idx, const long n,
                   const long Nv, float *__restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int p = idx[i];
    if (p < 0 || p >= Nv) continue;
    const float w = fmaxf(weight[i], 0.0f);
    if (w > 0.0f) atomicMax(reinterpret_cast<int *>(out) + p, __float_as_int(w));
  }
}

static inline unsigned flat_grid(long items) {
  long b = (items + 255) / 256;
  if (b < 1) b = 1;
  if (b > 256L * 32) b = 256L * 32;
  return (unsigned)b;
}

// ---- the reference's coarse stage (rasterize_points_coarse): bbox = centre -+ radius, skipped when z < 0
// (rasterize_coarse.cu:20-42); a point is listed in every bin its bbox overlaps, bins padded by half a pixel
// (:111-135); points are taken in chunks of 512 and a chunk that no longer fits a bin's M slots is dropped while the
// bin's counter still advances (:154-170).  One workgroup per (bin, batch element) walks the chunks in ascending
// order, so the lists are ascending in index and deterministic -- the reference's order between chunks is whatever
// its atomicAdd race produced.  NaN radii (negative column sums in convert_to_box) never overlap, as there.
__device__ inline float ndc_range(const int S1, const int S2) {      // rasterization_utils.cuh:15-23
  float range = 2.0f;
  if (S1 > S2) range = (S1 * range) / S2;
  return range;
}
__device__ inline float pix_to_ndc(const int i, const int S1, const int S2) {      // rasterization_utils.cuh:36-42
  const float range = ndc_range(S1, S2);
  const float offset = range / 2.0f;
  return -offset + (range * i + offset) / S1;
}

constexpr int kCoarseChunk = 512;

__global__ void __launch_bounds__(256)
coarse_bin_kernel(const float *__restrict__ points /* [P,3] */, const float *__restrict__ radius /* [P,2] */,
                  const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_points, const int P, const int H,
                  const int W, const int bin_size, const int M, int32_t *__restrict__ bin_elems /* [B,BH,BW,M], -1 filled here */) {
  __shared__ int wcnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbx = 1 + (W - 1) / bin_size;
  const int bx = blockIdx.x % nbx, by = blockIdx.x / nbx, b = blockIdx.y;
  int32_t *out = bin_elems + ((size_t)b * gridDim.x + blockIdx.x) * M;
  const float half_x = ndc_range(W, H) / 2.0f / W, half_y = ndc_range(H, W) / 2.0f / H;
  const float bin_x_min = pix_to_ndc(bx * bin_size, W, H) - half_x, bin_x_max = pix_to_ndc((bx + 1) * bin_size - 1, W, H) + half_x;
  const float bin_y_min = pix_to_ndc(by * bin_size, H, W) - half_y, bin_y_max = pix_to_ndc((by + 1) * bin_size - 1, H, W) + half_y;
  const long e0 = first_idx[b], e1 = e0 + num_points[b];
  int run = 0;      // the bin's counter (elems_per_bin), workgroup-uniform
  for (int lo = 0; lo < P; lo += kCoarseChunk) {
    bool in[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lo + u * 256 + tid;
      in[u] = false;
      if (e < P && e >= e0 && e < e1) {
        const float x = points[3 * (size_t)e], y = points[3 * (size_t)e + 1], z = points[3 * (size_t)e + 2];
        const float rx = radius[2 * (size_t)e], ry = radius[2 * (size_t)e + 1];
        const float xmin = x - rx, xmax = x + rx, ymin = y - ry, ymax = y + ry;
        in[u] = !(z < 0) && (ymin <= bin_y_max) && (bin_y_min < ymax) && (xmin <= bin_x_max) && (bin_x_min < xmax);
      }
    }
    const unsigned long long m0 = __ballot(in[0]), m1 = __ballot(in[1]);
    __syncthreads();      // wcnt of the previous chunk consumed
    if (lane == 0) wcnt[wave] = __popcll(m0) | (__popcll(m1) << 16);
    __syncthreads();
    // order inside the chunk: element index = u * 256 + wave * 64 + lane
    int before[2] = {0, 0}, total = 0;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      for (int w = 0; w < 4; ++w) {
        const int c = (wcnt[w] >> (16 * u)) & 0xffff;
        if (w < wave) before[u] += c;
        total += c;
      }
    const int c0 = (wcnt[0] & 0xffff) + (wcnt[1] & 0xffff) + (wcnt[2] & 0xffff) + (wcnt[3] & 0xffff);
    const int start = run;
    run += total;
    if (start + total > M) continue;      // this chunk does not fit: dropped, the counter keeps its advance
    if (in[0]) out[start + before[0] + __popcll(m0 & ((1ull << lane) - 1ull))] = lo + tid;
    if (in[1]) out[start + c0 + before[1] + __popcll(m1 & ((1ull << lane) - 1ull))] = lo + 256 + tid;
  }
}

}  // namespace voge

using namespace voge;

extern "C" int voge_ray_dense_fwd(const float *mus, const float *isigmas, const float *rays, int M, long N,
                                  float *len, float *act, float *dsd, voge_stream_t stream) {
  if (M < 0 || N < 0) return VOGE_ERR_BAD_ARG;
  if ((long)M * N == 0) return 0;
  if (!mus || !isigmas || !rays || !len || !act || !dsd) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(ray_dense_fwd_kernel, dim3(flat_grid(N * M)), dim3(256), 0, (hipStream_t)stream, mus, isigmas,
                     rays, M, N, len, act, dsd);
  return launch_status();
}

extern "C" int voge_ray_dense_bwd(const float *mus, const float *isigmas, const float *rays, const float *g_len,
                                  const float *g_act, const float *g_dsd, int M, long N, float *g_ray,
                                  float *g_mus, float *g_isg, voge_stream_t stream) {
  if (M < 0 || N < 0) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (N > 0) { if (!g_ray) return VOGE_ERR_BAD_ARG; e = voge_fill_async(g_ray, 0, sizeof(float) * 3 * (size_t)N, st); if (e != hipSuccess) return (int)e; }
  if (M > 0) {
    if (!g_mus || !g_isg) return VOGE_ERR_BAD_ARG;
    e = voge_fill_async(g_mus, 0, sizeof(float) * 3 * (size_t)M, st); if (e != hipSuccess) return (int)e;
    e = voge_fill_async(g_isg, 0, sizeof(float) * 9 * (size_t)M, st); if (e != hipSuccess) return (int)e;
  }
  if ((long)M * N == 0) return 0;
  if (!mus || !isigmas || !rays || !g_len || !g_act || !g_dsd) return VOGE_ERR_BAD_ARG;
  dim3 grid((M + 63) / 64, (unsigned)((N + kDenseRays - 1) / kDenseRays));
  hipLaunchKernelGGL(ray_dense_bwd_kernel, grid, dim3(64), 0, st, mus, isigmas, rays, g_len, g_act, g_dsd, M, N, g_ray,
                     g_mus, g_isg);
  return launch_status();
}

extern "C" int voge_find_nearest_k(const float *len_in, const float *act_in, const float *dsd_in, float thr_act,
                                   int M, int K, long N, int32_t *idx, float *len, float *act, float *dsd,
                                   voge_stream_t stream) {
  if (M < 0 || N < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K) return VOGE_ERR_K_TOO_LARGE;
  if (N == 0) return 0;
  if (!idx || !len || !act || !dsd || (M > 0 && (!len_in || !act_in || !dsd_in))) return VOGE_ERR_BAD_ARG;
  const size_t lds = sizeof(uint64_t) * (size_t)K * 64;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(nearest_k_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(nearest_k_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), lds, (hipStream_t)stream, len_in, act_in,
                     dsd_in, thr_act, M, K, N, idx, len, act, dsd);
  return launch_status();
}

extern "C" int voge_find_nearest_k_bwd(const int32_t *idx, const float *g_len, const float *g_act, const float *g_dsd,
                                       int M, int K, long N, float *gi_len, float *gi_act, float *gi_dsd,
                                       voge_stream_t stream) {
  if (M < 0 || N < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = sizeof(float) * (size_t)N * M;
  if (bytes > 0) {
    if (!gi_len || !gi_act || !gi_dsd) return VOGE_ERR_BAD_ARG;
    hipError_t e = voge_fill_async(gi_len, 0, bytes, st); if (e != hipSuccess) return (int)e;
    e = voge_fill_async(gi_act, 0, bytes, st); if (e != hipSuccess) return (int)e;
    e = voge_fill_async(gi_dsd, 0, bytes, st); if (e != hipSuccess) return (int)e;
  }
  if (N * K == 0 || M == 0) return 0;
  if (!idx || !g_len || !g_act || !g_dsd) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(nearest_k_bwd_kernel, dim3(flat_grid(N * K)), dim3(256), 0, st, idx, g_len, g_act, g_dsd, M, K, N,
                     gi_len, gi_act, gi_dsd);
  return launch_status();
}

extern "C" int voge_scatter_max(const float *weight, const int32_t *idx, long n, long Nv, float *out,
                                voge_stream_t stream) {
  if (n < 0 || Nv < 0) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (Nv > 0) {
    if (!out) return VOGE_ERR_BAD_ARG;
    hipError_t e = voge_fill_async(out, 0, sizeof(float) * (size_t)Nv, st);
    if (e != hipSuccess) return (int)e;
  }
  if (n == 0 || Nv == 0) return 0;
  if (!weight || !idx) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(scatter_max_kernel, dim3(flat_grid(n)), dim3(256), 0, st, weight, idx, n, Nv, out);
  return launch_status();
}

extern "C" int voge_bin_gaussians(const float *points, const int64_t *cloud_to_packed_first_idx,
                                  const int64_t *num_points_per_cloud, int B, int P, int H, int W, const float *radius,
                                  int bin_size, int max_points_per_bin, int32_t *bin_elems, voge_stream_t stream) {
  if (B < 0 || P < 0 || H <= 0 || W <= 0 || bin_size <= 0 || max_points_per_bin < 0) return VOGE_ERR_BAD_ARG;
  const int nbx = 1 + (W - 1) / bin_size, nby = 1 + (H - 1) / bin_size;
  if (nbx >= 66 || nby >= 66) return VOGE_ERR_BAD_ARG;      // kMaxItemsPerBin (rasterize_coarse.cu:213-219)
  const size_t n = (size_t)B * nby * nbx * max_points_per_bin;
  if (n == 0) return 0;
  if (!bin_elems || !cloud_to_packed_first_idx || !num_points_per_cloud || (P > 0 && (!points || !radius))) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const hipError_t e = voge_fill_async(bin_elems, 0xff, n * sizeof(int32_t), st);      // at::full(-1), :222
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(coarse_bin_kernel, dim3(nbx * nby, B), dim3(256), 0, st, points, radius, cloud_to_packed_first_idx,
                     num_points_per_cloud, P, H, W, bin_size, max_points_per_bin, bin_elems);
  return launch_status();
}
