// Fine ray trace forward for gfx950: per-Gaussian prep, cone-culled LDS-tiled sweep with a
// per-lane top-K in LDS, and the explicit-candidate-list variant.
//
// Reference behaviour being reproduced: RayTraceFineVogeKernel
// (VoGE/csrc/ray_trace_voge/ray_trace_voge.cu:135-217) + the host wrapper (:219-280) and the
// "-1" candidate list of VoGE/RayTracing.py:22-26.  Design notes are in DESIGN.md §Kernels.
#include "voge_common.h"

namespace voge {

// ------------------------------------------------------------------------------------------
// prep: one thread per Gaussian.  Reads mu (12 B) + A (36 B), writes cull (16 B) + eval (48 B).
// The reach uses the smallest eigenvalue of sym(A) (closed form, fp64 -- P-sized work).
// ------------------------------------------------------------------------------------------
__device__ inline double lambda_min_sym3(double a00, double a11, double a22, double a01,
                                         double a02, double a12) {
  const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
  if (p1 == 0.0) return fmin(a00, fmin(a11, a22));
  const double q = (a00 + a11 + a22) / 3.0;
  const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
  const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
  const double p = sqrt(p2 / 6.0);
  const double ip = 1.0 / p;
  const double c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip;
  const double c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
  double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) +
                    c02 * (c01 * c12 - c11 * c02));
  r = fmin(1.0, fmax(-1.0, r));
  const double phi = acos(r) / 3.0;
  return q + 2.0 * p * cos(phi + 2.0943951023931953);
}

__global__ void __launch_bounds__(256)
prep_kernel(const float *__restrict__ mus, const float *__restrict__ isg,
            const float *__restrict__ cam_fwd, const int N, const int P, const float thr_act,
            float4 *__restrict__ cull, float4 *__restrict__ evr) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P) return;
  const float mx = mus[3 * g + 0], my = mus[3 * g + 1], mz = mus[3 * g + 2];
  float A[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) A[i] = isg[9 * (size_t)g + i];
  const EvalRec e = make_eval(mx, my, mz, A);

  const double lmin = lambda_min_sym3(A[0], A[4], A[8], 0.5 * ((double)A[1] + A[3]),
                                      0.5 * ((double)A[2] + A[6]), 0.5 * ((double)A[5] + A[7]));
  const double lmax_bound = fabs((double)A[0]) + fabs((double)A[4]) + fabs((double)A[8]) +
                            fabs((double)e.s01) + fabs((double)e.s02) + fabs((double)e.s12);
  const double lsafe = lmin * (1.0 - 1e-6) - 1e-12 * lmax_bound;
  float reach = INFINITY;
  if (lsafe > 0.0 && lsafe < 1e300) {
    const double nb = sqrt((double)e.bx * e.bx + (double)e.by * e.by + (double)e.bz * e.bz);
    const double nk = sqrt((double)e.kx * e.kx + (double)e.ky * e.ky + (double)e.kz * e.kz);
    const double nm = sqrt((double)mx * mx + (double)my * my + (double)mz * mz);
    // act >= lmin*dist^2 - |len|*|k|, |len| <= |b|/lmin  ->  dist^2 <= (thr + |k||b|/lmin)/lmin
    const double thr2 = (double)thr_act + 1.000001 * nk * nb / lsafe;
    const double r = sqrt(fmax(thr2, 0.0) / lsafe) * (1.0 + 1e-5) + 1e-5 * nm + 1e-30;
    reach = (float)(r * (1.0 + 1e-6));
    if (!(reach >= 0.0f)) reach = INFINITY;  // NaN guard
  }
  if (cam_fwd != nullptr) {
    const float *f = cam_fwd + 3 * (g / N);
    if (fmaf(mz, f[2], fmaf(my, f[1], mx * f[0])) < 0.0f) reach = -1.0f;
  }
  cull[g] = make_float4(mx, my, mz, reach);
  evr[3 * (size_t)g + 0] = make_float4(e.s00, e.s11, e.s22, e.s01);
  evr[3 * (size_t)g + 1] = make_float4(e.s02, e.s12, e.bx, e.by);
  evr[3 * (size_t)g + 2] = make_float4(e.bz, e.kx, e.ky, e.kz);
}

__device__ __forceinline__ EvalRec unpack_eval(const float4 a, const float4 b, const float4 c) {
  EvalRec e;
  e.s00 = a.x; e.s11 = a.y; e.s22 = a.z; e.s01 = a.w;
  e.s02 = b.x; e.s12 = b.y; e.bx = b.z; e.by = b.w;
  e.bz = c.x; e.kx = c.y; e.ky = c.z; e.kz = c.w;
  return e;
}

// ------------------------------------------------------------------------------------------
// sweep.  One workgroup = WAVES waves = a TW x TH pixel tile (each wave an 8x8 sub-tile, one
// ray per lane).  The Gaussian stream of batch b is read once per workgroup in chunks of T:
//   fill   : thread i tests Gaussian base+i against the workgroup's bounding cone; survivors
//            are compacted (in index order) into LDS with their eval record;
//   consume: each wave re-tests the survivors against its own 8x8 cone, 64 at a time (one per
//            lane, ballot), then for every remaining candidate all 64 lanes evaluate their
//            ray against it (record broadcast from LDS) and insert into their LDS top-K list.
// Both culls are conservative (cone_keep), so the result equals the brute-force sweep.
// ------------------------------------------------------------------------------------------
template <int CAP>  // survivors buffered in LDS between fill and consume
struct TraceLds {
  // layout inside dynamic LDS, after the [K][T] key array
  float4 cull[CAP];
  float4 ev[CAP * 3];
  int32_t id[CAP];
  float red[4 * 8];
  int wcnt[2][4];
};

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
trace_fwd_kernel(const float4 *__restrict__ cull, const float4 *__restrict__ evr,
                 const float *__restrict__ rays, const int N, const int H, const int W,
                 const int K, const float thr_act, int32_t *__restrict__ out_idx,
                 float *__restrict__ out_len, float *__restrict__ out_act,
                 float *__restrict__ out_dsd) {
  constexpr int T = 64 * WAVES;
  constexpr int TW = (WAVES >= 2) ? 16 : 8;
  constexpr int TH = (WAVES == 4) ? 16 : 8;
  constexpr int kCap = 2 * T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t *keys = reinterpret_cast<uint64_t *>(smem_raw);
  TraceLds<kCap> &L = *reinterpret_cast<TraceLds<kCap> *>(smem_raw + sizeof(uint64_t) * (size_t)K * T);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + TW - 1) / TW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int px = tx * TW + (wave & 1) * 8 * (TW == 16) + (lane & 7);
  const int py = ty * TH + (wave >> 1) * 8 + (lane >> 3);
  const bool valid = (px < W) && (py < H);
  const int cpx = min(px, W - 1), cpy = min(py, H - 1);
  const size_t ray_id = ((size_t)b * H + cpy) * W + cpx;
  const float dx = rays[3 * ray_id + 0], dy = rays[3 * ray_id + 1], dz = rays[3 * ray_id + 2];
  const float qxx = dx * dx, qyy = dy * dy, qzz = dz * dz, qxy = dx * dy, qxz = dx * dz, qyz = dy * dz;

  // ---- bounding cones (wave, then workgroup) ---------------------------------------------
  const float dn2 = qxx + qyy + qzz;
  const float inv = 1.0f / sqrtf(dn2);
  const bool dir_ok = (dn2 > 0.0f) && (inv < 3e38f) && (inv == inv);
  const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
  const bool wave_dirs_ok = __all(dir_ok);
  auto make_cone = [&](float sx, float sy, float sz, bool all_ok, auto red_max, auto red_min) {
    Cone c;
    const float n = sqrtf(sx * sx + sy * sy + sz * sz);
    c.ax = sx / n; c.ay = sy / n; c.az = sz / n;
    const float cl = fmaf(uz, c.az, fmaf(uy, c.ay, ux * c.ax));
    const float rx = fmaf(-cl, c.ax, ux), ry = fmaf(-cl, c.ay, uy), rz = fmaf(-cl, c.az, uz);
    const float sl = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx * rx)));
    const float smax = red_max(dir_ok ? sl : 2.0f);
    const float cmin = red_min(dir_ok ? cl : -1.0f);
    c.sn = smax * (1.0f + 1e-5f) + 1e-7f;
    c.cs = cmin - 1e-6f;
    c.ok = all_ok && (n > 1e-3f) && (cmin > 0.05f) && (c.sn == c.sn);
    return c;
  };
  const float wsx = wave_sum(dir_ok ? ux : 0.f), wsy = wave_sum(dir_ok ? uy : 0.f),
              wsz = wave_sum(dir_ok ? uz : 0.f);
  const Cone wcone = make_cone(wsx, wsy, wsz, wave_dirs_ok, [](float v) { return wave_max(v); },
                               [](float v) { return wave_min(v); });
  Cone gcone = wcone;
  if (WAVES > 1) {
    if (lane == 0) {
      L.red[wave * 8 + 0] = wsx; L.red[wave * 8 + 1] = wsy; L.red[wave * 8 + 2] = wsz;
      L.red[wave * 8 + 3] = wave_dirs_ok ? 1.f : 0.f;
    }
    __syncthreads();
    float gx = 0, gy = 0, gz = 0; bool gok = true;
    for (int w = 0; w < WAVES; ++w) {
      gx += L.red[w * 8 + 0]; gy += L.red[w * 8 + 1]; gz += L.red[w * 8 + 2];
      gok = gok && (L.red[w * 8 + 3] != 0.f);
    }
    // per-wave extrema w.r.t. the workgroup axis, then across waves through LDS
    const Cone part = make_cone(gx, gy, gz, gok, [](float v) { return wave_max(v); },
                                [](float v) { return wave_min(v); });
    if (lane == 0) { L.red[wave * 8 + 4] = part.sn; L.red[wave * 8 + 5] = part.cs; L.red[wave * 8 + 6] = part.ok ? 1.f : 0.f; }
    __syncthreads();
    gcone = part;
    for (int w = 0; w < WAVES; ++w) {
      gcone.sn = fmaxf(gcone.sn, L.red[w * 8 + 4]);
      gcone.cs = fminf(gcone.cs, L.red[w * 8 + 5]);
      gcone.ok = gcone.ok && (L.red[w * 8 + 6] != 0.f);
    }
  }

  // ---- sweep -------------------------------------------------------------------------------
  uint64_t *mykeys = keys + tid;
  int cnt = 0;
  uint64_t worst = ~0ull;
  const float4 *cullb = cull + (size_t)b * N;
  const float4 *evrb = evr + (size_t)b * N * 3;
  const float4 cull_none = make_float4(0.f, 0.f, 0.f, -1.f);

  int base = 0, par = 0;
  float4 cnext = (tid < N) ? cullb[tid] : cull_none;
  while (base < N) {
    int nbuf = 0;
    // fill: append survivors until another full chunk might not fit
    while (base < N && nbuf + T <= kCap) {
      const int g = base + tid;
      const float4 c = cnext;
      const int gn = g + T;
      cnext = (gn < N) ? cullb[gn] : cull_none;
      const bool keep = cone_keep(c, gcone);
      const unsigned long long m = __ballot(keep);
      if (lane == 0) L.wcnt[par][wave] = __popcll(m);
      __syncthreads();
      int off = nbuf, tot = 0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const int cw = L.wcnt[par][w];
        if (w < wave) off += cw;
        tot += cw;
      }
      if (keep) {
        const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
        L.cull[slot] = c;
        L.id[slot] = g;
        L.ev[slot * 3 + 0] = evrb[(size_t)g * 3 + 0];
        L.ev[slot * 3 + 1] = evrb[(size_t)g * 3 + 1];
        L.ev[slot * 3 + 2] = evrb[(size_t)g * 3 + 2];
      }
      nbuf += tot;
      base += T;
      par ^= 1;
    }
    __syncthreads();
    // consume
    for (int c0 = 0; c0 < nbuf; c0 += 64) {
      const int i = c0 + lane;
      bool keep = false;
      if (i < nbuf) keep = (WAVES == 1) ? true : cone_keep(L.cull[i], wcone);
      unsigned long long m = __ballot(keep);
      while (m) {
        const int j = __builtin_ctzll(m);
        m &= m - 1ull;
        const int s = c0 + j;
        const float4 cc = L.cull[s];
        const EvalRec e = unpack_eval(L.ev[s * 3 + 0], L.ev[s * 3 + 1], L.ev[s * 3 + 2]);
        const PairOut o = pair_eval(cc.x, cc.y, cc.z, e, dx, dy, dz, qxx, qyy, qzz, qxy, qxz, qyz);
        if (valid && o.act < thr_act && o.len < VOGE_SENT_LEN) {
          const uint64_t key = ((uint64_t)f2ord(o.len) << 32) | (uint32_t)(L.id[s] + b * N);
          if (key < worst) topk_insert(mykeys, T, K, cnt, worst, key);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: decode the winners, recompute act / dsd with the same arithmetic ----------
  if (!valid) return;
  const size_t pix = ((size_t)b * H + py) * W + px;
  for (int s = 0; s < K; ++s) {
    int32_t oi = -1;
    float ol = VOGE_SENT_LEN, oa = VOGE_SENT_ACT, od = 0.0f;
    if (s < cnt) {
      const uint64_t key = mykeys[(size_t)s * T];
      oi = (int32_t)(uint32_t)key;
      const float4 cc = cull[oi];
      const EvalRec e = unpack_eval(evr[(size_t)oi * 3 + 0], evr[(size_t)oi * 3 + 1], evr[(size_t)oi * 3 + 2]);
      const PairOut o = pair_eval(cc.x, cc.y, cc.z, e, dx, dy, dz, qxx, qyy, qzz, qxy, qxz, qyz);
      ol = ord2f((uint32_t)(key >> 32));
      oa = o.act;
      od = o.dsd;
    }
    out_idx[pix * K + s] = oi;
    out_len[pix * K + s] = ol;
    out_act[pix * K + s] = oa;
    out_dsd[pix * K + s] = od;
  }
}

// ------------------------------------------------------------------------------------------
// explicit candidate lists (the reference's bin_points tensor): one ray per lane, each lane
// walks the list of the bin its pixel falls in.  Compatibility path, no culling.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
trace_list_fwd_kernel(const float *__restrict__ mus, const float *__restrict__ isg,
                      const float *__restrict__ rays, const int32_t *__restrict__ bins,
                      const int P, const int H, const int W, const int K, const int BH,
                      const int BW, const int M, const int bin_size, const float thr_act,
                      int32_t *__restrict__ out_idx, float *__restrict__ out_len,
                      float *__restrict__ out_act, float *__restrict__ out_dsd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t *keys = reinterpret_cast<uint64_t *>(smem_raw);
  const int lane = threadIdx.x;
  const int tiles_x = (W + 7) / 8;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
  if (px >= W || py >= H) return;  // no barriers below
  const size_t pix = ((size_t)b * H + py) * W + px;
  const float dx = rays[3 * pix + 0], dy = rays[3 * pix + 1], dz = rays[3 * pix + 2];
  const float qxx = dx * dx, qyy = dy * dy, qzz = dz * dz, qxy = dx * dy, qxz = dx * dz, qyz = dy * dz;
  const int by = min(py / bin_size, BH - 1), bx = min(px / bin_size, BW - 1);
  const int32_t *lst = bins + (((size_t)b * BH + by) * BW + bx) * M;
  uint64_t *mykeys = keys + lane;
  int cnt = 0;
  uint64_t worst = ~0ull;
  for (int m = 0; m < M; ++m) {
    const int p = lst[m];
    if (p < 0 || p >= P) continue;
    float A[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = isg[9 * (size_t)p + i];
    const float mx = mus[3 * (size_t)p], my = mus[3 * (size_t)p + 1], mz = mus[3 * (size_t)p + 2];
    const EvalRec e = make_eval(mx, my, mz, A);
    const PairOut o = pair_eval(mx, my, mz, e, dx, dy, dz, qxx, qyy, qzz, qxy, qxz, qyz);
    if (o.act < thr_act && o.len < VOGE_SENT_LEN) {
      const uint64_t key = ((uint64_t)f2ord(o.len) << 32) | (uint32_t)p;
      if (key < worst) topk_insert(mykeys, 64, K, cnt, worst, key);
    }
  }
  for (int s = 0; s < K; ++s) {
    int32_t oi = -1;
    float ol = VOGE_SENT_LEN, oa = VOGE_SENT_ACT, od = 0.0f;
    if (s < cnt) {
      const uint64_t key = mykeys[(size_t)s * 64];
      oi = (int32_t)(uint32_t)key;
      float A[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) A[i] = isg[9 * (size_t)oi + i];
      const float mx = mus[3 * (size_t)oi], my = mus[3 * (size_t)oi + 1], mz = mus[3 * (size_t)oi + 2];
      const EvalRec e = make_eval(mx, my, mz, A);
      const PairOut o = pair_eval(mx, my, mz, e, dx, dy, dz, qxx, qyy, qzz, qxy, qxz, qyz);
      ol = ord2f((uint32_t)(key >> 32));
      oa = o.act;
      od = o.dsd;
    }
    out_idx[pix * K + s] = oi;
    out_len[pix * K + s] = ol;
    out_act[pix * K + s] = oa;
    out_dsd[pix * K + s] = od;
  }
}

template <int WAVES>
static int launch_trace(const float4 *cull, const float4 *evr, const float *rays, int B, int N,
                        int H, int W, int K, float thr_act, int32_t *idx, float *len, float *act,
                        float *dsd, hipStream_t st) {
  constexpr int T = 64 * WAVES;
  constexpr int TW = (WAVES >= 2) ? 16 : 8;
  constexpr int TH = (WAVES == 4) ? 16 : 8;
  const size_t lds = sizeof(uint64_t) * (size_t)K * T + sizeof(TraceLds<2 * T>);
  auto kern = trace_fwd_kernel<WAVES>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid(((W + TW - 1) / TW) * ((H + TH - 1) / TH), B);
  hipLaunchKernelGGL(kern, grid, dim3(T), lds, st, cull, evr, rays, N, H, W, K, thr_act, idx, len, act, dsd);
  return launch_status();
}

}  // namespace voge

using namespace voge;

extern "C" size_t voge_trace_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return (size_t)B * N * 64;
}

extern "C" int voge_trace_topk_fwd(const float *mus, const float *isigmas, const float *rays,
                                   const float *cam_fwd, int B, int N, int H, int W, int K,
                                   float thr_act, void *workspace, size_t workspace_bytes,
                                   int32_t *idx, float *len, float *act, float *dsd,
                                   voge_stream_t stream) {
  if (B < 0 || N < 0 || H < 0 || W < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K) return VOGE_ERR_K_TOO_LARGE;
  if ((size_t)B * H * W == 0) return 0;  // numel == 0 early return (ray_trace_voge.cu:248-251)
  if (!rays || !idx || !len || !act || !dsd) return VOGE_ERR_BAD_ARG;
  if (N > 0 && (!mus || !isigmas || !workspace)) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < voge_trace_workspace_bytes(B, N)) return VOGE_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int P = B * N;
  float4 *cull = reinterpret_cast<float4 *>(workspace);
  float4 *evr = cull + P;
  if (P > 0) {
    hipLaunchKernelGGL(prep_kernel, dim3((P + 255) / 256), dim3(256), 0, st, mus, isigmas, cam_fwd, N, P,
                       thr_act, cull, evr);
    int rc = launch_status();
    if (rc) return rc;
  }
  // largest tile whose LDS footprint still lets two workgroups share a CU; else whatever fits
  auto fits = [&](int waves, size_t budget) {
    const size_t fixed = waves == 4 ? sizeof(TraceLds<512>) : waves == 2 ? sizeof(TraceLds<256>) : sizeof(TraceLds<128>);
    return sizeof(uint64_t) * (size_t)K * 64 * waves + fixed <= budget;
  };
  const size_t two_per_cu = 80 * 1024, one_per_cu = 160 * 1024;
  if (fits(4, two_per_cu)) return launch_trace<4>(cull, evr, rays, B, N, H, W, K, thr_act, idx, len, act, dsd, st);
  if (fits(2, two_per_cu)) return launch_trace<2>(cull, evr, rays, B, N, H, W, K, thr_act, idx, len, act, dsd, st);
  if (fits(1, two_per_cu)) return launch_trace<1>(cull, evr, rays, B, N, H, W, K, thr_act, idx, len, act, dsd, st);
  if (fits(1, one_per_cu)) return launch_trace<1>(cull, evr, rays, B, N, H, W, K, thr_act, idx, len, act, dsd, st);
  return VOGE_ERR_K_TOO_LARGE;
}

extern "C" int voge_trace_topk_list_fwd(const float *mus, const float *isigmas, const float *rays,
                                        const int32_t *bin_points, int B, int P, int H, int W, int K,
                                        int BH, int BW, int M, int bin_size, float thr_act,
                                        int32_t *idx, float *len, float *act, float *dsd,
                                        voge_stream_t stream) {
  if (B < 0 || P < 0 || H < 0 || W < 0 || K <= 0 || BH <= 0 || BW <= 0 || M < 0 || bin_size <= 0)
    return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K) return VOGE_ERR_K_TOO_LARGE;
  if ((size_t)B * H * W == 0) return 0;
  if (!rays || !idx || !len || !act || !dsd || (M > 0 && !bin_points)) return VOGE_ERR_BAD_ARG;
  if (P > 0 && (!mus || !isigmas)) return VOGE_ERR_BAD_ARG;
  const size_t lds = sizeof(uint64_t) * (size_t)K * 64;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(trace_list_fwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid(((W + 7) / 8) * ((H + 7) / 8), B);
  hipLaunchKernelGGL(trace_list_fwd_kernel, grid, dim3(64), lds, (hipStream_t)stream, mus, isigmas, rays,
                     bin_points, P, H, W, K, BH, BW, M, bin_size, thr_act, idx, len, act, dsd);
  return launch_status();
}
