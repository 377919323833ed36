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

__device__ __forceinline__ void prep_one(const int g, const float *__restrict__ mus, const float *__restrict__ isg,
                                         const float *__restrict__ cam_fwd, const int N, const float thr_act,
                                         const int iso_in, float4 *__restrict__ cull, float4 *__restrict__ evr,
                                         float4 *__restrict__ ms, float4 *__restrict__ ell, const IsoView view,
                                         float4 *__restrict__ pk = nullptr /* [P][3] packed (mu, A): kept by the caller */,
                                         const CamView cam = no_camera()) {
  float mx, my, mz;
  const int src = view.shared ? g % N : g;
  float fwd[3] = {0.f, 0.f, 0.f};
  bool has_fwd = false;
  if (cam.R != nullptr) {      // (round 6) centre and view axis from the camera, as rays_fwd_kernel / _view_axis make them
    const int b = g / N;
    const CamK ck = cam_load(cam, b);
    float ox, oy, oz;
    cam_origin(ck, cam.T + 3 * b, ox, oy, oz);
    mx = mus[3 * (size_t)src + 0] - ox; my = mus[3 * (size_t)src + 1] - oy; mz = mus[3 * (size_t)src + 2] - oz;
    if (cam.origin_out != nullptr && g == b * N) { cam.origin_out[3 * b] = ox; cam.origin_out[3 * b + 1] = oy; cam.origin_out[3 * b + 2] = oz; }
    if (cam.behind) { const float *r = cam.R + 9 * b; fwd[0] = r[2]; fwd[1] = r[5]; fwd[2] = r[8]; has_fwd = true; }
  } else if (view.origin != nullptr) {   // centring of Renderer.py:130 done here: the same single fp32 subtraction
    const float *o = view.origin + 3 * (g / N);
    mx = mus[3 * (size_t)src + 0] - o[0]; my = mus[3 * (size_t)src + 1] - o[1]; mz = mus[3 * (size_t)src + 2] - o[2];
  } else {
    mx = mus[3 * (size_t)src + 0]; my = mus[3 * (size_t)src + 1]; mz = mus[3 * (size_t)src + 2];
  }
  float A[9];
  if (iso_in) {   // isg holds one scalar per Gaussian: A = a I
    const float a = iso_view_a(isg[src], view.mode);
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = (i % 4 == 0) ? a : 0.0f;
  } else if (view.gen_kind == 1) {      // the user's per-axis sigmas: A = 2 diag(s) (general_preamble_fwd_kernel's operations)
    const float *sg = isg + 3 * (size_t)(view.sigma_shared ? g % N : g);
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = (i % 4 == 0) ? 2.0f * sg[i / 4] : 0.0f;
  } else if (view.gen_kind == 2) {      // the user's [3][3] sigmas: A = 2 S
    const float *sg = isg + 9 * (size_t)(view.sigma_shared ? g % N : g);
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = 2.0f * sg[i];
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = isg[9 * (size_t)g + i];
  }
  const EvalRec e = make_eval(mx, my, mz, A);
  if (pk != nullptr && view.gen_kind == 1) {      // per-axis sigmas: the compact (centred mu, a0, a1, a2) record, 32 bytes
    pk[2 * (size_t)g + 0] = make_float4(mx, my, mz, A[0]);
    pk[2 * (size_t)g + 1] = make_float4(A[4], A[8], 0.0f, 0.0f);
  } else if (pk != nullptr) {      // what the deferred composite and the fused backward gather (fragment_bwd.hip's record layout)
    pk[3 * (size_t)g + 0] = make_float4(mx, my, mz, A[0]);
    pk[3 * (size_t)g + 1] = make_float4(A[1], A[2], A[3], A[4]);
    pk[3 * (size_t)g + 2] = make_float4(A[5], A[6], A[7], A[8]);
  }

  const double lmin = lambda_min_sym3(A[0], A[4], A[8], 0.5 * ((double)A[1] + A[3]),
                                      0.5 * ((double)A[2] + A[6]), 0.5 * ((double)A[5] + A[7]));
  const double lmax_bound = fabs((double)A[0]) + fabs((double)A[4]) + fabs((double)A[8]) +
                            fabs((double)A[1] + A[3]) + fabs((double)A[2] + A[6]) + fabs((double)A[5] + A[7]);
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
    // Anisotropic Gaussians also get the ELLIPSOID every hit must touch: a ray with act < thr_act has
    // its peak point x = len d inside E = {x : (x-mu)^T S (x-mu) <= thr2}, S = sym(A) (same bound as
    // above, before lambda_min replaces S).  Record M = thr2 S^-1, so that the support function of E
    // is h(n) = sqrt(n^T M n); the bin kernels look for a plane that separates E from a ray cone.
    // The lowest mantissa bit of `reach` says whether the record exists (rounding reach UP is safe).
    uint32_t rb = __float_as_uint(reach);
    bool has_ell = false;
    if (!is_iso(e) && reach < 3e38f) {
      const double s00 = A[0], s11 = A[4], s22 = A[8], s01 = 0.5 * ((double)A[1] + A[3]),
                   s02 = 0.5 * ((double)A[2] + A[6]), s12 = 0.5 * ((double)A[5] + A[7]);
      const double c00 = s11 * s22 - s12 * s12, c01 = s02 * s12 - s01 * s22, c02 = s01 * s12 - s02 * s11;
      const double c11 = s00 * s22 - s02 * s02, c12 = s01 * s02 - s00 * s12, c22 = s00 * s11 - s01 * s01;
      const double det = s00 * c00 + s01 * c01 + s02 * c02;
      const double f = fmax(thr2, 0.0) * (1.0 + 1e-4) / det;
      const double m00 = c00 * f, m11 = c11 * f, m22 = c22 * f, m01 = c01 * f, m02 = c02 * f, m12 = c12 * f;
      const double msum = fabs(m00) + fabs(m11) + fabs(m22) + 2.0 * (fabs(m01) + fabs(m02) + fabs(m12));
      // a positive definite S has det > 0 and positive diagonal cofactors; anything else keeps the sphere only
      if (det > 0.0 && f > 0.0 && m00 > 0.0 && m11 > 0.0 && m22 > 0.0 && msum < 1e30) {
        ell[2 * (size_t)g + 0] = make_float4((float)m00, (float)m11, (float)m22, (float)m01);
        // .z: absolute slack for n^T M n evaluated in fp32 (|n| <= 1.001); .w: additive slack of h
        ell[2 * (size_t)g + 1] = make_float4((float)m02, (float)m12, (float)(8e-6 * msum) + 1e-30f,
                                             (float)(1e-5 * nm + 1e-5 * r) + 1e-30f);
        has_ell = true;
      }
    }
    rb = has_ell ? (rb | 1u) : ((rb + 1u) & ~1u);
    reach = __uint_as_float(rb);
  }
  if (cam_fwd != nullptr) {
    const float *f = cam_fwd + 3 * (g / N);
    if (fmaf(mz, f[2], fmaf(my, f[1], mx * f[0])) < 0.0f) reach = -1.0f;
  }
  if (has_fwd && fmaf(mz, fwd[2], fmaf(my, fwd[1], mx * fwd[0])) < 0.0f) reach = -1.0f;
  cull[g] = make_float4(mx, my, mz, reach);
  evr[3 * (size_t)g + 0] = make_float4(e.s00, e.s11, e.s22, e.s01);
  evr[3 * (size_t)g + 1] = make_float4(e.s02, e.s12, e.bx, e.by);
  evr[3 * (size_t)g + 2] = make_float4(e.bz, e.kx, e.ky, e.kz);
  // epilogue record: everything an isotropic Gaussian needs in one 16-byte gather; w = NaN sends
  // the reader to the full records
  ms[g] = make_float4(mx, my, mz, is_iso(e) ? e.s00 : __uint_as_float(0x7fc00000u));
}

__device__ __forceinline__ EvalRec unpack_eval(const float4 a, const float4 b, const float4 c) {
  EvalRec e;
  e.s00 = a.x; e.s11 = a.y; e.s22 = a.z; e.s01 = a.w;
  e.s02 = b.x; e.s12 = b.y; e.bx = b.z; e.by = b.w;
  e.bz = c.x; e.kx = c.y; e.ky = c.z; e.kz = c.w;
  return e;
}

// One thread per Gaussian: the general (3x3) entry point's records.  (The scalar-sigma entry points derive
// theirs inside binA.)
// Small sets (VOGE_SMALL_N Gaussians per batch element or fewer): no binA at all.  The prep pass marks every (super-tile,
// slice) segment as overflowed (count -1) and resets binB's pool counter, which is what binA does for a slice it could not
// hold -- binB then takes a quad's candidates straight from the per-Gaussian records (its tested fallback: N / 256 cone tests per
// thread and pass).  For a few thousand Gaussians that is cheaper than a launch whose workgroups are pure latency (binA: 8.4 us
// at cfg5's 2 562 Gaussians; profiles/r5_kernel_trace_cfg5_loop.txt).
__device__ __forceinline__ void small_set_marks(int *__restrict__ seg_count, const long n_seg, unsigned long long *__restrict__ pool_top) {
  if (seg_count == nullptr) return;
  const long t = ((long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
  const long nt = (long)gridDim.x * gridDim.y * blockDim.x;
  for (long i = t; i < n_seg; i += nt) seg_count[i] = -1;
  if (t == 0) *pool_top = 0ull;
}

__global__ void __launch_bounds__(256)
prep_kernel(const float *__restrict__ mus, const float *__restrict__ isg, const float *__restrict__ cam_fwd, const int N,
            const int P, const float thr_act, float4 *__restrict__ cull, float4 *__restrict__ evr,
            float4 *__restrict__ ms, float4 *__restrict__ ell, float4 *__restrict__ pk, const IsoView view, const CamView cam,
            int *__restrict__ seg_count = nullptr, const long n_seg = 0, unsigned long long *__restrict__ pool_top = nullptr) {
  small_set_marks(seg_count, n_seg, pool_top);
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < P) prep_one(g, mus, isg, cam_fwd, N, thr_act, 0, cull, evr, ms, ell, view, pk, cam);
}

}  // namespace voge

#include "trace_bin.h"
#include "composite_core.h"

#ifndef VOGE_SMALL_N
#define VOGE_SMALL_N 4096      // Gaussians per batch element up to which binA is skipped (small_set_marks)
#endif
#ifndef VOGE_ISO_PREP_SPLIT
#define VOGE_ISO_PREP_SPLIT 131072      // Gaussians per batch element from which on the scalar-sigma records get their own pass
#endif

namespace voge {

// The scalar-sigma records as a pass of their own: binA_kernel<true> derives them inside every (region, slice) workgroup -- one
// launch less, and free while a slice is one round (cfg3: 3 Gaussians per thread).  At 200k Gaussians every thread derives 12
// records -- as the same slice's workgroup of every other region does -- and the kernel is VALU-bound on it (5 of a workgroup's
// 15.8 us, VOGE_HIP_LIB=bt.so tools/bin_times.py cfg4_200k_1024): from VOGE_ISO_PREP_SPLIT Gaussians on this pass runs first and
// binA reads the records (cfg4: entry 297 -> 287 us, renderer form 272 -> 257).  Same functions, same bits.
__global__ void __launch_bounds__(256)
iso_prep_kernel(const float *__restrict__ mus, const float *__restrict__ isg, const float *__restrict__ cam_fwd, const int N,
                const float thr_act, const IsoView view, float4 *__restrict__ cull, float4 *__restrict__ ms, const CamView cam,
                int *__restrict__ seg_count = nullptr, const long n_seg = 0, unsigned long long *__restrict__ pool_top = nullptr) {
  small_set_marks(seg_count, n_seg, pool_top);
  const int g = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (g >= N) return;
  CamK ck;
  if (cam.R != nullptr) ck = cam_load(cam, b);      // (uniform)
  const BinAView V = binA_view<true>(b, cam_fwd, view, cam, &ck);
  if (cam.R != nullptr && cam.origin_out != nullptr && g == 0) {
    cam.origin_out[3 * b] = V.ox; cam.origin_out[3 * b + 1] = V.oy; cam.origin_out[3 * b + 2] = V.oz;
  }
  const BinARaw raw = binA_fetch<true>(g, b, N, nullptr, mus, isg, view);
  float a;
  const float4 c = binA_derive<true>(raw, true, V, thr_act, view, a);
  cull[(size_t)b * N + g] = c;
  ms[(size_t)b * N + g] = make_float4(c.x, c.y, c.z, a);
}

// ------------------------------------------------------------------------------------------
// sweep.  One workgroup = WAVES waves = a TW x TH pixel tile (each wave an 8x8 sub-tile, one
// ray per lane).  Its candidate stream is the sorted list of its super-tile (or, if that bin
// overflowed, every Gaussian of the batch element), read in chunks of T:
//   fill   : thread i gathers the cull record of stream entry base+i and tests it against the
//            workgroup's bounding cone; survivors are compacted IN ORDER into LDS together with
//            their eval record and len bound;
//   consume: each wave re-tests the survivors against its own 8x8 cone, 64 at a time (one per
//            lane, ballot), then for every remaining candidate all 64 lanes evaluate their ray
//            against it (record broadcast from LDS) and insert into their LDS top-K list.
//   exit   : once every lane of a wave holds K hits and the next candidate's len bound exceeds
//            the wave's largest kept len, nothing later in the (sorted) stream can enter.
// Both culls and the exit test are conservative, so the result equals the brute-force sweep.
// The epilogue re-maps lanes to (pixel, slot) so that all four outputs are written as
// contiguous runs of TW*K floats.
// ------------------------------------------------------------------------------------------
#ifdef VOGE_SWEEP_STATS
#define VOGE_SWEEP_TIMES 1
#endif
#ifdef VOGE_SWEEP_TIMES
__device__ unsigned long long g_sweep_stats[16];
__device__ unsigned long long g_sweep_times[8192 * 8];   // per WG: start, after cones, fill sum, consume sum, loop end, end, evals, smid
#endif
#ifndef VOGE_TRIP
#define VOGE_TRIP 4
#endif
constexpr int kTrip = VOGE_TRIP;   // candidates evaluated per trip of the sweep's inner loop

}  // namespace voge
#include "sweep_iso.h"      // round 4's scalar-sigma sweep (sweep_iso_kernel)
namespace voge {

template <int T, bool ISO>
struct TraceLds {
  // layout inside dynamic LDS, after the [K][T+1] key array.  ISO (the scalar-sigma entry point: every
  // candidate is isotropic) stages no full records; with the K = 40 key array that is 22.9 KB per
  // single-wave workgroup -- seven of them per CU instead of six.
  float4 cull[T > 64 ? T : 1];   // (mu, reach): the per-wave re-test of multi-wave tiles only
  float4 ms[T];         // (mu, s00 | NaN): all an isotropic evaluation needs
  float4 ev[ISO ? 1 : T * 3];    // full eval record, staged for anisotropic candidates only
  int32_t id[T];        // candidate ids of the staged chunk; per-ray hit counts during the epilogue
  float lb[T];
  float red[(T / 64) * 8];
  int wcnt[2][4];
  int done;
};

template <int WAVES, bool ISO>
__global__ void __launch_bounds__(64 * WAVES)
trace_fwd_kernel(const float4 *__restrict__ cull, const float4 *__restrict__ evr,
                 const float4 *__restrict__ ms, const float *__restrict__ rays, const int *__restrict__ bin_count,
                 const int32_t *__restrict__ bin_id, const float *__restrict__ bin_lb,
                 const int *__restrict__ tl_count, const int32_t *__restrict__ tl_id,
                 const float *__restrict__ tl_lb, const int32_t *__restrict__ pool_id, const float *__restrict__ pool_lb,
                 const int *__restrict__ tl_off, const int2 *__restrict__ order, const int tiles_per_img,
                 const int nstx, const int nst,
                 const int N, const int H,
                 const int W, const int K, const float thr_act, int32_t *__restrict__ out_idx,
                 float *__restrict__ out_len, float *__restrict__ out_act, float *__restrict__ out_dsd,
                 int32_t *__restrict__ out_cnt, const float occ, float *__restrict__ out_weight,
                 int64_t *__restrict__ out_valid) {
  constexpr int T = 64 * WAVES;
  constexpr int TP = T + 1;   // key row stride: the transposed epilogue read stays conflict-light
  constexpr int TW = (WAVES >= 2) ? 16 : 8;
  constexpr int TH = (WAVES == 4) ? 16 : 8;
  constexpr int kCap = T;     // one chunk of the candidate stream is staged at a time
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t *keys = reinterpret_cast<uint64_t *>(smem_raw);
  TraceLds<T, ISO> &L = *reinterpret_cast<TraceLds<T, ISO> *>(smem_raw + ((sizeof(uint64_t) * (size_t)(K + 1) * TP + 15) & ~(size_t)15));

#ifdef VOGE_SWEEP_TIMES
  const unsigned long long ts0 = wall_clock64();
  unsigned long long ts_fill = 0, ts_cons = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + TW - 1) / TW;
  // Heavy tiles first: workgroup i takes slot i % 16 of the super-tile with launch rank i / 16 (binB: super-tiles by
  // descending candidate count, inside them quad by quad, a quad's tiles by descending list length).  The sweep lasts as long as its longest
  // tile, so that one must not start late; everything shorter fills in behind it.
  const int2 slot = order[blockIdx.x];               // (tile [| kPoolFlag], length of its list | -1 = overflowed)
  if (slot.x < 0 || slot.y == 0) return;             // outside the image | nothing can hit it: binB wrote its outputs
  const bool pooled = (slot.x & kPoolFlag) != 0;     // a long list (binB's long path): it lives in the pool
  const int lin = slot.x & ~kPoolFlag;
  const int b = lin / tiles_per_img, bx = lin - b * tiles_per_img;
  const int tx = bx % tiles_x, ty = bx / tiles_x;
  const int px = tx * TW + (wave & 1) * 8 * (TW == 16) + (lane & 7);
  const int py = ty * TH + (wave >> 1) * 8 + (lane >> 3);
  const bool valid = (px < W) && (py < H);
  const int cpx = min(px, W - 1), cpy = min(py, H - 1);
  const size_t ray_id = ((size_t)b * H + cpy) * W + cpx;
  const float dx = rays[3 * ray_id + 0], dy = rays[3 * ray_id + 1], dz = rays[3 * ray_id + 2];
  const float qxx = dx * dx, qyy = dy * dy, qzz = dz * dz, qxy = dx * dy, qxz = dx * dz, qyz = dy * dz;
  if (tid == 0) L.done = 0;

  // ---- bounding cones (wave, then workgroup) ---------------------------------------------
  const RayDir u = ray_dir(dx, dy, dz);
  const bool wave_dirs_ok = __all(u.ok);
  const bool unit_rays = __all(!u.ok || u.unit);
  const float wsx = wave_sum_dpp(u.ok ? u.ux : 0.f), wsy = wave_sum_dpp(u.ok ? u.uy : 0.f), wsz = wave_sum_dpp(u.ok ? u.uz : 0.f);
  Cone wcone, gcone;
  {
    const float n = sqrtf(fmaf(wsz, wsz, fmaf(wsy, wsy, wsx * wsx)));
    const float ax = wsx / n, ay = wsy / n, az = wsz / n;
    float smax = 0.f, cmin = 1.f;
    cone_partial(u, ax, ay, az, smax, cmin);
    wcone = cone_finish(ax, ay, az, n, wave_max(smax), wave_min(cmin), wave_dirs_ok);
    gcone = wcone;
  }
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
    const float n = sqrtf(fmaf(gz, gz, fmaf(gy, gy, gx * gx)));
    const float ax = gx / n, ay = gy / n, az = gz / n;
    float smax = 0.f, cmin = 1.f;
    cone_partial(u, ax, ay, az, smax, cmin);
    smax = wave_max(smax); cmin = wave_min(cmin);
    if (lane == 0) { L.red[wave * 8 + 4] = smax; L.red[wave * 8 + 5] = cmin; }
    __syncthreads();
    for (int w = 0; w < WAVES; ++w) { smax = fmaxf(smax, L.red[w * 8 + 4]); cmin = fminf(cmin, L.red[w * 8 + 5]); }
    gcone = cone_finish(ax, ay, az, n, smax, cmin, gok);
  } else {
    __syncthreads();
  }

#ifdef VOGE_SWEEP_TIMES
  const unsigned long long ts1 = wall_clock64();
#endif
  // ---- candidate stream of this tile -----------------------------------------------------
  // tile list (bin2) -> super-tile list (bin) -> every Gaussian of the batch element
  const int tile = lin;
  // (fallback of an overflowed tile list: the ordered list of the tile's 16x16-pixel quad, which binB then spilled)
  const int bin = (b * nst + ((ty * TH) / kST) * nstx + (tx * TW) / kST) * 4 + (((ty * TH) / kQuad) & 1) * 2 + (((tx * TW) / kQuad) & 1);
  const int tc = slot.y;
  const int bc = (tc >= 0) ? tc : ((bin_count != nullptr) ? bin_count[bin] : -1);
  const bool binned = bc >= 0;
  const int src_n = binned ? bc : N;
  const size_t list_at = pooled ? (size_t)tl_off[tile] : (size_t)tile * kTileCap;
  const int32_t *src_id = (tc >= 0) ? (pooled ? pool_id : tl_id) + list_at : (binned ? bin_id + (size_t)bin * kQCap : nullptr);
  const float *src_lb = (tc >= 0) ? (pooled ? pool_lb : tl_lb) + list_at : (binned ? bin_lb + (size_t)bin * kQCap : nullptr);
  const float4 *cullb = cull + (size_t)b * N;
  const float4 *evrb = evr + (size_t)b * N * 3;
  const float4 cull_none = make_float4(0.f, 0.f, 0.f, -1.f);
  auto load_id = [&](int g) { return (g < src_n) ? (binned ? src_id[g] : g) : -1; };
  auto load_lb = [&](int g) { return (binned && g < src_n) ? src_lb[g] : -INFINITY; };
  auto load_rec = [&](int id) { return (id >= 0) ? cullb[id] : cull_none; };
  const float4 *msb = ms + (size_t)b * N;
  auto load_ms = [&](int id) { return (id >= 0) ? msb[id] : cull_none; };
  // the tile's own list was already filtered with this tile's cone (bin2): no second test
  const bool prefiltered = (WAVES == 1) && (tc >= 0);

  uint64_t *mykeys = keys + tid;
  int cnt = 0;
  uint64_t worst = valid ? ((uint64_t)f2ord(VOGE_SENT_LEN) << 32) : 0ull, tail = 0ull;
  bool wdone = false, reported = false;
#ifdef VOGE_SWEEP_STATS
  unsigned st_staged = 0, st_eval = 0, st_trips = 0, st_slow = 0, st_shift = 0, st_hits = 0, st_batches = 0;
#endif

  int base = 0, par = 0;
  // two-deep software pipeline: ids two chunks ahead, cull / ms records one chunk ahead
  int id0 = load_id(tid);
  float lb0 = load_lb(tid);
  float4 c0r = prefiltered ? cull_none : load_rec(id0);
  float4 m0r = load_ms(id0);
  int id1 = load_id(T + tid);
  float lb1 = load_lb(T + tid);
  bool tile_gen = false;      // an anisotropic candidate was staged at some point (workgroup-uniform)
  while (base < src_n) {
    int nbuf = 0;
    bool chunk_iso = true;   // every staged candidate of this buffer is isotropic (wave-uniform)
    bool chunk_gen = !ISO;   // ... or every one is anisotropic
#ifdef VOGE_SWEEP_TIMES
    const unsigned long long tsa = wall_clock64();
#endif
    while (base < src_n && nbuf + T <= kCap) {
      const int id = id0;
      const float lbv = lb0;
      const float4 c = c0r;
      const float4 mrec = m0r;
      id0 = id1; lb0 = lb1;
      c0r = prefiltered ? cull_none : load_rec(id0);
      m0r = load_ms(id0);
      id1 = load_id(base + 2 * T + tid);
      lb1 = load_lb(base + 2 * T + tid);
      const bool keep = prefiltered ? (id >= 0) : cone_keep(c, gcone);
      const unsigned long long m = __ballot(keep);
      if (!ISO) {
        chunk_iso = chunk_iso && __all(!keep || (mrec.w == mrec.w));
        chunk_gen = chunk_gen && __all(!keep || !(mrec.w == mrec.w));
      }
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
        if (WAVES > 1) L.cull[slot] = c;
        L.ms[slot] = mrec;
        L.id[slot] = id;
        L.lb[slot] = lbv;
        if (!ISO && !(mrec.w == mrec.w)) {   // anisotropic: the full record (dependent gather, not prefetched)
          L.ev[slot * 3 + 0] = evrb[(size_t)id * 3 + 0];
          L.ev[slot * 3 + 1] = evrb[(size_t)id * 3 + 1];
          L.ev[slot * 3 + 2] = evrb[(size_t)id * 3 + 2];
        }
      }
      nbuf += tot;
      base += T;
      par ^= 1;
    }
#ifdef VOGE_SWEEP_STATS
    st_staged += nbuf;
#endif
    tile_gen = tile_gen || !chunk_iso;
    __syncthreads();
#ifdef VOGE_SWEEP_TIMES
    const unsigned long long tsb = wall_clock64();
    ts_fill += tsb - tsa;
#endif
    // consume
    if (!wdone) {
      for (int c0 = 0; c0 < nbuf && !wdone; c0 += 64) {
        const int i = c0 + lane;
        bool keep = false;
        if (i < nbuf) keep = (WAVES == 1) ? true : cone_keep(L.cull[i], wcone);
        unsigned long long m = __ballot(keep);
        // Exit test, once per 64-candidate batch.  The bound is refreshed here only: a stale
        // (larger) bound merely delays the exit, because a lane's worst key only ever decreases.
        // The list bounds are monotone, so "first candidate past the bound" cuts the batch.
        bool last_batch = false;
        if (binned && unit_rays && __all(!valid || cnt == K)) {
          const float wmax = wave_max(valid ? ord2f((uint32_t)(worst >> 32)) : -INFINITY);
          const unsigned long long ex = __ballot(i < nbuf && L.lb[i] > wmax);
          if (ex) {
            m &= (1ull << __builtin_ctzll(ex)) - 1ull;
            last_batch = true;
          }
        }
        // (gid = the candidate's global id, read from L.id by the caller: the fast loops fetch the ids of a
        // trip together with its records, so no LDS latency sits between two commits)
        auto commit = [&](const PairOut &o, const int gid, const bool on) {
          const uint64_t key = ((uint64_t)f2ord(o.len) << 32) | (uint32_t)gid;
          // (rays outside the image start with worst = 0, the others with the key of len = 1e10: `key < worst`
          // also says "a ray of the image" and "len below the sentinel")
          const bool take = on & (o.act < thr_act) & (key < worst);
#ifdef VOGE_SWEEP_STATS
          {
            const bool app = take && (cnt < K) && (key >= tail);
            const bool slow = take && !app;
            st_hits += __popcll(__ballot(take));
            if (__any(slow)) {
              ++st_slow;
              int steps = 0;
              if (slow) { int pos = min(cnt, K - 1); while (pos > 0 && mykeys[(pos - 1) * TP] > key) { --pos; ++steps; } }
              st_shift += (unsigned)wave_max((float)steps);
            }
          }
#endif
#ifndef VOGE_ISO_ROWS4
#define VOGE_ISO_ROWS4 0
#endif
          topk_commit<(!ISO) || (VOGE_ISO_ROWS4 != 0)>(mykeys, TP, K, cnt, worst, tail, key, take);
        };
#ifdef VOGE_SWEEP_STATS
        st_eval += __popcll(m); ++st_batches;
#endif
        if (WAVES == 1 && chunk_iso) {
          // Single-wave tile, all-isotropic chunk (the common case): the surviving candidates are
          // the contiguous range [c0, c0 + n) -- no bit scanning, no per-candidate isotropy test.
          const int s_end = c0 + __popcll(m);
          for (int s0 = c0; s0 < s_end; s0 += kTrip) {
#ifdef VOGE_SWEEP_STATS
            ++st_trips;
#endif
            float4 cc[kTrip];
            int gid[kTrip];
            PairOut o[kTrip];
#pragma unroll
            for (int q = 0; q < kTrip; ++q) {
              cc[q] = L.ms[min(s0 + q, s_end - 1)];
              gid[q] = L.id[min(s0 + q, s_end - 1)] + b * N;
            }
#pragma unroll
            for (int q = 0; q < kTrip; ++q)
              o[q] = pair_eval_iso(cc[q].x, cc[q].y, cc[q].z, cc[q].w, dx, dy, dz, qxx, qyy, qzz);
            // The evaluations must finish as one block of four interleaved chains: without this
            // the compiler sinks each one behind its own commit's predicate and the wave (alone on
            // its SIMD) runs four dependent chains back to back.
#pragma unroll
            for (int q = 0; q < kTrip; ++q) asm volatile("" : "+v"(o[q].len), "+v"(o[q].act));
#pragma unroll
            for (int q = 0; q < kTrip; ++q) commit(o[q], gid[q], s0 + q < s_end);
          }
          m = 0ull;
        } else if (!ISO && WAVES == 1 && chunk_gen) {
          // the same contiguous-range loop for an all-anisotropic chunk (full records from LDS)
          const int s_end = c0 + __popcll(m);
          for (int s0 = c0; s0 < s_end; s0 += kTrip) {
#ifdef VOGE_SWEEP_STATS
            ++st_trips;
#endif
            PairOut o[kTrip];
            int gid[kTrip];
#pragma unroll
            for (int q = 0; q < kTrip; ++q) {
              const int sidx = min(s0 + q, s_end - 1);
              const float4 cc = L.ms[sidx];
              gid[q] = L.id[sidx] + b * N;
              o[q] = pair_eval_gen(cc.x, cc.y, cc.z, unpack_eval(L.ev[sidx * 3], L.ev[sidx * 3 + 1], L.ev[sidx * 3 + 2]), dx, dy,
                                   dz, qxx, qyy, qzz, qxy, qxz, qyz);
            }
#pragma unroll
            for (int q = 0; q < kTrip; ++q) asm volatile("" : "+v"(o[q].len), "+v"(o[q].act));
#pragma unroll
            for (int q = 0; q < kTrip; ++q) commit(o[q], gid[q], s0 + q < s_end);
          }
          m = 0ull;
        }
        while (m) {
#ifdef VOGE_SWEEP_STATS
          ++st_trips;
#endif
          // four candidates per trip: their evaluations are independent instruction streams
          int sq[kTrip];
          int nt = 0;
#pragma unroll
          for (int q = 0; q < kTrip; ++q) {
            sq[q] = c0 + (m ? __builtin_ctzll(m) : 0);
            if (m) { ++nt; m &= m - 1ull; }
          }
          // The four evaluations form ONE straight-line block (the isotropic / general choice is
          // made per batch, on scalar registers), so the scheduler interleaves their chains.
          PairOut o[kTrip];
          float4 cc[kTrip], e0[kTrip];
          int gid[kTrip];
          bool iso = true, any_iso = false;
          bool fiso[kTrip];
#pragma unroll
          for (int q = 0; q < kTrip; ++q) {
            cc[q] = L.ms[sq[q]];
            gid[q] = L.id[sq[q]] + b * N;
            const bool f = fiso[q] = (__builtin_amdgcn_readfirstlane(__float_as_uint(cc[q].w)) & 0x7fffffffu) <= 0x7f800000u;
            iso = iso && f;
            any_iso = any_iso || f;
          }
          if (ISO || iso) {
#pragma unroll
            for (int q = 0; q < kTrip; ++q)
              o[q] = pair_eval_iso(cc[q].x, cc[q].y, cc[q].z, cc[q].w, dx, dy, dz, qxx, qyy, qzz);
          } else {
            float4 e1[kTrip], e2[kTrip];
#pragma unroll
            for (int q = 0; q < kTrip; ++q) {
              e0[q] = L.ev[sq[q] * 3]; e1[q] = L.ev[sq[q] * 3 + 1]; e2[q] = L.ev[sq[q] * 3 + 2];
            }
            if (!any_iso) {
#pragma unroll
              for (int q = 0; q < kTrip; ++q)
                o[q] = pair_eval_gen(cc[q].x, cc[q].y, cc[q].z, unpack_eval(e0[q], e1[q], e2[q]), dx, dy, dz, qxx, qyy,
                                     qzz, qxy, qxz, qyz);
            } else {  // mixed batch: per-candidate dispatch (same arithmetic, just not interleaved)
#pragma unroll
              for (int q = 0; q < kTrip; ++q) {
                if (fiso[q])   // uniform: the flag came through readfirstlane
                  o[q] = pair_eval_iso(cc[q].x, cc[q].y, cc[q].z, cc[q].w, dx, dy, dz, qxx, qyy, qzz);
                else
                  o[q] = pair_eval_gen(cc[q].x, cc[q].y, cc[q].z, unpack_eval(e0[q], e1[q], e2[q]), dx, dy, dz, qxx, qyy,
                                       qzz, qxy, qxz, qyz);
              }
            }
          }
#pragma unroll
          for (int q = 0; q < kTrip; ++q) commit(o[q], gid[q], q < nt);
        }
        if (last_batch) wdone = true;
      }
    }
#ifdef VOGE_SWEEP_TIMES
    ts_cons += wall_clock64() - tsb;
#endif
    if (wdone && !reported) {
      reported = true;
      if (lane == 0) atomicAdd(&L.done, 1);
    }
    __syncthreads();
    if (L.done == WAVES) break;
  }

#ifdef VOGE_SWEEP_TIMES
  const unsigned long long ts2 = wall_clock64();
#endif
#ifdef VOGE_SWEEP_STATS
  if (lane == 0) {
    atomicAdd(&g_sweep_stats[0], 1ull);
    atomicAdd(&g_sweep_stats[1], (unsigned long long)(wave == 0 ? st_staged : 0));
    atomicAdd(&g_sweep_stats[2], (unsigned long long)st_eval);
    atomicAdd(&g_sweep_stats[3], (unsigned long long)st_trips);
    atomicAdd(&g_sweep_stats[4], (unsigned long long)st_slow);
    atomicAdd(&g_sweep_stats[5], (unsigned long long)st_shift);
    atomicAdd(&g_sweep_stats[6], (unsigned long long)st_hits);
    atomicAdd(&g_sweep_stats[7], (unsigned long long)(wave == 0 ? src_n : 0));
    atomicAdd(&g_sweep_stats[8], (unsigned long long)(wave == 0 ? min(base, src_n) : 0));
    atomicAdd(&g_sweep_stats[9], (unsigned long long)st_batches);
  }
#endif
  // ---- epilogue: lanes re-mapped to (pixel, slot); act / dsd recomputed with pair_eval ------
  __syncthreads();            // every wave is done with the staged ids: the array now holds the hit counts
  L.id[tid] = cnt;
#ifdef VOGE_SWEEP_TIMES
  const int cnt_dbg = (int)__popcll(__ballot(cnt > 0));   // rays of the tile with at least one hit
#endif
  if (out_cnt != nullptr && valid) out_cnt[((size_t)b * H + py) * W + px] = cnt;
  __syncthreads();
  const int tw = min(TW, W - tx * TW);
  const int row_items = tw * K;
  auto slot_value = [&](const int r, const int x, const int s, const size_t pix, int32_t &oi, float &ol, float &oa,
                        float &od) {
    const int owner = ((x >> 3) + (TW / 8) * (r >> 3)) * 64 + (x & 7) + 8 * (r & 7);
    oi = -1; ol = VOGE_SENT_LEN; oa = VOGE_SENT_ACT; od = 0.0f;
    if (s < L.id[owner]) {
      const uint64_t key = keys[(size_t)s * TP + owner];
      oi = (int32_t)(uint32_t)key;
      const float *ry = rays + pix * 3;
      const float ex = ry[0], ey = ry[1], ez = ry[2];
      const float4 cc = ms[oi];      // (centre, a | NaN): an isotropic Gaussian needs nothing else
      PairOut o;
      if (ISO || cc.w == cc.w) {
        o = pair_eval_iso(cc.x, cc.y, cc.z, cc.w, ex, ey, ez, ex * ex, ey * ey, ez * ez);
      } else {
        const EvalRec e = unpack_eval(evr[(size_t)oi * 3 + 0], evr[(size_t)oi * 3 + 1], evr[(size_t)oi * 3 + 2]);
        o = pair_eval(cc.x, cc.y, cc.z, e, ex, ey, ez, ex * ex, ey * ey, ez * ez, ex * ey, ex * ez, ey * ez);
      }
      ol = ord2f((uint32_t)(key >> 32));
      oa = o.act;
      od = o.dsd;
    }
  };
  const bool vec4 = ((K & 3) == 0);   // rows of K floats stay 16-byte aligned: 16-byte stores
  if (WAVES == 1 && out_weight != nullptr) {
    // ---- fused epilogue: fragments AND their composite weights (VoGE/Aggregation.py:82-107).  The depth-ordered
    // list of every ray is in LDS right now; instead of writing (idx, len, act, dsd) and letting a second kernel read
    // them back (126 MB at cfg3) the wave composites here: a lane owns four consecutive slots of a pixel, 64 / (K/4)
    // pixels per round, the row pass is the stand-alone kernel's (composite_core.h: bit-identical weights).  The
    // keys, the (mu, a) gathers and the ray of round r + 1 are requested before round r is computed.  (host: K % 4 == 0)
    constexpr int NS = 4;
    const int LP = K >> 2, pw = 64 / LP;
    const int rows = compn_rows(K, NS, 64, true);
    float *const Llen = reinterpret_cast<float *>(smem_raw + ((sizeof(uint64_t) * (size_t)(K + 1) * TP + 15) & ~(size_t)15) + ((sizeof(TraceLds<T, ISO>) + 15) & ~(size_t)15));
    float *const Lsp = Llen + rows, *const LE = Lsp + rows;
    const int RS = compn_stride(K, NS);
    const int pl = __float2int_rz(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)LP)), q = lane - pl * LP;
    const bool in_wg = pl < pw;
    const int k0 = NS * q, seg_lo = lane - q;
    const int d0 = (in_wg ? pl : 0) * RS + 2 + (in_wg ? k0 : 0);
    if (in_wg && q < 2) {      // the sentinel pairs in front of and behind every pixel's row: written once
      const int r0 = pl * RS;
      for (int t2 = q; t2 < 2; t2 += LP) {
        Llen[r0 + t2] = -kBig; Lsp[r0 + t2] = 1.0f; LE[r0 + t2] = 0.0f;
        const int eb = r0 + RS - 2 + t2;
        Llen[eb] = kBig; Lsp[eb] = 1.0f; LE[eb] = 0.0f;
      }
    }
    const int th = min(TH, H - ty * TH);
    struct Req {
      uint64_t key[NS];
      float4 rec[NS];
      float ex, ey, ez;
      int nv, cntp;
      bool on;
      size_t pix;
    };
    auto request = [&](const int round, Req &r) {
      const int pixl = round * pw + pl;                  // pixel of the tile: x = pixl & 7, row = pixl >> 3
      r.on = in_wg && pixl < 64 && (pixl & 7) < tw && (pixl >> 3) < th;
      r.pix = ((size_t)b * H + ty * TH + (pixl >> 3)) * W + (size_t)tx * TW + (pixl & 7);
      r.cntp = r.on ? L.id[pixl & 63] : 0;
      r.nv = max(0, min(NS, r.cntp - k0));
      r.ex = r.ey = r.ez = 0.0f;
#pragma unroll
      for (int a = 0; a < NS; ++a) r.key[a] = (a < r.nv) ? keys[(size_t)(k0 + a) * TP + (pixl & 63)] : 0ull;
      if (r.nv > 0) { r.ex = rays[r.pix * 3]; r.ey = rays[r.pix * 3 + 1]; r.ez = rays[r.pix * 3 + 2]; }
#pragma unroll
      for (int a = 0; a < NS; ++a) r.rec[a] = (a < r.nv) ? ms[(uint32_t)r.key[a]] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const int nround = (64 + pw - 1) / pw;
    Req cur, nxt;
    request(0, cur);
    for (int round = 0; round < nround; ++round) {
      if (round + 1 < nround) request(round + 1, nxt);
      int32_t oi[NS];
      float ol[NS], oa[NS], od[NS], lm[NS], sm[NS], em[NS];
      const float qxx = cur.ex * cur.ex, qyy = cur.ey * cur.ey, qzz = cur.ez * cur.ez;
      bool gen_any = false;
      if (!ISO) {
#pragma unroll
        for (int a = 0; a < NS; ++a) gen_any = gen_any || ((a < cur.nv) && !(cur.rec[a].w == cur.rec[a].w));
        gen_any = __any(gen_any);
      }
      float4 g0[NS], g1[NS], g2[NS];
      if (!ISO && gen_any) {      // anisotropic entries (w = NaN): their full records, all in flight together
#pragma unroll
        for (int a = 0; a < NS; ++a) {
          g0[a] = g1[a] = g2[a] = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((a < cur.nv) && !(cur.rec[a].w == cur.rec[a].w)) {
            const size_t eo = (size_t)(uint32_t)cur.key[a] * 3;
            g0[a] = evr[eo]; g1[a] = evr[eo + 1]; g2[a] = evr[eo + 2];
          }
        }
      }
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        oi[a] = -1; ol[a] = VOGE_SENT_LEN; oa[a] = VOGE_SENT_ACT; od[a] = 0.0f;
        lm[a] = VOGE_SENT_LEN; sm[a] = 1e-5f; em[a] = 0.0f;      // what the stand-alone kernel takes an empty slot for
        if (a < cur.nv) {
          oi[a] = (int32_t)(uint32_t)cur.key[a];
          ol[a] = ord2f((uint32_t)(cur.key[a] >> 32));
          PairOut o;
          if (ISO || cur.rec[a].w == cur.rec[a].w) {
            o = pair_eval_iso_at(cur.rec[a].x, cur.rec[a].y, cur.rec[a].z, cur.rec[a].w, ol[a], cur.ex, cur.ey, cur.ez,
                                 (qxx + qyy) + qzz);     // len is in the key: no second division
          } else {
            const EvalRec e = unpack_eval(g0[a], g1[a], g2[a]);
            o = pair_eval(cur.rec[a].x, cur.rec[a].y, cur.rec[a].z, e, cur.ex, cur.ey, cur.ez, qxx, qyy, qzz, cur.ex * cur.ey,
                          cur.ex * cur.ez, cur.ey * cur.ez);
          }
          oa[a] = o.act; od[a] = o.dsd;
          lm[a] = ol[a]; em[a] = FAST_EXP(-oa[a]); sm[a] = FAST_SQRT(od[a] + 1e-10f);
        }
      }
      if (in_wg) {
#pragma unroll
        for (int h2 = 0; h2 < NS / 2; ++h2) {
          const int a = 2 * h2;
          *reinterpret_cast<v2f *>(Llen + d0 + a) = (v2f){lm[a], lm[a + 1]};
          *reinterpret_cast<v2f *>(Lsp + d0 + a) = (v2f){sm[a] * kCs, sm[a + 1] * kCs};
          *reinterpret_cast<v2f *>(LE + d0 + a) = (v2f){em[a], em[a + 1]};
        }
      }
      wave_lds_sync();
      float wgt[NS];
      compn_fwd_rows<NS>(lm, sm, em, Llen, Lsp, LE, d0, k0, K, q, LP, LP, in_wg, cur.on, true, seg_lo, occ, wgt);
      wave_lds_sync();      // the rows are rewritten by the next round
      if (cur.on) {
        const size_t ob = cur.pix * K + k0;
        st16i<(VOGE_NT_STORES & 2) != 0>(out_idx + ob, oi[0], oi[1], oi[2], oi[3]);      // (write-once, 16 B per slot: non-temporal, voge_common.h)
        st16f<(VOGE_NT_STORES & 2) != 0>(out_len + ob, ol[0], ol[1], ol[2], ol[3]);
        st16f<(VOGE_NT_STORES & 2) != 0>(out_act + ob, oa[0], oa[1], oa[2], oa[3]);
        st16f<(VOGE_NT_STORES & 2) != 0>(out_dsd + ob, od[0], od[1], od[2], od[3]);
        *reinterpret_cast<float4 *>(out_weight + ob) = make_float4(wgt[0], wgt[1], wgt[2], wgt[3]);
        if (q == 0 && out_valid != nullptr) out_valid[cur.pix] = (int64_t)cur.cntp;
      }
      cur = nxt;
    }
  } else if (vec4) {
    // All rows of the tile as one item space; an item = 4 consecutive slots of one pixel.  kEpiU
    // items per thread go through the stages together -- LDS keys, then one 16-byte gather per
    // slot (isotropic Gaussians need nothing more), then arithmetic and the 16-byte stores -- so
    // a thread has up to 4 * kEpiU gathers in flight instead of one dependent chain per slot.
#ifndef VOGE_EPI_U
#define VOGE_EPI_U 4
#endif
    constexpr int kEpiU = VOGE_EPI_U;
    const int th = min(TH, H - ty * TH);
    const int ipr = row_items >> 2;
    const int nitem = th * ipr;
    const float inv_ipr = 1.0f / (float)ipr, invK = 1.0f / (float)K;
    // Tiles that staged anisotropic candidates: centre and full record (4 gathers per slot) are issued
    // together for kEpiG items -- one round trip per round instead of "centre, then 3 more per slot".
    // (pair_eval dispatches on the record, so an isotropic entry in such a tile is still exact.)
    constexpr int kEpiG = 2;
    const bool want_ad = out_act != nullptr;
    for (int it0 = tid; tile_gen && want_ad && it0 < nitem; it0 += T * kEpiG) {
      uint64_t key[kEpiG][4];
      float4 rc[kEpiG][4], g0[kEpiG][4], g1[kEpiG][4], g2[kEpiG][4];
      float ex[kEpiG], ey[kEpiG], ez[kEpiG];
      size_t ob[kEpiG];
      int nv[kEpiG];
#pragma unroll
      for (int u = 0; u < kEpiG; ++u) {
        const int it = it0 + u * T;
        nv[u] = -1;
        ob[u] = 0;
        ex[u] = ey[u] = ez[u] = 0.0f;
        if (it < nitem) {
          const int r = __float2int_rz(((float)it + 0.5f) * inv_ipr);
          const int j = (it - r * ipr) * 4;
          const int x = __float2int_rz(((float)j + 0.5f) * invK);
          const int sl = j - x * K;
          const int owner = ((x >> 3) + (TW / 8) * (r >> 3)) * 64 + (x & 7) + 8 * (r & 7);
          const size_t pix = ((size_t)b * H + ty * TH + r) * W + (size_t)tx * TW + x;
          ob[u] = pix * K + sl;
          nv[u] = max(0, min(4, L.id[owner] - sl));
#pragma unroll
          for (int q = 0; q < 4; ++q) key[u][q] = (q < nv[u]) ? keys[(size_t)(sl + q) * TP + owner] : 0ull;
          if (nv[u] > 0) { ex[u] = rays[pix * 3]; ey[u] = rays[pix * 3 + 1]; ez[u] = rays[pix * 3 + 2]; }
        }
      }
#pragma unroll
      for (int u = 0; u < kEpiG; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          rc[u][q] = g0[u][q] = g1[u][q] = g2[u][q] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (q < nv[u]) {
            const size_t gi = (uint32_t)key[u][q];
            rc[u][q] = ms[gi]; g0[u][q] = evr[gi * 3]; g1[u][q] = evr[gi * 3 + 1]; g2[u][q] = evr[gi * 3 + 2];
          }
        }
#pragma unroll
      for (int u = 0; u < kEpiG; ++u) {
        if (nv[u] < 0) continue;
        int32_t oi[4];
        float ol[4], oa[4], od[4];
        const float qxx = ex[u] * ex[u], qyy = ey[u] * ey[u], qzz = ez[u] * ez[u];
        const float qxy = ex[u] * ey[u], qxz = ex[u] * ez[u], qyz = ey[u] * ez[u];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          oi[q] = -1; ol[q] = VOGE_SENT_LEN; oa[q] = VOGE_SENT_ACT; od[q] = 0.0f;
          if (q < nv[u]) {
            oi[q] = (int32_t)(uint32_t)key[u][q];
            ol[q] = ord2f((uint32_t)(key[u][q] >> 32));
            const PairOut o = pair_eval(rc[u][q].x, rc[u][q].y, rc[u][q].z, unpack_eval(g0[u][q], g1[u][q], g2[u][q]),
                                        ex[u], ey[u], ez[u], qxx, qyy, qzz, qxy, qxz, qyz);
            oa[q] = o.act;
            od[q] = o.dsd;
          }
        }
        st16i<(VOGE_NT_STORES & 2) != 0>(out_idx + ob[u], oi[0], oi[1], oi[2], oi[3]);      // (write-once, 16 B per slot: non-temporal, voge_common.h)
        st16f<(VOGE_NT_STORES & 2) != 0>(out_len + ob[u], ol[0], ol[1], ol[2], ol[3]);
        st16f<(VOGE_NT_STORES & 2) != 0>(out_act + ob[u], oa[0], oa[1], oa[2], oa[3]);
        st16f<(VOGE_NT_STORES & 2) != 0>(out_dsd + ob[u], od[0], od[1], od[2], od[3]);
      }
    }
    // Fragment mode without act / dsd (out_act == NULL; voge_fragments_fwd_iso*): index and len are the key itself --
    // no gather, no ray, no arithmetic; the composite kernel behind the sweep derives act / dsd from the same
    // records with the same operations (composite.hip), at its own, much higher residency.
    for (int it0 = tid; !want_ad && it0 < nitem; it0 += T * kEpiU) {
#pragma unroll
      for (int u = 0; u < kEpiU; ++u) {
        const int it = it0 + u * T;
        if (it >= nitem) break;
        const int r = __float2int_rz(((float)it + 0.5f) * inv_ipr);
        const int j = (it - r * ipr) * 4;
        const int x = __float2int_rz(((float)j + 0.5f) * invK);
        const int sl = j - x * K;
        const int owner = ((x >> 3) + (TW / 8) * (r >> 3)) * 64 + (x & 7) + 8 * (r & 7);
        const size_t pix = ((size_t)b * H + ty * TH + r) * W + (size_t)tx * TW + x;
        const int nv = max(0, min(4, L.id[owner] - sl));
        int32_t oi[4];
        float ol[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint64_t key = (q < nv) ? keys[(size_t)(sl + q) * TP + owner] : 0ull;
          oi[q] = (q < nv) ? (int32_t)(uint32_t)key : -1;
          ol[q] = (q < nv) ? ord2f((uint32_t)(key >> 32)) : VOGE_SENT_LEN;
        }
        *reinterpret_cast<int4 *>(out_idx + pix * K + sl) = make_int4(oi[0], oi[1], oi[2], oi[3]);
        *reinterpret_cast<float4 *>(out_len + pix * K + sl) = make_float4(ol[0], ol[1], ol[2], ol[3]);
      }
    }
    for (int it0 = tid; !tile_gen && want_ad && it0 < nitem; it0 += T * kEpiU) {
      uint64_t key[kEpiU][4];
      float4 rec[kEpiU][4];
      float ex[kEpiU], ey[kEpiU], ez[kEpiU];
      size_t ob[kEpiU];
      int nv[kEpiU];
#pragma unroll
      for (int u = 0; u < kEpiU; ++u) {
        const int it = it0 + u * T;
        nv[u] = -1;
        ob[u] = 0;
        ex[u] = ey[u] = ez[u] = 0.0f;
        if (it < nitem) {
          const int r = __float2int_rz(((float)it + 0.5f) * inv_ipr);
          const int j = (it - r * ipr) * 4;
          const int x = __float2int_rz(((float)j + 0.5f) * invK);
          const int sl = j - x * K;
          const int owner = ((x >> 3) + (TW / 8) * (r >> 3)) * 64 + (x & 7) + 8 * (r & 7);
          const size_t pix = ((size_t)b * H + ty * TH + r) * W + (size_t)tx * TW + x;
          ob[u] = pix * K + sl;
          nv[u] = max(0, min(4, L.id[owner] - sl));
#pragma unroll
          for (int q = 0; q < 4; ++q) key[u][q] = (q < nv[u]) ? keys[(size_t)(sl + q) * TP + owner] : 0ull;
          if (nv[u] > 0) { ex[u] = rays[pix * 3]; ey[u] = rays[pix * 3 + 1]; ez[u] = rays[pix * 3 + 2]; }
        }
      }
#pragma unroll
      for (int u = 0; u < kEpiU; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rec[u][q] = (q < nv[u]) ? ms[(uint32_t)key[u][q]] : make_float4(0.f, 0.f, 0.f, 0.f);
      // anisotropic entries (w = NaN) need their full record: 3 more gathers each.  They are issued for
      // all four slots of an item before any is used (12 in flight per lane instead of 3).
      bool gen_any = false;
#pragma unroll
      for (int u = 0; u < kEpiU; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) gen_any = gen_any || ((q < nv[u]) && !(rec[u][q].w == rec[u][q].w));
      gen_any = __any(gen_any);
#pragma unroll
      for (int u = 0; u < kEpiU; ++u) {
        if (nv[u] < 0) continue;
        int32_t oi[4];
        float ol[4], oa[4], od[4];
        const float qxx = ex[u] * ex[u], qyy = ey[u] * ey[u], qzz = ez[u] * ez[u];
        float4 g0[4], g1[4], g2[4];
        if (gen_any) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            g0[q] = g1[q] = g2[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((q < nv[u]) && !(rec[u][q].w == rec[u][q].w)) {
              const size_t eo = (size_t)(uint32_t)key[u][q] * 3;
              g0[q] = evr[eo]; g1[q] = evr[eo + 1]; g2[q] = evr[eo + 2];
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          oi[q] = -1; ol[q] = VOGE_SENT_LEN; oa[q] = VOGE_SENT_ACT; od[q] = 0.0f;
          if (q < nv[u]) {
            oi[q] = (int32_t)(uint32_t)key[u][q];
            ol[q] = ord2f((uint32_t)(key[u][q] >> 32));
            PairOut o;
            if (rec[u][q].w == rec[u][q].w) {
              o = pair_eval_iso_at(rec[u][q].x, rec[u][q].y, rec[u][q].z, rec[u][q].w, ol[q], ex[u], ey[u], ez[u],
                                   (qxx + qyy) + qzz);     // len is in the key: no second division
            } else {
              const EvalRec e = unpack_eval(g0[q], g1[q], g2[q]);
              o = pair_eval(rec[u][q].x, rec[u][q].y, rec[u][q].z, e, ex[u], ey[u], ez[u], qxx, qyy, qzz, ex[u] * ey[u],
                            ex[u] * ez[u], ey[u] * ez[u]);
            }
            oa[q] = o.act;
            od[q] = o.dsd;
          }
        }
        st16i<(VOGE_NT_STORES & 2) != 0>(out_idx + ob[u], oi[0], oi[1], oi[2], oi[3]);      // (write-once, 16 B per slot: non-temporal, voge_common.h)
        st16f<(VOGE_NT_STORES & 2) != 0>(out_len + ob[u], ol[0], ol[1], ol[2], ol[3]);
        st16f<(VOGE_NT_STORES & 2) != 0>(out_act + ob[u], oa[0], oa[1], oa[2], oa[3]);
        st16f<(VOGE_NT_STORES & 2) != 0>(out_dsd + ob[u], od[0], od[1], od[2], od[3]);
      }
    }
  }
  if (!vec4 && out_weight == nullptr && out_act == nullptr) {
    // K not a multiple of four, fragments without act / dsd (ShapeFitting's max_assign = 25): index and len are the keys
    // themselves -- every slot of the tile is one independent LDS read and two 4-byte stores, no gather, no arithmetic
    const int th = min(TH, H - ty * TH);
    const float inv_ri = 1.0f / (float)row_items;
    for (int it = tid; it < th * row_items; it += T) {
      const int r = __float2int_rz(((float)it + 0.5f) * inv_ri);
      const int j = it - r * row_items;
      const int x = j / K, sl = j - x * K;
      const int owner = ((x >> 3) + (TW / 8) * (r >> 3)) * 64 + (x & 7) + 8 * (r & 7);
      const bool in = sl < L.id[owner];
      const uint64_t key = in ? keys[(size_t)sl * TP + owner] : 0ull;
      const size_t o = (((size_t)b * H + ty * TH + r) * W + (size_t)tx * TW) * K + j;
      out_idx[o] = in ? (int32_t)(uint32_t)key : -1;
      out_len[o] = in ? ord2f((uint32_t)(key >> 32)) : VOGE_SENT_LEN;
    }
  }
  if (!vec4 && out_weight == nullptr && out_act != nullptr) {
    // K not a multiple of four, act / dsd wanted: one slot per lane and trip over the whole tile (rows x pixels x slots
    // flattened, so a wave makes th * tw * K / 64 trips instead of th * ceil(tw * K / 64)), four trips in flight
    const int th = min(TH, H - ty * TH);
    const float inv_ri = 1.0f / (float)row_items;
    const int nit = th * row_items;
    for (int it0 = tid; it0 < nit; it0 += 4 * T) {
      int32_t oi[4];
      float ol[4], oa[4], od[4];
      size_t oo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = it0 + u * T;
        oo[u] = 0; oi[u] = -1; ol[u] = VOGE_SENT_LEN; oa[u] = VOGE_SENT_ACT; od[u] = 0.0f;
        if (it < nit) {
          const int r = __float2int_rz(((float)it + 0.5f) * inv_ri);
          const int j = it - r * row_items;
          const int x = j / K, sl = j - x * K;
          const size_t pix = ((size_t)b * H + ty * TH + r) * W + (size_t)tx * TW + x;
          slot_value(r, x, sl, pix, oi[u], ol[u], oa[u], od[u]);
          oo[u] = pix * K + sl;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (it0 + u * T < nit) { out_idx[oo[u]] = oi[u]; out_len[oo[u]] = ol[u]; out_act[oo[u]] = oa[u]; out_dsd[oo[u]] = od[u]; }
      }
    }
  }
#ifdef VOGE_SWEEP_TIMES
  if (tid == 0 && b == 0 && bx < 8192) {
    unsigned long long *o = g_sweep_times + 8 * (size_t)bx;
    o[0] = ts0; o[1] = ts1; o[2] = ts_fill; o[3] = ts_cons; o[4] = ts2; o[5] = wall_clock64();
#ifdef VOGE_SWEEP_STATS
    o[6] = st_eval;
#else
    o[6] = (unsigned long long)cnt_dbg;
#endif
    o[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
  }
#endif
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
                      float *__restrict__ out_act, float *__restrict__ out_dsd, int32_t *__restrict__ out_cnt) {
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
  uint64_t worst = ~0ull, tail = 0ull;
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
      if (key < worst) topk_insert(mykeys, 64, K, cnt, worst, tail, key);
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
  if (out_cnt != nullptr) out_cnt[pix] = cnt;
}

struct TraceWs {
  float4 *cull, *evr, *ms, *ell;     // per-Gaussian records
  ConeRec *cones;                     // per super-tile
  int *seg_count;                     // binA -> binB: per (super-tile, slice) segment
  int32_t *seg_id;
  int *q_count;                       // quad lists in memory: only the fallback of an overflowed tile list
  int32_t *q_id;
  float *q_lb;
  int *tl_count;                      // per sweep tile
  int32_t *tl_id;
  float *tl_lb;
  float4 *seg_rec;
  int2 *order;                        // launch order of the sweep: (tile, list length) by super-tile rank and slot
  unsigned long long *pool_top;       // binB's long path: lists of quads with more than kQCap candidates
  int32_t *pool_id;
  float *pool_lb;
  int *tl_off;                        // per sweep tile: start of its pooled list
  int pool_cap;
  int *seg_ext;                       // binA -> binB: extensions of segments with more than kSegCap entries
  int32_t *ext_id;
  int ext_arena;                      // ids per (region, slice) workgroup of binA
  int nstx, nsty, nst0x, nst0y, nbin;
};

// Entries of the list pool: a Gaussian sits in the list of every tile its (conservative) footprint touches -- a handful
// for the small footprints that make a quad overflow in the first place.
static size_t trace_pool_entries(const size_t P) {
  const size_t want = 32 * P;
  return want < ((size_t)1 << 20) ? ((size_t)1 << 20) : (want > ((size_t)1 << 30) ? ((size_t)1 << 30) : want);
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// The list pool's counters: one per chunk of the batch an entry point walks (chunk c uses slot min(c, kPoolSlots - 1)), in the
// LAST kPoolTail bytes of the scratch the caller handed over -- the same addresses whatever a chunk's view count, so that
// voge_trace_pool_usage finds every chunk's counter behind the call, and out of the way of a caller who lets later, smaller
// entry points reuse the buffer's head (voge_amd.ops._workspace does: at the head of the scratch the fused backward's
// records overwrote them).
constexpr int kPoolSlots = 64;
constexpr size_t kPoolTail = 512;
static unsigned long long *trace_pool_slots(void *workspace, const size_t workspace_bytes) {
  return reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(workspace) + ((workspace_bytes - kPoolTail) & ~(size_t)255));
}

static size_t trace_ws_layout(int B, int N, int H, int W, void *base, TraceWs *ws) {
  const size_t P = (size_t)B * N;
  const int nstx = (W + kST - 1) / kST, nsty = (H + kST - 1) / kST;
  const size_t nbin = (size_t)B * nstx * nsty;
  const size_t ntile = (size_t)B * ((W + 7) / 8) * ((H + 7) / 8);
  size_t off = 0;
  char *p = reinterpret_cast<char *>(base);
  auto take = [&](size_t bytes) { char *q = p ? p + off : nullptr; off += align256(bytes); return q; };
  char *c = take(P * 16), *e = take(P * 48), *m4 = take(P * 16), *el = take(P * 32), *cn = take(cone_records(1, nbin) * sizeof(ConeRec)),
       *sc = take(nbin * kParts * 4), *si = take(nbin * kParts * (size_t)kSegCap * 4), *bc = take(nbin * 4 * 4),
       *bi = take(nbin * 4 * kQCap * 4), *bl = take(nbin * 4 * kQCap * 4), *tc = take(ntile * 4),
       *ti = take(ntile * kTileCap * 4), *tl = take(ntile * kTileCap * 4), *sr = take(nbin * kParts * (size_t)kSegCap * 16),
       *cq = take(nbin * kTilesPerBin * 8 * 2);      // (second half: the exactly sorted copy of VOGE_EXACT_ORDER builds)
  const size_t npool = trace_pool_entries(P);
  char *pi = take(npool * 4), *pl = take(npool * 4), *to = take(ntile * 4);
  const int nst0x = (W + kST0 - 1) / kST0, nst0y = (H + kST0 - 1) / kST0;
  size_t arena = (size_t)kExtMul * slice_cap(N);
  // (seg_ext holds 32-bit offsets into ext_id: a batch so large that the extensions would pass 2^31 ids goes without them
  // -- such segments then count as overflowed and binB re-tests their slice, as before round 3)
  if ((size_t)B * nst0x * nst0y * kParts * arena > (size_t)0x7fffffff) arena = 0;
  char *se = take(nbin * kParts * kExtChunks * 4), *ei = take((size_t)B * nst0x * nst0y * kParts * arena * 4);
  if (ws) {
    ws->seg_ext = reinterpret_cast<int *>(se); ws->ext_id = reinterpret_cast<int32_t *>(ei); ws->ext_arena = (int)arena;
    ws->pool_top = nullptr /* (trace_pool_slots: the scratch's tail) */; ws->pool_id = reinterpret_cast<int32_t *>(pi);
    ws->pool_lb = reinterpret_cast<float *>(pl); ws->tl_off = reinterpret_cast<int *>(to);
    ws->pool_cap = (int)npool;
    ws->cull = reinterpret_cast<float4 *>(c); ws->evr = reinterpret_cast<float4 *>(e);
    ws->ms = reinterpret_cast<float4 *>(m4); ws->ell = reinterpret_cast<float4 *>(el);
    ws->cones = reinterpret_cast<ConeRec *>(cn);
    ws->seg_count = reinterpret_cast<int *>(sc); ws->seg_id = reinterpret_cast<int32_t *>(si);
    ws->q_count = reinterpret_cast<int *>(bc); ws->q_id = reinterpret_cast<int32_t *>(bi);
    ws->q_lb = reinterpret_cast<float *>(bl);
    ws->tl_count = reinterpret_cast<int *>(tc); ws->tl_id = reinterpret_cast<int32_t *>(ti);
    ws->tl_lb = reinterpret_cast<float *>(tl);
    ws->seg_rec = reinterpret_cast<float4 *>(sr); ws->order = reinterpret_cast<int2 *>(cq);
    ws->nstx = nstx; ws->nsty = nsty;
    ws->nst0x = (W + kST0 - 1) / kST0; ws->nst0y = (H + kST0 - 1) / kST0;
    ws->nbin = (int)nbin;
  }
  return off;
}

#ifndef VOGE_EXACT_ORDER
#define VOGE_EXACT_ORDER 0        // (experiment) re-sort the launch order exactly by list length with one extra launch
#endif
// One workgroup: counting sort of the n launch slots by descending list length (overflowed lists first, empty tiles
// and slots outside the image last).
__global__ void __launch_bounds__(1024) order_sort_kernel(const int2 *__restrict__ in, int2 *__restrict__ out, const int n) {
  constexpr int NB = 3072;      // >= kTileCap + 3 buckets, 3 per thread
  __shared__ int hist[NB];
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NB; i += 1024) hist[i] = 0;
  __syncthreads();
  auto bucket = [](const int2 v) { return (v.x < 0 || v.y == 0) ? kTileCap + 2 : (v.y < 0 ? 0 : 1 + (kTileCap - min(v.y, kTileCap))); };
  for (int i = tid; i < n; i += 1024) atomicAdd(&hist[bucket(in[i])], 1);
  __syncthreads();
  const int a = hist[3 * tid], b = hist[3 * tid + 1], c = hist[3 * tid + 2];
  int x = a + b + c;
  const int own = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int off = x - own;
  for (int w = 0; w < wave; ++w) off += wsum[w];
  hist[3 * tid] = off; hist[3 * tid + 1] = off + a; hist[3 * tid + 2] = off + a + b;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const int2 v = in[i];
    out[atomicAdd(&hist[bucket(v)], 1)] = v;
  }
}

#ifndef VOGE_FUSED_EPILOGUE
#define VOGE_FUSED_EPILOGUE 0      // (experiment, see trace_chunk_fwd) composite inside the sweep's epilogue
#endif
#ifndef VOGE_SWEEP_LDS_PAD
#define VOGE_SWEEP_LDS_PAD 0      // (occupancy experiments: extra dynamic LDS per workgroup)
#endif
// -DVOGE_AB builds only (libvoge_hip_ab.so, never the product library): a process-wide switch between sweep_iso_kernel (0)
// and round 3's scalar-sigma sweep, trace_fwd_kernel<1, true> (1) -- for A/B timing and the bit-for-bit comparison in
// tests/test_gpu_configs.py (voge_debug_sweep_variant).  The product library has neither the switch nor the old kernel.
#ifdef VOGE_AB
static std::atomic<int> g_sweep_variant{0};
#endif
// binB + the sweep (one wave = one 8x8-pixel tile per workgroup)
template <bool ISO>
static int launch_trace(const TraceWs &ws, const ConeRec *cones, const float *rays, int B, int N, int H, int W, int K,
                        float thr_act, int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt, float occ,
                        float *weight, int64_t *valid_num, hipStream_t st, const CamView &cam, const bool diag = false) {
  constexpr int T = 64;
  // (diag: every general form of the launch is a per-axis one -- the frame path's gen_kind 1 -- sweep_iso_kernel<2>)
  const int gen = ISO ? 0 : (diag ? 2 : 1);
  const auto sweep = ISO ? sweep_iso_kernel<0> : (diag ? sweep_iso_kernel<2> : sweep_iso_kernel<1>);
  // (fused composite epilogue: three padded per-pixel rows (len, s', E) for one round of 64 / (K/4) pixels)
  const size_t comp = (weight != nullptr) ? sizeof(float) * 3 * (size_t)compn_rows(K, 4, 64, true) : 0;
  const size_t lds = ((sizeof(uint64_t) * (size_t)(K + 1) * (T + 1) + 15) & ~(size_t)15) + ((sizeof(TraceLds<T, ISO>) + 15) & ~(size_t)15) +
                     comp + VOGE_SWEEP_LDS_PAD;
  // no composite inside the epilogue: sweep_iso_kernel (sweep_iso.h: float-compare commits, 6-byte list entries) -- <false> for
  // scalar sigmas (round 4), <true> for the general forms (round 5)
#ifdef VOGE_AB
  const bool v2 = VOGE_SWEEP_V2 && weight == nullptr && g_sweep_variant.load(std::memory_order_relaxed) != 1;
  constexpr bool kOld = true;
#else
  const bool v2 = VOGE_SWEEP_V2 && weight == nullptr;
  constexpr bool kOld = !VOGE_SWEEP_V2 || VOGE_FUSED_EPILOGUE;      // (the product keeps round 3's sweeps out of the library)
  if (!v2 && !kOld) return VOGE_ERR_BAD_ARG;
#endif
  const size_t lds2 = sweep2_lds_bytes(K, gen) + VOGE_SWEEP_LDS_PAD;
  {
    static DynLdsCache cache2[2];      // (one per kernel of this instantiation)
    if (v2) {
      const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(sweep), lds2, cache2[diag ? 1 : 0]);
      if (rc) return rc;
    }
  }
  if constexpr (kOld) if (!v2) {
    static DynLdsCache cache;
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(trace_fwd_kernel<1, ISO && kOld>), lds, cache);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(binB_kernel<!ISO>, dim3(ws.nstx * ws.nsty * 4, B), dim3(kQT), 0, st, ws.cull, ws.ell, ws.seg_count, ws.seg_id, ws.seg_rec,
                     cones, N, H, W, ws.nstx, ws.nsty, ws.nbin, ws.q_count, ws.q_id, ws.q_lb, ws.tl_count, ws.tl_id, ws.tl_lb,
                     ws.order, ws.pool_top, ws.pool_cap, ws.pool_id, ws.pool_lb, ws.tl_off, ws.seg_ext, ws.ext_id, K,
                     (v2 && act == nullptr) ? nullptr : idx /* (sweep_iso_kernel writes the empty tiles itself) */, len, act, dsd, cnt,
                     weight, valid_num, cam);
  {
    int rc = launch_status();
    if (rc) return rc;
  }
  const int2 *order = ws.order;
#if VOGE_EXACT_ORDER
  hipLaunchKernelGGL(order_sort_kernel, dim3(1), dim3(1024), 0, st, ws.order, ws.order + (size_t)ws.nbin * kTilesPerBin, ws.nbin * kTilesPerBin);
  order = ws.order + (size_t)ws.nbin * kTilesPerBin;
#endif
  dim3 grid(ws.nbin * kTilesPerBin);     // one workgroup per tile slot of every super-tile (slots outside the image exit)
  if (v2) {
    hipLaunchKernelGGL(sweep, grid, dim3(T), lds2, st, ws.cull, ws.ms, ws.evr, rays, ws.q_count, ws.q_id, ws.q_lb, ws.tl_id,
                       ws.tl_lb, ws.pool_id, ws.pool_lb, ws.tl_off, order, ((W + 7) / 8) * ((H + 7) / 8), ws.nstx, ws.nstx * ws.nsty, N, H, W, K,
                       thr_act, idx, len, act, dsd, cnt, cam);
    return launch_status();
  }
  if (cam.R != nullptr) return VOGE_ERR_BAD_ARG;      // (round 3's sweeps read the bundle)
  if constexpr (kOld) {
    hipLaunchKernelGGL((trace_fwd_kernel<1, ISO && kOld>), grid, dim3(T), lds, st, ws.cull, ws.evr, ws.ms, rays, ws.q_count, ws.q_id, ws.q_lb,
                       ws.tl_count, ws.tl_id, ws.tl_lb, ws.pool_id, ws.pool_lb, ws.tl_off, order, ((W + 7) / 8) * ((H + 7) / 8), ws.nstx,
                       ws.nstx * ws.nsty, N, H, W, K, thr_act, idx, len, act, dsd, cnt, occ, weight, valid_num);
  }
  return launch_status();
}

}  // namespace voge

using namespace voge;

#ifdef VOGE_SWEEP_TIMES
// debug builds only (tools/sweep_stats.py): read and clear the sweep counters
extern "C" int voge_debug_sweep_stats(unsigned long long *out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(voge::g_sweep_stats), sizeof(unsigned long long) * 16);
  if (e != hipSuccess) return (int)e;
  unsigned long long z[16] = {0};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(voge::g_sweep_stats), z, sizeof(z));
}
#endif
#ifdef VOGE_SWEEP_TIMES
extern "C" int voge_debug_sweep_times(unsigned long long *out, int n_wg) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(voge::g_sweep_times), sizeof(unsigned long long) * 8 * (size_t)n_wg);
}
#endif

#ifdef VOGE_BIN_TIMES
extern "C" int voge_debug_bin_wave(unsigned long long *out, int n_wg) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(voge::g_bin_wave), sizeof(unsigned long long) * 64 * (size_t)n_wg);
}
extern "C" int voge_debug_bin_times(unsigned long long *out, int which, int n_wg) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(voge::g_bin_times), sizeof(unsigned long long) * 8 * (size_t)n_wg,
                                  sizeof(unsigned long long) * 8 * 1024 * (size_t)which);
}
#endif

#ifdef VOGE_AB
extern "C" int voge_debug_sweep_variant(int variant) {
  if (variant < 0 || variant > 1) return VOGE_ERR_BAD_ARG;
  voge::g_sweep_variant.store(variant, std::memory_order_relaxed);
  return 0;
}
#endif

// Round 5: the scratch is sized for a CHUNK of the batch, not for all of it (it used to be B x 165 MB at 512^2, B x 774 MB at
// 1024^2: 6.2 GB for eight 1024^2 views).  Views are independent in every stage, so the entry points walk a batch in
// chunks of as many views as the scratch they were given holds -- stream-ordered launches on the same buffers -- and the
// size asked for here is what the largest chunk under kTraceWsCap needs (never less than one view).  A caller that wants a
// big batch in ONE chunk passes more: any size >= this is accepted and used.
constexpr size_t kTraceWsCap = (size_t)1 << 30;
static size_t trace_ws_bytes(const int nb, const int N, const int H, const int W) {      // a chunk's arrays + the counters' tail
  return trace_ws_layout(nb, N, H, W, nullptr, nullptr) + 256 + kPoolTail;
}
static int trace_views_that_fit(const int B, const int N, const int H, const int W, const size_t bytes) {
  int nb = 1;      // (layout is monotone in the view count: the largest nb whose layout fits)
  for (int step = B; step >= 1; step >>= 1)
    while (nb + step <= B && trace_ws_bytes(nb + step, N, H, W) <= bytes) nb += step;
  return nb;
}
extern "C" size_t voge_trace_workspace_bytes(int B, int N, int H, int W) {
  if (B <= 0 || N < 0 || H <= 0 || W <= 0) return 0;
  return trace_ws_bytes(trace_views_that_fit(B, N, H, W, kTraceWsCap), N, H, W);
}

extern "C" int voge_trace_pool_usage(const void *workspace, size_t workspace_bytes, int B, int N, int H, int W, int *used,
                                     int *capacity) {
  if (!workspace || !used || !capacity || B <= 0 || N < 0 || H <= 0 || W <= 0) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < trace_ws_bytes(1, N, H, W)) return VOGE_ERR_WORKSPACE;
  // the chunks the entry point walked with THIS scratch (the same rule: trace_topk_fwd_impl), each with a counter of its own in
  // the scratch's tail; reported: the largest use of any chunk, against the smallest capacity
  const int per = trace_views_that_fit(B, N, H, W, workspace_bytes);
  const int nchunks = (B + per - 1) / per;
  unsigned long long tops[kPoolSlots];
  const hipError_t e = hipMemcpy(tops, trace_pool_slots(const_cast<void *>(workspace), workspace_bytes), sizeof(tops), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  unsigned long long top = 0ull;
  for (int c = 0; c < nchunks && c < kPoolSlots; ++c) top = tops[c] > top ? tops[c] : top;
  TraceWs ws;
  trace_ws_layout(per, N, H, W, const_cast<void *>(workspace), &ws);
  *capacity = ws.pool_cap;
  if (nchunks > 1 && B % per != 0) {      // (the last, shorter chunk: its pool is sized for its own Gaussians)
    trace_ws_layout(B % per, N, H, W, const_cast<void *>(workspace), &ws);
    *capacity = ws.pool_cap < *capacity ? ws.pool_cap : *capacity;
  }
  *used = top > 0x7fffffffull ? 0x7fffffff : (int)top;
  return 0;
}

extern "C" int voge_ray_cones(const float *rays, int B, int H, int W, float *cones, voge_stream_t stream);   // rays.hip
extern "C" int voge_composite_fwd_iso(const int32_t *idx, const int32_t *cnt, const float *len, const float *records,
                                      const float *rays, float occ, long npix, int K, float *weight, int64_t *valid_num,
                                      voge_stream_t stream);                                                     // composite.hip

// idx[i] += off where idx[i] >= 0 (a chunk's indices are local to its first view: see trace_topk_fwd_impl)
__global__ void __launch_bounds__(256) idx_rebase_kernel(int32_t *__restrict__ idx, const size_t n4, const int off) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    int4 v = reinterpret_cast<int4 *>(idx)[i];
    v.x = v.x >= 0 ? v.x + off : v.x; v.y = v.y >= 0 ? v.y + off : v.y; v.z = v.z >= 0 ? v.z + off : v.z; v.w = v.w >= 0 ? v.w + off : v.w;
    reinterpret_cast<int4 *>(idx)[i] = v;
  }
}
__global__ void __launch_bounds__(256) idx_rebase1_kernel(int32_t *__restrict__ idx, const size_t n, const int off) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int v = idx[i];
    if (v >= 0) idx[i] = v + off;
  }
}

static int trace_chunk_fwd(const int iso_in, const IsoView view, const float *mus, const float *isigmas, const float *rays,
                           const float *cam_fwd, const float *cones_in, int B, int N, int H, int W, int K,
                           float thr_act, void *workspace, int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                           voge_stream_t stream, float occ, float *weight, int64_t *valid_num, float *records, const CamView &cam,
                           unsigned long long *pool_top);

static int trace_topk_fwd_impl(const int iso_in, const IsoView view, const float *mus, const float *isigmas, const float *rays,
                               const float *cam_fwd, const float *cones_in, int B, int N, int H, int W, int K,
                               float thr_act, void *workspace, size_t workspace_bytes,
                               int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                               voge_stream_t stream, float occ = 1.0f, float *weight = nullptr, int64_t *valid_num = nullptr,
                               float *records = nullptr, const CamView cam = no_camera()) {
  if (B < 0 || N < 0 || H < 0 || W < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K) return VOGE_ERR_K_TOO_LARGE;
  if ((size_t)B * H * W == 0) return 0;  // numel == 0 early return (ray_trace_voge.cu:248-251)
  // (with a camera the kernels make the rays themselves; the bundle they leave in cam.rays_out is what a composite behind the
  //  sweep reads.  That form keeps no act / dsd: the sweep's epilogue would have to read the bundle back.)
  if (cam.R != nullptr) {
    if (rays || cones_in || cam_fwd || act || dsd || !cam.T || !cam.focal || !cam.pp || cam.stripe_h <= 0 || cam.pitch < 0 ||
        cam.h != H || cam.W != W || (weight != nullptr && !cam.rays_out))
      return VOGE_ERR_BAD_ARG;
  } else if (!rays) {
    return VOGE_ERR_BAD_ARG;
  }
  if (!idx || !len || !workspace || (act == nullptr) != (dsd == nullptr)) return VOGE_ERR_BAD_ARG;
  // act / dsd may be omitted by the scalar-sigma fragment entry points only (they are re-derived where needed)
  // (with weights: composited behind the sweep; without: records kept, the caller composites later)
  // (general forms: trace only, with the packed (mu, A) records kept for the deferred composite -- voge_trace_lean_fwd)
  if (act == nullptr && !((iso_in ? (weight != nullptr || records != nullptr) : (weight == nullptr && records != nullptr)) &&
                          cnt != nullptr && (long)B * N < (1l << 26)))
    return VOGE_ERR_BAD_ARG;
  if (N > 0 && (!mus || !isigmas)) return VOGE_ERR_BAD_ARG;
  // (one view's scratch is the least that works; voge_trace_workspace_bytes(B, ...) is what to allocate)
  if (workspace_bytes < trace_ws_bytes(1, N, H, W)) return VOGE_ERR_WORKSPACE;
  // the top-K lists of one 8x8 tile must fit the CU's LDS: validated before anything is enqueued
  if (sizeof(uint64_t) * (size_t)(K + 1) * 65 + 16 + sizeof(TraceLds<64, false>) > 160 * 1024) return VOGE_ERR_K_TOO_LARGE;
  // ---- the batch in chunks of as many views as the scratch holds (all of them, when it was sized for that): every array the
  // caller sees is offset to the chunk's first view, the chunk runs as a batch of its own, and the indices it wrote -- local
  // to that first view -- are moved up by b0 N afterwards (one pass over idx, chunks behind the first only)
  const int per = trace_views_that_fit(B, N, H, W, workspace_bytes);
  const size_t npv = (size_t)H * W;      // pixels per view
  const size_t nst = (size_t)((W + kST - 1) / kST) * ((H + kST - 1) / kST);
  const size_t stride_mu = view.shared ? 0 : (size_t)N * 3;
  const size_t stride_sg = iso_in ? (view.shared ? 0 : (size_t)N)
                                  : (view.gen_kind ? (view.sigma_shared ? 0 : (size_t)N * (view.gen_kind == 1 ? 3 : 9)) : (size_t)N * 9);
  for (int b0 = 0; b0 < B; b0 += per) {
    const int nb = (B - b0 < per) ? B - b0 : per;
    IsoView v = view;
    if (v.origin != nullptr) v.origin += 3 * (size_t)b0;
    auto at = [&](auto *p, const size_t per_view) { return p ? p + (size_t)b0 * per_view : p; };
    if (cam.R != nullptr) v.cam_origin = 1;
    CamView cv = cam;
    cv.R = at(cam.R, 9); cv.T = at(cam.T, 3); cv.focal = at(cam.focal, 2); cv.pp = at(cam.pp, 2);
    cv.origin_out = at(cam.origin_out, 3); cv.rays_out = at(cam.rays_out, npv * 3);
    const int rc = trace_chunk_fwd(iso_in, v, at(mus, stride_mu), at(isigmas, stride_sg), at(rays, npv * 3), at(cam_fwd, 3),
                                   at(cones_in, nst * kConeRecsPerST * (sizeof(ConeRec) / sizeof(float))), nb, N, H, W, K, thr_act, workspace,
                                   at(idx, npv * K), at(len, npv * K), at(act, npv * K), at(dsd, npv * K), at(cnt, npv), stream, occ,
                                   at(weight, npv * K), at(valid_num, npv), at(records, (size_t)N * (iso_in ? 4 : (view.gen_kind == 1 ? 8 : 12))), cv,
                                   trace_pool_slots(workspace, workspace_bytes) + ((b0 / per) < kPoolSlots ? (b0 / per) : kPoolSlots - 1));
    if (rc) return rc;
    if (b0 > 0 && N > 0) {
      const size_t n = (size_t)nb * npv * K;
      int32_t *ic = idx + (size_t)b0 * npv * K;
      if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(ic) & 15) == 0) {
        const size_t n4 = n >> 2;
        hipLaunchKernelGGL(idx_rebase_kernel, dim3((unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096)), dim3(256), 0,
                           (hipStream_t)stream, ic, n4, b0 * N);
      } else {
        hipLaunchKernelGGL(idx_rebase1_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0,
                           (hipStream_t)stream, ic, n, b0 * N);
      }
      const int rc2 = launch_status();
      if (rc2) return rc2;
    }
  }
  return 0;
}

static int trace_chunk_fwd(const int iso_in, const IsoView view, const float *mus, const float *isigmas, const float *rays,
                           const float *cam_fwd, const float *cones_in, int B, int N, int H, int W, int K,
                           float thr_act, void *workspace, int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                           voge_stream_t stream, float occ, float *weight, int64_t *valid_num, float *records, const CamView &cam,
                           unsigned long long *pool_top) {
  hipStream_t st = (hipStream_t)stream;
  const int P = B * N;
  TraceWs ws;
  trace_ws_layout(B, N, H, W, workspace, &ws);
  ws.pool_top = pool_top;      // (this chunk's own counter, in the scratch's tail: voge_trace_pool_usage)
  if (records != nullptr && iso_in) ws.ms = reinterpret_cast<float4 *>(records);      // the caller keeps the (centre, a) records (backward)
  // super-tile cones: the caller's (voge_rays_fwd makes them while it makes the rays), or one more launch here
  const ConeRec *cones = reinterpret_cast<const ConeRec *>(cones_in);
  if (cones == nullptr && cam.R == nullptr) {
    const int rc = voge_ray_cones(rays, B, H, W, reinterpret_cast<float *>(ws.cones), stream);
    if (rc) return rc;
    cones = ws.cones;
  }
  const dim3 gridA(ws.nst0x * ws.nst0y * kParts, B);
  // (the scan of tools/small_set_scan.py, profiles/r5_small_set_scan.txt: at 128^2 / 256^2 skipping binA pays up to ~4 096 Gaussians,
  //  at 512^2 -- four times the quads, each reading every record -- up to ~2 000)
  const bool small_set = N > 0 && N <= ((long)H * W <= 131072 ? VOGE_SMALL_N : VOGE_SMALL_N / 2);
  const long n_seg = (long)B * ws.nstx * ws.nsty * kParts;
  if (small_set && iso_in) {
    // a few thousand Gaussians: records + "every segment overflowed" marks, no binA (small_set_marks)
    hipLaunchKernelGGL(iso_prep_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, mus, isigmas, cam_fwd, N, thr_act, view, ws.cull, ws.ms,
                       cam, ws.seg_count, n_seg, ws.pool_top);
  } else if (small_set) {
    hipLaunchKernelGGL(prep_kernel, dim3((P + 255) / 256), dim3(256), 0, st, mus, isigmas, cam_fwd, N, P, thr_act, ws.cull,
                       ws.evr, ws.ms, ws.ell, reinterpret_cast<float4 *>(records), view, cam, ws.seg_count, n_seg, ws.pool_top);
  } else if (iso_in && N >= VOGE_ISO_PREP_SPLIT) {
    // scalar sigmas, slices of more than two rounds: the records in a pass of their own (iso_prep_kernel)
    hipLaunchKernelGGL(iso_prep_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, mus, isigmas, cam_fwd, N, thr_act, view, ws.cull, ws.ms, cam);
    hipLaunchKernelGGL(binA_kernel<false>, gridA, dim3(kBinThreads), 0, st, cones, ws.nstx, ws.nsty, ws.nst0x, mus, isigmas,
                       cam_fwd, N, thr_act, view, ws.cull, ws.ms, ws.seg_count, ws.seg_id, ws.seg_rec, ws.pool_top, ws.seg_ext, ws.ext_id, ws.ext_arena, cam);
  } else if (iso_in) {
    // scalar sigmas: binA derives the per-Gaussian records itself -- two launches in front of the sweep
    hipLaunchKernelGGL(binA_kernel<true>, gridA, dim3(kBinThreads), 0, st, cones, ws.nstx, ws.nsty, ws.nst0x, mus, isigmas,
                       cam_fwd, N, thr_act, view, ws.cull, ws.ms, ws.seg_count, ws.seg_id, ws.seg_rec, ws.pool_top, ws.seg_ext, ws.ext_id, ws.ext_arena, cam);
  } else {
    if (P > 0)
      hipLaunchKernelGGL(prep_kernel, dim3((P + 255) / 256), dim3(256), 0, st, mus, isigmas, cam_fwd, N, P, thr_act, ws.cull,
                         ws.evr, ws.ms, ws.ell, reinterpret_cast<float4 *>(records), view, cam);
    hipLaunchKernelGGL(binA_kernel<false>, gridA, dim3(kBinThreads), 0, st, cones, ws.nstx, ws.nsty, ws.nst0x, mus, isigmas,
                       cam_fwd, N, thr_act, view, ws.cull, ws.ms, ws.seg_count, ws.seg_id, ws.seg_rec, ws.pool_top, ws.seg_ext, ws.ext_id, ws.ext_arena, cam);
  }
  {
    int rc = launch_status();
    if (rc) return rc;
  }
  // One wave (an 8x8 pixel tile) per sweep workgroup.  Residency is set by the LDS top-K lists (~7 waves per CU at
  // K = 40), and independent single-wave workgroups measured 4-10 % faster than 16x8 / 16x16 tiles in round 1.
  // Fragments wanted as well (weight != NULL): the stand-alone composite kernel runs behind the sweep.  Compositing
  // inside the sweep's epilogue (VOGE_FUSED_EPILOGUE=1 builds, K % 4 == 0; same bits, tests/test_gpu_parity.py) was
  // built and measured: the sweep holds ~1.5 waves per SIMD (its top-K lists fill the LDS), so the composite's row
  // walks run latency-bound there -- sweep 64 -> 142 us at cfg3 against 52 us for the kernel it would replace, which
  // does the same instructions at eight waves per SIMD and reads its 126 MB mostly from the Infinity Cache.
  const bool fused = VOGE_FUSED_EPILOGUE && weight != nullptr && (K & 3) == 0 && K <= 256 && cnt != nullptr;
  int rc;
#ifndef VOGE_NO_ISO_SWEEP
  if (iso_in) rc = launch_trace<true>(ws, cones, rays, B, N, H, W, K, thr_act, idx, len, act, dsd, cnt, occ, fused ? weight : nullptr, fused ? valid_num : nullptr, st, cam);
  else
#endif
  rc = launch_trace<false>(ws, cones, rays, B, N, H, W, K, thr_act, idx, len, act, dsd, cnt, occ, fused ? weight : nullptr, fused ? valid_num : nullptr, st, cam,
                           view.gen_kind == 1);
  if (rc || weight == nullptr || fused) return rc;
  if (act == nullptr)
    return voge_composite_fwd_iso(idx, cnt, len, reinterpret_cast<const float *>(ws.ms), cam.R != nullptr ? cam.rays_out : rays, occ,
                                  (long)B * H * W, K, weight, valid_num, stream);
  return voge_composite_fwd(idx, cnt, act, len, dsd, occ, (long)B * H * W, K, weight, valid_num, stream);
}

extern "C" int voge_trace_topk_fwd(const float *mus, const float *isigmas, const float *rays,
                                   const float *cam_fwd, const float *cones, int B, int N, int H, int W, int K,
                                   float thr_act, void *workspace, size_t workspace_bytes,
                                   int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                                   voge_stream_t stream) {
  return trace_topk_fwd_impl(0, IsoView{nullptr, 0, 0}, mus, isigmas, rays, cam_fwd, cones, B, N, H, W, K, thr_act, workspace,
                             workspace_bytes, idx, len, act, dsd, cnt, stream);
}

extern "C" int voge_trace_topk_fwd_iso(const float *mus, const float *a, const float *rays,
                                       const float *cam_fwd, const float *cones, int B, int N, int H, int W, int K,
                                       float thr_act, void *workspace, size_t workspace_bytes,
                                       int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                                       voge_stream_t stream) {
  return trace_topk_fwd_impl(1, IsoView{nullptr, 0, 0}, mus, a, rays, cam_fwd, cones, B, N, H, W, K, thr_act, workspace,
                             workspace_bytes, idx, len, act, dsd, cnt, stream);
}

extern "C" int voge_trace_topk_list_fwd(const float *mus, const float *isigmas, const float *rays,
                                        const int32_t *bin_points, int B, int P, int H, int W, int K,
                                        int BH, int BW, int M, int bin_size, float thr_act,
                                        int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                                        voge_stream_t stream) {
  if (B < 0 || P < 0 || H < 0 || W < 0 || K <= 0 || BH <= 0 || BW <= 0 || M < 0 || bin_size <= 0)
    return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K) return VOGE_ERR_K_TOO_LARGE;
  if ((size_t)B * H * W == 0) return 0;
  if (!rays || !idx || !len || !act || !dsd || (M > 0 && !bin_points)) return VOGE_ERR_BAD_ARG;
  if (P > 0 && (!mus || !isigmas)) return VOGE_ERR_BAD_ARG;
  const size_t lds = sizeof(uint64_t) * (size_t)K * 64;
  {
    static DynLdsCache cache;
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(trace_list_fwd_kernel), lds, cache);
    if (rc) return rc;
  }
  dim3 grid(((W + 7) / 8) * ((H + 7) / 8), B);
  hipLaunchKernelGGL(trace_list_fwd_kernel, grid, dim3(64), lds, (hipStream_t)stream, mus, isigmas, rays,
                     bin_points, P, H, W, K, BH, BW, M, bin_size, thr_act, idx, len, act, dsd, cnt);
  return launch_status();
}

extern "C" int voge_trace_topk_fwd_iso_view(const float *verts, const float *sigmas, const float *origin, int shared,
                                            int sigma_mode, const float *rays, const float *cam_fwd, const float *cones,
                                            int B, int N, int H, int W, int K, float thr_act, void *workspace,
                                            size_t workspace_bytes, int32_t *idx, float *len, float *act, float *dsd,
                                            int32_t *cnt, voge_stream_t stream) {
  if (sigma_mode < 0 || sigma_mode > 2) return VOGE_ERR_BAD_ARG;
  return trace_topk_fwd_impl(1, IsoView{origin, shared ? 1 : 0, sigma_mode}, verts, sigmas, rays, cam_fwd, cones, B, N, H, W, K,
                             thr_act, workspace, workspace_bytes, idx, len, act, dsd, cnt, stream);
}

// ---- the general trace alone (no act / dsd, no composite): index, len, hit counts, and the packed (mu, A) records
// [B*N][12] the deferred composite (voge_composite_fwd_rec / voge_composite_shade_fwd_rec) and the fused backward read ----
extern "C" int voge_trace_lean_fwd(const float *mus, const float *isigmas, const float *rays, const float *cam_fwd,
                                   const float *cones, int B, int N, int H, int W, int K, float thr_act, void *workspace,
                                   size_t workspace_bytes, int32_t *idx, float *len, int32_t *cnt, float *records,
                                   voge_stream_t stream) {
  if (!cnt || !records) return VOGE_ERR_BAD_ARG;
  return trace_topk_fwd_impl(0, IsoView{nullptr, 0, 0}, mus, isigmas, rays, cam_fwd, cones, B, N, H, W, K, thr_act, workspace,
                             workspace_bytes, idx, len, nullptr, nullptr, cnt, stream, 1.0f, nullptr, nullptr, records);
}

// ---- trace + composite in one call: fragments (weight, idx, valid_num, len) plus act / dsd / cnt for the backward ----
extern "C" int voge_fragments_fwd(const float *mus, const float *isigmas, const float *rays, const float *cam_fwd,
                                  const float *cones, int B, int N, int H, int W, int K, float thr_act, float occ,
                                  void *workspace, size_t workspace_bytes, int32_t *idx, float *len, float *act, float *dsd,
                                  int32_t *cnt, float *weight, int64_t *valid_num, voge_stream_t stream) {
  if (!weight || !valid_num || !cnt) return VOGE_ERR_BAD_ARG;
  return trace_topk_fwd_impl(0, IsoView{nullptr, 0, 0}, mus, isigmas, rays, cam_fwd, cones, B, N, H, W, K, thr_act, workspace,
                             workspace_bytes, idx, len, act, dsd, cnt, stream, occ, weight, valid_num);
}

extern "C" int voge_fragments_fwd_iso(const float *mus, const float *a, const float *rays, const float *cam_fwd,
                                      const float *cones, int B, int N, int H, int W, int K, float thr_act, float occ,
                                      void *workspace, size_t workspace_bytes, int32_t *idx, float *len, float *act,
                                      float *dsd, int32_t *cnt, float *weight, int64_t *valid_num, float *records,
                                      voge_stream_t stream) {
  if (!cnt || (weight == nullptr) != (valid_num == nullptr) || (!weight && (!records || act || dsd))) return VOGE_ERR_BAD_ARG;
  return trace_topk_fwd_impl(1, IsoView{nullptr, 0, 0}, mus, a, rays, cam_fwd, cones, B, N, H, W, K, thr_act, workspace,
                             workspace_bytes, idx, len, act, dsd, cnt, stream, occ, weight, valid_num, records);
}

extern "C" int voge_fragments_fwd_iso_view(const float *verts, const float *sigmas, const float *origin, int shared,
                                           int sigma_mode, const float *rays, const float *cam_fwd, const float *cones,
                                           int B, int N, int H, int W, int K, float thr_act, float occ, void *workspace,
                                           size_t workspace_bytes, int32_t *idx, float *len, float *act, float *dsd,
                                           int32_t *cnt, float *weight, int64_t *valid_num, float *records,
                                           voge_stream_t stream) {
  if (sigma_mode < 0 || sigma_mode > 2 || !cnt || (weight == nullptr) != (valid_num == nullptr) ||
      (!weight && (!records || act || dsd)))
    return VOGE_ERR_BAD_ARG;
  return trace_topk_fwd_impl(1, IsoView{origin, shared ? 1 : 0, sigma_mode}, verts, sigmas, rays, cam_fwd, cones, B, N, H, W, K,
                             thr_act, workspace, workspace_bytes, idx, len, act, dsd, cnt, stream, occ, weight, valid_num, records);
}

// ---- Round 6: the renderer's trace with the CAMERA as its input (GaussianRenderer.forward, Renderer.py:102-150, up to and
// including ray_tracing: the ray bundle of :124-128, the centring of :130, the sigma rule of :133-137, RayTracing.py:12-30 and
// ray_trace_voge.cu:135-217) as three launches -- binA, binB, sweep -- with no ray-generation launch in front: every kernel
// makes the rays, cones, camera centre and view axis it needs from (R, T, focal, principal point) with rays_fwd_kernel's own
// operations (voge_common.h: CamView).  Outputs: idx, len, cnt, records as voge_fragments_fwd_iso_view's trace-only form, plus
// the ray bundle `rays` [B,h,W,3] (written by the sweep: the composite and the backward read it) and `origin` [B,3].
// rows: the band is h stacked rows; stacked row i is image row row0 + (i / stripe_h) * pitch + i % stripe_h (a contiguous
// band: stripe_h >= h, pitch 0; a rank's interleaved stripes: voge_rays_striped_fwd's meaning).  behind != 0: the reference's
// "skip z < 0" candidate rule (rasterize_coarse.cu:35), the view axis being column 2 of R.
extern "C" int voge_frame_trace_fwd_iso(const float *verts, const float *sigmas, int shared, int sigma_mode, const float *R,
                                        const float *T, const float *focal, const float *pp, int row0, int stripe_h, int pitch,
                                        int behind, int B, int N, int h, int W, int K, float thr_act, void *workspace,
                                        size_t workspace_bytes, int32_t *idx, float *len, int32_t *cnt, float *records,
                                        float *rays, float *origin, voge_stream_t stream) {
  if (sigma_mode < 0 || sigma_mode > 2 || !cnt || !records || !R || !T || !focal || !pp || stripe_h <= 0 || pitch < 0)
    return VOGE_ERR_BAD_ARG;      // (origin may be NULL: nobody on the frame's path reads it)
  if ((size_t)B * h * W > 0 && !rays) return VOGE_ERR_BAD_ARG;
  const CamView cam{R, T, focal, pp, row0, stripe_h, pitch, h, W, behind ? 1 : 0, origin, rays};
  return trace_topk_fwd_impl(1, IsoView{nullptr, shared ? 1 : 0, sigma_mode}, verts, sigmas, nullptr, nullptr, nullptr, B, N, h, W, K,
                             thr_act, workspace, workspace_bytes, idx, len, nullptr, nullptr, cnt, stream, 1.0f, nullptr, nullptr, records,
                             cam);
}

// ... and for (N,3) / (N,3,3) sigmas (Renderer.py:130-137 with Aggregation.py:144-175: centred = verts - origin[b], A = 2 *
// expend_sigma(sigmas); no inverse_sigma here): the general trace with the camera AND the user's own arrays as inputs -- the
// record pass does the centring and the expansion (general_preamble_fwd_kernel's operations: the same bits), so neither the ray
// launch nor the preamble launch stands in front of the frame.  kind 1: sigmas [N | B*N][3], kind 2: [N | B*N][3][3];
// shared_verts / shared_sigmas: one set seen by every view.  records: what the deferred composite and voge_frame_bwd_gen read --
// kind 2: the packed (centred mu, A) [B*N][12]; kind 1: the compact (centred mu, a0, a1, a2, 0, 0) [B*N][8].
extern "C" int voge_frame_trace_fwd_gen(const float *verts, const float *sigmas, int shared_verts, int shared_sigmas, int kind,
                                        const float *R, const float *T, const float *focal, const float *pp, int row0, int stripe_h,
                                        int pitch, int behind, int B, int N, int h, int W, int K, float thr_act, void *workspace,
                                        size_t workspace_bytes, int32_t *idx, float *len, int32_t *cnt, float *records, float *rays,
                                        float *origin, voge_stream_t stream) {
  if ((kind != 1 && kind != 2) || !cnt || !records || !R || !T || !focal || !pp || stripe_h <= 0 || pitch < 0) return VOGE_ERR_BAD_ARG;
  if ((size_t)B * h * W > 0 && !rays) return VOGE_ERR_BAD_ARG;
  const CamView cam{R, T, focal, pp, row0, stripe_h, pitch, h, W, behind ? 1 : 0, origin, rays};
  IsoView view{nullptr, shared_verts ? 1 : 0, 0};
  view.gen_kind = kind; view.sigma_shared = shared_sigmas ? 1 : 0;
  return trace_topk_fwd_impl(0, view, verts, sigmas, nullptr, nullptr, nullptr, B, N, h, W, K, thr_act, workspace, workspace_bytes, idx, len,
                             nullptr, nullptr, cnt, stream, 1.0f, nullptr, nullptr, records, cam);
}
