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
                                         float4 *__restrict__ ms, float4 *__restrict__ ell, const IsoView view) {
  float mx, my, mz;
  const int src = view.shared ? g % N : g;
  if (view.origin != nullptr) {   // centring of Renderer.py:130 done here: the same single fp32 subtraction
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
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = isg[9 * (size_t)g + i];
  }
  const EvalRec e = make_eval(mx, my, mz, A);

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

// ------------------------------------------------------------------------------------------
// Bounding cone of a set of rays, workgroup-wide (used by the bin and the sweep kernels).
// ------------------------------------------------------------------------------------------
struct RayDir {
  float ux, uy, uz;
  bool ok;      // finite, non-zero direction
  bool unit;    // |d| == 1 within 1e-4 (the depth bound of the early exit assumes unit rays)
};
__device__ __forceinline__ RayDir ray_dir(const float dx, const float dy, const float dz) {
  RayDir r;
  const float dn2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  const float inv = 1.0f / sqrtf(dn2);
  r.ok = (dn2 > 0.0f) && (inv < 3e38f) && (inv == inv);
  r.unit = fabsf(dn2 - 1.0f) < 1e-4f;
  r.ux = dx * inv; r.uy = dy * inv; r.uz = dz * inv;
  return r;
}
// Partial (per-wave) extrema of one ray w.r.t. a given axis sum; finish with cone_finish().
__device__ __forceinline__ void cone_partial(const RayDir &u, const float ax, const float ay, const float az,
                                             float &smax, float &cmin) {
  const float cl = fmaf(u.uz, az, fmaf(u.uy, ay, u.ux * ax));
  const float rx = fmaf(-cl, ax, u.ux), ry = fmaf(-cl, ay, u.uy), rz = fmaf(-cl, az, u.uz);
  const float sl = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx * rx)));
  smax = fmaxf(smax, u.ok ? sl : 2.0f);
  cmin = fminf(cmin, u.ok ? cl : -1.0f);
}
__device__ __forceinline__ Cone cone_finish(const float ax, const float ay, const float az, const float n,
                                            const float smax, const float cmin, const bool all_ok) {
  Cone c;
  c.ax = ax; c.ay = ay; c.az = az;
  c.sn = smax * (1.0f + 1e-5f) + 1e-7f;
  c.cs = cmin - 1e-6f;
  c.ok = all_ok && (n > 1e-3f) && (cmin > 0.05f) && (c.sn == c.sn);
  return c;
}

// Depth key of a candidate for the front-to-back order of a bin: kappa = +|mu| for a Gaussian in
// front of the camera, -|mu| otherwise (behind it, or too close / cone too wide to tell), and
// -inf for an unbounded reach.  For every UNIT ray d of the cone that can hit the Gaussian
// (its line passes within `reach` R of mu): len = mu.d - v.d with |v| <= R, hence
//   front (|mu| > 4R, mu.axis > 0): len >= sqrt(|mu|^2 - R^2) - R >= kappa - 1.13 R
//   otherwise                     : len >= -|mu| - R               = kappa - R
// so  kappa - 1.13 * Rmax  (Rmax = largest finite reach in the bin) is a lower bound of len that
// is MONOTONE in kappa -- what the sweep's early exit needs.
__device__ __forceinline__ float depth_key(const float4 c, const Cone &k) {
  const float nm = sqrtf(fmaf(c.z, c.z, fmaf(c.y, c.y, c.x * c.x)));
  const float R = c.w;
  if (!(R < 3e38f) || !(nm < 3e38f)) return -INFINITY;
  float kappa = -nm;
  if (k.ok && k.cs >= 0.5f && nm > 4.0f * R) {
    const float p = fmaf(c.z, k.az, fmaf(c.y, k.ay, c.x * k.ax));
    if (p > 0.0f) kappa = nm;
  }
  return kappa;
}

// ------------------------------------------------------------------------------------------
// bin: one 1024-thread workgroup per kST x kST-pixel super-tile.  Tests every Gaussian on its
// parent region's list (bin0) against the super-tile's bounding cone (conservative), then orders the
// survivors front to back by depth key (counting sort + per-bucket insertion sort, LDS) and
// writes the (id, monotone len lower bound) list.
// More than kBinCap survivors -> count = -1 and the sweep falls back to the full stream.
// ------------------------------------------------------------------------------------------
constexpr int kST = 32;
constexpr int kBinCap = 8192;
constexpr int kBinThreads = 1024;

constexpr int kBuckets = 1024;  // == kBinThreads (one scan lane per bucket)
struct BinLds {
  uint64_t keys[kBinCap];     // (ord(depth key) << 32 | id), unordered
  uint64_t sorted[kBinCap];   // the same, front to back
  float red[16 * 4];
  int hist[kBuckets];
  uint32_t bmin[kBuckets];   // per bucket: smallest own len bound of the entries with an ellipsoid record (ord)
  int start[kBuckets + 1];
  int wsum[16];
  int count;
  int nflag;      // entries with an ellipsoid record
};

__device__ __forceinline__ void block_reduce16(float *red, const int wave, const int lane, float &a, float &b,
                                               float &c, float &d, const int mode /*0 sum,1 max/min*/) {
  // a,b,c: sum (mode 0) or a: max, b: min (mode 1); d: AND-flag as float
  if (mode == 0) { a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); }
  else { a = wave_max(a); b = wave_min(b); }
  d = wave_min(d);
  __syncthreads();
  if (lane == 0) { red[wave * 4 + 0] = a; red[wave * 4 + 1] = b; red[wave * 4 + 2] = c; red[wave * 4 + 3] = d; }
  __syncthreads();
  float ra = red[0], rb = red[1], rc = red[2], rd = red[3];
  for (int w = 1; w < kBinThreads / 64; ++w) {
    if (mode == 0) { ra += red[w * 4 + 0]; rb += red[w * 4 + 1]; rc += red[w * 4 + 2]; }
    else { ra = fmaxf(ra, red[w * 4 + 0]); rb = fminf(rb, red[w * 4 + 1]); }
    rd = fminf(rd, red[w * 4 + 3]);
  }
  a = ra; b = rb; c = rc; d = rd;
}

// Bounding cone of the rays of a pixel rectangle, computed by a whole 1024-thread workgroup.
template <int kCU>
__device__ __forceinline__ Cone region_cone(const float *__restrict__ rays, const int b, const int H, const int W,
                                            const int x0, const int y0, const int rw, const int rh, float *red,
                                            const int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int npx = rw * rh;
  const float inv_rw = 1.0f / (float)rw;
  // kCU rays per thread are loaded before any is used: a 128 x 128 region is 16 rays per thread
  // and two passes, i.e. 32 dependent round trips if taken one at a time
  auto load_rays = [&](const int i0, float (&rx)[kCU], float (&ry)[kCU], float (&rz)[kCU]) {
#pragma unroll
    for (int u = 0; u < kCU; ++u) {
      const int i = i0 + u * kBinThreads;
      rx[u] = 0.f; ry[u] = 0.f; rz[u] = 0.f;
      if (i < npx) {
        const int y = __float2int_rz(((float)i + 0.5f) * inv_rw), x = i - y * rw;
        const float *r = rays + (((size_t)b * H + y0 + y) * W + x0 + x) * 3;
        rx[u] = r[0]; ry[u] = r[1]; rz[u] = r[2];
      }
    }
  };
  // Pass 1: axis = direction of the plain vector sum (any axis gives a valid cone; rays of a
  // pinhole camera have near-equal lengths, so this is the mean direction).  Pass 2: extrema of
  // the axial cosine and of the SQUARED radial sine -- one v_rsq per ray, one sqrt per workgroup.
  // (The libm sqrt / divide per ray made a 128 x 128 region cost ~20 us of VALU time.)
  float sx = 0.f, sy = 0.f, sz = 0.f, okf = 1.f;
  for (int i0 = tid; i0 < npx; i0 += kCU * kBinThreads) {
    float rx[kCU], ry[kCU], rz[kCU];
    load_rays(i0, rx, ry, rz);
#pragma unroll
    for (int u = 0; u < kCU; ++u) {
      if (i0 + u * kBinThreads < npx) {
        const float dn2 = fmaf(rz[u], rz[u], fmaf(ry[u], ry[u], rx[u] * rx[u]));
        if (dn2 > 1e-30f && dn2 < 1e30f) { sx += rx[u]; sy += ry[u]; sz += rz[u]; } else okf = 0.f;
      }
    }
  }
  block_reduce16(red, wave, lane, sx, sy, sz, okf, 0);
  const float n = sqrtf(fmaf(sz, sz, fmaf(sy, sy, sx * sx)));
  const float ax = sx / n, ay = sy / n, az = sz / n;
  float s2max = 0.f, cmin = 1.f, dummy = 0.f, okf2 = 1.f;
  for (int i0 = tid; i0 < npx; i0 += kCU * kBinThreads) {
    float rx[kCU], ry[kCU], rz[kCU];
    load_rays(i0, rx, ry, rz);
#pragma unroll
    for (int u = 0; u < kCU; ++u) {
      if (i0 + u * kBinThreads < npx) {
        const float dn2 = fmaf(rz[u], rz[u], fmaf(ry[u], ry[u], rx[u] * rx[u]));
        const bool ok = dn2 > 1e-30f && dn2 < 1e30f;
        const float inv = __builtin_amdgcn_rsqf(dn2);
        const float da = fmaf(rz[u], az, fmaf(ry[u], ay, rx[u] * ax));
        const float qx = fmaf(-da, ax, rx[u]), qy = fmaf(-da, ay, ry[u]), qz = fmaf(-da, az, rz[u]);
        const float s2 = fmaf(qz, qz, fmaf(qy, qy, qx * qx)) * (inv * inv);
        s2max = fmaxf(s2max, ok ? s2 : 4.0f);
        cmin = fminf(cmin, ok ? da * inv : -1.0f);
      }
    }
  }
  block_reduce16(red, wave, lane, s2max, cmin, dummy, okf2, 1);
  // v_rsq is good to ~1 ulp: pad the bounds by 4e-7 relative on top of cone_finish's margins
  return cone_finish(ax, ay, az, n / (float)max(npx, 1), sqrtf(s2max) * (1.0f + 4e-7f) + 4e-7f, cmin - 4e-7f, okf != 0.f);
}

// Cones of all kST x kST super-tiles, computed once: bin_kernel reads its own, bin0_kernel
// composes its region's cone from the <= 16 children instead of touching 16k rays again.
struct ConeRec {
  float ax, ay, az, cs, sn, ok, pad0, pad1;
};

// One launch for the two independent preparation steps: the first blocks derive the per-Gaussian
// records, the rest compute super-tile cones (the first of them also clears the region counters).
__global__ void __launch_bounds__(kBinThreads)
prep_cone_kernel(const float *__restrict__ rays, const int H, const int W, const int nstx, const int nst, const int B,
                 ConeRec *__restrict__ cones /* [B][nst] */, int *__restrict__ c_count, const int n_count,
                 const float *__restrict__ mus, const float *__restrict__ isg, const float *__restrict__ cam_fwd,
                 const int N, const int P, const float thr_act, const int iso_in, float4 *__restrict__ cull,
                 float4 *__restrict__ evr, float4 *__restrict__ ms, float4 *__restrict__ ell, const IsoView view) {
  __shared__ float red[16 * 4];
  const int tid = threadIdx.x;
  const int nprep = (P + kBinThreads - 1) / kBinThreads;
  if ((int)blockIdx.x < nprep) {     // record blocks first: they are the longer ones (fp64 eigenvalue)
    const int g = (int)blockIdx.x * kBinThreads + tid;
    if (g < P) prep_one(g, mus, isg, cam_fwd, N, thr_act, iso_in, cull, evr, ms, ell, view);
    return;
  }
  const int cb = (int)blockIdx.x - nprep;
  if (cb == 0)
    for (int i = tid; i < n_count; i += kBinThreads) c_count[i] = 0;
  const int b = cb / nst, bin = cb - b * nst;
  const int stx = bin % nstx, sty = bin / nstx;
  const int x0 = stx * kST, y0 = sty * kST;
  const Cone c = region_cone<1>(rays, b, H, W, x0, y0, min(kST, W - x0), min(kST, H - y0), red, tid);   // 1 ray per thread
  if (tid == 0) cones[cb] = ConeRec{c.ax, c.ay, c.az, c.cs, c.sn, c.ok ? 1.f : 0.f, 0.f, 0.f};
}

__device__ __forceinline__ Cone load_cone(const ConeRec &r) {
  Cone c;
  c.ax = r.ax; c.ay = r.ay; c.az = r.az; c.cs = r.cs; c.sn = r.sn; c.ok = r.ok != 0.f;
  return c;
}

// Conservative union of child cones: a ray of child i makes at most alpha_i + theta_i with the
// parent axis (alpha_i = angle between the axes), so
//   cos >= cos(alpha_i) cs_i - sin(alpha_i) sn_i ,   sin <= sin(alpha_i) + cos(alpha_i) sn_i .
__device__ __forceinline__ Cone compose_cones(const ConeRec *__restrict__ ch, const int nstx, const int cx0, const int cy0,
                                              const int ncx, const int ncy) {
  float sx = 0.f, sy = 0.f, sz = 0.f;
  bool ok = true;
  for (int j = 0; j < ncy; ++j)
    for (int i = 0; i < ncx; ++i) {
      const ConeRec r = ch[(cy0 + j) * nstx + cx0 + i];
      sx += r.ax; sy += r.ay; sz += r.az;
      ok = ok && (r.ok != 0.f);
    }
  const float n = sqrtf(fmaf(sz, sz, fmaf(sy, sy, sx * sx)));
  const float ax = sx / n, ay = sy / n, az = sz / n;
  float smax = 0.f, cmin = 1.f;
  for (int j = 0; j < ncy; ++j)
    for (int i = 0; i < ncx; ++i) {
      const ConeRec r = ch[(cy0 + j) * nstx + cx0 + i];
      const float ca = fmaf(r.az, az, fmaf(r.ay, ay, r.ax * ax));
      const float qx = fmaf(-ca, ax, r.ax), qy = fmaf(-ca, ay, r.ay), qz = fmaf(-ca, az, r.az);
      const float sa = sqrtf(fmaf(qz, qz, fmaf(qy, qy, qx * qx))) * (1.0f + 1e-6f) + 1e-7f;
      const float cl = fminf(ca, 1.0f) - 1e-7f;
      if (!(cl > 0.0f)) ok = false;
      cmin = fminf(cmin, fmaf(cl, r.cs, -sa * r.sn));
      smax = fmaxf(smax, fmaf(fminf(ca + 1e-7f, 1.0f), r.sn, sa));
    }
  return cone_finish(ax, ay, az, n / (float)(ncx * ncy), smax, cmin, ok);
}

// ------------------------------------------------------------------------------------------
// bin0: the coarse level.  A kST0 x kST0-pixel region is covered by kBin0Split workgroups, each
// testing its slice of the batch element's Gaussians against the region's cone and appending the
// survivors' ids (unordered) to the region's list.  bin_kernel then scans its parent region's
// list (~N/10) instead of all N Gaussians: the scan work drops from nst*N to nst0*N + nst*N/10.
// ------------------------------------------------------------------------------------------
constexpr int kST0 = 128;
#ifndef VOGE_BIN0_SPLIT
#define VOGE_BIN0_SPLIT 8
#endif
constexpr int kBin0Split = VOGE_BIN0_SPLIT;

__global__ void __launch_bounds__(kBinThreads)
bin0_kernel(const float4 *__restrict__ cull, const ConeRec *__restrict__ cones, const int nstx, const int nsty,
            const int N, const int nst0x, int *__restrict__ c_count /* zeroed */,
            int32_t *__restrict__ c_id /* [B*nst0][N] */) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int region = blockIdx.x / kBin0Split, part = blockIdx.x - region * kBin0Split, b = blockIdx.y;
  const int stx = region % nst0x, sty = region / nst0x;
  constexpr int kCh = kST0 / kST;   // children per side
  const int cx0 = stx * kCh, cy0 = sty * kCh;
  const Cone cone = compose_cones(cones + (size_t)b * nstx * nsty, nstx, cx0, cy0, min(kCh, nstx - cx0), min(kCh, nsty - cy0));
  const int nreg = gridDim.x / kBin0Split;
  int *cnt = c_count + (size_t)b * nreg + region;
  int32_t *out = c_id + ((size_t)b * nreg + region) * N;
  const float4 *cullb = cull + (size_t)b * N;
  const int per = (N + kBin0Split - 1) / kBin0Split;
  const int g0 = part * per, g1 = min(N, g0 + per);
  // Rounds of up to 32 candidates per thread: test them (keep flags in one register, 8 loads in
  // flight), reserve the round's output range with ONE global atomic per workgroup, then write
  // the ids wave by wave.  (One atomic per wave-ballot on the region counter serialises in L2.)
  __shared__ int wtot[kBinThreads / 64];
  __shared__ int wg_base;
  const int wave = tid >> 6;
  constexpr int kRound = 32, kU = 8;
  for (int r0 = g0; r0 < g1; r0 += kRound * kBinThreads) {
    unsigned flags = 0;
    int wcount = 0;   // wave-uniform
#pragma unroll
    for (int q0 = 0; q0 < kRound; q0 += kU) {
      if (r0 + q0 * kBinThreads >= g1) break;   // uniform
      float4 c[kU];
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        const int g = r0 + (q0 + q) * kBinThreads + tid;
        c[q] = (g < g1) ? cullb[g] : make_float4(0.f, 0.f, 0.f, -1.f);
      }
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        const bool keep = cone_keep(c[q], cone);
        flags |= keep ? (1u << (q0 + q)) : 0u;
        wcount += __popcll(__ballot(keep));
      }
    }
    __syncthreads();   // previous round's wtot / wg_base fully consumed
    if (lane == 0) wtot[wave] = wcount;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < kBinThreads / 64; ++w) t += wtot[w];
      wg_base = (t > 0) ? atomicAdd(cnt, t) : 0;
    }
    __syncthreads();
    int off = wg_base;
    for (int w = 0; w < wave; ++w) off += wtot[w];
    for (int q = 0; q < kRound; ++q) {
      if (r0 + q * kBinThreads >= g1) break;   // uniform
      const bool keep = (flags >> q) & 1u;
      const unsigned long long m = __ballot(keep);
      if (keep) out[off + __popcll(m & ((1ull << lane) - 1ull))] = r0 + q * kBinThreads + tid;
      off += __popcll(m);
    }
  }
}

#ifdef VOGE_BIN_TIMES
__device__ unsigned long long g_bin_times[1024 * 8];   // per super-tile: start, scan done, reduced, hist, scanned, scattered, end
#define BIN_TS(k) if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 1024) g_bin_times[8 * blockIdx.x + (k)] = wall_clock64()
#else
#define BIN_TS(k)
#endif
#ifndef VOGE_ELL_KEY
#define VOGE_ELL_KEY 0
#endif
__global__ void __launch_bounds__(kBinThreads)
bin_kernel(const float4 *__restrict__ cull, const float4 *__restrict__ ell, const float4 *__restrict__ evr,
           const ConeRec *__restrict__ cones, const int *__restrict__ c_count,
           const int32_t *__restrict__ c_id, const int nst0x, const int nst0, const int N, const int H,
           const int W, const int nstx, int *__restrict__ bin_count, int32_t *__restrict__ bin_id,
           float *__restrict__ bin_lb, float4 *__restrict__ bin_rec) {
  __shared__ BinLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int stx = blockIdx.x % nstx, sty = blockIdx.x / nstx, b = blockIdx.y;
  const int x0 = stx * kST, y0 = sty * kST;
  const Cone cone = load_cone(cones[(size_t)b * gridDim.x + blockIdx.x]);

  BIN_TS(0);
  // ---- scan all Gaussians of this batch element, keep (bound, id) of the survivors ----
  if (tid == 0) { L.count = 0; L.nflag = 0; }
  __syncthreads();
  const float4 *cullb = cull + (size_t)b * N;
  const float4 *ellb = ell + (size_t)b * N * 2;
  float rmax = 0.0f;   // largest finite reach among this thread's survivors WITHOUT an ellipsoid record
  float klo = INFINITY, khi = -INFINITY;   // extrema of the finite order keys this thread appended
  constexpr int kScanU = 8;
  // candidates: the parent region's list, or (small problems, no coarse level) every Gaussian
  const int parent = b * nst0 + (y0 / kST0) * nst0x + x0 / kST0;
  const int n_src = (c_count != nullptr) ? c_count[parent] : N;
  const int32_t *src = (c_count != nullptr) ? c_id + (size_t)parent * N : nullptr;
  for (int base = 0; base < n_src; base += kScanU * kBinThreads) {
    // eight independent gathers in flight per lane: the scan is latency-, not compute-bound
    float4 c[kScanU];
    int gid[kScanU];
#pragma unroll
    for (int q = 0; q < kScanU; ++q) {
      const int i = base + q * kBinThreads + tid;
      gid[q] = (i < n_src) ? (src != nullptr ? src[i] : i) : -1;
    }
#pragma unroll
    for (int q = 0; q < kScanU; ++q) c[q] = (gid[q] >= 0) ? cullb[gid[q]] : make_float4(0.f, 0.f, 0.f, -1.f);
    bool kp[kScanU], el[kScanU];
#pragma unroll
    for (int q = 0; q < kScanU; ++q) { kp[q] = cone_keep(c[q], cone); el[q] = kp[q] && cull_has_ell(c[q]); }
    float gkey[kScanU];
#pragma unroll
    for (int q = 0; q < kScanU; ++q) gkey[q] = 0.0f;
    if (__any(el[0] || el[1] || el[2] || el[3] || el[4] || el[5] || el[6] || el[7])) {
      // anisotropic survivors of the sphere test: separating-plane test against their ellipsoid,
      // four at a time (the records of all eight would not fit the 128 registers of a 1024-thread workgroup)
#pragma unroll
      for (int h = 0; h < kScanU; h += 4) {
        float4 e0[4], e1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (el[h + q]) { e0[q] = ellb[2 * (size_t)gid[h + q]]; e1[q] = ellb[2 * (size_t)gid[h + q] + 1]; }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (el[h + q]) {
            kp[h + q] = cone_keep_ell(c[h + q], e0[q], e1[q], cone);
            el[h + q] = kp[h + q];
            // order key: VOGE_ELL_KEY 0 = the centre's depth along the axis, 1 = the entry's own lower bound of len
            const float pa = fmaf(c[h + q].z, cone.az, fmaf(c[h + q].y, cone.ay, c[h + q].x * cone.ax));
#if VOGE_ELL_KEY == 1
            gkey[h + q] = pa - ell_support(e0[q], e1[q], cone.ax, cone.ay, cone.az) + 0.0f;
#else
            gkey[h + q] = pa + 0.0f;
#endif
          }
        asm volatile("" ::: "memory");
      }
    }
    // ONE LDS atomic per wave and trip reserves the slots of all kScanU ballots (it was one per ballot: a chain
    // of eight LDS round trips per trip); the keys' extrema ride along in registers for the bucket mapping.
    unsigned long long mq[kScanU];
    int wtot = 0;
    bool wave_flag = false;
#pragma unroll
    for (int q = 0; q < kScanU; ++q) {
      mq[q] = __ballot(kp[q]);
      wtot += __popcll(mq[q]);
      wave_flag = wave_flag || (kp[q] && el[q]);
    }
    wave_flag = __any(wave_flag);
    if (wtot) {       // uniform
      int start = 0;
      if (lane == 0) {
        start = atomicAdd(&L.count, wtot);
        if (wave_flag) L.nflag = 1;
      }
      start = __shfl(start, 0, 64);
#pragma unroll
      for (int q = 0; q < kScanU; ++q) {
        const int slot = start + __popcll(mq[q] & ((1ull << lane) - 1ull));
        start += __popcll(mq[q]);
        if (kp[q]) {
          if (!el[q] && c[q].w < 3e38f) rmax = fmaxf(rmax, c[q].w);
          // bit 31 of the id word: the entry has an ellipsoid record (its len bound is its own)
          const float key = el[q] ? gkey[q] : depth_key(c[q], cone);
          if (key > -INFINITY) { klo = fminf(klo, key); khi = fmaxf(khi, key); }
          if (slot < kBinCap) L.keys[slot] = ((uint64_t)f2ord(key) << 32) | (uint32_t)gid[q] | (el[q] ? 0x80000000u : 0u);
        }
      }
    }
  }
  __syncthreads();
  BIN_TS(1);
  const int total = L.count;
  const int bin = b * gridDim.x + blockIdx.x;
  if (total > kBinCap) {
    if (tid == 0) bin_count[bin] = -1;
    return;
  }
  // ---- order the survivors front to back: counting sort on the depth key, then an insertion
  // sort inside every bucket (a handful of entries each).  Exact order is not needed for
  // correctness (the sweep's top-K insertion is order independent) but it turns nearly every
  // insertion into an append.  Bucket 0 collects the -inf keys (unbounded reach).
  // one workgroup reduction for (largest key, smallest key, largest sphere reach)
  float hi = wave_max(khi), lo = wave_min(klo), rm = wave_max(rmax);
  if (lane == 0) { L.red[wave * 4 + 0] = hi; L.red[wave * 4 + 1] = lo; L.red[wave * 4 + 2] = rm; }
  __syncthreads();
  hi = L.red[0]; lo = L.red[1]; rm = L.red[2];
  for (int w = 1; w < kBinThreads / 64; ++w) {
    hi = fmaxf(hi, L.red[w * 4 + 0]); lo = fminf(lo, L.red[w * 4 + 1]); rm = fmaxf(rm, L.red[w * 4 + 2]);
  }
  BIN_TS(2);
  const float span = fmaxf(hi - lo, 1e-20f);
  const float scale = (float)(kBuckets - 2) / span;
  const bool flagged = L.nflag != 0;     // any entry with an ellipsoid record (workgroup-uniform)
  for (int i = tid; i < kBuckets; i += kBinThreads) { L.hist[i] = 0; L.bmin[i] = f2ord(INFINITY); }
  __syncthreads();
  for (int i = tid; i < total; i += kBinThreads) {
    const uint64_t k = L.keys[i];
    const float v = ord2f((uint32_t)(k >> 32));
    int q = 0;
    if (v > -INFINITY) q = 1 + min(kBuckets - 2, max(0, (int)((v - lo) * scale)));
    atomicAdd(&L.hist[q], 1);
    if ((uint32_t)k & 0x80000000u) {
      // own lower bound of len: the peak point x = len d of a hit lies in the ellipsoid, so
      // len (d.a) = x.a >= t = mu.a - h(a), and d.a is in [cs, 1]
      const int g = (int)((uint32_t)k & 0x7fffffffu);
      const float4 c = cullb[g], e0 = ellb[2 * (size_t)g], e1 = ellb[2 * (size_t)g + 1];
      float bnd = -INFINITY;
      if (cone.ok) {
        const float pa = fmaf(c.z, cone.az, fmaf(c.y, cone.ay, c.x * cone.ax));
        const float nm1 = fabsf(c.x) + fabsf(c.y) + fabsf(c.z);
        const float t = pa - ell_support(e0, e1, cone.ax, cone.ay, cone.az) - 4e-6f * nm1;
        bnd = (t >= 0.0f) ? t : t / cone.cs;
        bnd = bnd - 1e-5f * fabsf(bnd) - 1e-30f;
      }
      atomicMin(&L.bmin[q], f2ord(bnd));
    }
  }
  __syncthreads();
  BIN_TS(3);
  // exclusive scan of kBuckets (== kBinThreads) counters: wave scan + wave offsets
  {
    const int v = L.hist[tid];
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) L.wsum[wave] = x;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += L.wsum[w];
    L.hist[tid] = off + x - v;
    L.start[tid] = off + x - v;
    if (tid == kBuckets - 1) L.start[kBuckets] = off + x;
  }
  __syncthreads();
  BIN_TS(4);
  for (int i = tid; i < total; i += kBinThreads) {
    const uint64_t k = L.keys[i];
    const float v = ord2f((uint32_t)(k >> 32));
    int q = 0;
    if (v > -INFINITY) q = 1 + min(kBuckets - 2, max(0, (int)((v - lo) * scale)));
    L.sorted[atomicAdd(&L.hist[q], 1)] = k;
  }
  __syncthreads();
  BIN_TS(5);
#ifdef VOGE_BIN_FULLSORT
  {  // thread q orders bucket q (keys are unique: the id is in the low word)
    const int s0 = L.start[tid], s1 = L.start[tid + 1];
    for (int i = s0 + 1; i < s1; ++i) {
      const uint64_t k = L.sorted[i];
      int j = i;
      while (j > s0 && L.sorted[j - 1] > k) { L.sorted[j] = L.sorted[j - 1]; --j; }
      L.sorted[j] = k;
    }
  }
  __syncthreads();
#endif
  // Entries are ordered by BUCKET only (1022 buckets over the bin's depth range, i.e. a few
  // thousandths of a scene unit each -- far finer than the reach that separates kappa from the
  // actual len), and every entry carries its bucket's lower edge as the len bound: monotone along
  // the list, which is all the sweep's early exit needs.  The order inside a bucket is whatever
  // the LDS atomics produced; the sweep's top-K is order independent.
  // suffix minimum over the buckets of the flagged entries' own bounds (tid <-> bucket)
  if (flagged) {
    uint32_t x = L.bmin[tid];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_down(x, o, 64);
      if (lane + o < 64) x = min(x, y);
    }
    if (lane == 0) L.wsum[wave] = (int)x;
    __syncthreads();
    for (int w = wave + 1; w < kBinThreads / 64; ++w) x = min(x, (uint32_t)L.wsum[w]);
    __syncthreads();
    L.bmin[tid] = x;
  }
  __syncthreads();
  int32_t *oid = bin_id + (size_t)bin * kBinCap;
  float *olb = bin_lb + (size_t)bin * kBinCap;
  float4 *orec = bin_rec + (size_t)bin * kBinCap;
  const float slack = 1.13f * rm * (1.0f + 1e-5f);
  const float inv_scale = span / (float)(kBuckets - 2);
  for (int i = tid; i < total; i += kBinThreads) {
    const uint64_t k = L.sorted[i];
    const float v = ord2f((uint32_t)(k >> 32));
    float edge = -INFINITY;
    int qb = 0;
    if (v > -INFINITY) {
      const int q = min(kBuckets - 2, max(0, (int)((v - lo) * scale)));
      edge = fminf(v, lo + (float)q * inv_scale - 4e-6f * span);
      qb = 1 + q;
    }
    oid[i] = (int32_t)((uint32_t)k & 0x7fffffffu);
    // the entry's cull record rides along in list order: the 16 sweep tiles of this super-tile then
    // stream it (16 coalesced bytes per entry) instead of gathering it by id once each
    orec[i] = cullb[(uint32_t)k & 0x7fffffffu];
    // sphere-only entries: bucket edge minus the largest sphere reach; entries with an ellipsoid
    // record: the smallest own bound from this bucket on.  Both are monotone along the list.
    const float lb_sphere = (edge > -INFINITY) ? edge - slack - 1e-5f * fabsf(edge) - 1e-30f : -INFINITY;
    olb[i] = flagged ? fminf(lb_sphere, ord2f(L.bmin[qb])) : lb_sphere;
  }
  BIN_TS(6);
  if (tid == 0) bin_count[bin] = total;
}

// ------------------------------------------------------------------------------------------
// bin2: one 256-thread workgroup per sweep tile (TW x TH pixels).  Filters the sorted list of
// the tile's super-tile with the tile's own bounding cone, keeping the order, so that the sweep
// streams only the ~10 % of the super-tile list that can touch its pixels.  More than kTileCap
// survivors (or an overflowed parent) -> count = -1 and the sweep falls back to the parent list.
// ------------------------------------------------------------------------------------------
constexpr int kTileCap = 2048;

__global__ void __launch_bounds__(256)
bin2_kernel(const float4 *__restrict__ bin_rec, const float4 *__restrict__ ell, const float *__restrict__ rays, const int *__restrict__ bin_count,
            const int32_t *__restrict__ bin_id, const float *__restrict__ bin_lb, const int nstx, const int nst,
            const int N, const int H, const int W, const int TW, const int TH, int *__restrict__ tl_count,
            int32_t *__restrict__ tl_id, float *__restrict__ tl_lb, const int K, int32_t *__restrict__ out_idx,
            float *__restrict__ out_len, float *__restrict__ out_act, float *__restrict__ out_dsd,
            int32_t *__restrict__ out_cnt) {
  __shared__ float red[4 * 8];
  __shared__ int wcnt[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + TW - 1) / TW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int tile = b * gridDim.x + blockIdx.x;
  const int bin = b * nst + ((ty * TH) / kST) * nstx + (tx * TW) / kST;
  const int bc = bin_count[bin];
  if (bc < 0) {
    if (tid == 0) tl_count[tile] = -1;
    return;
  }
  // The first trip's ids and records do not depend on the cone: their two dependent round trips
  // overlap the ray loads and the two reductions of the cone.
  const int32_t *src_id = bin_id + (size_t)bin * kBinCap;
  const float4 *src_rec = bin_rec + (size_t)bin * kBinCap;
  int id[4];
  float4 c[4];
  auto load_trip = [&](const int base) {   // ids and records are both streams: one round trip
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int g = base + q * 256 + tid;
      id[q] = (g < bc) ? src_id[g] : -1;
      c[q] = (g < bc) ? src_rec[g] : make_float4(0.f, 0.f, 0.f, -1.f);
    }
  };
  load_trip(0);
  // ---- cone of the tile's rays (thread <-> pixel, clamped at the image border) ----
  const int lx = tid % TW, ly = tid / TW;
  const bool has = ly < TH;
  const int px = min(tx * TW + lx, W - 1), py = min(ty * TH + min(ly, TH - 1), H - 1);
  const float *r = rays + (((size_t)b * H + py) * W + px) * 3;
  const RayDir u = ray_dir(r[0], r[1], r[2]);
  float sx = wave_sum((has && u.ok) ? u.ux : 0.f), sy = wave_sum((has && u.ok) ? u.uy : 0.f),
        sz = wave_sum((has && u.ok) ? u.uz : 0.f);
  const bool wok = __all(!has || u.ok);
  if (lane == 0) { red[wave * 8 + 0] = sx; red[wave * 8 + 1] = sy; red[wave * 8 + 2] = sz; red[wave * 8 + 3] = wok ? 1.f : 0.f; }
  __syncthreads();
  sx = red[0] + red[8] + red[16] + red[24];
  sy = red[1] + red[9] + red[17] + red[25];
  sz = red[2] + red[10] + red[18] + red[26];
  const bool all_ok = (red[3] != 0.f) && (red[11] != 0.f) && (red[19] != 0.f) && (red[27] != 0.f);
  const float n = sqrtf(fmaf(sz, sz, fmaf(sy, sy, sx * sx)));
  const float ax = sx / n, ay = sy / n, az = sz / n;
  float smax = 0.f, cmin = 1.f;
  if (has) cone_partial(u, ax, ay, az, smax, cmin);
  smax = wave_max(smax); cmin = wave_min(cmin);
  if (lane == 0) { red[wave * 8 + 4] = smax; red[wave * 8 + 5] = cmin; }
  __syncthreads();
  for (int w = 0; w < 4; ++w) { smax = fmaxf(smax, red[w * 8 + 4]); cmin = fminf(cmin, red[w * 8 + 5]); }
  const Cone cone = cone_finish(ax, ay, az, n, smax, cmin, all_ok);

  // ---- ordered filter of the parent list ----
  const float *src_lb = bin_lb + (size_t)bin * kBinCap;
  const float4 *ellb = ell + (size_t)b * N * 2;
  int32_t *oid = tl_id + (size_t)tile * kTileCap;
  float *olb = tl_lb + (size_t)tile * kTileCap;
  int total = 0, par = 0;
  for (int base = 0; base < bc; base += 1024) {
    // four chunks per trip: 4 independent (id -> record) chains in flight per lane
    if (base > 0) load_trip(base);
    bool kp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) kp[q] = cone_keep(c[q], cone);
    // anisotropic survivors of the sphere test: the ellipsoid test, two candidates at a time (their
    // records in flight together; all four would cost the kernel three waves per SIMD of occupancy)
    bool el[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) el[q] = kp[q] && cull_has_ell(c[q]);
#pragma unroll
    for (int h = 0; h < 4; h += 2) {
      if (__any(el[h] || el[h + 1])) {
        float4 e0[2], e1[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (el[h + q]) { e0[q] = ellb[2 * (size_t)id[h + q]]; e1[q] = ellb[2 * (size_t)id[h + q] + 1]; }
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (el[h + q]) kp[h + q] = cone_keep_ell(c[h + q], e0[q], e1[q], cone);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (base + q * 256 >= bc) break;  // uniform
      const bool keep = kp[q];
      const unsigned long long m = __ballot(keep);
      if (lane == 0) wcnt[par][wave] = __popcll(m);
      __syncthreads();
      int off = total, tot = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int cw = wcnt[par][w];
        if (w < wave) off += cw;
        tot += cw;
      }
      if (keep) {
        const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < kTileCap) { oid[pos] = id[q]; olb[pos] = src_lb[base + q * 256 + tid]; }   // (a survivor's bound: loaded only now)
      }
      total += tot;
      par ^= 1;
    }
  }
  if (tid == 0) tl_count[tile] = (total > kTileCap) ? -1 : total;
  // A tile nothing can hit gets its outputs right here: this kernel runs at full occupancy and
  // is latency-bound, so the stores are free, whereas in the sweep (a few waves per CU) the empty
  // tiles were a pure write phase of ~8 us each.  The sweep returns at once on a zero count.
  if (total == 0) {
    const int tw = min(TW, W - tx * TW), th = min(TH, H - ty * TH);
    const int row_items = tw * K;
    for (int r = 0; r < th; ++r) {
      const size_t pix0 = ((size_t)b * H + ty * TH + r) * W + (size_t)tx * TW;
      if ((K & 3) == 0) {
        for (int j4 = tid; j4 < row_items / 4; j4 += 256) {
          const size_t o = pix0 * K + (size_t)j4 * 4;
          *reinterpret_cast<int4 *>(out_idx + o) = make_int4(-1, -1, -1, -1);
          *reinterpret_cast<float4 *>(out_len + o) = make_float4(VOGE_SENT_LEN, VOGE_SENT_LEN, VOGE_SENT_LEN, VOGE_SENT_LEN);
          *reinterpret_cast<float4 *>(out_act + o) = make_float4(VOGE_SENT_ACT, VOGE_SENT_ACT, VOGE_SENT_ACT, VOGE_SENT_ACT);
          *reinterpret_cast<float4 *>(out_dsd + o) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        for (int j = tid; j < row_items; j += 256) {
          const size_t o = pix0 * K + j;
          out_idx[o] = -1; out_len[o] = VOGE_SENT_LEN; out_act[o] = VOGE_SENT_ACT; out_dsd[o] = 0.0f;
        }
      }
      if (out_cnt != nullptr && tid < tw) out_cnt[pix0 + tid] = 0;
    }
  }
}

// ------------------------------------------------------------------------------------------
// tile_order: launch order of the sweep = tiles by descending candidate count (counting sort into
// 256 buckets of 8).  The sweep's residency is small (LDS top-K lists), tile costs span 10-80 us,
// and a heavy tile that starts late is the kernel's tail; longest-first removes that tail.  The
// sweep's results do not depend on the order.
// ------------------------------------------------------------------------------------------
#ifndef VOGE_ORDER_BITS
#define VOGE_ORDER_BITS 6
#endif
constexpr int kOrderBits = VOGE_ORDER_BITS;
constexpr int kOrderClasses = 1 << kOrderBits;
constexpr int kOrderPer = 16;   // counts per thread kept in registers (one memory round trip)
__global__ void __launch_bounds__(1024)
tile_order_kernel(const int *__restrict__ tl_count, const int ntile, int *__restrict__ order) {
  // Partition into kOrderClasses (64) classes of the count range, heaviest first (16 -> 64 classes: sweep
  // 67 -> 64 us at cfg3, the order is closer to longest-first), with ballots only (no atomics:
  // thousands of tiles with near-equal counts would serialise on a handful of LDS addresses).
  // Position = class base + this wave's base within the class + rank inside the ballot: a pure
  // function of the counts (deterministic).  Chunks of 16384 tiles; chunks are ordered one
  // after the other (a frame has more than one only beyond 1024^2 pixels or in batches).
  __shared__ int red[16];
  __shared__ float redf[16];
  __shared__ int wtot[kOrderClasses][16];
  __shared__ int ctot[kOrderClasses];
  __shared__ int cexc[kOrderClasses];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c0 = 0; c0 < ntile; c0 += 1024 * kOrderPer) {
    const int n = min(ntile - c0, 1024 * kOrderPer);
    int cnt[kOrderPer];
#pragma unroll
    for (int u = 0; u < kOrderPer; ++u) {
      const int i = u * 1024 + tid;
      cnt[u] = (i < n) ? tl_count[c0 + i] : 0;
    }
    int cmax = 0;
    float csum = 0.0f;
#pragma unroll
    for (int u = 0; u < kOrderPer; ++u) {
      cmax = max(cmax, cnt[u] < 0 ? (1 << 20) : cnt[u]);
      csum += (float)max(cnt[u], 0);
    }
    cmax = (int)wave_max((float)cmax);
    csum = wave_sum(csum);
    __syncthreads();   // previous chunk done with the shared arrays
    if (lane == 0) { red[wave] = cmax; redf[wave] = csum; }
    __syncthreads();
    cmax = red[0];
    csum = redf[0];
    for (int w = 1; w < 16; ++w) { cmax = max(cmax, red[w]); csum += redf[w]; }
    // Balanced chunk (mean list at least half the longest): keep the spatial launch order -- there
    // is no tail to remove and neighbouring tiles share their super-tile's list and Gaussians in L2.
    if (csum * 2.0f >= (float)cmax * (float)n) {
#pragma unroll
      for (int u = 0; u < kOrderPer; ++u)
        if (u * 1024 + tid < n) order[c0 + u * 1024 + tid] = c0 + u * 1024 + tid;
      continue;   // uniform
    }
    const float scale = (float)kOrderClasses / ((float)cmax + 1.0f);
    int cls[kOrderPer];
    // lanes of the wave that hold the same class as this lane, from the kOrderBits bit-ballots of the
    // class id (instead of one ballot per class)
    auto same_class = [&](const int c) {
      unsigned long long same = __ballot(c >= 0);
#pragma unroll
      for (int bit = 0; bit < kOrderBits; ++bit) {
        const unsigned long long mb = __ballot((c >> bit) & 1);
        same &= ((c >> bit) & 1) ? mb : ~mb;
      }
      return same;
    };
    for (int i = tid; i < kOrderClasses * 16; i += 1024) (&wtot[0][0])[i] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kOrderPer; ++u) {
      cls[u] = (u * 1024 + tid >= n) ? -1
               : (cnt[u] < 0)        ? 0
                                     : (kOrderClasses - 1) - min(kOrderClasses - 1, (int)((float)cnt[u] * scale));
      if (u * 1024 < n) {   // uniform
        const unsigned long long same = same_class(cls[u]);
        if (cls[u] >= 0 && (same & ((1ull << lane) - 1ull)) == 0) wtot[cls[u]][wave] += __popcll(same);   // class leader
      }
    }
    __syncthreads();
    // exclusive prefix over (class, wave): one thread per class scans its 16 waves, then thread 0
    // chains the class totals (two short dependent chains instead of one of 256 LDS round trips)
    if (tid < kOrderClasses) {
      int v[16], run = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) v[w] = wtot[tid][w];
#pragma unroll
      for (int w = 0; w < 16; ++w) { const int c = v[w]; v[w] = run; run += c; }
#pragma unroll
      for (int w = 0; w < 16; ++w) wtot[tid][w] = v[w];
      ctot[tid] = run;
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int q = 0; q < kOrderClasses; ++q) { cexc[q] = run; run += ctot[q]; }
    }
    __syncthreads();
    // wtot[class][wave] now holds this wave's running write offset inside the class
#pragma unroll
    for (int u = 0; u < kOrderPer; ++u) {
      if (u * 1024 >= n) break;   // uniform
      const unsigned long long same = same_class(cls[u]);
      if (cls[u] >= 0) {
        const int rank = __popcll(same & ((1ull << lane) - 1ull));
        const int off = wtot[cls[u]][wave];
        order[c0 + cexc[cls[u]] + off + rank] = c0 + u * 1024 + tid;
        if (rank == 0) wtot[cls[u]][wave] = off + __popcll(same);
      }
    }
  }
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
                 const float *__restrict__ tl_lb, const int *__restrict__ tile_order, const int nstx, const int nst,
                 const int N, const int H,
                 const int W, const int K, const float thr_act, int32_t *__restrict__ out_idx,
                 float *__restrict__ out_len, float *__restrict__ out_act, float *__restrict__ out_dsd,
                 int32_t *__restrict__ out_cnt) {
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
  // heavy tiles first: the launch order is a permutation of (batch element, tile) sorted by list length
  int lin = blockIdx.y * gridDim.x + blockIdx.x;
  if (tile_order != nullptr) lin = tile_order[lin];
  if (tl_count != nullptr && tl_count[lin] == 0) return;   // bin2 already wrote this tile's all-sentinel outputs
  const int b = lin / (int)gridDim.x, bx = lin - b * (int)gridDim.x;
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
  const float wsx = wave_sum(u.ok ? u.ux : 0.f), wsy = wave_sum(u.ok ? u.uy : 0.f), wsz = wave_sum(u.ok ? u.uz : 0.f);
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
  const int bin = b * nst + ((ty * TH) / kST) * nstx + (tx * TW) / kST;
  const int tc = (tl_count != nullptr) ? tl_count[tile] : -1;
  const int bc = (tc >= 0) ? tc : ((bin_count != nullptr) ? bin_count[bin] : -1);
  const bool binned = bc >= 0;
  const int src_n = binned ? bc : N;
  const int32_t *src_id = (tc >= 0) ? tl_id + (size_t)tile * kTileCap : (binned ? bin_id + (size_t)bin * kBinCap : nullptr);
  const float *src_lb = (tc >= 0) ? tl_lb + (size_t)tile * kTileCap : (binned ? bin_lb + (size_t)bin * kBinCap : nullptr);
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
          topk_commit(mykeys, TP, K, cnt, worst, tail, key, take);
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
      const float4 cc = cull[oi];
      const EvalRec e = unpack_eval(evr[(size_t)oi * 3 + 0], evr[(size_t)oi * 3 + 1], evr[(size_t)oi * 3 + 2]);
      const PairOut o = pair_eval(cc.x, cc.y, cc.z, e, ex, ey, ez, ex * ex, ey * ey, ez * ez, ex * ey, ex * ez, ey * ez);
      ol = ord2f((uint32_t)(key >> 32));
      oa = o.act;
      od = o.dsd;
    }
  };
  const bool vec4 = ((K & 3) == 0);   // rows of K floats stay 16-byte aligned: 16-byte stores
  if (vec4) {
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
    for (int it0 = tid; tile_gen && it0 < nitem; it0 += T * kEpiG) {
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
        *reinterpret_cast<int4 *>(out_idx + ob[u]) = make_int4(oi[0], oi[1], oi[2], oi[3]);
        *reinterpret_cast<float4 *>(out_len + ob[u]) = make_float4(ol[0], ol[1], ol[2], ol[3]);
        *reinterpret_cast<float4 *>(out_act + ob[u]) = make_float4(oa[0], oa[1], oa[2], oa[3]);
        *reinterpret_cast<float4 *>(out_dsd + ob[u]) = make_float4(od[0], od[1], od[2], od[3]);
      }
    }
    for (int it0 = tid; !tile_gen && it0 < nitem; it0 += T * kEpiU) {
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
        *reinterpret_cast<int4 *>(out_idx + ob[u]) = make_int4(oi[0], oi[1], oi[2], oi[3]);
        *reinterpret_cast<float4 *>(out_len + ob[u]) = make_float4(ol[0], ol[1], ol[2], ol[3]);
        *reinterpret_cast<float4 *>(out_act + ob[u]) = make_float4(oa[0], oa[1], oa[2], oa[3]);
        *reinterpret_cast<float4 *>(out_dsd + ob[u]) = make_float4(od[0], od[1], od[2], od[3]);
      }
    }
  }
  for (int r = 0; r < TH && !vec4; ++r) {
    const int gy = ty * TH + r;
    if (gy >= H) break;
    const size_t pix0 = ((size_t)b * H + gy) * W + (size_t)tx * TW;
    {
      for (int j = tid; j < row_items; j += T) {
        const int x = j / K, s = j - x * K;
        int32_t oi;
        float ol, oa, od;
        slot_value(r, x, s, pix0 + x, oi, ol, oa, od);
        const size_t o = pix0 * K + j;
        out_idx[o] = oi;
        out_len[o] = ol;
        out_act[o] = oa;
        out_dsd[o] = od;
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
  float4 *cull, *evr, *ms, *ell;
  int *bin_count;
  int32_t *bin_id;
  float *bin_lb;
  float4 *bin_rec;     // the cull records of the list entries, in list order (bin2 streams them)
  int *tl_count;
  int32_t *tl_id;
  float *tl_lb;
  int *c_count;        // coarse regions (bin0)
  int32_t *c_id;
  ConeRec *cones;      // per super-tile
  int *tile_order;     // per sweep tile
  int nstx, nsty, nst0x, nst0y;
};

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t trace_ws_layout(int B, int N, int H, int W, void *base, TraceWs *ws) {
  const size_t P = (size_t)B * N;
  const int nstx = (W + kST - 1) / kST, nsty = (H + kST - 1) / kST;
  const size_t nbin = (size_t)B * nstx * nsty;
  size_t off = 0;
  char *p = reinterpret_cast<char *>(base);
  auto take = [&](size_t bytes) { char *q = p ? p + off : nullptr; off += align256(bytes); return q; };
  // sweep tiles are at least 8x8 pixels: size the tile lists for that worst case
  const size_t ntile = (size_t)B * ((W + 7) / 8) * ((H + 7) / 8);
  char *c = take(P * 16), *e = take(P * 48), *m4 = take(P * 16), *bc = take(nbin * 4), *bi = take(nbin * kBinCap * 4),
       *bl = take(nbin * kBinCap * 4), *tc = take(ntile * 4), *ti = take(ntile * kTileCap * 4),
       *tl = take(ntile * kTileCap * 4);
  const int nst0x = (W + kST0 - 1) / kST0, nst0y = (H + kST0 - 1) / kST0;
  const size_t nreg = (size_t)B * nst0x * nst0y;
  char *cc = take(nreg * 4), *ci = take(nreg * (size_t)N * 4), *cn = take(nbin * sizeof(ConeRec)), *to = take(ntile * 4), *el = take(P * 32), *br = take(nbin * kBinCap * 16);
  if (ws) {
    ws->ell = reinterpret_cast<float4 *>(el);
    ws->bin_rec = reinterpret_cast<float4 *>(br);
    ws->c_count = reinterpret_cast<int *>(cc); ws->c_id = reinterpret_cast<int32_t *>(ci);
    ws->cones = reinterpret_cast<ConeRec *>(cn);
    ws->tile_order = reinterpret_cast<int *>(to);
    ws->nst0x = nst0x; ws->nst0y = nst0y;
    ws->tl_count = reinterpret_cast<int *>(tc); ws->tl_id = reinterpret_cast<int32_t *>(ti);
    ws->tl_lb = reinterpret_cast<float *>(tl);
    ws->cull = reinterpret_cast<float4 *>(c); ws->evr = reinterpret_cast<float4 *>(e);
    ws->ms = reinterpret_cast<float4 *>(m4);
    ws->bin_count = reinterpret_cast<int *>(bc); ws->bin_id = reinterpret_cast<int32_t *>(bi);
    ws->bin_lb = reinterpret_cast<float *>(bl); ws->nstx = nstx; ws->nsty = nsty;
  }
  return off;
}

#ifndef VOGE_SWEEP_LDS_PAD
#define VOGE_SWEEP_LDS_PAD 0      // (occupancy experiments: extra dynamic LDS per workgroup)
#endif
template <int WAVES, bool ISO>
static int launch_trace(const TraceWs &ws, const float *rays, int B, int N, int H, int W, int K,
                        float thr_act, int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                        hipStream_t st) {
  constexpr int T = 64 * WAVES;
  constexpr int TW = (WAVES >= 2) ? 16 : 8;
  constexpr int TH = (WAVES == 4) ? 16 : 8;
  const size_t lds = ((sizeof(uint64_t) * (size_t)(K + 1) * (T + 1) + 15) & ~(size_t)15) + sizeof(TraceLds<T, ISO>) + VOGE_SWEEP_LDS_PAD;
  auto kern = trace_fwd_kernel<WAVES, ISO>;
  {
    static DynLdsCache cache;
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, cache);
    if (rc) return rc;
  }
  dim3 grid(((W + TW - 1) / TW) * ((H + TH - 1) / TH), B);
  hipLaunchKernelGGL(bin2_kernel, grid, dim3(256), 0, st, ws.bin_rec, ws.ell, rays, ws.bin_count, ws.bin_id, ws.bin_lb, ws.nstx,
                     ws.nstx * ws.nsty, N, H, W, TW, TH, ws.tl_count, ws.tl_id, ws.tl_lb, K, idx, len, act, dsd, cnt);
  {
    int rc = launch_status();
    if (rc) return rc;
  }
  // ordering only matters when the tiles do not all fit on the chip at once
  #ifndef VOGE_ORDER_MIN_TILES
#define VOGE_ORDER_MIN_TILES 2048
#endif
  const bool ordered = (size_t)grid.x * grid.y > VOGE_ORDER_MIN_TILES;
  if (ordered)
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, st, ws.tl_count, (int)(grid.x * grid.y), ws.tile_order);
  hipLaunchKernelGGL(kern, grid, dim3(T), lds, st, ws.cull, ws.evr, ws.ms, rays, ws.bin_count, ws.bin_id, ws.bin_lb,
                     ws.tl_count, ws.tl_id, ws.tl_lb, ordered ? ws.tile_order : nullptr, ws.nstx, ws.nstx * ws.nsty, N, H, W, K, thr_act, idx, len, act, dsd, cnt);
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
#ifdef VOGE_BIN_TIMES
extern "C" int voge_debug_bin_times(unsigned long long *out, int n_wg) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(voge::g_bin_times), sizeof(unsigned long long) * 8 * (size_t)n_wg);
}
#endif
#ifdef VOGE_SWEEP_TIMES
extern "C" int voge_debug_sweep_times(unsigned long long *out, int n_wg) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(voge::g_sweep_times), sizeof(unsigned long long) * 8 * (size_t)n_wg);
}
#endif

extern "C" size_t voge_trace_workspace_bytes(int B, int N, int H, int W) {
  if (B <= 0 || N < 0 || H <= 0 || W <= 0) return 0;
  return trace_ws_layout(B, N, H, W, nullptr, nullptr);
}

static int trace_topk_fwd_impl(const int iso_in, const IsoView view, const float *mus, const float *isigmas, const float *rays,
                               const float *cam_fwd, int B, int N, int H, int W, int K,
                               float thr_act, void *workspace, size_t workspace_bytes,
                               int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                               voge_stream_t stream) {
  if (B < 0 || N < 0 || H < 0 || W < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K) return VOGE_ERR_K_TOO_LARGE;
  if ((size_t)B * H * W == 0) return 0;  // numel == 0 early return (ray_trace_voge.cu:248-251)
  if (!rays || !idx || !len || !act || !dsd || !workspace) return VOGE_ERR_BAD_ARG;
  if (N > 0 && (!mus || !isigmas)) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < voge_trace_workspace_bytes(B, N, H, W)) return VOGE_ERR_WORKSPACE;
  // the top-K lists of one 8x8 tile must fit the CU's LDS: validated before anything is enqueued
  if (sizeof(uint64_t) * (size_t)(K + 1) * 65 + 16 + sizeof(TraceLds<64, false>) > 160 * 1024) return VOGE_ERR_K_TOO_LARGE;
  hipStream_t st = (hipStream_t)stream;
  const int P = B * N;
  TraceWs ws;
  trace_ws_layout(B, N, H, W, workspace, &ws);
  // The coarse level pays off once the per-super-tile scans of all N dominate: either many
  // super-tiles (nst * N tests in total) or simply a long scan per workgroup (a band of a sharded
  // frame has few super-tiles but every one of them would still walk all N: 44 us at N = 50k).
#ifndef VOGE_COARSE_MIN_N
#define VOGE_COARSE_MIN_N 16384
#endif
  const bool coarse = ((size_t)ws.nstx * ws.nsty * (size_t)N >= ((size_t)1 << 21)) ||
                      (N >= VOGE_COARSE_MIN_N && ws.nstx * ws.nsty >= 4);
  {
    const int nst = ws.nstx * ws.nsty;
    const int nprep = (P + kBinThreads - 1) / kBinThreads;
    hipLaunchKernelGGL(prep_cone_kernel, dim3(nst * B + nprep), dim3(kBinThreads), 0, st, rays, H, W, ws.nstx, nst, B,
                       ws.cones, ws.c_count, B * ws.nst0x * ws.nst0y, mus, isigmas, cam_fwd, N, P, thr_act, iso_in, ws.cull,
                       ws.evr, ws.ms, ws.ell, view);
    int rc = launch_status();
    if (rc) return rc;
  }
  if (coarse) {
    hipLaunchKernelGGL(bin0_kernel, dim3(ws.nst0x * ws.nst0y * kBin0Split, B), dim3(kBinThreads), 0, st, ws.cull, ws.cones,
                       ws.nstx, ws.nsty, N, ws.nst0x, ws.c_count, ws.c_id);
  }
  hipLaunchKernelGGL(bin_kernel, dim3(ws.nstx * ws.nsty, B), dim3(kBinThreads), 0, st, ws.cull, ws.ell, ws.evr, ws.cones,
                     coarse ? ws.c_count : nullptr, ws.c_id, ws.nst0x, ws.nst0x * ws.nst0y, N, H, W, ws.nstx, ws.bin_count,
                     ws.bin_id, ws.bin_lb, ws.bin_rec);
  {
    int rc = launch_status();
    if (rc) return rc;
  }
  // One wave (an 8x8 pixel tile) per workgroup.  Residency is set by the LDS top-K lists either
  // way (~6 waves per CU at K = 40), and independent single-wave workgroups measured 4-10 % faster
  // than 16x8 / 16x16 tiles on all three BASELINE configs (no barriers, finer load balance,
  // tighter per-tile candidate lists).  The multi-wave instantiations remain for experiments.
#ifdef VOGE_FORCE_WAVES
  return launch_trace<VOGE_FORCE_WAVES, false>(ws, rays, B, N, H, W, K, thr_act, idx, len, act, dsd, cnt, st);
#endif
#ifndef VOGE_NO_ISO_SWEEP
  if (iso_in) return launch_trace<1, true>(ws, rays, B, N, H, W, K, thr_act, idx, len, act, dsd, cnt, st);
#endif
  return launch_trace<1, false>(ws, rays, B, N, H, W, K, thr_act, idx, len, act, dsd, cnt, st);
}

extern "C" int voge_trace_topk_fwd(const float *mus, const float *isigmas, const float *rays,
                                   const float *cam_fwd, int B, int N, int H, int W, int K,
                                   float thr_act, void *workspace, size_t workspace_bytes,
                                   int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                                   voge_stream_t stream) {
  return trace_topk_fwd_impl(0, IsoView{nullptr, 0, 0}, mus, isigmas, rays, cam_fwd, B, N, H, W, K, thr_act, workspace, workspace_bytes, idx, len,
                             act, dsd, cnt, stream);
}

extern "C" int voge_trace_topk_fwd_iso(const float *mus, const float *a, const float *rays,
                                       const float *cam_fwd, int B, int N, int H, int W, int K,
                                       float thr_act, void *workspace, size_t workspace_bytes,
                                       int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                                       voge_stream_t stream) {
  return trace_topk_fwd_impl(1, IsoView{nullptr, 0, 0}, mus, a, rays, cam_fwd, B, N, H, W, K, thr_act, workspace, workspace_bytes, idx, len, act,
                             dsd, cnt, stream);
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
                                            int sigma_mode, const float *rays, const float *cam_fwd, int B, int N, int H,
                                            int W, int K, float thr_act, void *workspace, size_t workspace_bytes,
                                            int32_t *idx, float *len, float *act, float *dsd, int32_t *cnt,
                                            voge_stream_t stream) {
  if (sigma_mode < 0 || sigma_mode > 2) return VOGE_ERR_BAD_ARG;
  return trace_topk_fwd_impl(1, IsoView{origin, shared ? 1 : 0, sigma_mode}, verts, sigmas, rays, cam_fwd, B, N, H, W, K,
                             thr_act, workspace, workspace_bytes, idx, len, act, dsd, cnt, stream);
}
