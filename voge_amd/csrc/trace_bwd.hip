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
// 2. A wave owns an 8x8 pixel tile and streams it row by row, ONE LANE PER SLOT: the 8*K slots of
//    a tile row are one contiguous run of the flat [npix*K] arrays, read 64 at a time with the
//    loads of four batches in flight.  The 12 per-Gaussian sums are accumulated in a wave-private
//    LDS table keyed by Gaussian index with plain read-modify-write (lanes of one instruction
//    that share a Gaussian elect one writer per round), then flushed with
//    lane-coalesced global atomics: 12 adjacent lanes add to 12 adjacent floats of one 64-byte
//    line of a padded [P][16] accumulator.  g_ray is pixel-owned: segmented wave sum, plain stores.
#include "voge_common.h"

namespace voge {

#ifndef VOGE_BWD_WAVES
#define VOGE_BWD_WAVES 1
#endif
constexpr int kBwdWaves = VOGE_BWD_WAVES;    // waves per workgroup (each wave is independent)
#ifndef VOGE_BWD_TH
#define VOGE_BWD_TH 2
#endif
#ifndef VOGE_BWD_NE
#define VOGE_BWD_NE 128
#endif
constexpr int kBwdTH = VOGE_BWD_TH;       // tile height: a wave owns an 8 x kBwdTH pixel tile
constexpr int kBwdNE = VOGE_BWD_NE;     // table entries per wave (an 8x4 tile touches ~50-100 Gaussians)
#ifndef VOGE_BWD_U
#define VOGE_BWD_U 1
#endif
constexpr int kBwdU = VOGE_BWD_U;        // 64-slot batches whose loads are issued together

struct BwdWaveLds {
  WaveTable<kBwdNE, 3> tab;     // key = Gaussian index, values = g_mu (3) + g_A (9)
  float ray[8 * kBwdTH * 3];    // g_ray of the tile's pixels
  int cntv[64];                 // number of leading valid slots per pixel
};

// Slot batch = 64 consecutive slots of one 8-pixel tile row; a GROUP is kBwdU batches whose loads
// are issued together:
//   A(g): idx / g_len / g_act / g_dsd of the slots      (coalesced streams)
//   B(g): packed (mu, A) record of each slot's Gaussian  (gather, needs A(g)) and the pixel's ray
//   C(g): gradient terms + table accumulation            (LDS)
struct BwdGroupA {
  int p[kBwdU], pl[kBwdU];
  float gl[kBwdU], ga[kBwdU], gd[kBwdU];
};
struct BwdGroupB {
  float4 r0[kBwdU], r1[kBwdU], r2[kBwdU];
  float dx[kBwdU], dy[kBwdU], dz[kBwdU];
  bool live[kBwdU];
};

__global__ void __launch_bounds__(64 * kBwdWaves)
trace_bwd_kernel(const float4 *__restrict__ rec, const float *__restrict__ rays,
                 const int32_t *__restrict__ idx, const int32_t *__restrict__ cnt,
                 const float *__restrict__ g_len, const float *__restrict__ g_act,
                 const float *__restrict__ g_dsd, const int P, const long nrows, const int W, const int K,
                 float *__restrict__ g_ray, float *__restrict__ acc /* [P][16]: g_mu (3), g_A (9), pad (4) */) {
  __shared__ BwdWaveLds Ls[kBwdWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  BwdWaveLds &L = Ls[wave];
  const int tiles_x = (W + 7) / 8;
  const long ntiles = (long)tiles_x * ((nrows + kBwdTH - 1) / kBwdTH);
  const long tile = (long)blockIdx.x * kBwdWaves + wave;
  if (tile >= ntiles) return;  // waves never synchronise with each other
  const int x0 = (int)(tile % tiles_x) * 8;
  const long y0 = (tile / tiles_x) * kBwdTH;
  const int tw = min(8, W - x0);
  const int th = (int)min((long)kBwdTH, nrows - y0);
  {   // per-pixel slot counts (all K when the caller has none)
    const int lx = lane & 7, ly = lane >> 3;
    int c = 0;
    if (lx < tw && ly < th) c = (cnt != nullptr) ? min(K, max(0, cnt[(y0 + ly) * W + x0 + lx])) : K;
    L.cntv[lane] = c;
    if (__all(c == 0)) {   // nothing was hit in this tile
      if (g_ray != nullptr && lx < tw && ly < th) {
        float *o = g_ray + ((y0 + ly) * W + x0 + lx) * 3;
        o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
      }
      return;
    }
  }
  wt_clear(L.tab, lane);
  for (int i = lane; i < 8 * kBwdTH * 3; i += 64) L.ray[i] = 0.0f;
  const int n_items = tw * K;              // slots of one 8-pixel row: one contiguous run
  const int nit = (n_items + 63) >> 6;
  const int nb = th * nit;                 // batches of the tile
  const int ng = (nb + kBwdU - 1) / kBwdU;
  const float invK = 1.0f / (float)K;

  auto loadA = [&](const int g, BwdGroupA &a) {
#pragma unroll
    for (int u = 0; u < kBwdU; ++u) {
      const int bb = g * kBwdU + u;
      const int r = bb / nit, it = bb - r * nit;
      const int j = it * 64 + lane;
      const int lx = __float2int_rz(((float)j + 0.5f) * invK);
      const int k = j - lx * K;
      const int pl = r * 8 + lx;
      const bool ok = (bb < nb) && (j < n_items) && (k < L.cntv[min(pl, 63)]);
      const long pid = ((y0 + r) * W + x0) * (long)K + j;
      a.p[u] = ok ? idx[pid] : -1;
      a.gl[u] = ok ? g_len[pid] : 0.0f;
      a.ga[u] = ok ? g_act[pid] : 0.0f;
      a.gd[u] = ok ? g_dsd[pid] : 0.0f;
      a.pl[u] = ok ? pl : 64 + lane;       // inactive lanes: private segment keys
    }
  };
  auto loadB = [&](const BwdGroupA &a, BwdGroupB &b) {
#pragma unroll
    for (int u = 0; u < kBwdU; ++u) {
      b.live[u] = (a.p[u] >= 0) && (a.p[u] < P) && !(a.gl[u] == 0.0f && a.ga[u] == 0.0f && a.gd[u] == 0.0f);
      if (b.live[u]) {
        b.r0[u] = rec[3 * (size_t)a.p[u]]; b.r1[u] = rec[3 * (size_t)a.p[u] + 1]; b.r2[u] = rec[3 * (size_t)a.p[u] + 2];
        const float *ry = rays + ((y0 + (a.pl[u] >> 3)) * W + x0 + (a.pl[u] & 7)) * 3;
        b.dx[u] = ry[0]; b.dy[u] = ry[1]; b.dz[u] = ry[2];
      }
    }
  };
  auto compute = [&](const BwdGroupA &a, const BwdGroupB &b) {
#pragma unroll
    for (int u = 0; u < kBwdU; ++u) {
      float4 val[3] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
      float rx = 0.f, ryv = 0.f, rz = 0.f;
      if (b.live[u]) {
        const float dx = b.dx[u], dy = b.dy[u], dz = b.dz[u];
        const float mx = b.r0[u].x, my = b.r0[u].y, mz = b.r0[u].z;
        const float A[9] = {b.r0[u].w, b.r1[u].x, b.r1[u].y, b.r1[u].z, b.r1[u].w, b.r2[u].x, b.r2[u].y, b.r2[u].z, b.r2[u].w};
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
        float vx = fmaf(-t, dx, mx), vy = fmaf(-t, dy, my), vz = fmaf(-t, dz, mz);
        {   // (v^T A d = 0 by construction: what is left of it is t's rounding error; one projection along d removes it)
          const float rr = fmaf(vz, adz, fmaf(vy, ady, vx * adx)) * ik;
          vx = fmaf(-rr, dx, vx); vy = fmaf(-rr, dy, vy); vz = fmaf(-rr, dz, vz);
        }
        const float avx = fmaf(A[2], vz, fmaf(A[1], vy, A[0] * vx));
        const float avy = fmaf(A[5], vz, fmaf(A[4], vy, A[3] * vx));
        const float avz = fmaf(A[8], vz, fmaf(A[7], vy, A[6] * vx));
        const float tvx = fmaf(A[6], vz, fmaf(A[3], vy, A[0] * vx));
        const float tvy = fmaf(A[7], vz, fmaf(A[4], vy, A[1] * vx));
        const float tvz = fmaf(A[8], vz, fmaf(A[5], vy, A[2] * vx));
        const float c1 = a.gl[u] * ik;
        const float g_a = a.ga[u], g_d = a.gd[u];
        const float gat = g_a * t;
        float o[12];
        o[0] = fmaf(c1, adx, g_a * (avx + tvx + t * (tdx - adx)));
        o[1] = fmaf(c1, ady, g_a * (avy + tvy + t * (tdy - ady)));
        o[2] = fmaf(c1, adz, g_a * (avz + tvz + t * (tdz - adz)));
        const float d[3] = {dx, dy, dz}, v[3] = {vx, vy, vz};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int c = 0; c < 3; ++c)
            o[3 + 3 * i + c] = fmaf(g_a, fmaf(v[i], v[c], t * (d[i] * v[c] - v[i] * d[c])),
                                    fmaf(g_d, d[i] * d[c], c1 * (v[i] * d[c])));
        val[0] = make_float4(o[0], o[1], o[2], o[3]);
        val[1] = make_float4(o[4], o[5], o[6], o[7]);
        val[2] = make_float4(o[8], o[9], o[10], o[11]);
        rx = fmaf(g_d, adx + tdx, fmaf(c1, fmaf(-t, adx, tvx), gat * fmaf(t, adx - tdx, -2.0f * tvx)));
        ryv = fmaf(g_d, ady + tdy, fmaf(c1, fmaf(-t, ady, tvy), gat * fmaf(t, ady - tdy, -2.0f * tvy)));
        rz = fmaf(g_d, adz + tdz, fmaf(c1, fmaf(-t, adz, tvz), gat * fmaf(t, adz - tdz, -2.0f * tvz)));
      }
      if (!__any(b.live[u])) continue;   // uniform
      const int slot = wt_find(L.tab, a.p[u], b.live[u]);
      wt_add(L.tab, slot, val, b.live[u] && slot >= 0, lane);
      if (b.live[u] && slot < 0) {  // table full: rare, straight to HBM
        const float o[12] = {val[0].x, val[0].y, val[0].z, val[0].w, val[1].x, val[1].y,
                             val[1].z, val[1].w, val[2].x, val[2].y, val[2].z, val[2].w};
#pragma unroll
        for (int c = 0; c < 12; ++c) unsafeAtomicAdd(acc + 16 * (size_t)a.p[u] + c, o[c]);
      }
      if (g_ray != nullptr) {  // pixel-owned: segmented sum over the lanes of each pixel
        rx = seg_sum_key(rx, a.pl[u], lane);
        ryv = seg_sum_key(ryv, a.pl[u], lane);
        rz = seg_sum_key(rz, a.pl[u], lane);
        const int prev = __shfl_up(a.pl[u], 1, 64);
        if ((lane == 0 || prev != a.pl[u]) && a.pl[u] < 8 * kBwdTH) {
          L.ray[a.pl[u] * 3 + 0] += rx;
          L.ray[a.pl[u] * 3 + 1] += ryv;
          L.ray[a.pl[u] * 3 + 2] += rz;
        }
      }
    }
  };

  // No register double-buffering across groups: at 80 VGPRs ten waves share a CU (the LDS tables
  // are the limit) and hide each other's latency; a two-deep software pipeline needed 176-256
  // VGPRs and measured slower (254 / 362 us vs 223 us on MI355X).
  for (int g = 0; g < ng; ++g) {
    BwdGroupA a;
    BwdGroupB b;
    loadA(g, a);
    loadB(a, b);
    compute(a, b);
  }

  if (g_ray != nullptr) {
    for (int j = lane; j < 8 * kBwdTH * 3; j += 64) {
      const int i = j / 3, c = j - i * 3;
      const int px = i & 7, py = i >> 3;
      if (px < tw && py < th) g_ray[((y0 + py) * W + x0 + px) * 3 + c] = L.ray[j];
    }
  }
  // flush: 16 adjacent lanes per table entry add 12 adjacent floats of ONE 64-byte line of
  // acc[p][16] -- lane-coalesced atomics run ~15x faster than 64 scattered ones
  // (tools/atomic_bench.hip: 330 vs 21 Gatomic/s).
  {
    const int c = lane & 15;
    const float *vals = reinterpret_cast<const float *>(L.tab.vals);
    const int n = wt_compact(L.tab, lane);
    const lds_vint *list = lds_volatile(L.tab.owner);
    for (int i = lane >> 4; i < n; i += 4) {
      const int s = list[i];
      const int p = L.tab.keys[s];
      if (c < 12) unsafeAtomicAdd(acc + 16 * (size_t)p + c, vals[s * 12 + c]);
    }
  }
}

// mus [P,3] + isigmas [P,9] -> 3 x float4 per Gaussian, so the sweep gathers with dwordx4 loads
__global__ void __launch_bounds__(256)
bwd_pack_kernel(const float *__restrict__ mus, const float *__restrict__ isg, const int P,
                float4 *__restrict__ rec, float4 *__restrict__ acc /* [P][4] float4, zeroed here */) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[4 * (size_t)g + q] = make_float4(0.f, 0.f, 0.f, 0.f);
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
  if (g >= P || c >= 12) return;
  const float v = acc[t];
  if (c < 3) g_mus[3 * (size_t)g + c] = v; else g_isg[9 * (size_t)g + (c - 3)] = v;
}

// ------------------------------------------------------------------------------------------
// Isotropic Gaussians, A = a I with ONE scalar a per Gaussian (the reference's (N,) sigma form,
// Aggregation.py:155-157 / Cuboid.py:51-52).  len = mu.d/|d|^2, v = mu - len d, act = a |v|^2,
// dsd = a |d|^2, so per slot
//   g_mu  = g_len d/|d|^2 + 2 a g_act v ,      g_a = g_act |v|^2 + g_dsd |d|^2 ,
//   g_ray = g_len (mu - 2 len d)/|d|^2 - 2 a len g_act v + 2 a g_dsd d .
// Four sums per Gaussian instead of twelve: one float4 per table entry (a third of the LDS, of
// the flush atomics and of the arithmetic of the general kernel); the caller gets the gradient
// of the scalar directly instead of a 3x3 block it would reduce again.
// ------------------------------------------------------------------------------------------
#ifndef VOGE_BWDI_NE
#define VOGE_BWDI_NE 128
#endif
#ifndef VOGE_BWDI_WAVES
#define VOGE_BWDI_WAVES 1
#endif
#ifndef VOGE_BWDI_U
#define VOGE_BWDI_U 2
#endif
constexpr int kBwdINE = VOGE_BWDI_NE, kBwdIWaves = VOGE_BWDI_WAVES, kBwdIU = VOGE_BWDI_U;

struct BwdIsoLds {
  WaveTable<kBwdINE, 1> tab;   // key = Gaussian index, value = (g_mu, g_a)
  float ray[8 * kBwdTH * 3];
  int cntv[64];
};

__global__ void __launch_bounds__(64 * kBwdIWaves)
trace_bwd_iso_kernel(const float4 *__restrict__ rec /* (mu, a) */, const float *__restrict__ rays,
                     const int32_t *__restrict__ idx, const int32_t *__restrict__ cnt,
                     const float *__restrict__ g_len, const float *__restrict__ g_act,
                     const float *__restrict__ g_dsd, const int P, const long nrows, const int W, const int K,
                     float *__restrict__ g_ray, float *__restrict__ acc /* [P][4] */) {
  __shared__ BwdIsoLds Ls[kBwdIWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  BwdIsoLds &L = Ls[wave];
  const int tiles_x = (W + 7) / 8;
  const long ntiles = (long)tiles_x * ((nrows + kBwdTH - 1) / kBwdTH);
  const long tile = (long)blockIdx.x * kBwdIWaves + wave;
  if (tile >= ntiles) return;  // waves never synchronise with each other
  const int x0 = (int)(tile % tiles_x) * 8;
  const long y0 = (tile / tiles_x) * kBwdTH;
  const int tw = min(8, W - x0);
  const int th = (int)min((long)kBwdTH, nrows - y0);
  {
    const int lx = lane & 7, ly = lane >> 3;
    int c = 0;
    if (lx < tw && ly < th) c = (cnt != nullptr) ? min(K, max(0, cnt[(y0 + ly) * W + x0 + lx])) : K;
    L.cntv[lane] = c;
    if (__all(c == 0)) {
      if (g_ray != nullptr && lx < tw && ly < th) {
        float *o = g_ray + ((y0 + ly) * W + x0 + lx) * 3;
        o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
      }
      return;
    }
  }
  wt_clear(L.tab, lane);
  for (int i = lane; i < 8 * kBwdTH * 3; i += 64) L.ray[i] = 0.0f;
  const int n_items = tw * K;
  const int nit = (n_items + 63) >> 6;
  const int nb = th * nit;
  const float invK = 1.0f / (float)K;
  for (int g0 = 0; g0 < nb; g0 += kBwdIU) {
    int p[kBwdIU], pl[kBwdIU];
    float gl[kBwdIU], ga[kBwdIU], gd[kBwdIU];
#pragma unroll
    for (int u = 0; u < kBwdIU; ++u) {
      const int bb = g0 + u;
      const int r = bb / nit, it = bb - r * nit;
      const int j = it * 64 + lane;
      const int lx = __float2int_rz(((float)j + 0.5f) * invK);
      const int k = j - lx * K;
      const int pq = r * 8 + lx;
      const bool ok = (bb < nb) && (j < n_items) && (k < L.cntv[min(pq, 63)]);
      const long pid = ((y0 + r) * W + x0) * (long)K + j;
      p[u] = ok ? idx[pid] : -1;
      gl[u] = ok ? g_len[pid] : 0.0f;
      ga[u] = ok ? g_act[pid] : 0.0f;
      gd[u] = ok ? g_dsd[pid] : 0.0f;
      pl[u] = ok ? pq : 64 + lane;
    }
    float4 rc[kBwdIU];
    float dx[kBwdIU], dy[kBwdIU], dz[kBwdIU];
    bool live[kBwdIU];
#pragma unroll
    for (int u = 0; u < kBwdIU; ++u) {
      live[u] = (p[u] >= 0) && (p[u] < P) && !(gl[u] == 0.0f && ga[u] == 0.0f && gd[u] == 0.0f);
      rc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      dx[u] = dy[u] = dz[u] = 0.0f;
      if (live[u]) {
        rc[u] = rec[p[u]];
        const float *ry = rays + ((y0 + (pl[u] >> 3)) * W + x0 + (pl[u] & 7)) * 3;
        dx[u] = ry[0]; dy[u] = ry[1]; dz[u] = ry[2];
      }
    }
#pragma unroll
    for (int u = 0; u < kBwdIU; ++u) {
      if (g0 + u >= nb) break;   // uniform
      float4 val[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
      float rx = 0.f, ryv = 0.f, rz = 0.f;
      if (live[u]) {
        const float mx = rc[u].x, my = rc[u].y, mz = rc[u].z, a = rc[u].w;
        const float dn2 = fmaf(dz[u], dz[u], fmaf(dy[u], dy[u], dx[u] * dx[u]));
        const float idn = __builtin_amdgcn_rcpf(dn2);
        const float t = fmaf(mz, dz[u], fmaf(my, dy[u], mx * dx[u])) * idn;
        float vx = fmaf(-t, dx[u], mx), vy = fmaf(-t, dy[u], my), vz = fmaf(-t, dz[u], mz);
        {   // v is orthogonal to d by construction; what is left along d is t's fp32 rounding (|mu| ~ 6, |v| ~ 1e-3 on
            // the bunny: 2e-4 of v), and g_mu multiplies it by 2 a g_act ~ 1e6.  One projection removes it.
          const float rr = fmaf(vz, dz[u], fmaf(vy, dy[u], vx * dx[u])) * idn;
          vx = fmaf(-rr, dx[u], vx); vy = fmaf(-rr, dy[u], vy); vz = fmaf(-rr, dz[u], vz);
        }
        const float c1 = gl[u] * idn, c2 = 2.0f * a * ga[u];
        val[0] = make_float4(fmaf(c1, dx[u], c2 * vx), fmaf(c1, dy[u], c2 * vy), fmaf(c1, dz[u], c2 * vz),
                             fmaf(ga[u], fmaf(vz, vz, fmaf(vy, vy, vx * vx)), gd[u] * dn2));
        if (g_ray != nullptr) {
          const float c3 = 2.0f * a * gd[u], c4 = -c2 * t;
          rx = fmaf(c3, dx[u], fmaf(c4, vx, c1 * fmaf(-2.0f * t, dx[u], mx)));
          ryv = fmaf(c3, dy[u], fmaf(c4, vy, c1 * fmaf(-2.0f * t, dy[u], my)));
          rz = fmaf(c3, dz[u], fmaf(c4, vz, c1 * fmaf(-2.0f * t, dz[u], mz)));
        }
      }
      if (__any(live[u])) {
        const int slot = wt_find(L.tab, p[u], live[u]);
        wt_add(L.tab, slot, val, live[u] && slot >= 0, lane);
        if (live[u] && slot < 0) {   // table full: rare, straight to HBM
          const float o[4] = {val[0].x, val[0].y, val[0].z, val[0].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) unsafeAtomicAdd(acc + 4 * (size_t)p[u] + c, o[c]);
        }
      }
      if (g_ray != nullptr) {
        rx = seg_sum_key(rx, pl[u], lane);
        ryv = seg_sum_key(ryv, pl[u], lane);
        rz = seg_sum_key(rz, pl[u], lane);
        const int prev = __shfl_up(pl[u], 1, 64);
        if ((lane == 0 || prev != pl[u]) && pl[u] < 8 * kBwdTH) {
          L.ray[pl[u] * 3 + 0] += rx;
          L.ray[pl[u] * 3 + 1] += ryv;
          L.ray[pl[u] * 3 + 2] += rz;
        }
      }
    }
  }
  if (g_ray != nullptr) {
    for (int j = lane; j < 8 * kBwdTH * 3; j += 64) {
      const int i = j / 3, c = j - i * 3;
      const int px = i & 7, py = i >> 3;
      if (px < tw && py < th) g_ray[((y0 + py) * W + x0 + px) * 3 + c] = L.ray[j];
    }
  }
  {   // flush: 4 adjacent lanes per entry -> the 16 bytes of acc[p]: lane-coalesced atomics
    const int c = lane & 3;
    const float *vals = reinterpret_cast<const float *>(L.tab.vals);
    const int n = wt_compact(L.tab, lane);
    const lds_vint *list = lds_volatile(L.tab.owner);
    for (int i = lane >> 2; i < n; i += 16) {
      const int s = list[i];
      unsafeAtomicAdd(acc + 4 * (size_t)L.tab.keys[s] + c, vals[s * 4 + c]);
    }
  }
}

__global__ void __launch_bounds__(256)
bwd_pack_iso_kernel(const float *__restrict__ mus, const float *__restrict__ a, const int P, const int N,
                    const IsoView view, float4 *__restrict__ rec, float4 *__restrict__ acc) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P) return;
  const int src = view.shared ? g % N : g;
  float mx = mus[3 * (size_t)src], my = mus[3 * (size_t)src + 1], mz = mus[3 * (size_t)src + 2];
  if (view.origin != nullptr) {
    const float *o = view.origin + 3 * (g / N);
    mx -= o[0]; my -= o[1]; mz -= o[2];
  }
  rec[g] = make_float4(mx, my, mz, iso_view_a(a[src], view.mode));
  acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// acc [P] = (g_mu, g_a) per (batch element, Gaussian)  ->  gradients of what the caller passed in:
// plain arrays: g_mus [P,3], g_a [P]; view: the chain rule through a = 2 sigma / 2 / sigma, and the sum
// over the batch when one (verts, sigmas) set is shared by all views.
__global__ void __launch_bounds__(256)
bwd_unpack_iso_kernel(const float4 *__restrict__ acc, const float *__restrict__ a_in, const int P, const int N,
                      const int B, const IsoView view, float *__restrict__ g_mus, float *__restrict__ g_a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_out = view.shared ? N : P;
  if (g >= n_out) return;
  float4 v = acc[g];
  if (view.shared)
    for (int b = 1; b < B; ++b) {
      const float4 w = acc[(size_t)b * N + g];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
  g_mus[3 * (size_t)g] = v.x; g_mus[3 * (size_t)g + 1] = v.y; g_mus[3 * (size_t)g + 2] = v.z;
  float ga = v.w;
  if (view.mode == 1) ga = 2.0f * ga;
  else if (view.mode == 2) { const float s = a_in[g]; ga = -2.0f * ga / (s * s); }
  g_a[g] = ga;
}

}  // namespace voge


using namespace voge;

extern "C" size_t voge_trace_bwd_workspace_bytes(int P) {
  return P <= 0 ? 0 : (size_t)P * (48 + 64);
}

extern "C" int voge_trace_bwd(const float *mus, const float *isigmas, const float *rays,
                              const int32_t *idx, const int32_t *cnt, const float *g_len, const float *g_act,
                              const float *g_dsd, int P, long nrows, int W, int K, void *workspace,
                              size_t workspace_bytes, float *g_ray, float *g_mus, float *g_isg,
                              voge_stream_t stream) {
  if (P < 0 || nrows < 0 || W < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0) {
    if (g_ray && nrows * W > 0) return (int)voge_fill_async(g_ray, 0, sizeof(float) * 3 * (size_t)(nrows * W), st);
    return 0;
  }
  if (!g_mus || !g_isg || !mus || !isigmas || !workspace) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < voge_trace_bwd_workspace_bytes(P)) return VOGE_ERR_WORKSPACE;
  float *acc = reinterpret_cast<float *>(workspace);
  float4 *rec = reinterpret_cast<float4 *>(reinterpret_cast<char *>(workspace) + (size_t)P * 64);
  if (nrows * W > 0 && (!rays || !idx || !g_len || !g_act || !g_dsd)) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(bwd_pack_kernel, dim3((P + 255) / 256), dim3(256), 0, st, mus, isigmas, P, rec,
                     reinterpret_cast<float4 *>(acc));
  if (nrows * W > 0) {
    const long tiles = (long)((W + 7) / 8) * ((nrows + kBwdTH - 1) / kBwdTH);
    hipLaunchKernelGGL(trace_bwd_kernel, dim3((unsigned)((tiles + kBwdWaves - 1) / kBwdWaves)), dim3(64 * kBwdWaves),
                       0, st, rec, rays, idx, cnt, g_len, g_act, g_dsd, P, nrows, W, K, g_ray, acc);
  }
  hipLaunchKernelGGL(bwd_unpack_kernel, dim3((unsigned)(((size_t)P * 16 + 255) / 256)), dim3(256), 0, st, acc, P,
                     g_mus, g_isg);
  return launch_status();
}

extern "C" size_t voge_trace_bwd_iso_workspace_bytes(int P) { return P <= 0 ? 0 : (size_t)P * 32; }

static int trace_bwd_iso_impl(const IsoView view, const float *mus, const float *a, const float *rays, const int32_t *idx,
                              const int32_t *cnt, const float *g_len, const float *g_act, const float *g_dsd,
                              int B, int N, long nrows, int W, int K, void *workspace, size_t workspace_bytes,
                              float *g_ray, float *g_mus, float *g_a, voge_stream_t stream) {
  const int P = B * N;
  if (P < 0 || nrows < 0 || W < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0) {
    if (g_ray && nrows * W > 0) return (int)voge_fill_async(g_ray, 0, sizeof(float) * 3 * (size_t)(nrows * W), st);
    return 0;
  }
  if (!g_mus || !g_a || !mus || !a || !workspace) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < voge_trace_bwd_iso_workspace_bytes(P)) return VOGE_ERR_WORKSPACE;
  if (nrows * W > 0 && (!rays || !idx || !g_len || !g_act || !g_dsd)) return VOGE_ERR_BAD_ARG;
  float4 *acc = reinterpret_cast<float4 *>(workspace);
  float4 *rec = acc + P;
  hipLaunchKernelGGL(bwd_pack_iso_kernel, dim3((P + 255) / 256), dim3(256), 0, st, mus, a, P, N, view, rec, acc);
  if (nrows * W > 0) {
    const long tiles = (long)((W + 7) / 8) * ((nrows + kBwdTH - 1) / kBwdTH);
    hipLaunchKernelGGL(trace_bwd_iso_kernel, dim3((unsigned)((tiles + kBwdIWaves - 1) / kBwdIWaves)),
                       dim3(64 * kBwdIWaves), 0, st, rec, rays, idx, cnt, g_len, g_act, g_dsd, P, nrows, W, K, g_ray,
                       reinterpret_cast<float *>(acc));
  }
  const int n_out = view.shared ? N : P;
  hipLaunchKernelGGL(bwd_unpack_iso_kernel, dim3((n_out + 255) / 256), dim3(256), 0, st, acc, a, P, N, B, view, g_mus, g_a);
  return launch_status();
}

extern "C" int voge_trace_bwd_iso(const float *mus, const float *a, const float *rays, const int32_t *idx,
                                  const int32_t *cnt, const float *g_len, const float *g_act, const float *g_dsd,
                                  int P, long nrows, int W, int K, void *workspace, size_t workspace_bytes,
                                  float *g_ray, float *g_mus, float *g_a, voge_stream_t stream) {
  return trace_bwd_iso_impl(IsoView{nullptr, 0, 0}, mus, a, rays, idx, cnt, g_len, g_act, g_dsd, 1, P, nrows, W, K, workspace,
                            workspace_bytes, g_ray, g_mus, g_a, stream);
}

extern "C" int voge_trace_bwd_iso_view(const float *verts, const float *sigmas, const float *origin, int shared,
                                       int sigma_mode, const float *rays, const int32_t *idx, const int32_t *cnt,
                                       const float *g_len, const float *g_act, const float *g_dsd, int B, int N, long nrows,
                                       int W, int K, void *workspace, size_t workspace_bytes, float *g_ray,
                                       float *g_verts, float *g_sigmas, voge_stream_t stream) {
  if (sigma_mode < 0 || sigma_mode > 2 || B < 0 || N < 0) return VOGE_ERR_BAD_ARG;
  return trace_bwd_iso_impl(IsoView{origin, shared ? 1 : 0, sigma_mode}, verts, sigmas, rays, idx, cnt, g_len, g_act, g_dsd, B, N,
                            nrows, W, K, workspace, workspace_bytes, g_ray, g_verts, g_sigmas, stream);
}
