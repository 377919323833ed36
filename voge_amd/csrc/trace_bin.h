// Candidate binning in front of the fine-trace sweep (gfx950): two launches.
//
//   binA  (one 1024-thread workgroup per 128x128-pixel REGION and Gaussian slice): the region's cone as the
//         conservative union of its <= 16 super-tile cones (32x32 pixels each; the cones come with the rays --
//         voge_rays_fwd / voge_ray_cones), then every Gaussian of the slice against the region cone and -- survivors
//         only, compacted in LDS -- against the 16 child cones.  Survivors are appended to per-(super-tile, slice)
//         SEGMENTS whose positions come from ballots and in-wave prefix sums: no global atomics, no zero-initialised
//         counters, and the list order is a pure function of the inputs.  The scalar-sigma entry points also derive
//         the per-Gaussian records here (ISO_PREP), so the chain in front of the sweep is binA -> binB.
//   binB  (one 1024-thread workgroup per super-tile): gathers the <= 16 segments, orders the survivors front to
//         back (counting sort on a depth key, LDS), and then each of its 16 waves filters the ordered list --
//         still in LDS, records included -- with the bounding cone of its own 8x8-pixel sweep tile and writes the
//         tile's candidate list.  It also fixes the sweep's launch order without any exchange: super-tiles by
//         descending candidate count (every workgroup ranks itself against the segment counts of all), tiles inside
//         a super-tile by descending list length.
//
// What this replaces (round 1): prep_cone -> bin0 -> bin -> bin2 -> tile_order, five dependent launches
// (75 us at 50k Gaussians / 512^2) that exchanged region lists, super-tile lists and their records through HBM.
// Both tests are conservative (cone_keep / cone_keep_ell, voge_common.h), so the sweep's result equals the
// brute-force "-1" candidate list of VoGE/RayTracing.py:22-26.
#pragma once
#ifndef VOGE_BINA_REGION_RECT
#define VOGE_BINA_REGION_RECT 1
#endif
#ifndef VOGE_XCD_CHAIN
#define VOGE_XCD_CHAIN 0     // 1: a region's 16 binA slices AND its 64 binB quads on XCD (region % 8): binB finds the segments in the L2 binA wrote them through
#endif
#include <type_traits>

#include "voge_common.h"

namespace voge {

constexpr int kST0 = 128;          // region edge
constexpr int kCh = kST0 / kST;    // super-tiles per region side
constexpr int kBinThreads = 1024;
constexpr int kParts = 16;         // Gaussian slices per region (chunks of 1024 Gaussians dealt round-robin)
constexpr int kSegCap = 512;       // entries of one (super-tile, slice) segment (ids + their cull records)
constexpr int kTileCap = 2048;     // entries of a sweep tile's list
constexpr int kTilesPerBin = (kST / 8) * (kST / 8);   // sweep tiles (8x8 pixels) of a super-tile
constexpr int kRankMaxBins = 4096;  // beyond this many super-tiles (a batch of big frames) the sweep keeps the spatial order
static_assert(kCh * kCh == kBinThreads / 64, "one wave per super-tile of a region / per sweep tile of a super-tile");

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

// The cull record of an isotropic Gaussian (A = a I) from (centre, a): the fp32 form of prep_one's reach with the
// margins doubled to cover the fp32 rounding of the square root and the division (a conservative reach only ever
// keeps more candidates).  act = a |v|^2 >= a dist^2, so a hit needs dist^2 < thr_act / a.
__device__ __forceinline__ float4 iso_cull_record(const float mx, const float my, const float mz, const float a,
                                                  const float thr_act) {
  float reach = INFINITY;
  if (a > 0.0f && a < 3e38f) {
    // (hardware square root and reciprocal, ~1 ulp each, far inside the 2e-5 margins: the IEEE sequences were 40 of the
    //  instructions every workgroup of binA spends per Gaussian before its first test, 16 times over per Gaussian)
    const float nm = __builtin_amdgcn_sqrtf(fmaf(mz, mz, fmaf(my, my, mx * mx)));
    const float r = __builtin_amdgcn_sqrtf(fmaxf(thr_act, 0.0f) * __builtin_amdgcn_rcpf(a * (1.0f - 4e-6f))) * (1.0f + 2e-5f) + 2e-5f * nm + 1e-30f;
    reach = r;
    if (!(reach >= 0.0f)) reach = INFINITY;  // NaN guard
    // the lowest mantissa bit of a finite reach says "has an ellipsoid record": never for A = a I (round up to even)
    if (reach < 3e38f) reach = __uint_as_float((__float_as_uint(reach) + 1u) & ~1u);
  }
  return make_float4(mx, my, mz, reach);
}

#ifdef VOGE_BIN_TIMES   // debug builds (tools/bin_times.py): per-workgroup phase timestamps of binA (0) and binB (1)
__device__ unsigned long long g_bin_times[2][1024 * 8];
__device__ unsigned long long g_bin_wave[1024 * 16 * 4];   // binB: per (workgroup, wave): filter start, filter end, registered, done; count in [3] low bits
#define BIN_WTS(k) \
  if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.x < 1024) g_bin_wave[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 4 + (k)] = wall_clock64()
#define BIN_TS(which, k) \
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 1024) g_bin_times[which][8 * blockIdx.x + (k)] = wall_clock64()
#else
#define BIN_TS(which, k)
#define BIN_WTS(k)
#endif

// ------------------------------------------------------------------------------------------
// binA
// ------------------------------------------------------------------------------------------
constexpr int kRoundChunks = 4;                           // 1024-Gaussian chunks a workgroup tests per round
constexpr int kRoundCap = kRoundChunks * kBinThreads;     // candidates (hence at most survivors) of a round

// A segment that outgrows its kSegCap inline entries continues in an EXTENSION: ids only, in chunks of kExtChunk taken
// on demand from an arena of kExtMul * slice_cap(N) ids that belongs to the (region, slice) workgroup (an LDS counter: no
// global one, nothing to reset); up to kExtChunks chunks per segment, their offsets in seg_ext.  A Gaussian falls into one
// or two of a region's super-tiles, so the arena holds whatever the slice produces; only a segment beyond
// kSegCap + kExtChunks * kExtChunk entries (or a used-up arena) still counts as overflowed (binB then re-tests its slice).
constexpr int kExtMul = 3;
constexpr int kExtChunk = 1024;
constexpr int kExtChunks = 16;
__host__ __device__ inline int slice_cap(const int N) {
  return ((N + kParts * kBinThreads - 1) / (kParts * kBinThreads)) * kBinThreads;      // the most Gaussians a slice can hold
}
struct BinALds {
  int arena_top;               // ids of the workgroup's extension arena handed out so far
  int extc[kCh * kCh][kExtChunks];      // per child: where its extension chunks start in the arena
  int extn[kCh * kCh];         // per child: chunks it holds (-1: one could not be had -- the segment counts as overflowed)
  float4 srec[kRoundCap];      // survivors of the region test, this round: record ...
  int sid[kRoundCap];          // ... and Gaussian id, in (chunk, wave, lane) order
  ConeRec child[kCh * kCh];
  ConeRec region;
  ConeRec rowc[kCh];           // cones of the region's kCh rows of super-tiles (the stacked-stripes case, see binA_kernel)
  int use_rows;                // the region's own cone is much wider than its rows': test the rows instead
  int cnt0[kRoundChunks][16];  // [chunk][wave] survivors of the region test -> exclusive prefix
  int base[kCh * kCh];         // entries written so far per child
  int nS;
};

// Gaussian g of batch element b as (centre, reach) -- from the prepared records, or derived on the fly from the
// scalar-sigma inputs (ISO_PREP), in which case `a_out` also receives a.  Two steps, so that a round's loads are all in
// flight before the first one is waited for (round 5: as one function per Gaussian the compiler had serialised the four
// Gaussians of a thread -- load, wait, scalar loads of the view, wait, derive, next -- and a workgroup's first round took
// 3.7 us from kernel entry to "records derived": profiles/r5_bin_times.txt):
//   binA_fetch: the Gaussian's raw words (an index of -1 reads Gaussian 0: every load is unconditional);
//   binA_derive: the record from them and the workgroup-uniform view terms (loaded once, BinAView).
struct BinARaw {
  float4 v;      // ISO_PREP: (mu as stored, sigma word); else the prepared cull record
};
struct BinAView {
  float ox, oy, oz;      // the batch element's origin (centring of Renderer.py:130), or 0
  float fx, fy, fz;      // the camera's forward axis (rasterize_coarse.cu:35), valid if has_f
  bool has_f;
};
template <bool ISO_PREP>
__device__ __forceinline__ BinAView binA_view(const int b, const float *__restrict__ cam_fwd, const IsoView view,
                                              const CamView &cam = no_camera(), const CamK *ck = nullptr) {
  BinAView V;
  V.ox = V.oy = V.oz = 0.f; V.fx = V.fy = V.fz = 0.f; V.has_f = false;
  if (ISO_PREP) {
    if (view.origin != nullptr) { const float *o = view.origin + 3 * b; V.ox = o[0]; V.oy = o[1]; V.oz = o[2]; }
    if (cam_fwd != nullptr) { const float *f = cam_fwd + 3 * b; V.fx = f[0]; V.fy = f[1]; V.fz = f[2]; V.has_f = true; }
    if (cam.R != nullptr) {      // (round 6: the camera itself -- centre and view axis exactly as rays_fwd_kernel / _view_axis make them)
      cam_origin(*ck, cam.T + 3 * b, V.ox, V.oy, V.oz);
      if (cam.behind) { const float *r = cam.R + 9 * b; V.fx = r[2]; V.fy = r[5]; V.fz = r[8]; V.has_f = true; }
    }
  }
  return V;
}
template <bool ISO_PREP>
__device__ __forceinline__ BinARaw binA_fetch(const int g, const int b, const int N, const float4 *__restrict__ cull,
                                              const float *__restrict__ mus, const float *__restrict__ isg, const IsoView view) {
  BinARaw r;
  r.v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (N <= 0) return r;      // (uniform: an empty batch element has nothing to read)
  const int gc = max(g, 0);
  if (!ISO_PREP) { r.v = cull[(size_t)b * N + gc]; return r; }
  const size_t src = view.shared ? (size_t)gc : (size_t)b * N + gc;
  r.v = make_float4(mus[3 * src + 0], mus[3 * src + 1], mus[3 * src + 2], isg[src]);
  return r;
}
template <bool ISO_PREP>
__device__ __forceinline__ float4 binA_derive(const BinARaw r, const bool valid, const BinAView &V, const float thr_act,
                                              const IsoView view, float &a_out) {
  a_out = 0.f;
  if (!valid) return make_float4(0.f, 0.f, 0.f, -1.f);      // (a padding record: reach -1, never kept)
  if (!ISO_PREP) return r.v;
  float mx = r.v.x, my = r.v.y, mz = r.v.z;
  if (view.origin != nullptr || view.cam_origin) { mx -= V.ox; my -= V.oy; mz -= V.oz; }   // centring of Renderer.py:130: the same single fp32 subtraction
  const float a = iso_view_a(r.v.w, view.mode);
  a_out = a;
  float4 c = iso_cull_record(mx, my, mz, a, thr_act);
  if (V.has_f && fmaf(mz, V.fz, fmaf(my, V.fy, mx * V.fx)) < 0.0f) c.w = -1.0f;   // rasterize_coarse.cu:35 ("skip z < 0")
  return c;
}

// How binA deals the N Gaussians of a batch element to its kParts slices (see binA_kernel), for binB, which re-tests a
// slice's members from their records when that slice's segment of a super-tile overflowed kSegCap.
__host__ __device__ inline bool deal_interleaved(const int N) { return N < kParts * kBinThreads; }
__device__ __forceinline__ int slice_of(const int g, const int N) {
  return deal_interleaved(N) ? g % kParts : (g / kBinThreads) % kParts;
}

template <bool ISO_PREP>
__global__ void __launch_bounds__(kBinThreads)
binA_kernel(const ConeRec *__restrict__ cones /* [B][nst] */, const int nstx, const int nsty, const int nst0x,
            const float *__restrict__ mus, const float *__restrict__ isg, const float *__restrict__ cam_fwd, const int N,
            const float thr_act, const IsoView view, float4 *__restrict__ cull, float4 *__restrict__ ms,
            int *__restrict__ seg_count /* [B*nst][kParts] */, int32_t *__restrict__ seg_id /* [B*nst][kParts][kSegCap] */,
            float4 *__restrict__ seg_rec /* the same shape: the entries' cull records */,
            unsigned long long *__restrict__ pool_top /* binB's list pool: reset here, one launch ahead of its first use */,
            int *__restrict__ seg_ext /* [B*nst][kParts][kExtChunks]: starts of the segment's extension chunks in ext_id */,
            int32_t *__restrict__ ext_id /* [B][regions][kParts][ext_arena] */, const int ext_arena,
            const CamView cam /* R != NULL: cones, centre and view axis from the camera; `cones` is not read */) {
  __shared__ BinALds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int region = blockIdx.x / kParts, part = blockIdx.x - region * kParts;
  const int b = blockIdx.y;
#if VOGE_XCD_CHAIN
  if (((gridDim.x / kParts) & 7) == 0) {      // workgroup id % 8 = the XCD (round-robin dispatch): all of a region's slices on XCD region % 8
    const int j = blockIdx.x >> 3;
    region = (blockIdx.x & 7) + 8 * (j / kParts);
    part = j % kParts;
  }
#endif
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *pool_top = 0ull;
  const int rx = region % nst0x, ry = region / nst0x;
  const int nst = nstx * nsty;
  if (tid < kCh * kCh) { L.base[tid] = 0; L.extn[tid] = 0; }
  if (tid == 0) L.arena_top = 0;
  int32_t *const arena = ext_id + ((size_t)(b * (int)(gridDim.x / kParts) + region) * kParts + part) * (size_t)ext_arena;
  int ext_have = 0;      // (wave cc <-> child cc for the whole kernel) extension chunks held; -1: one could not be had
  BIN_TS(0, 0);
  // How the N Gaussians are dealt to the kParts slices.  Large sets: whole 1024-Gaussian chunks round-robin (coalesced
  // loads; a spatially ordered input still spreads over the slices).  Small sets (fewer than kParts chunks) would leave
  // most slices empty and put 1024 candidates into a 512-entry segment as soon as half of them fall into one
  // super-tile -- the fitted ShapeFitting sphere (2562 Gaussians on a 50-pixel object) did exactly that and sent every tile
  // of the image to the stream-everything fallback (sweep 80 -> 780 us).  They are dealt Gaussian by Gaussian instead
  // (g = slot * kParts + part): every slice gets N / kParts of them.
  const bool interleaved = deal_interleaved(N);
  const int nchunks = interleaved ? kParts * ((N + kParts * kBinThreads - 1) / (kParts * kBinThreads))
                                  : (N + kBinThreads - 1) / kBinThreads;
  const bool writes_records = ISO_PREP && region == 0;   // region 0's slices cover every Gaussian exactly once
  // the first round's candidates do not depend on the cones: their loads go out first
  float4 c[kRoundChunks];
  float av[kRoundChunks];
  int gq[kRoundChunks];
  CamK ck;
  if (cam.R != nullptr) ck = cam_load(cam, b);      // (uniform)
  const BinAView V = binA_view<ISO_PREP>(b, cam_fwd, view, cam, &ck);
  if (cam.R != nullptr && cam.origin_out != nullptr && blockIdx.x == 0 && tid == 0) {
    cam.origin_out[3 * b] = V.ox; cam.origin_out[3 * b + 1] = V.oy; cam.origin_out[3 * b + 2] = V.oz;
  }
  BinARaw raw[kRoundChunks];
  auto fetch_round = [&](const int j0) {      // (all of the round's loads go out together ...)
#pragma unroll
    for (int q = 0; q < kRoundChunks; ++q) {
      const int j = j0 + q * kParts;      // (j % kParts == part)
      const int g = interleaved ? ((j / kParts) * kBinThreads + tid) * kParts + part : j * kBinThreads + tid;
      gq[q] = (j < nchunks && g < N) ? g : -1;
      raw[q] = binA_fetch<ISO_PREP>(gq[q], b, N, cull, mus, isg, view);
    }
  };
  auto derive_round = [&]() {                 // (... and the records are derived once they are here)
#pragma unroll
    for (int q = 0; q < kRoundChunks; ++q) c[q] = binA_derive<ISO_PREP>(raw[q], gq[q] >= 0, V, thr_act, view, av[q]);
  };
  auto load_round = [&](const int j0) { fetch_round(j0); derive_round(); };
  fetch_round(part);
  // ---- the region's child cones, and its own cone as their conservative union (wave 0, lane <-> child): a ray of
  // child i makes at most alpha_i + theta_i with the parent axis (alpha_i = angle between the axes), so
  //   cos >= cos(alpha_i) cs_i - sin(alpha_i) sn_i ,   sin <= sin(alpha_i) + cos(alpha_i) sn_i .
  if (wave == 0) {
    const int cc = lane & (kCh * kCh - 1);
    const int cx = rx * kCh + (cc & (kCh - 1)), cy = ry * kCh + cc / kCh;
    ConeRec r = {0.f, 0.f, 1.f, 1.f, 0.f, -1.f, 0.f, 0.f};      // ok = -1: no such super-tile
    // (VOGE_BINA_REGION_RECT, round 6: with the camera at hand and a contiguous band the region's OWN cone comes from its own four
    //  corner rays -- lane 16 makes it in the same call that makes the sixteen children -- instead of the conservative union of the
    //  children below: tighter, and the union's chain of wave reductions leaves the head of every workgroup)
    const bool region_rect = (VOGE_BINA_REGION_RECT != 0) && cam.R != nullptr && cam.h <= cam.stripe_h;      // (uniform)
    if (cam.R != nullptr) {      // (uniform branch)
      const bool reg = region_rect && lane == kCh * kCh;
      const int j0 = reg ? rx * kCh * kST : cx * kST, i0 = reg ? ry * kCh * kST : cy * kST, ext = reg ? kCh * kST : kST;
      if ((cx < nstx && cy < nsty) || reg) r = cam_rect_cone(ck, cam, j0, j0 + ext - 1, i0, i0 + ext - 1);
    } else if (cx < nstx && cy < nsty) {
      r = cones[cone_super_at(b, nst, cy * nstx + cx)];
    }
    if (lane < kCh * kCh) L.child[lane] = r;
    if (region_rect) {
      if (lane == kCh * kCh) L.region = r;      // (ok = -1 cannot happen: the region holds at least one super-tile of the image)
      if (lane == 0) L.use_rows = 0;
    } else {
    const bool present = lane < kCh * kCh && r.ok >= 0.f;
    const float sx = wave_sum_dpp(present ? r.ax : 0.f), sy = wave_sum_dpp(present ? r.ay : 0.f), sz = wave_sum_dpp(present ? r.az : 0.f);
    const float npres = wave_sum_dpp(present ? 1.f : 0.f);
    bool ok = __all(!present || r.ok > 0.f);
    const float n = sqrtf(fmaf(sz, sz, fmaf(sy, sy, sx * sx)));
    const float ax = sx / n, ay = sy / n, az = sz / n;
    // a child's extrema with respect to an axis (ax, ay, az): cos >= cos(alpha) cs - sin(alpha) sn, sin <= sin(alpha) + cos(alpha) sn
    auto child_extrema = [&](const float ax_, const float ay_, const float az_, float &smax_, float &cmin_, bool &ok_) {
      smax_ = 0.f; cmin_ = 1.f;
      if (present) {
        const float ca = fmaf(r.az, az_, fmaf(r.ay, ay_, r.ax * ax_));
        const float qx = fmaf(-ca, ax_, r.ax), qy = fmaf(-ca, ay_, r.ay), qz = fmaf(-ca, az_, r.az);
        const float sa = sqrtf(fmaf(qz, qz, fmaf(qy, qy, qx * qx))) * (1.0f + 1e-6f) + 1e-7f;
        const float cl = fminf(ca, 1.0f) - 1e-7f;
        if (!(cl > 0.0f)) ok_ = false;
        cmin_ = fmaf(cl, r.cs, -sa * r.sn);
        smax_ = fmaf(fminf(ca + 1e-7f, 1.0f), r.sn, sa);
      }
    };
    float smax, cmin;
    child_extrema(ax, ay, az, smax, cmin, ok);
    ok = __all(ok);
    smax = wave_max(smax); cmin = wave_min(cmin);
    const Cone cn = cone_finish(ax, ay, az, n / fmaxf(npres, 1.f), smax, cmin, ok);
    if (lane == 0) L.region = ConeRec{cn.ax, cn.ay, cn.az, cn.cs, cn.sn, cn.ok ? 1.f : 0.f, 0.f, 0.f};
    // ---- the same union per ROW of super-tiles (lanes 4g .. 4g+3 = a DPP quad).  A frame's region is compact and its
    // own cone is what culls; but the rows of a STACKED image (a rank's interleaved stripes of a frame dealt over several
    // GPUs: voge_rays_striped_fwd) come from distant parts of the frame: the region's cone then spans all of them and
    // keeps nearly every Gaussian (binA 31 -> 75 us per rank at 200k / 1024^2 / 8 ranks), while each row's cone is tight.
    // (computed only when the region's cone is far wider than a compact 4x4 block of its children would make it --
    // uniform: a frame's regions never take this branch)
    const float widest_child = wave_max(present ? (r.ok > 0.f ? r.sn : 2.0f) : 0.f);
    if (lane == 0) L.use_rows = 0;
    if (__builtin_expect((!cn.ok || cn.sn > 8.0f * widest_child) && N < (1 << 27), 0)) {      // (the row mask rides in the id's top bits)
      auto quad_sum = [](float v) { v += VOGE_DPP(v, 0xB1); v += VOGE_DPP(v, 0x4E); return v; };
      auto quad_max = [](float v) { v = fmaxf(v, VOGE_DPP(v, 0xB1)); v = fmaxf(v, VOGE_DPP(v, 0x4E)); return v; };
      auto quad_min = [](float v) { v = fminf(v, VOGE_DPP(v, 0xB1)); v = fminf(v, VOGE_DPP(v, 0x4E)); return v; };
      static_assert(kCh == 4, "a row of super-tiles is a DPP quad");
      const float qsx = quad_sum(present ? r.ax : 0.f), qsy = quad_sum(present ? r.ay : 0.f), qsz = quad_sum(present ? r.az : 0.f);
      const float qn = quad_sum(present ? 1.f : 0.f);
      const float nn = sqrtf(fmaf(qsz, qsz, fmaf(qsy, qsy, qsx * qsx)));
      const float rax = qsx / nn, ray = qsy / nn, raz = qsz / nn;
      float rs, rc;
      bool rok = !present || r.ok > 0.f;
      child_extrema(rax, ray, raz, rs, rc, rok);
      rs = quad_max(rs); rc = quad_min(rc);
      const float okf = quad_min(rok ? 1.f : 0.f);
      const Cone rcn = cone_finish(rax, ray, raz, nn / fmaxf(qn, 1.f), rs, rc, okf != 0.f);
      // (a row without super-tiles: ok = -1, nothing is kept for it)
      if (lane < kCh * kCh && (lane & 3) == 0) L.rowc[lane >> 2] = ConeRec{rcn.ax, rcn.ay, rcn.az, rcn.cs, rcn.sn, qn > 0.f ? (rcn.ok ? 1.f : 0.f) : -1.f, 0.f, 0.f};
      const float widest_row = wave_max((lane < kCh * kCh && qn > 0.f) ? (rcn.ok ? rcn.sn : 2.0f) : 0.f);
      if (lane == 0) L.use_rows = (!cn.ok || cn.sn > 1.6f * widest_row) ? 1 : 0;
    }
    }      // (!region_rect)
  }
  BIN_TS(0, 6);      // (wave 0: the region's cone is in LDS)
  // (the first round's records: derived behind the cone block, so that wave 0's cone loads -- requested right after its
  // Gaussians -- and the region's cone overlap the Gaussians' way in instead of following it)
  derive_round();
  BIN_TS(0, 7);      // (the first round's Gaussians have arrived and their records are derived)
  __syncthreads();
  const Cone rcone = load_cone(L.region);
  const bool use_rows = L.use_rows != 0;      // (uniform)
  BIN_TS(0, 1);

  // ---- the slice's Gaussians: chunks of 1024 dealt round-robin to the kParts slices (so a spatially ordered
  // input, e.g. mesh vertices, still spreads evenly over the segments), kRoundChunks chunks per round ----
  // (two copies of the round loop, chosen once per workgroup: the stacked-rows form carries a row mask with every
  // survivor; a frame's regions run the original code, instruction for instruction)
  auto rounds = [&](auto rows_tag) {
    constexpr bool ROWS = decltype(rows_tag)::value;
    for (int j0 = part; j0 < nchunks; j0 += kParts * kRoundChunks) {
      // (1) every candidate of the round against the region's cone; the survivors' records go to LDS in a fixed order
      if (j0 != part) load_round(j0);
      bool k0[kRoundChunks];
      unsigned rowm[kRoundChunks];
  #pragma unroll
      for (int q = 0; q < kRoundChunks; ++q) {
        if (writes_records && gq[q] >= 0) {
          cull[(size_t)b * N + gq[q]] = c[q];
          ms[(size_t)b * N + gq[q]] = make_float4(c[q].x, c[q].y, c[q].z, av[q]);
        }
        k0[q] = cone_keep(c[q], rcone);     // (a padding record has reach -1: never kept)
        rowm[q] = 0xFu;
        if (ROWS) {      // the rows' cones instead (see above): bit g of the mask = row g can be hit
          rowm[q] = 0u;
  #pragma unroll
          for (int g = 0; g < kCh; ++g) {
            const ConeRec rr = L.rowc[g];
            if (rr.ok >= 0.f && cone_keep(c[q], load_cone(rr))) rowm[q] |= 1u << g;
          }
          k0[q] = rowm[q] != 0u;
        }
      }
      unsigned long long m0[kRoundChunks];
  #pragma unroll
      for (int q = 0; q < kRoundChunks; ++q) {
        m0[q] = __ballot(k0[q]);
        if (lane == 0) L.cnt0[q][wave] = __popcll(m0[q]);
      }
      __syncthreads();
      // lane <-> (chunk, wave): exclusive prefix of the 64 counts -- by EVERY wave for itself (round 5: wave 0 used to scan
      // them for all and hand the offsets back through LDS behind one more barrier)
      static_assert(kRoundChunks * 16 == 64, "one lane per (chunk, wave)");
      int nS;
      int my_off[kRoundChunks];
      {
        const int v = (&L.cnt0[0][0])[lane];
        const int x = wave_incl_scan_i32(v);
        nS = __builtin_amdgcn_readlane(x, 63);
  #pragma unroll
        for (int q = 0; q < kRoundChunks; ++q) my_off[q] = __shfl(x - v, q * 16 + wave, 64);
      }
  #pragma unroll
      for (int q = 0; q < kRoundChunks; ++q)
        if (k0[q]) {
          const int pos = my_off[q] + __popcll(m0[q] & ((1ull << lane) - 1ull));
          L.srec[pos] = c[q];
          L.sid[pos] = ROWS ? (gq[q] | (int)(rowm[q] << 27)) : gq[q];      // (ids stay below 2^27: the row mask rides in bits 27..30)
        }
      __syncthreads();
      BIN_TS(0, 2);
      // (2) survivors x children: wave cc walks ALL survivors against child cc's cone (in registers), four 64-survivor
      // batches per trip.  It is the only writer of that child's segment, so its running count IS the fill position:
      // one pass, no counters in LDS, no barrier, and the order is a pure function of the inputs.
      {
        const int cc = wave;
        const ConeRec cr = L.child[cc];
        if (cr.ok >= 0.f) {                // (uniform) the super-tile exists
          const Cone ck = load_cone(cr);
          const int ccx = rx * kCh + (cc & (kCh - 1)), ccy = ry * kCh + cc / kCh;
          const size_t seg0 = (((size_t)b * nst + ccy * nstx + ccx) * kParts + part) * kSegCap;
          int32_t *seg = seg_id + seg0;
          float4 *segr = seg_rec + seg0;      // (the record rides along: binB then streams it instead of gathering by id)
          int fill = L.base[cc];
          constexpr int kU = 4;
          for (int s0 = 0; s0 < nS; s0 += 64 * kU) {
            float4 r[kU];
            int id[kU];
  #pragma unroll
            for (int q = 0; q < kU; ++q) {
              const int si = s0 + q * 64 + lane;
              r[q] = (si < nS) ? L.srec[si] : make_float4(0.f, 0.f, 0.f, -1.f);
              id[q] = (si < nS) ? L.sid[si] : -1;
            }
            bool kp[kU];
  #pragma unroll
            for (int q = 0; q < kU; ++q) {
              bool in_row = true;
              if (ROWS) {      // (a frame's regions never carry a mask)
                in_row = id[q] >= 0 && (((unsigned)id[q] >> (27 + (cc >> 2))) & 1u) != 0u;      // its row of super-tiles can be hit
                id[q] = (id[q] >= 0) ? (id[q] & 0x07ffffff) : -1;
              }
              kp[q] = in_row && cone_keep(r[q], ck);      // (padding: reach -1, never kept)
            }
            const int fill0 = fill;
  #pragma unroll
            for (int q = 0; q < kU; ++q) {
              const unsigned long long m = __ballot(kp[q]);
              if (kp[q]) {
                const int pos = fill + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < kSegCap) { seg[pos] = id[q]; segr[pos] = r[q]; }
              }
              fill += __popcll(m);
            }
  #ifndef VOGE_NO_SEG_EXT      // (A/B builds: no extensions -- a segment beyond kSegCap counts as overflowed)
            if (__builtin_expect(fill > kSegCap, 0)) {      // (uniform, rare) this trip's entries beyond the inline part
              const int need = (fill - kSegCap + kExtChunk - 1) / kExtChunk;
              while (ext_have >= 0 && ext_have < need) {      // (uniform) one more chunk from the workgroup's arena
                int at = 0;
                if (lane == 0) at = atomicAdd(&L.arena_top, kExtChunk);
                at = __builtin_amdgcn_readfirstlane(at);
                if (ext_have < kExtChunks && at + kExtChunk <= ext_arena) {
                  if (lane == 0) L.extc[cc][ext_have] = at;
                  ++ext_have;
                } else {
                  ext_have = -1;
                }
              }
              wave_lds_sync();
              if (ext_have >= 0) {
                int f = fill0;
  #pragma unroll
                for (int q = 0; q < kU; ++q) {
                  const unsigned long long m = __ballot(kp[q]);
                  const int e = f + __popcll(m & ((1ull << lane) - 1ull)) - kSegCap;
                  if (kp[q] && e >= 0) arena[*lds_volatile(&L.extc[cc][e / kExtChunk]) + (e % kExtChunk)] = id[q];
                  f += __popcll(m);
                }
              }
            }
  #endif
          }
          if (lane == 0) { L.base[cc] = fill; L.extn[cc] = ext_have; }
        }
      }
      BIN_TS(0, 3);
      __syncthreads();     // srec / sid / cnt are rewritten by the next round
      BIN_TS(0, 4);
    }
  };
  if (__builtin_expect(!use_rows, 1)) rounds(std::false_type{}); else rounds(std::true_type{});
  BIN_TS(0, 5);
  if (tid < kCh * kCh) {
    const int ccx = rx * kCh + (tid & (kCh - 1)), ccy = ry * kCh + tid / kCh;
    if (ccx < nstx && ccy < nsty) {
      const size_t so = ((size_t)b * nst + ccy * nstx + ccx) * kParts + part;
      const bool over = L.base[tid] > kSegCap;
      seg_count[so] = (over && L.extn[tid] < 0) ? -1 : L.base[tid];      // (> kSegCap: the rest is in the extension chunks)
      if (over && L.extn[tid] > 0)
        for (int q = 0; q < L.extn[tid]; ++q) seg_ext[so * kExtChunks + q] = (int)(arena - ext_id) + L.extc[tid][q];
    }
  }
}

// ------------------------------------------------------------------------------------------
// binB: one 256-thread workgroup per QUAD (16x16 pixels = 2x2 sweep tiles, a quarter of a super-tile), one wave
// per sweep tile.  A whole super-tile per workgroup (round 2's first form) left the chip as imbalanced as the
// scene: the 16 tile filters of a central super-tile (5000 candidates) kept ONE CU busy for 18 us while the
// others idled.  A quad's workgroup first cuts the super-tile's candidates down with the quad's own cone (~40 %
// survive), so its sort and its four tile filters touch a fraction of the list, and four times as many
// workgroups share the work.
// ------------------------------------------------------------------------------------------
constexpr int kQuad = 16;           // quad edge (pixels)
constexpr int kQT = 256;            // threads: 4 waves <-> the quad's 2x2 sweep tiles
constexpr int kQCap = 3008;         // entries of a quad's ordered list (LDS sort capacity; 39.5 KB of LDS: 4 workgroups per CU)
constexpr int kQRec = kQCap / 2;       // ... whose cull records are kept in LDS for the tile filters (the rest: gathered)
constexpr int kBuckets = 512;       // depth buckets of the counting sort
constexpr int kTilesPerQuad = 4;
// order[].x of a tile whose list lives in the POOL (a quad with more than kQCap candidates: binB's long path) carries this
// flag; the list then starts at pool entry tl_off[tile] and may be longer than kTileCap.
constexpr int kPoolFlag = 1 << 30;
struct BinLds {
  union {
    uint64_t keys[kQCap];     // (ord(depth key) << 32 | flag << 31 | id) of the quad's candidates, unordered
    float4 rec[kQRec];        // after the sort: cull records of the ordered list's first kQRec entries
    int hist4[kTilesPerQuad][kBuckets];      // long path: per-tile bucket counters / cursors
  };
  uint32_t sorted[kQCap];     // (flag << 31 | id), front to back (an entry's depth key is recomputed from its record)
  float red[4 * 8];
  union {
    int hist[kBuckets];
    int exto[kParts][kExtChunks];      // (while the sources are streamed) where each segment's extension chunks start in ext_id
  };
  uint32_t bmin[kBuckets];   // per bucket: smallest own len bound of the entries with an ellipsoid record (ord)
  int wsum[8];
  int segn[kParts + 1];      // exclusive prefix of the segments' INLINE counts (an overflowed segment counts as empty here)
  int extn[kParts + 1];      // exclusive prefix of the segments' extension counts (entries beyond kSegCap, ids only)
  unsigned ovf;              // bit p: slice p's segment overflowed kSegCap -- its Gaussians are re-tested from their records
  int count;
  int nflag;      // entries with an ellipsoid record
  int spill;      // a tile list overflowed kTileCap: the quad's ordered list goes to memory as its fallback
  float tc[kTilesPerQuad][8];      // long path: the four tile cones (ax, ay, az, cs, sn, ok)
  int toff[kTilesPerQuad];         // long path: where each tile's list starts in the pool (-1: the pool is exhausted)
};

#ifndef VOGE_ELL_KEY
#define VOGE_ELL_KEY 0
#endif

#ifndef VOGE_BINB_XCD
#define VOGE_BINB_XCD 0      // 1: the four quads of a super-tile on one XCD (measured: entry 69.8 -> 70.9 us, lean equal -- off)
#endif
#ifndef VOGE_BINB_GU
#define VOGE_BINB_GU 8
#endif
constexpr int kGU = VOGE_BINB_GU;      // gathers in flight per thread of binB's source passes

// What one pass over a quad's sources needs: the super-tile's segments (as binA filled them), the per-Gaussian records
// for the slices whose segment overflowed, and the quad's cone.
struct BinStream {
  const float4 *cullb, *ellb, *segr;
  const int32_t *segs, *ext_id;
  const int *exto;      // LDS: [kParts][kExtChunks] starts of the extension chunks
  int n_src, n_ext, N;
  unsigned ovf;
  Cone qcone;
};
// The quad-cone test of one batch (kGU entries per thread), then `sink(ids, records, kept, has ellipsoid, ellipsoid key)`.
// ELL (here and below): Gaussians with an ellipsoid record can occur (the general entry points); the scalar-sigma ones
// instantiate everything without that code.
template <bool ELL, class Sink>
__device__ __forceinline__ void bin_test_batch(const BinStream &S, const int (&gid)[kGU], const float4 (&c)[kGU], Sink &&sink) {
  bool kp[kGU], el[kGU];
  float gkey[kGU];
#pragma unroll
  for (int j = 0; j < kGU; ++j) {
    kp[j] = cone_keep(c[j], S.qcone);         // (padding: reach -1, never kept)
    el[j] = ELL && kp[j] && cull_has_ell(c[j]);
    gkey[j] = 0.0f;
    if (ELL && __any(el[j])) {
      if (el[j]) {
        const float4 e0 = S.ellb[2 * (size_t)gid[j]], e1 = S.ellb[2 * (size_t)gid[j] + 1];
        kp[j] = cone_keep_ell(c[j], e0, e1, S.qcone);
        el[j] = kp[j];
        // order key: VOGE_ELL_KEY 0 = the centre's depth along the axis, 1 = the entry's own lower bound of len
        const float pa = fmaf(c[j].z, S.qcone.az, fmaf(c[j].y, S.qcone.ay, c[j].x * S.qcone.ax));
#if VOGE_ELL_KEY == 1
        gkey[j] = pa - ell_support(e0, e1, S.qcone.ax, S.qcone.ay, S.qcone.az) + 0.0f;
#else
        gkey[j] = pa + 0.0f;
#endif
      }
    }
  }
  sink(gid, c, kp, el, gkey);
}
// One pass over the quad's sources, kGU entries per thread at a time; the sink sees every batch, wave-uniformly.
template <bool ELL, class Sink>
__device__ __forceinline__ void bin_stream_sources(const BinStream &S, const int *segn /* LDS: exclusive prefix of the segment counts;
                                                   behind it (BinLds): extn[kParts + 1] */, const int tid, Sink &&sink) {
  // (A thread's stream indices only ever grow, so the segment an index falls into is carried along: one LDS read and a
  // compare per entry.  Round 3 searched all 15 boundaries for every entry -- 15 LDS reads and 30 VALU instructions of the
  // ~80 a streamed entry cost: the source pass of a central quad, 7.3 us of its 16, was bound by exactly that.)
  int seg_p = 0, seg_lo = 0, seg_hi = segn[1];
  for (int base = 0; base < S.n_src; base += kQT * kGU) {      // the segments binA filled
    int gid[kGU];
    float4 c[kGU];
#pragma unroll
    for (int j = 0; j < kGU; ++j) {     // ids and records are two streams: one round trip, nothing dependent
      const int i = base + j * kQT + tid;
      gid[j] = -1;
      c[j] = make_float4(0.f, 0.f, 0.f, -1.f);
      if (i < S.n_src) {
        while (i >= seg_hi) { seg_lo = seg_hi; ++seg_p; seg_hi = segn[seg_p + 1]; }      // (i < n_src = segn[kParts]: ends)
        const int o = seg_p * kSegCap + (i - seg_lo);
        gid[j] = S.segs[o];
        c[j] = S.segr[o];
      }
    }
    bin_test_batch<ELL>(S, gid, c, sink);
  }
  if (S.n_ext > 0) {      // (rare) the part of long segments that binA put into their extensions: ids, records by gather
    const int *extn = segn + (kParts + 1), *exto = S.exto;
    for (int base = 0; base < S.n_ext; base += kQT * kGU) {
      int gid[kGU];
      float4 c[kGU];
#pragma unroll
      for (int j = 0; j < kGU; ++j) {
        const int i = base + j * kQT + tid;
        gid[j] = -1;
        if (i < S.n_ext) {
          int p = 0;
#pragma unroll
          for (int q = 1; q < kParts; ++q) p += (i >= extn[q]) ? 1 : 0;
          const int e = i - extn[p];
          gid[j] = S.ext_id[(size_t)exto[p * kExtChunks + e / kExtChunk] + (size_t)(e % kExtChunk)];
        }
      }
#pragma unroll
      for (int j = 0; j < kGU; ++j) c[j] = (gid[j] >= 0) ? S.cullb[gid[j]] : make_float4(0.f, 0.f, 0.f, -1.f);
      bin_test_batch<ELL>(S, gid, c, sink);
    }
  }
  if (S.ovf != 0u) {      // (rarer) slices whose segment overflowed: their Gaussians straight from the per-Gaussian records
    for (int base = 0; base < S.N; base += kQT * kGU) {
      int gid[kGU];
      float4 c[kGU];
#pragma unroll
      for (int j = 0; j < kGU; ++j) {
        const int g = base + j * kQT + tid;
        const bool ok = g < S.N && ((S.ovf >> slice_of(g, S.N)) & 1u) != 0u;
        gid[j] = ok ? g : -1;
        c[j] = ok ? S.cullb[g] : make_float4(0.f, 0.f, 0.f, -1.f);
      }
      bin_test_batch<ELL>(S, gid, c, sink);
    }
  }
}
// Slots of a batch's survivors in the workgroup's compacted order: ONE LDS atomic per wave and batch (rounds 1-2: one per
// wave and ENTRY -- eight dependent atomic + broadcast chains per batch).
__device__ __forceinline__ void bin_batch_slots(int *count /* LDS */, const int lane, const bool (&kp)[kGU], int (&slot)[kGU]) {
  unsigned long long m[kGU];
  int cw = 0;
#pragma unroll
  for (int j = 0; j < kGU; ++j) { m[j] = __ballot(kp[j]); cw += __popcll(m[j]); }
  int start = 0;
  if (cw > 0) {      // uniform
    if (lane == 0) start = atomicAdd(count, cw);
    start = __builtin_amdgcn_readfirstlane(start);
  }
#pragma unroll
  for (int j = 0; j < kGU; ++j) {
    slot[j] = start + __popcll(m[j] & ((1ull << lane) - 1ull));
    start += __popcll(m[j]);
  }
}
// The depth buckets of a quad's candidates and the len bounds they stand for.
struct BinKeys {
  float lo, scale, span, slack;      // bucket b covers keys [lo + (b - 1) / scale, ...); slack: 1.13 x the largest sphere reach
  bool flagged;                      // some entry has an ellipsoid record: L.bmin holds suffix minima of their own bounds
};
__device__ __forceinline__ int bin_bucket_of(const BinKeys &Kk, const float v) {
  return (v > -INFINITY) ? 1 + min(kBuckets - 2, max(0, (int)((v - Kk.lo) * Kk.scale))) : 0;
}
// own lower bound of len of an entry with an ellipsoid record: the peak point x = len d of a hit lies in the
// ellipsoid, so len (d.a) = x.a >= t = mu.a - h(a), and d.a is in [cs, 1]
__device__ __forceinline__ float bin_own_bound(const BinStream &S, const size_t g) {
  const float4 cj = S.cullb[g], e0 = S.ellb[2 * g], e1 = S.ellb[2 * g + 1];
  float bnd = -INFINITY;
  if (S.qcone.ok) {
    const float pa = fmaf(cj.z, S.qcone.az, fmaf(cj.y, S.qcone.ay, cj.x * S.qcone.ax));
    const float nm1 = fabsf(cj.x) + fabsf(cj.y) + fabsf(cj.z);
    const float t = pa - ell_support(e0, e1, S.qcone.ax, S.qcone.ay, S.qcone.az) - 4e-6f * nm1;
    bnd = (t >= 0.0f) ? t : t / S.qcone.cs;
    bnd = bnd - 1e-5f * fabsf(bnd) - 1e-30f;
  }
  return bnd;
}
// Suffix minimum over the buckets of the flagged entries' own bounds (thread <-> two buckets).  (one barrier inside)
__device__ __forceinline__ void bin_bmin_suffix(uint32_t *bmin, int *wsum, const int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t m4[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) m4[q] = bmin[2 * tid + q];
  m4[0] = min(m4[0], m4[1]);
  uint32_t x = m4[0];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_down(x, o, 64);
    if (lane + o < 64) x = min(x, y);
  }
  if (lane == 0) wsum[4 + wave] = (int)x;
  __syncthreads();
  // x = minimum from this thread's first bucket to the end of the wave; beyond: the later waves' minima
  uint32_t later = 0xffffffffu;      // minimum over the lanes / waves behind this thread
  {
    const uint32_t nxt = __shfl_down(x, 1, 64);
    later = (lane < 63) ? nxt : 0xffffffffu;
    for (int w = wave + 1; w < 4; ++w) later = min(later, (uint32_t)wsum[4 + w]);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) bmin[2 * tid + q] = min(m4[q], later);
}
// the len bound of an entry from its depth key: its bucket's lower edge (see binB_kernel)
__device__ __forceinline__ float bin_len_bound(const BinKeys &Kk, const uint32_t *bmin, const float v) {
  float edge = -INFINITY;
  int qb = 0;
  if (v > -INFINITY) {
    const int q = min(kBuckets - 2, max(0, (int)((v - Kk.lo) * Kk.scale)));
    edge = fminf(v, Kk.lo + (float)q * (Kk.span / (float)(kBuckets - 2)) - 4e-6f * Kk.span);
    qb = 1 + q;
  }
  // sphere-only entries: bucket edge minus the largest sphere reach; entries with an ellipsoid record: the
  // smallest own bound from this bucket on.  Both are monotone along the list.
  const float lb_sphere = (edge > -INFINITY) ? edge - Kk.slack - 1e-5f * fabsf(edge) - 1e-30f : -INFINITY;
  return Kk.flagged ? fminf(lb_sphere, ord2f(bmin[qb])) : lb_sphere;
}

// ---- binB's LONG PATH: a quad with more candidates than the LDS sort holds (tens of thousands of Gaussians behind a few
// dozen pixels: a small object, a zoomed-out view).  Rounds 1-3 sent such a quad's tiles to the stream-everything fallback
// (trace 0.1 -> 3 ms).  Here the sources are streamed once more to put the quad's survivors into a POOL
// (their number is known from the kernel's first pass); two walks over the survivors then count, per tile and depth
// bucket, the candidates that pass the tile's cone (and collect the flagged entries' bounds), and scatter them -- bucket
// by bucket, i.e. front to back -- into per-tile lists of exactly the needed length, also from the pool.  Inside a bucket
// the order is arbitrary, as in the LDS sort; every entry carries its bucket's bound.  Only an exhausted pool still falls
// back.  
struct BinLong {
  BinStream S;
  BinKeys Kk;
  Cone tcone;
  bool tile_ok;
  int tile, quad, total;
  unsigned long long *pool_top;      // (64 bits: the counter runs on past an exhausted pool and must not wrap)
  int *tl_off, *tl_count, *q_count;
  const int *seg_ext;
  int bin;
  int pool_cap;
  int32_t *pool_id;
  float *pool_lb;
  int2 *my_order;
};
template <bool ELL>
__device__ __forceinline__ void binB_long_path(const BinLong A, BinLds &L) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const BinStream &S = A.S;
  const int total = A.total;
  __syncthreads();                                   // (everyone is done with L.red / L.count of pass one)
  for (int i = tid; i < kTilesPerQuad * kBuckets; i += kQT) (&L.hist4[0][0])[i] = 0;
  if (S.n_ext > 0) {      // the extension chunk tables again (they share their LDS with the histogram the kernel just cleared)
    const int p = tid / kExtChunks, q = tid % kExtChunks;
    const int ne = L.extn[p + 1] - L.extn[p];
    L.exto[p][q] = (q * kExtChunk < ne) ? A.seg_ext[((size_t)A.bin * kParts + p) * kExtChunks + q] : 0;
  }
  if (lane == 0) {
    L.tc[wave][0] = A.tcone.ax; L.tc[wave][1] = A.tcone.ay; L.tc[wave][2] = A.tcone.az; L.tc[wave][3] = A.tcone.cs;
    L.tc[wave][4] = A.tcone.sn; L.tc[wave][5] = (A.tcone.ok || !A.tile_ok) ? 1.0f : 0.0f;      // (a tile outside the image keeps nothing)
    L.tc[wave][6] = A.tile_ok ? 1.0f : 0.0f;
  }
  if (tid == 0) {
    // the quad's survivors -- (id | flag, depth key) -- go to the pool first: the sources (up to N entries where
    // segments overflowed) are streamed ONCE more, everything after that walks the `total` survivors only
    const unsigned long long at64 = atomicAdd(A.pool_top, (unsigned long long)total);
    int at = (at64 + (unsigned long long)total <= (unsigned long long)A.pool_cap) ? (int)at64 : -1;      // -1: exhausted (the counter runs on)
    L.toff[0] = at;
    L.count = 0;
  }
  __syncthreads();
  const int surv_at = L.toff[0];
  bool pooled = surv_at >= 0;
  int tile_n[kTilesPerQuad] = {0, 0, 0, 0};
  if (pooled) {
    int32_t *sv_id = A.pool_id + surv_at;
    float *sv_key = A.pool_lb + surv_at;
    bin_stream_sources<ELL>(S, L.segn, tid, [&](const int (&g)[kGU], const float4 (&cj)[kGU], const bool (&kp)[kGU],
                                            const bool (&el)[kGU], const float (&gkey)[kGU]) {
      int slot[kGU];
      bin_batch_slots(&L.count, lane, kp, slot);
#pragma unroll
      for (int j = 0; j < kGU; ++j) {
        if (kp[j]) {      // (the same tests on the same data as pass one: slot < total)
          sv_id[slot[j]] = (int32_t)((uint32_t)g[j] | (el[j] ? 0x80000000u : 0u));
          sv_key[slot[j]] = el[j] ? gkey[j] : depth_key(cj[j], S.qcone);
        }
      }
    });
    __threadfence_block();
    __syncthreads();
    BIN_TS(1, 2);
    auto tile_mask = [&](const size_t g, const float4 cj, const bool el) {
      unsigned mk = 0u;
#pragma unroll 1
      for (int t = 0; t < kTilesPerQuad; ++t) {
        Cone tk;
        tk.ax = L.tc[t][0]; tk.ay = L.tc[t][1]; tk.az = L.tc[t][2]; tk.cs = L.tc[t][3]; tk.sn = L.tc[t][4];
        tk.ok = L.tc[t][5] != 0.0f;
        bool k2 = (L.tc[t][6] != 0.0f) && cone_keep(cj, tk);
        if (k2 && el) k2 = cone_keep_ell(cj, S.ellb[2 * g], S.ellb[2 * g + 1], tk);
        mk |= k2 ? (1u << t) : 0u;
      }
      return mk;
    };
    // count, per tile and depth bucket, the survivors of the tile's cone; the flagged entries' own bounds
    for (int i = tid; i < total; i += kQT) {
      const uint32_t word = (uint32_t)sv_id[i];
      const size_t g = word & 0x7fffffffu;
      const bool el = ELL && (word & 0x80000000u) != 0u;
      const int q = bin_bucket_of(A.Kk, sv_key[i]);
      const unsigned mk = tile_mask(g, S.cullb[g], el);
#pragma unroll
      for (int t = 0; t < kTilesPerQuad; ++t)
        if ((mk >> t) & 1u) atomicAdd(&L.hist4[t][q], 1);
      if (el) atomicMin(&L.bmin[q], f2ord(bin_own_bound(S, g)));
    }
    __syncthreads();
    // exclusive scans of the four tiles' bucket counters (two buckets per thread); totals -> tile_n
    for (int t = 0; t < kTilesPerQuad; ++t) {
      const int2 v = *reinterpret_cast<const int2 *>(&L.hist4[t][2 * tid]);
      const int s4 = v.x + v.y;
      int x = s4;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
      }
      if (lane == 63) L.wsum[wave] = x;
      __syncthreads();
      int off = x - s4;
      for (int w = 0; w < wave; ++w) off += L.wsum[w];
      tile_n[t] = L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
      *reinterpret_cast<int2 *>(&L.hist4[t][2 * tid]) = make_int2(off, off + v.x);
      __syncthreads();
    }
    if (ELL && A.Kk.flagged) bin_bmin_suffix(L.bmin, L.wsum, tid);
    if (tid == 0) {
      const int need = tile_n[0] + tile_n[1] + tile_n[2] + tile_n[3];
      const unsigned long long at64 = (need > 0) ? atomicAdd(A.pool_top, (unsigned long long)need) : 0ull;
      int at = (at64 + (unsigned long long)need <= (unsigned long long)A.pool_cap) ? (int)at64 : -1;
#pragma unroll
      for (int t = 0; t < kTilesPerQuad; ++t) { L.toff[t] = at; if (at >= 0) at += tile_n[t]; }
    }
    __syncthreads();
    pooled = L.toff[0] >= 0;
    BIN_TS(1, 3);
    if (pooled) {      // scatter: bucket by bucket, i.e. front to back; inside a bucket in whatever order the atomics give
      for (int i = tid; i < total; i += kQT) {
        const uint32_t word = (uint32_t)sv_id[i];
        const size_t g = word & 0x7fffffffu;
        const bool el = ELL && (word & 0x80000000u) != 0u;
        const float kv = sv_key[i];
        const int q = bin_bucket_of(A.Kk, kv);
        const unsigned mk = tile_mask(g, S.cullb[g], el);
        if (mk == 0u) continue;
        const float lbv = bin_len_bound(A.Kk, L.bmin, kv);
#pragma unroll
        for (int t = 0; t < kTilesPerQuad; ++t) {
          if ((mk >> t) & 1u) {
            const size_t pos = (size_t)L.toff[t] + (size_t)atomicAdd(&L.hist4[t][q], 1);
            A.pool_id[pos] = (int32_t)g;
            A.pool_lb[pos] = lbv;
          }
        }
      }
    }
  }
  if (lane == 0 && A.tile_ok) {
    A.tl_count[A.tile] = pooled ? tile_n[wave] : -1;
    A.tl_off[A.tile] = pooled ? L.toff[wave] : -1;
  }
  __syncthreads();
  // launch slots: longest list first (an overflowed one counts as longest), tiles outside the image last
  if (lane == 0) L.wsum[wave] = A.tile_ok ? (pooled ? tile_n[wave] : 0x7fffffff) : -1;
  __syncthreads();
  if (lane == 0) {
    const int mine = L.wsum[wave];
    int slot = 0;
    for (int w = 0; w < kTilesPerQuad; ++w) {
      const int o = L.wsum[w];
      slot += (o > mine || (o == mine && w < wave)) ? 1 : 0;
    }
    A.my_order[slot] = make_int2(A.tile_ok ? (pooled ? (A.tile | kPoolFlag) : A.tile) : -1, pooled ? mine : -1);
  }
  if (tid == 0) A.q_count[A.quad] = pooled ? -2 : -1;
  BIN_TS(1, 4);
}


#ifndef VOGE_BINB_WPE
#define VOGE_BINB_WPE 4
#endif
template <bool ELL>
__global__ void __launch_bounds__(kQT) __attribute__((amdgpu_waves_per_eu(VOGE_BINB_WPE, VOGE_BINB_WPE)))
binB_kernel(const float4 *__restrict__ cull, const float4 *__restrict__ ell, const int *__restrict__ seg_count,
            const int32_t *__restrict__ seg_id, const float4 *__restrict__ seg_rec,
            const ConeRec *__restrict__ cones /* the hierarchy of voge_cones_floats: super-tiles, quads, tiles */,
            const int N, const int H, const int W,
            const int nstx, const int nsty, const int nbin_total,
            int *__restrict__ q_count, int32_t *__restrict__ q_id, float *__restrict__ q_lb,
            int *__restrict__ tl_count, int32_t *__restrict__ tl_id, float *__restrict__ tl_lb,
            int2 *__restrict__ order /* [nbin_total][16]: (tile, list length) by launch rank */,
            unsigned long long *__restrict__ pool_top, const int pool_cap, int32_t *__restrict__ pool_id, float *__restrict__ pool_lb,
            int *__restrict__ tl_off, const int *__restrict__ seg_ext, const int32_t *__restrict__ ext_id,
            const int K, int32_t *__restrict__ out_idx, float *__restrict__ out_len, float *__restrict__ out_act,
            float *__restrict__ out_dsd, int32_t *__restrict__ out_cnt, float *__restrict__ out_weight,
            int64_t *__restrict__ out_valid, const CamView cam /* R != NULL: the cones from the camera; `cones` is not read */) {
  __shared__ BinLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  BIN_TS(1, 7);      // (kernel entry)
  const int b = blockIdx.y;
  // super-tile of this batch element, quad inside it.  (VOGE_BINB_XCD=1 puts the four quads of a super-tile, which read the same
  // 16 segments, 8 workgroup ids apart = on one XCD's L2: fewer HBM reads, but the entry is 1 us SLOWER with it and the
  // renderer's form unchanged -- HISTORY R5 -- so it is off.)
  int binl = blockIdx.x >> 2, qq = blockIdx.x & 3;
#if VOGE_BINB_XCD
  if ((int)blockIdx.x < ((nstx * nsty) >> 3) << 5) { binl = ((blockIdx.x >> 5) << 3) + (blockIdx.x & 7); qq = (blockIdx.x >> 3) & 3; }
#endif
#if VOGE_XCD_CHAIN
  if ((nstx & 3) == 0 && (nsty & 3) == 0 && (((nstx >> 2) * (nsty >> 2)) & 7) == 0) {      // (the same rule as binA's: whole regions, a multiple of 8 of them)
    const int j = blockIdx.x >> 3, region = (blockIdx.x & 7) + 8 * (j >> 6), w = j & 63, nst0x = nstx >> 2;
    qq = w & 3;
    binl = ((region / nst0x) * 4 + (w >> 4)) * nstx + (region % nst0x) * 4 + ((w >> 2) & 3);
  }
#endif
  const int stx = binl % nstx, sty = binl / nstx;
  const int bin = b * nstx * nsty + binl;
  const int quad = bin * 4 + qq;
  const int tiles_x = (W + 7) >> 3, tiles_y = (H + 7) >> 3;
  // this wave's sweep tile
  const int tx = stx * (kST / 8) + (qq & 1) * 2 + (wave & 1), ty = sty * (kST / 8) + (qq >> 1) * 2 + (wave >> 1);
  const bool tile_ok = tx < tiles_x && ty < tiles_y;
  const int tile = b * tiles_x * tiles_y + ty * tiles_x + tx;

  // ---- the wave's tile cone and the quad's cone: since round 5 both come with the rays (voge_common.h,
  // block_cones_hier256: one pass over the super-tile's rays in the ray kernel makes all 21 records).  The workgroup used to
  // derive them here from its 256 rays -- a memory round trip, four wave reductions and two barriers in front of its first
  // useful load (entry -> cones done: 3.8 us of the 11.5 us a workgroup lasts, profiles/r5_bin_times.txt).
  // (everything the prologue needs from memory is requested here, before the first use: the two cones, this super-tile's 16
  // segment counts, and the counts the launch rank is estimated from)
  ConeRec tcr, qcr;
  if (cam.R != nullptr) {      // (uniform; round 6) both cones from four corner rays each -- no load at all
    const CamK ck = cam_load(cam, b);
    cam_two_cones(ck, cam, tx * 8, ty * 8, 8, stx * kST + (qq & 1) * kQuad, sty * kST + (qq >> 1) * kQuad, kQuad, tcr, qcr);
  } else {
    tcr = cones[cone_tile_at(b, (size_t)nstx * nsty, binl, ((qq >> 1) * 2 + (wave >> 1)) * 4 + (qq & 1) * 2 + (wave & 1))];
    qcr = cones[cone_quad_at(b, (size_t)nstx * nsty, binl, qq)];
  }
  const int my_cnt = (tid < kParts) ? seg_count[(size_t)bin * kParts + tid] : 0;
  const bool ranked = nbin_total <= kRankMaxBins;
  const int4 est0 = (ranked && tid < nbin_total) ? *reinterpret_cast<const int4 *>(seg_count + (size_t)tid * kParts) : make_int4(0, 0, 0, 0);
  const int4 est_mine = ranked ? *reinterpret_cast<const int4 *>(seg_count + (size_t)bin * kParts) : make_int4(0, 0, 0, 0);
  const Cone tcone = load_cone(tcr), qcone = load_cone(qcr);
  {
    if (tid == 0) { L.count = 0; L.nflag = 0; L.spill = 0; }
    // A segment that overflowed its kSegCap entries (more than 512 of a slice's Gaussians in one super-tile: a small
    // object far away, a zoomed-out view) does not send the quad to the stream-everything fallback any more: the
    // slice's Gaussians -- N / kParts of them -- are tested against the quad's cone right here, from the records
    // binA / prep left per Gaussian.  Only a quad list beyond kQCap still falls back.
    // (The 16 counts were requested at the kernel's entry, one per lane, together with the cones and the ranking's counts.
    // Their prefix sums are a 16-lane scan.)
    if (wave == 0) {
      int c = (lane < kParts) ? my_cnt : 0;
      const unsigned long long ofm = __ballot(lane < kParts && c < 0);      // (its Gaussians come from the records, last loop of bin_stream_sources)
      if (c < 0) c = 0;
      int e = 0;
      if (c > kSegCap) { e = c - kSegCap; c = kSegCap; }
      int xc = c, xe = e;
#pragma unroll
      for (int o = 1; o < kParts; o <<= 1) {
        const int yc = __shfl_up(xc, o, 64), ye = __shfl_up(xe, o, 64);
        if (lane >= o) { xc += yc; xe += ye; }
      }
      if (lane < kParts) { L.segn[lane] = xc - c; L.extn[lane] = xe - e; }
      if (lane == kParts - 1) { L.segn[kParts] = xc; L.extn[kParts] = xe; }
      if (lane == 0) L.ovf = (unsigned)ofm;
    }
    __syncthreads();
    auto load_ext_tables = [&]() {      // the chunk tables of the segments with an extension: thread <-> (slice, chunk)
      static_assert(kParts * kExtChunks == kQT, "one thread per (slice, chunk)");
      const int p = tid / kExtChunks, q = tid % kExtChunks;
      const int ne = L.extn[p + 1] - L.extn[p];
      L.exto[p][q] = (q * kExtChunk < ne) ? seg_ext[((size_t)bin * kParts + p) * kExtChunks + q] : 0;
    };
    if (L.extn[kParts] > 0) load_ext_tables();      // (uniform, rare; the tables share their LDS with the sort's histogram)
  }
  BIN_TS(1, 0);

  // ---- launch rank of the super-tile among all of them: by descending candidate count, estimated from the first
  // four segments of every super-tile (a quarter of the Gaussians, dealt round-robin: proportional to the total).
  // Every workgroup derives the rank from the same numbers, so the ranks are a permutation: no exchange.  The sweep's
  // workgroup number lin then finds its tile (and the length of its list) in order[lin]: one load. ----
  BIN_TS(1, 6);      // (cones done)
  int rank = bin;
  {
    auto est_of = [](const int4 v) {      // (an overflowed segment counts as full: such super-tiles go first)
      return (v.x < 0 ? kSegCap : v.x) + (v.y < 0 ? kSegCap : v.y) + (v.z < 0 ? kSegCap : v.z) + (v.w < 0 ? kSegCap : v.w);
    };
    auto estimate = [&](const int q) { return est_of(*reinterpret_cast<const int4 *>(seg_count + (size_t)q * kParts)); };
    if (ranked) {
      const int mine = est_of(est_mine);
      int ahead = 0;
      for (int q = tid; q < nbin_total; q += kQT) {
        const int e = (q == tid) ? est_of(est0) : estimate(q);      // (the first one was requested at the kernel's entry)
        ahead += (e > mine || (e == mine && q < bin)) ? 1 : 0;
      }
      ahead = (int)wave_sum_dpp((float)ahead);      // (< 2^24: exact in fp32)
      if (lane == 0) L.wsum[wave] = ahead;
      __syncthreads();
      rank = L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
      __syncthreads();
    }
  }
  int2 *my_order = order + (size_t)rank * kTilesPerBin + qq * kTilesPerQuad;     // this quad's four launch slots
  const int n_src = L.segn[kParts];
  const unsigned ovf = L.ovf;
  // ---- the super-tile's segments against the quad's cone; survivors' keys are compacted in LDS.  Anisotropic
  // candidates are tested with their ellipsoid as well (binA tested bounding spheres only). ----
  const float4 *cullb = cull + (size_t)b * N;
  const float4 *ellb = ell + (size_t)b * N * 2;
  const int32_t *segs = seg_id + (size_t)bin * kParts * kSegCap;
  const float4 *segr = seg_rec + (size_t)bin * kParts * kSegCap;
  float rmax = 0.0f;   // largest finite reach among this thread's survivors WITHOUT an ellipsoid record
  float klo = INFINITY, khi = -INFINITY;   // extrema of the finite order keys of this thread's entries
  bool any_el = false;
  BinStream S;
  S.cullb = cullb; S.ellb = ellb; S.segr = segr; S.segs = segs; S.n_src = n_src; S.N = N; S.ovf = ovf; S.qcone = qcone;
  S.ext_id = ext_id; S.n_ext = L.extn[kParts]; S.exto = &L.exto[0][0];
  // pass one: the quad's survivors -- keys compacted in LDS (the first kQCap of them), extrema, the largest sphere reach
  bin_stream_sources<ELL>(S, L.segn, tid, [&](const int (&g)[kGU], const float4 (&cj)[kGU], const bool (&kp)[kGU],
                                          const bool (&el)[kGU], const float (&gkey)[kGU]) {
    int slot[kGU];
    bin_batch_slots(&L.count, lane, kp, slot);
#pragma unroll
    for (int j = 0; j < kGU; ++j) {
      if (kp[j]) {
        if (!el[j] && cj[j].w < 3e38f) rmax = fmaxf(rmax, cj[j].w);
        // bit 31 of the id word: the entry has an ellipsoid record (its len bound is its own)
        const float kv = el[j] ? gkey[j] : depth_key(cj[j], qcone);
        if (kv > -INFINITY) { klo = fminf(klo, kv); khi = fmaxf(khi, kv); }
        if (slot[j] < kQCap) L.keys[slot[j]] = ((uint64_t)f2ord(kv) << 32) | (uint32_t)g[j] | (el[j] ? 0x80000000u : 0u);
        any_el = any_el || el[j];
      }
    }
  });
  BIN_TS(1, 1);
  if (__any(any_el) && lane == 0) L.nflag = 1;
  // ---- order the survivors front to back: counting sort on the depth key.  Exact order is not needed for
  // correctness (the sweep's top-K insertion is order independent) but it turns nearly every insertion into an
  // append.  Bucket 0 collects the -inf keys (unbounded reach).
  float hi = wave_max(khi), lo = wave_min(klo), rm = wave_max(rmax);
  if (lane == 0) { L.red[wave * 8 + 0] = hi; L.red[wave * 8 + 1] = lo; L.red[wave * 8 + 2] = rm; }
  // (the histogram shares its LDS with the extension chunk tables: every wave must be through with its source pass
  // before the first one clears it -- without this barrier a fast wave zeroed table entries a slow one still read, and a
  // candidate went missing once in a few hundred dense scenes: tools/soak.py, seed 502)
  if (S.n_ext > 0) __syncthreads();      // (uniform, rare)
  for (int i = tid; i < kBuckets; i += kQT) { L.hist[i] = 0; L.bmin[i] = f2ord(INFINITY); }
  __syncthreads();
  const int total = L.count;
  hi = fmaxf(fmaxf(L.red[0], L.red[8]), fmaxf(L.red[16], L.red[24]));
  lo = fminf(fminf(L.red[1], L.red[9]), fminf(L.red[17], L.red[25]));
  rm = fmaxf(fmaxf(L.red[2], L.red[10]), fmaxf(L.red[18], L.red[26]));
  const float span = fmaxf(hi - lo, 1e-20f);
  const float scale = (float)(kBuckets - 2) / span;
  const bool flagged = ELL && L.nflag != 0;     // any entry with an ellipsoid record (workgroup-uniform)
  BinKeys Kk;
  Kk.lo = lo; Kk.scale = scale; Kk.span = span; Kk.slack = 1.13f * rm * (1.0f + 1e-5f); Kk.flagged = flagged;
  auto bucket_of = [&](const uint64_t k) { return bin_bucket_of(Kk, ord2f((uint32_t)(k >> 32))); };
#ifdef VOGE_NO_LONG      // (A/B builds: the pre-pool behaviour -- such quads stream every Gaussian in the sweep)
  if (total > kQCap) {
    if (tid == 0) q_count[quad] = -1;
    if (lane == 0) {
      if (tile_ok) tl_count[tile] = -1;
      my_order[wave] = make_int2(tile_ok ? tile : -1, -1);
    }
    return;
  }
  {
#else
  if (__builtin_expect(total > kQCap, 0)) {
    BinLong A;
    A.S = S; A.Kk = Kk; A.tcone = tcone; A.tile_ok = tile_ok; A.tile = tile; A.quad = quad; A.total = total;
    A.pool_top = pool_top; A.tl_off = tl_off; A.tl_count = tl_count; A.q_count = q_count; A.pool_cap = pool_cap;
    A.pool_id = pool_id; A.pool_lb = pool_lb; A.my_order = my_order; A.seg_ext = seg_ext; A.bin = bin;
    binB_long_path<ELL>(A, L);
  } else {
#endif
  for (int i = tid; i < total; i += kQT) {
    const uint64_t kk = L.keys[i];
    const int q = bucket_of(kk);
    atomicAdd(&L.hist[q], 1);
    if (ELL && ((uint32_t)kk & 0x80000000u)) atomicMin(&L.bmin[q], f2ord(bin_own_bound(S, (size_t)((uint32_t)kk & 0x7fffffffu))));
  }
  __syncthreads();
  // exclusive scan of the kBuckets counters: two consecutive buckets per thread, wave scan, wave offsets
  {
    static_assert(kBuckets == 2 * kQT, "two buckets per thread");
    const int2 v = *reinterpret_cast<const int2 *>(&L.hist[2 * tid]);
    const int s4 = v.x + v.y;
    int x = s4;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) L.wsum[wave] = x;
    __syncthreads();
    int off = x - s4;
    for (int w = 0; w < wave; ++w) off += L.wsum[w];
    *reinterpret_cast<int2 *>(&L.hist[2 * tid]) = make_int2(off, off + v.x);
  }
  __syncthreads();
  BIN_TS(1, 2);
  for (int i = tid; i < total; i += kQT) {
    const uint64_t kk = L.keys[i];
    L.sorted[atomicAdd(&L.hist[bucket_of(kk)], 1)] = (uint32_t)kk;
  }
  // Entries are ordered by BUCKET only (1022 buckets over the quad's depth range, i.e. a few thousandths of a scene
  // unit each -- far finer than the reach that separates kappa from the actual len), and every entry carries its
  // bucket's lower edge as the len bound: monotone along the list, which is all the sweep's early exit needs.
  // The order inside a bucket is whatever the LDS atomics produced; the sweep's top-K is order independent.
  // Suffix minimum over the buckets of the flagged entries' own bounds (thread <-> two buckets).
  if (flagged) {
    __syncthreads();
    bin_bmin_suffix(L.bmin, L.wsum, tid);
  }
  __syncthreads();
  // the cull records of the list's first kQRec entries, in list order, over the (now dead) unordered keys
  for (int i0 = 0; i0 < min(total, kQRec); i0 += kQT * kGU) {
    float4 c[kGU];
#pragma unroll
    for (int j = 0; j < kGU; ++j) {
      const int i = i0 + j * kQT + tid;
      c[j] = (i < min(total, kQRec)) ? cullb[L.sorted[i] & 0x7fffffffu] : make_float4(0.f, 0.f, 0.f, -1.f);
    }
#pragma unroll
    for (int j = 0; j < kGU; ++j) {
      const int i = i0 + j * kQT + tid;
      if (i < min(total, kQRec)) L.rec[i] = c[j];
    }
  }
  __syncthreads();
  BIN_TS(1, 3);
  // the len bound of list entry `k` (its bucket's lower edge, see above)
  auto len_bound = [&](const uint32_t word, const float4 cr) {
    // the entry's depth key, as the gather computed it (same record, same cone: the same bits)
    const float v = (ELL && (word & 0x80000000u)) ? fmaf(cr.z, qcone.az, fmaf(cr.y, qcone.ay, cr.x * qcone.ax)) + 0.0f
                                         : depth_key(cr, qcone);
    return bin_len_bound(Kk, L.bmin, v);
  };

  // ---- the four sweep tiles: wave w filters the ordered list (in LDS, records included) with the bounding cone of
  // its own 8x8 pixels, keeping the order; four 64-entry batches per trip (their LDS reads and tests are independent)
  int kept = 0;
  if (tile_ok) {
    int32_t *oid = tl_id + (size_t)tile * kTileCap;
    float *olb = tl_lb + (size_t)tile * kTileCap;
    BIN_WTS(0);
#ifndef VOGE_BINB_FU
#define VOGE_BINB_FU 4
#endif
    constexpr int kFU = VOGE_BINB_FU;
    for (int base = 0; base < total; base += 64 * kFU) {
      uint32_t k[kFU];
      float4 cr[kFU];
#pragma unroll
      for (int q = 0; q < kFU; ++q) {
        const int i = base + q * 64 + lane;
        k[q] = (i < total) ? L.sorted[i] : 0u;
        cr[q] = make_float4(0.f, 0.f, 0.f, -1.f);
        // (an LDS read at a clamped index, and -- uniform, rare: a quad list beyond kQRec entries -- a gather for the rest.  As
        //  one `in LDS ? L.rec[i] : cullb[id]` expression this was a select of two pointers and a FLAT load in the filter's loop.)
        if (i < total) cr[q] = L.rec[min(i, kQRec - 1)];
        if (__any(i >= kQRec && i < total)) {
          if (i >= kQRec && i < total) cr[q] = cullb[k[q] & 0x7fffffffu];
        }
      }
      bool kp[kFU];
#pragma unroll
      for (int q = 0; q < kFU; ++q) kp[q] = cone_keep(cr[q], tcone);     // (padding: reach -1, never kept)
#pragma unroll
      for (int q = 0; q < kFU; ++q) {
        if (base + q * 64 >= total) break;      // uniform
        const uint32_t word = k[q];
        const bool el = ELL && kp[q] && (word & 0x80000000u);      // sphere survivors with an ellipsoid record: that test too
        if (__any(el)) {
          if (el) {
            const size_t g = word & 0x7fffffffu;
            kp[q] = cone_keep_ell(cr[q], ellb[2 * g], ellb[2 * g + 1], tcone);
          }
        }
        const unsigned long long m = __ballot(kp[q]);
        if (kp[q]) {
          const int pos = kept + __popcll(m & ((1ull << lane) - 1ull));
          if (pos < kTileCap) { oid[pos] = (int32_t)(word & 0x7fffffffu); olb[pos] = len_bound(word, cr[q]); }
        }
        kept += __popcll(m);
      }
    }
    BIN_WTS(1);
    if (lane == 0) {
      if (kept > kTileCap) { tl_count[tile] = -1; L.spill = 1; }
      else tl_count[tile] = kept;
    }
  }
  // launch slots of the four tiles inside the quad: longest list first (an overflowed one counts as longest), tiles
  // outside the image last
  if (lane == 0) L.wsum[wave] = tile_ok ? min(kept, kTileCap + 1) : -1;
  __syncthreads();
  BIN_TS(1, 4);
  if (lane == 0) {
    const int mine = L.wsum[wave];
    int slot = 0;
    for (int w = 0; w < kTilesPerQuad; ++w) {
      const int o = L.wsum[w];
      slot += (o > mine || (o == mine && w < wave)) ? 1 : 0;
    }
    my_order[slot] = make_int2(tile_ok ? tile : -1, (mine > kTileCap) ? -1 : mine);
  }
  // a tile list overflowed: its sweep falls back to the quad's ordered list, which now has to exist in memory
  if (L.spill) {
    int32_t *oid = q_id + (size_t)quad * kQCap;
    float *olb = q_lb + (size_t)quad * kQCap;
    for (int i = tid; i < total; i += kQT) {
      const uint32_t word = L.sorted[i];
      oid[i] = (int32_t)(word & 0x7fffffffu);
      float4 crec = L.rec[min(i, kQRec - 1)];
      if (i >= kQRec) crec = cullb[word & 0x7fffffffu];
      olb[i] = len_bound(word, crec);
    }
  }
  if (tid == 0) q_count[quad] = L.spill ? total : -2;     // (-2: never read -- every tile of this quad has its own list)
  }      // (short path)
  // Tiles nothing can hit get their all-sentinel outputs (ray_trace_voge.cu:244-247) here, written by the whole
  // workgroup: 40 KB per tile at K = 40, 62 MB per frame at cfg3 -- HBM-write-bound wherever it happens.  The
  // quads with empty tiles are the ones with short lists, i.e. the workgroups that would otherwise finish long
  // before the kernel does; the sweep (a few waves per CU, latency-bound) never sees these tiles.
  // (out_idx == NULL: the sweep that follows writes them itself -- sweep_iso_kernel without act / dsd, see there)
  for (int w = 0; w < kTilesPerQuad && out_idx != nullptr; ++w) {
    if (L.wsum[w] != 0) continue;      // uniform
    const int ftx = stx * (kST / 8) + (qq & 1) * 2 + (w & 1), fty = sty * (kST / 8) + (qq >> 1) * 2 + (w >> 1);
    const int tw = min(8, W - ftx * 8), th = min(8, H - fty * 8);
    const int row_items = tw * K;
    if ((K & 3) == 0) {
      const int ipr = row_items >> 2;
      for (int it = tid; it < th * ipr; it += kQT) {
        const int rr = it / ipr, j4 = it - rr * ipr;
        const size_t o = (((size_t)b * H + fty * 8 + rr) * W + (size_t)ftx * 8) * K + (size_t)j4 * 4;
        st16i<(VOGE_NT_STORES & 1) != 0>(out_idx + o, -1, -1, -1, -1);
        st16f<(VOGE_NT_STORES & 1) != 0>(out_len + o, VOGE_SENT_LEN, VOGE_SENT_LEN, VOGE_SENT_LEN, VOGE_SENT_LEN);
        if (out_weight != nullptr) {
          // fused trace + composite: the fragments are (weight, idx, len); act / dsd only serve the backward, which
          // never reads a pixel without hits -- a quarter of the empty tiles' bytes stays unwritten
          *reinterpret_cast<float4 *>(out_weight + o) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (out_act != nullptr) {      // (NULL: the scalar-sigma fragment entry points keep no act / dsd)
          st16f<(VOGE_NT_STORES & 1) != 0>(out_act + o, VOGE_SENT_ACT, VOGE_SENT_ACT, VOGE_SENT_ACT, VOGE_SENT_ACT);
          st16f<(VOGE_NT_STORES & 1) != 0>(out_dsd + o, 0.f, 0.f, 0.f, 0.f);
        }
      }
    } else {
      for (int it = tid; it < th * row_items; it += kQT) {
        const int rr = it / row_items, j = it - rr * row_items;
        const size_t o = (((size_t)b * H + fty * 8 + rr) * W + (size_t)ftx * 8) * K + j;
        out_idx[o] = -1; out_len[o] = VOGE_SENT_LEN;
        if (out_act != nullptr) { out_act[o] = VOGE_SENT_ACT; out_dsd[o] = 0.0f; }
        if (out_weight != nullptr) out_weight[o] = 0.0f;
      }
    }
    if (tid < th * 8) {
      const int rr = tid >> 3, x = tid & 7;
      const size_t pix = ((size_t)b * H + fty * 8 + rr) * W + (size_t)ftx * 8 + x;
      if (x < tw && out_cnt != nullptr) out_cnt[pix] = 0;
      if (x < tw && out_valid != nullptr) out_valid[pix] = 0;
    }
  }
  BIN_TS(1, 5);
}

}  // namespace voge
