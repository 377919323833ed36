// Fused backward of the fragment pipeline for gfx950: shade (merge + silhouette + blend) -> composite -> fine trace,
// ONE pass over the fragments.
//
// Reference behaviour being reproduced (all by autograd there):
//   VoGE/Renderer.py:157-171 (get_silhouette, to_colored_background) and VoGE/Aggregation.py:111-141 (merge_final),
//   VoGE/Aggregation.py:30-107 (aggregation), VoGE/csrc/ray_trace_voge/ray_trace_voge.cu:283-332 (trace backward),
// for the renderer's common case: isotropic Gaussians (one scalar each), <= 4 colour channels, fixed cameras.
//
// The three stand-alone backward kernels (voge_shade_bwd, voge_composite_bwd, voge_trace_bwd_iso) exchange g_weight
// and g_len / g_act / g_dsd through memory (336 MB at cfg3), each reads the index lists again, and two of them build a
// per-Gaussian accumulation table of their own.  Here a lane owns two consecutive slots of a pixel (the composite
// backward's layout, composite_core.h); for them it
//   1. gathers the slots' colours:       g_w[k] = <g_rgb, colour[idx_k]> + g_sum_w          (shade backward)
//   2. runs the composite's closed-form backward on u = g_w * w:  g_act, g_len, g_dsd        (registers)
//   3. gathers the slots' (mu, a) and forms the trace's per-slot terms for g_mu (3) and g_a  (trace backward)
//   4. adds (g_mu, g_a | w g_rgb) -- EIGHT sums -- to ONE wave-private LDS table entry per Gaussian,
// and the table is flushed with lane-coalesced atomics into acc[P][8].  A finishing pass turns acc into the
// gradients of what the caller passed in (verts / sigmas through the view's chain rule, colours).
// Everything a stage hands to the next stays in registers; by linearity in g_weight, gradients reaching the weights
// from other consumers (e.g. a silhouette loss) still take the stand-alone kernels and simply add up.
#include "composite_core.h"

namespace voge {

#ifndef VOGE_FB_NE
#define VOGE_FB_NE 128
#endif
constexpr int kFbNE = VOGE_FB_NE;  // table entries per wave
// A wave's group of pixels: GW x GH (<= 64 pixels; lane g of the wave holds pixel (g % GW, g / GW)'s hit count).  The
// lanes are handed out by those counts (composite_core.h, "Lane packing by hit count"): ceil(count / 2) lanes per
// pixel, as many consecutive pixels per round as fit the wave.
#ifndef VOGE_FB_GW
#define VOGE_FB_GW 4
#endif
#ifndef VOGE_FB_GH
#define VOGE_FB_GH 3
#endif
constexpr int kFbGW = VOGE_FB_GW, kFbGH = VOGE_FB_GH, kFbG = kFbGW * kFbGH;
static_assert(kFbG <= 64 && (kFbGW & (kFbGW - 1)) == 0, "a group's pixels are the lanes of one wave; GW a power of two");
// -DVOGE_FB_TIMES: per-section cycle counters of the fused backward (tools/fb_sections.py).  Sections: 0 set-up + lane
// packing, 1 streams + gathers + shade terms, 2 LDS rows + composite backward, 3 trace terms, 4 table, 5 flush.
#ifdef VOGE_FB_TIMES
constexpr int kFbTimesWaves = 1 << 16;
__device__ unsigned long long g_fb_times[kFbTimesWaves][8];      // one row per workgroup (plain stores: no same-address atomics)
__device__ unsigned long long g_fb_tab[kFbTimesWaves][4];       // accumulations (wave-wide calls), their election rounds, lanes taking part
__device__ unsigned long long g_fb_wall[kFbTimesWaves][2];       // its first and last stamp (s_memrealtime, 100 MHz): tools/fb_wall.py
#define FB_T0() const unsigned long long fb_w0_ = __builtin_amdgcn_s_memrealtime(); unsigned long long fb_t_ = __builtin_readcyclecounter(), fb_acc_[6] = {0, 0, 0, 0, 0, 0}
#define FB_TICK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); fb_acc_[i] += n_ - fb_t_; fb_t_ = n_; } while (0)
#define FB_TOUT() do { if (threadIdx.x == 0 && blockIdx.x < kFbTimesWaves) { for (int i_ = 0; i_ < 6; ++i_) g_fb_times[blockIdx.x][i_] = fb_acc_[i_]; g_fb_times[blockIdx.x][6] = 1ull; g_fb_wall[blockIdx.x][0] = fb_w0_; g_fb_wall[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define FB_T0() do {} while (0)
#define FB_TICK(i) do {} while (0)
#define FB_TOUT() do {} while (0)
#endif

#ifndef VOGE_BWD_RCOL
#define VOGE_BWD_RCOL 1
#endif
template <int NV4, int NS>
struct FragBwdLds {
  // key = Gaussian index; values: (g_mu, g_a), (w g_rgb, -) for A = a I [NV4 = 2];
  //                               (g_mu, g_A[0]), g_A[1..4], g_A[5..8], (w g_rgb, -) for a full 3x3 A [NV4 = 4];
  //                               without the colour term (SRC = 1): NV4 = 1 | 3
  static constexpr int kRows = NS * 64 + 4 * kFbG;      // a round's padded rows: 64 lanes' slots + two sentinel pairs per pixel
  WaveTable<kFbNE, NV4> tab;
  float len[kRows], sp[kRows], E[kRows], u[kRows];
#if VOGE_BWD_RCOL
  float r[kRows];                 // row sums of the composite backward, accumulated by the column walks (compn_bwd_wave<NS, true>)
#endif
  unsigned rmax[kFbG];            // per pixel of the round: its window radius (compn_bwd_wave)
};

#ifndef VOGE_FB_ABL
#define VOGE_FB_ABL 0
#endif
#ifndef VOGE_FB_LDS_RMAX
#define VOGE_FB_LDS_RMAX 1
#endif
#ifndef VOGE_FB_PAIR_TABLE
#define VOGE_FB_PAIR_TABLE 1      // a lane's two table accumulations share one election loop (wt_add2)
#endif
#ifndef VOGE_FB_TABLE_BY_PIXEL
#define VOGE_FB_TABLE_BY_PIXEL 1  // the table is taken pixel by pixel (no election: wt_add2_by_group) instead of by wt_add2's elections
#endif
#ifndef VOGE_FB_WPE
#define VOGE_FB_WPE 4      // capping the registers for 5 / 6 waves per SIMD spills and is slower
#endif
// SRC: where the gradient of the weights comes from.
//   0: the shade stage is part of the pass (colours, rgb, wsum, bg, g_img given): g_w = <g_rgb, colour> + g_sum_w, and the
//      colours' own gradient w g_rgb rides in the table (C: colour channels, 1..4);
//   1: g_weight [pix][K] is given by whoever consumed the weights (merge_final, get_silhouette, any torch expression),
//      element (p, k) at g_weight[p * gw_stride_pix + k * gw_stride_k] (K, 1 contiguous; (1, 0) for a per-pixel value
//      broadcast over the slots, which is what a silhouette loss alone produces), plus the gradient of vert_hit_length
//      (g_hitlen, contiguous or NULL) added to the trace's g_len; C = 0, no colour term.
// NS: slots per lane (2; 4 for lists of more than 128 slots, so that a pixel's lanes still fit one wave).
// OffT: uint32_t when every BYTE offset into the [pix][K] arrays fits 32 bits (the loads
// then take scalar base + 32-bit lane offset and the address arithmetic leaves the vector unit), else size_t.
// ISO: every Gaussian is A = a I (rec = [P] (mu, a)); otherwise rec = [P][3] packed (mu, A) as in trace_bwd.hip.
// NOAD: the forward kept no act / dsd (voge_fragments_fwd_iso* with act = dsd = NULL, voge_trace_lean_fwd); they are re-derived
// from the records, the ray and len with the forward's own operations (pair_eval_iso_at / pair_eval): 8 bytes per slot less to read.
// K need not be a multiple of NS: `vec` (uniform) says whether a lane's group is aligned and inside its pixel's row
// (wide loads) or is read slot by slot.
// DIAG (round 6; !ISO, NOAD): rec = [P][2] compact per-axis records (mu, a0 | a1, a2, 0, 0) of voge_frame_trace_fwd_gen (kind 1):
// em / sm from pair_eval_diag, the trace terms of the diagonal alone (the general terms' bits on such a form: their other
// coefficients multiply zeros), table entries of 12 | 8 floats instead of 16 | 12 -- (g_mu, g_A00), (g_A11, g_A22, w g_rgb ...).
template <int SRC, int C, int NS, typename OffT, bool ISO, bool NOAD, bool DIAG = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NS == 4 ? 2 : (ISO ? VOGE_FB_WPE : 3))))
fragment_bwd_kernel(const float4 *__restrict__ rec, const float *__restrict__ rays,
                        const float *__restrict__ colors, const int32_t *__restrict__ idx, const int32_t *__restrict__ cnt,
                        const float *__restrict__ weight, const float *__restrict__ act, const float *__restrict__ len,
                        const float *__restrict__ dsd, const float *__restrict__ rgb, const float *__restrict__ wsum,
                        const float *__restrict__ bg, const float thr, const float *__restrict__ g_img, const long gs_pix,
                        const long gs_c, const float *__restrict__ g_hitlen, const float occ,
                        const int P, const long nrows, const int W, const int K, const long Nattr,
                        float *__restrict__ acc /* [P][NACC], zeroed */) {
  static_assert(NS == 2 || NS == 4, "a lane owns one or two aligned pairs of slots");
  static_assert(!DIAG || (!ISO && NOAD && C <= 4), "the per-axis form keeps no act / dsd");
  constexpr int NV4 = SRC == 0 ? (ISO ? 2 : (DIAG ? 3 : 4)) : (ISO ? 1 : (DIAG ? 2 : 3)), NACC = 4 * NV4;
  constexpr int NRC = ISO ? 1 : (DIAG ? 2 : 3);      // float4s per record
  // sum c (0 .. NACC - 1) of Gaussian `key`.  (Round 6 tried adding the table straight into the caller's gradient arrays --
  // g_verts [N][3], g_sigmas [N], g_colors [N][C], zeroed by the forward -- instead of acc [P][8] + a finishing pass: one launch
  // less, and the kernel went from 107 to 137 us.  An entry's eight sums are ONE 32-byte atomic request here and three requests
  // into three arrays there; the kernel's tail is bound by those requests.  acc stays; the forward zeroes it on its way.)
  auto sum_add = [&](const int key, const int c, const float v) { unsafeAtomicAdd(acc + NACC * (size_t)key + c, v); };
  __shared__ __attribute__((aligned(16))) FragBwdLds<NV4, NS> L;
  float *const Llen = L.len, *const Lsp = L.sp, *const LE = L.E, *const Lu = L.u;
#if VOGE_BWD_RCOL
  float *const LR = L.r;
#endif
  const int lane = threadIdx.x;
  FB_T0();
  const int blocks_x = (W + kFbGW - 1) / kFbGW;
  const long blk = blockIdx.x;
  const int x0 = (int)(blk % blocks_x) * kFbGW;
  const long y0 = (blk / blocks_x) * kFbGH;
  const bool vec = (K % NS) == 0;
  // the group's hit counts -> lanes per pixel
  const int gx = x0 + (lane & (kFbGW - 1));
  const long gy = y0 + lane / kFbGW;
  const bool pv = lane < kFbG && gx < W && gy < nrows;
  const int lead_g = pv ? min(K, max(0, cnt[gy * W + gx])) : 0;
  const int need = (lead_g + NS - 1) / NS;
  const int incl = wave_incl_scan(need, lane);
  if (__builtin_amdgcn_readlane(incl, 63) == 0) return;     // nothing was hit in these pixels
  wt_clear(L.tab, lane);
  int pc = 0, off = 0;
  while (pc < kFbG) {
    PackLane pk;
    int npm;
    const int pe = pack_round(need, incl, kFbG, lane, pc, off, pk, npm);
    off = __builtin_amdgcn_readlane(incl, pe - 1);
    pc = pe;
    if (npm == 0) continue;       // (uniform) a run of empty pixels
    FB_TICK(0);
    const bool on = pk.p >= 0;
    const int q = on ? lane - pk.s0 : 0, LP = on ? pk.np : 1;
    const int lead = __shfl(lead_g, on ? pk.p : 0, 64);
    const int k0 = NS * q;
    const OffT pix = on ? (OffT)((y0 + pk.p / kFbGW) * W + x0 + (pk.p & (kFbGW - 1))) : (OffT)0;
    const OffT fb = (pix * (OffT)K + (OffT)k0) * (OffT)4;      // byte offset of the lane's group in the [pix][K] arrays
    const int r0 = NS * pk.s0 + 4 * pk.ord, RS = NS * LP + 4;
    const int d0 = on ? r0 + 2 + k0 : 2;
    // ---- the lane's slots ----
    // Round 5: the loads of a round go out in TWO waves of requests -- everything addressed by the pixel (ray, the slots'
    // idx / weight / len [/ act / dsd / g_weight], the pixel's forward sums and upstream gradient), then everything addressed
    // by the slots' Gaussian ids (records, colours) -- each unconditional at a clamped address, masks applied to the VALUES
    // afterwards.  As `if (on) load ...; if (live) load ...` the compiler had to join every exec-mask region before the next:
    // ray -> wait -> streams -> wait -> record 0 -> wait -> record 1 -> wait -> colours -> pixel data, six dependent round trips
    // per round (SQ_WAIT_ANY: 46 % of the kernel's wave cycles, profiles/r5_pmc_sq_counters_head.txt).
    int id[NS];
    float wv[NS], lm[NS], sm[NS], em[NS], gwv[NS];
    bool live[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      id[a] = -1; wv[a] = 0.f; lm[a] = VOGE_SENT_LEN; sm[a] = 1e-5f; em[a] = 0.f; gwv[a] = 0.f;
      live[a] = on && (k0 + a < lead);
    }
    // (a lane that is off reads pixel 0, slot 0: pix = 0 and q = 0 there)
    const float3 dv = at_bytes<float3>(rays, pix * (OffT)12);
    const float dx = on ? dv.x : 0.f, dy = on ? dv.y : 0.f, dz = on ? dv.z : 0.f;
    float4 rc[NS][NRC];
    float av[NS], dv2[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) { av[a] = 0.f; dv2[a] = 0.f; }
    {
      int idr[NS];
      float wr[NS], lr[NS], ar[NS], dr[NS], gr_[NS];
#pragma unroll
      for (int a = 0; a < NS; ++a) { idr[a] = -1; wr[a] = 0.f; lr[a] = VOGE_SENT_LEN; ar[a] = 0.f; dr[a] = 0.f; gr_[a] = 0.f; }
      const bool gw_wide = SRC == 1 && g_img != nullptr && vec && gs_c == 1 && gs_pix == (long)K;
      if (vec) {                // (uniform) wide accesses: the group is aligned and inside the pixel's row
#pragma unroll
        for (int h2 = 0; h2 < NS / 2; ++h2) {
          const OffT fo = fb + (OffT)(8 * h2);
          const int2 i2 = at_bytes<int2>(idx, fo);
          const v2f w2 = at_bytes<v2f>(weight, fo), l2 = at_bytes<v2f>(len, fo);
          idr[2 * h2] = i2.x; idr[2 * h2 + 1] = i2.y; wr[2 * h2] = w2.x; wr[2 * h2 + 1] = w2.y; lr[2 * h2] = l2.x; lr[2 * h2 + 1] = l2.y;
          if (!NOAD) {
            const v2f a2 = at_bytes<v2f>(act, fo), d2 = at_bytes<v2f>(dsd, fo);
            ar[2 * h2] = a2.x; ar[2 * h2 + 1] = a2.y; dr[2 * h2] = d2.x; dr[2 * h2 + 1] = d2.y;
          }
          if (gw_wide) {        // (uniform)
            const v2f g2 = at_bytes<v2f>(g_img, fo);
            gr_[2 * h2] = g2.x; gr_[2 * h2 + 1] = g2.y;
          }
        }
      } else {                  // slot by slot; a slot behind the pixel's row re-reads the row's last one (never live)
#pragma unroll
        for (int a = 0; a < NS; ++a) {
          const OffT fo = (pix * (OffT)K + (OffT)min(k0 + a, K - 1)) * (OffT)4;
          idr[a] = at_bytes<int>(idx, fo); wr[a] = at_bytes<float>(weight, fo); lr[a] = at_bytes<float>(len, fo);
          if (!NOAD) { ar[a] = at_bytes<float>(act, fo); dr[a] = at_bytes<float>(dsd, fo); }
        }
      }
      if (SRC == 1 && g_img != nullptr && !gw_wide) {      // (uniform) the consumers' gradient of the weights, any strides
#pragma unroll
        for (int a = 0; a < NS; ++a) gr_[a] = g_img[(long)pix * gs_pix + (long)min(k0 + a, K - 1) * gs_c];
      }
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        if (live[a]) { id[a] = idr[a]; wv[a] = wr[a]; lm[a] = lr[a]; av[a] = ar[a]; dv2[a] = dr[a]; gwv[a] = gr_[a]; }
      }
    }
    // the pixel's forward sums and upstream gradient (the shade stage), requested with the slots
    float gr[4] = {0.f, 0.f, 0.f, 0.f}, g_sum_w = 0.0f;
    float px_ws = 0.f, px_rgb[4] = {0.f, 0.f, 0.f, 0.f}, px_g[4] = {0.f, 0.f, 0.f, 0.f};
    if (SRC == 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) px_g[c] = g_img[(long)pix * gs_pix + c * gs_c];
      if (wsum != nullptr) px_ws = at_bytes<float>(wsum, pix * (OffT)4);
      if (bg != nullptr) {
#pragma unroll
        for (int c = 0; c < C; ++c) px_rgb[c] = at_bytes<float>(rgb, pix * (OffT)(4 * C) + (OffT)(4 * c));
      } else if (rgb != nullptr) {      // (merge form, round 6: `rgb` = the forward's weight sums [pix], `wsum` = the SILHOUETTE's gradient)
        px_rgb[0] = at_bytes<float>(rgb, pix * (OffT)4);
      }
    }
    // ---- second wave of requests: what the slots' Gaussians carry (records, colours), at clamped ids ----
    bool okv[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) okv[a] = live[a] && id[a] >= 0 && id[a] < P;
    if (NOAD) {      // (act / dsd re-derived from the records; with act / dsd given the records are only needed by the trace terms)
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const uint32_t ro = (uint32_t)(okv[a] ? id[a] : 0) * (16u * NRC);      // (P < 2^26: host)
#pragma unroll
        for (int r = 0; r < NRC; ++r) rc[a][r] = !(VOGE_FB_ABL & 8) ? at_bytes<float4>(rec, ro + 16u * r) : make_float4(0.f, 0.f, 0.f, 0.f);      // (P >= 1, Nattr >= 1 here: host)
      }
    }
    float col[NS][4];
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      col[a][0] = col[a][1] = col[a][2] = col[a][3] = 0.0f;
      if (SRC == 0 && !(VOGE_FB_ABL & 4)) {
        const bool cok = okv[a] && id[a] < Nattr;
        const uint32_t o = (uint32_t)(cok ? id[a] : 0) * (uint32_t)(4 * C);      // (bytes; Nattr * C < 2^30: host)
        if (C == 3) {
          const float3 v = at_bytes<float3>(colors, o);
          col[a][0] = cok ? v.x : 0.f; col[a][1] = cok ? v.y : 0.f; col[a][2] = cok ? v.z : 0.f;
        } else if (C == 4) {
          const float4 v = at_bytes<float4>(colors, o);
          col[a][0] = cok ? v.x : 0.f; col[a][1] = cok ? v.y : 0.f; col[a][2] = cok ? v.z : 0.f; col[a][3] = cok ? v.w : 0.f;
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) { const float v = at_bytes<float>(colors, o + 4u * c); col[a][c] = cok ? v : 0.f; }
        }
      }
    }
    // ---- em / sm of the slots: from act / dsd, or (NOAD) re-derived from the records with the forward's own operations ----
    if (!NOAD) {
#pragma unroll
      for (int a = 0; a < NS; ++a)
        if (live[a]) { em[a] = FAST_EXP(-av[a]); sm[a] = FAST_SQRT(dv2[a] + 1e-10f); }
    }
    if (NOAD && ISO) {
      const float dn2f = (dx * dx + dy * dy) + dz * dz;      // the forward's association
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const PairOut o = pair_eval_iso_at(rc[a][0].x, rc[a][0].y, rc[a][0].z, rc[a][0].w, lm[a], dx, dy, dz, dn2f);
        em[a] = okv[a] ? FAST_EXP(-o.act) : em[a]; sm[a] = okv[a] ? FAST_SQRT(o.dsd + 1e-10f) : sm[a];
      }
    }
    if (NOAD && DIAG) {         // per-axis forms: the same from the compact records
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const float4 r0 = rc[a][0], r1 = rc[a][NRC - 1];
        const PairOut o = pair_eval_diag(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, dx, dy, dz);
        em[a] = okv[a] ? FAST_EXP(-o.act) : em[a]; sm[a] = okv[a] ? FAST_SQRT(o.dsd + 1e-10f) : sm[a];
      }
    }
    if (NOAD && !ISO && !DIAG) {         // general forms: the same from the packed (mu, A) records, with the forward's operations
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const float4 r0 = rc[a][0], r1 = rc[a][NRC == 3 ? 1 : 0], r2 = rc[a][NRC == 3 ? 2 : 0];
        const float A[9] = {r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
        const PairOut o = pair_eval(r0.x, r0.y, r0.z, make_eval(r0.x, r0.y, r0.z, A), dx, dy, dz, dx * dx, dy * dy, dz * dz,
                                    dx * dy, dx * dz, dy * dz);
        em[a] = okv[a] ? FAST_EXP(-o.act) : em[a]; sm[a] = okv[a] ? FAST_SQRT(o.dsd + 1e-10f) : sm[a];
      }
    }
#pragma unroll
    for (int a = 0; a < NS; ++a) live[a] = okv[a];
    if (SRC == 0 && on && bg == nullptr) {
      // merge_final alone (interpolate_attr): the image's gradient is the merged attributes' own, and `wsum` carries the
      // gradient of the per-pixel weight sum (what a get_silhouette on the same fragments hands back), or is NULL
#pragma unroll
      for (int c = 0; c < C; ++c) gr[c] = px_g[c];
      if (wsum != nullptr) g_sum_w = px_ws;
      // (get_silhouette = min(sum w, 1) behind the same fragments: its gradient passes like torch.minimum's -- all of it below 1,
      //  half at a tie, none above)
      if (rgb != nullptr) g_sum_w *= (px_rgb[0] < 1.0f ? 1.0f : (px_rgb[0] == 1.0f ? 0.5f : 0.0f));
    } else if (SRC == 0 && on) {
      const float ws = px_ws;
      float sil = fminf(ws, 1.0f);
      const float pass_s = (thr > 0.0f) ? 0.0f : (ws < 1.0f ? 1.0f : (ws == 1.0f ? 0.5f : 0.0f));
      if (thr > 0.0f) sil = sil > thr ? 1.0f : 0.0f;
      float g_mask = 0.0f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float xc = fmaf(1.0f - sil, bg[c], px_rgb[c]);
        gr[c] = px_g[c] * (xc < 1.0f ? 1.0f : (xc == 1.0f ? 0.5f : 0.0f));     // min(x, 1) passes like torch.min
        g_mask = fmaf(-gr[c], bg[c], g_mask);
      }
      g_sum_w = g_mask * pass_s;
    }
    // ---- shade backward: g_w of the slots; u = g_w w ----
    float um[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      float gw;
      if (SRC == 0) gw = live[a] ? fmaf(gr[3], col[a][3], fmaf(gr[2], col[a][2], fmaf(gr[1], col[a][1], fmaf(gr[0], col[a][0], g_sum_w)))) : 0.0f;
      else gw = live[a] ? gwv[a] : 0.0f;
      um[a] = gw * wv[a];
    }
    // ---- composite backward (closed form, composite_core.h) ----
    FB_TICK(1);
    if (on) {
#pragma unroll
      for (int h2 = 0; h2 < NS / 2; ++h2) {
        const int a = 2 * h2;
        *reinterpret_cast<v2f *>(Llen + d0 + a) = (v2f){lm[a], lm[a + 1]};
        *reinterpret_cast<v2f *>(Lsp + d0 + a) = (v2f){sm[a] * kCs, sm[a + 1] * kCs};
        *reinterpret_cast<v2f *>(LE + d0 + a) = (v2f){em[a] * (sm[a] * kCs), em[a + 1] * (sm[a + 1] * kCs)};
        *reinterpret_cast<v2f *>(Lu + d0 + a) = splat(0.0f);
#if VOGE_BWD_RCOL
        *reinterpret_cast<v2f *>(LR + d0 + a) = splat(0.0f);
#endif
      }
      if (q < 2) {      // the sentinel pair in front of the pixel's row and the one behind it
        for (int t2 = q; t2 < 2; t2 += LP) {
          Llen[r0 + t2] = -kBig; Lsp[r0 + t2] = 1.0f; LE[r0 + t2] = 0.0f; Lu[r0 + t2] = 0.0f;
          const int eb = r0 + RS - 2 + t2;
          Llen[eb] = kBig; Lsp[eb] = 1.0f; LE[eb] = 0.0f; Lu[eb] = 0.0f;
#if VOGE_BWD_RCOL
          LR[r0 + t2] = 0.0f; LR[eb] = 0.0f;
#endif
        }
      }
    }
    wave_lds_sync();
    float ga[NS], gl[NS], gd[NS];
#if VOGE_FB_ABL & 2       // (timing experiment: no composite)
    for (int a = 0; a < NS; ++a) { ga[a] = um[a]; gl[a] = um[a] * sm[a]; gd[a] = um[a] * em[a]; }
#else
#if VOGE_BWD_RCOL
    compn_bwd_wave<NS, true>(lm, sm, em, um, Llen, Lsp, LE, Lu, d0, k0, NS * LP, q, LP, npm, on, on, true, pk.s0, occ, ga, gl, gd,
                             VOGE_FB_LDS_RMAX ? &L.rmax[on ? pk.ord : 0] : nullptr, LR);
#else
    compn_bwd_wave<NS>(lm, sm, em, um, Llen, Lsp, LE, Lu, d0, k0, NS * LP, q, LP, npm, on, on, true, pk.s0, occ, ga, gl, gd,
                       VOGE_FB_LDS_RMAX ? &L.rmax[on ? pk.ord : 0] : nullptr);
#endif
#endif
    wave_lds_sync();      // the rows are rewritten by the next round
    FB_TICK(2);
    if (SRC == 1 && g_hitlen != nullptr) {      // vert_hit_length is the trace's len itself (Aggregation.py:107)
#pragma unroll
      for (int a = 0; a < NS; ++a)
        if (live[a]) gl[a] += at_bytes<float>(g_hitlen, fb + (OffT)(4 * a));
    }
    // ---- trace backward terms (isotropic: trace_bwd.hip) + the colour term, one table entry per Gaussian.  (The
    // (mu, a) records are gathered only now: held across the composite they cost the kernel a wave per SIMD.) ----
    // (scalar sigmas without act / dsd: the records gathered above serve here too.  Otherwise they are gathered only now:
    //  held across the composite they cost the kernel a wave per SIMD -- 24 registers for the full forms)
    if (!NOAD || !ISO)
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      const uint32_t ro = (uint32_t)(live[a] ? id[a] : 0) * (16u * NRC);      // (P < 2^26: host)
#pragma unroll
      for (int r = 0; r < NRC; ++r)
        rc[a][r] = !(VOGE_FB_ABL & 8) ? at_bytes<float4>(rec, ro + 16u * r) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float dn2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float idn = __builtin_amdgcn_rcpf(dn2);
    // (two slots per lane: both slots' terms first, then ONE election loop for both -- wt_add2)
    constexpr bool PAIR = NS == 2 && VOGE_FB_PAIR_TABLE;
    float4 valp[PAIR ? 2 : 1][NV4];
    bool gop[2] = {false, false};
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      float4 val[NV4];
#pragma unroll
      for (int r = 0; r < NV4; ++r) val[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      bool go = live[a];
      if (go) {
        if (ISO) {
          const float mx = rc[a][0].x, my = rc[a][0].y, mz = rc[a][0].z, aa = rc[a][0].w;
          const float t = fmaf(mz, dz, fmaf(my, dy, mx * dx)) * idn;
          float vx = fmaf(-t, dx, mx), vy = fmaf(-t, dy, my), vz = fmaf(-t, dz, mz);
          {   // (trace_bwd.hip: the component of v along d is t's rounding error; project it out before 2 a g_act scales it)
            const float rr = fmaf(vz, dz, fmaf(vy, dy, vx * dx)) * idn;
            vx = fmaf(-rr, dx, vx); vy = fmaf(-rr, dy, vy); vz = fmaf(-rr, dz, vz);
          }
          const float c1 = gl[a] * idn, c2 = 2.0f * aa * ga[a];
          val[0] = make_float4(fmaf(c1, dx, c2 * vx), fmaf(c1, dy, c2 * vy), fmaf(c1, dz, c2 * vz),
                               fmaf(ga[a], fmaf(vz, vz, fmaf(vy, vy, vx * vx)), gd[a] * dn2));
        } else if (DIAG) {
          // A = diag(a0, a1, a2): the general terms below with every product by a zero coefficient left out (they add +-0 there)
          const float4 r0 = rc[a][0], r1 = rc[a][NRC - 1];
          const float mx = r0.x, my = r0.y, mz = r0.z;
          const float adx = r0.w * dx, ady = r1.x * dy, adz = r1.y * dz;
          const float ksk = fmaf(dz, adz, fmaf(dy, ady, dx * adx)), msk = fmaf(mz, adz, fmaf(my, ady, mx * adx));
          const float ik = __builtin_amdgcn_rcpf(ksk);
          const float t = msk * ik;
          float vx = fmaf(-t, dx, mx), vy = fmaf(-t, dy, my), vz = fmaf(-t, dz, mz);
          {
            const float rr = fmaf(vz, adz, fmaf(vy, ady, vx * adx)) * ik;
            vx = fmaf(-rr, dx, vx); vy = fmaf(-rr, dy, vy); vz = fmaf(-rr, dz, vz);
          }
          const float avx = r0.w * vx, avy = r1.x * vy, avz = r1.y * vz;
          const float c1 = gl[a] * ik, g_a = ga[a], g_d = gd[a];
          const float o0 = fmaf(c1, adx, g_a * (avx + avx)), o1 = fmaf(c1, ady, g_a * (avy + avy)), o2 = fmaf(c1, adz, g_a * (avz + avz));
          const float o3 = fmaf(g_a, vx * vx, fmaf(g_d, dx * dx, c1 * (vx * dx)));
          const float o4 = fmaf(g_a, vy * vy, fmaf(g_d, dy * dy, c1 * (vy * dy)));
          const float o5 = fmaf(g_a, vz * vz, fmaf(g_d, dz * dz, c1 * (vz * dz)));
          val[0] = make_float4(o0, o1, o2, o3);
          if (SRC == 0) {
            val[NV4 >= 2 ? 1 : 0] = make_float4(o4, o5, wv[a] * gr[0], wv[a] * gr[1]);
            val[NV4 >= 3 ? 2 : 0] = make_float4(wv[a] * gr[2], wv[a] * gr[3], 0.f, 0.f);
          } else {
            val[NV4 >= 2 ? 1 : 0] = make_float4(o4, o5, 0.f, 0.f);
          }
        } else {
          // the merged per-target terms of trace_bwd.hip (header there): g_mu (3) and the unsymmetrised g_A (9)
          const float4 r0 = rc[a][0], r1 = rc[a][NRC == 3 ? 1 : 0], r2 = rc[a][NRC == 3 ? 2 : 0];
          const float mx = r0.x, my = r0.y, mz = r0.z;
          const float A[9] = {r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
          const float adx = fmaf(A[2], dz, fmaf(A[1], dy, A[0] * dx)), ady = fmaf(A[5], dz, fmaf(A[4], dy, A[3] * dx)),
                      adz = fmaf(A[8], dz, fmaf(A[7], dy, A[6] * dx));
          const float tdx = fmaf(A[6], dz, fmaf(A[3], dy, A[0] * dx)), tdy = fmaf(A[7], dz, fmaf(A[4], dy, A[1] * dx)),
                      tdz = fmaf(A[8], dz, fmaf(A[5], dy, A[2] * dx));
          const float ksk = fmaf(dz, adz, fmaf(dy, ady, dx * adx)), msk = fmaf(mz, adz, fmaf(my, ady, mx * adx));
          const float ik = __builtin_amdgcn_rcpf(ksk);
          const float t = msk * ik;
          float vx = fmaf(-t, dx, mx), vy = fmaf(-t, dy, my), vz = fmaf(-t, dz, mz);
          {   // (v^T A d = 0 by construction: what is left of it is t's rounding error; one projection along d removes it)
            const float rr = fmaf(vz, adz, fmaf(vy, ady, vx * adx)) * ik;
            vx = fmaf(-rr, dx, vx); vy = fmaf(-rr, dy, vy); vz = fmaf(-rr, dz, vz);
          }
          const float avx = fmaf(A[2], vz, fmaf(A[1], vy, A[0] * vx)), avy = fmaf(A[5], vz, fmaf(A[4], vy, A[3] * vx)),
                      avz = fmaf(A[8], vz, fmaf(A[7], vy, A[6] * vx));
          const float tvx = fmaf(A[6], vz, fmaf(A[3], vy, A[0] * vx)), tvy = fmaf(A[7], vz, fmaf(A[4], vy, A[1] * vx)),
                      tvz = fmaf(A[8], vz, fmaf(A[5], vy, A[2] * vx));
          const float c1 = gl[a] * ik, g_a = ga[a], g_d = gd[a];
          float o[12];
          o[0] = fmaf(c1, adx, g_a * (avx + tvx + t * (tdx - adx)));
          o[1] = fmaf(c1, ady, g_a * (avy + tvy + t * (tdy - ady)));
          o[2] = fmaf(c1, adz, g_a * (avz + tvz + t * (tdz - adz)));
          const float d[3] = {dx, dy, dz}, v[3] = {vx, vy, vz};
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c)
              o[3 + 3 * i + c] = fmaf(g_a, fmaf(v[i], v[c], t * (d[i] * v[c] - v[i] * d[c])), fmaf(g_d, d[i] * d[c], c1 * (v[i] * d[c])));
          val[0] = make_float4(o[0], o[1], o[2], o[3]);
          val[NV4 >= 3 ? 1 : 0] = make_float4(o[4], o[5], o[6], o[7]);
          val[NV4 >= 3 ? 2 : 0] = make_float4(o[8], o[9], o[10], o[11]);
        }
        if (SRC == 0 && !DIAG) val[NV4 - 1] = make_float4(wv[a] * gr[0], wv[a] * gr[1], wv[a] * gr[2], wv[a] * gr[3]);
        go = false;
#pragma unroll
        for (int r = 0; r < NV4; ++r) go = go || (val[r].x != 0.f || val[r].y != 0.f || val[r].z != 0.f || val[r].w != 0.f);
      }
      if (PAIR) {
#pragma unroll
        for (int r = 0; r < NV4; ++r) valp[PAIR ? a : 0][r] = val[r];
        gop[a] = go;
        continue;
      }
      if (!__any(go)) continue;     // uniform
      FB_TICK(3);
#if VOGE_FB_ABL & 1       // (timing experiment: no table, nothing accumulated)
      if (go && val[0].x == 1.2345f && val[NV4 - 1].y == 3.21f) acc[id[a]] = val[0].w + val[NV4 - 1].x;
      continue;
#endif
      const int slot = wt_find(L.tab, id[a], go);
#ifdef VOGE_FB_TIMES      // (election rounds per accumulation: tools/fb_sections.py)
      {
        lds_vint *owner = lds_volatile(L.tab.owner);
        bool pending = go && slot >= 0;
        const unsigned long long m0 = __ballot(pending);
        unsigned rounds = 0;
        while (__any(pending)) {
          if (pending) owner[slot] = lane;
          if (pending && owner[slot] == lane) pending = false;
          ++rounds;
        }
        if (lane == 0 && blockIdx.x < kFbTimesWaves) {
          g_fb_tab[blockIdx.x][0] += 1; g_fb_tab[blockIdx.x][1] += rounds; g_fb_tab[blockIdx.x][2] += (unsigned)__popcll(m0);
        }
      }
#endif
      wt_add(L.tab, slot, val, go && slot >= 0, lane);
      if (go && slot < 0) {         // table full: rare, straight to memory
#pragma unroll
        for (int r = 0; r < NV4; ++r) {
          const float o[4] = {val[r].x, val[r].y, val[r].z, val[r].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) sum_add(id[a], 4 * r + c, o[c]);
        }
      }
      FB_TICK(4);
    }
    if (PAIR && __any(gop[0] | gop[1])) {
      FB_TICK(3);
      int slot0, slot1;
      wt_find2(L.tab, id[0], gop[0], id[NS - 1], gop[1], slot0, slot1);
#if VOGE_FB_TABLE_BY_PIXEL
      wt_add2_by_group(L.tab, slot0, valp[0], gop[0] && slot0 >= 0, slot1, valp[PAIR ? 1 : 0], gop[1] && slot1 >= 0, pk.ord);
#else
      wt_add2(L.tab, slot0, valp[0], gop[0] && slot0 >= 0, slot1, valp[PAIR ? 1 : 0], gop[1] && slot1 >= 0, lane);
#endif
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int sl = a == 0 ? slot0 : slot1;
        if (gop[a] && sl < 0) {         // table full: rare, straight to memory
#pragma unroll
          for (int r = 0; r < NV4; ++r) {
            const float o[4] = {valp[PAIR ? a : 0][r].x, valp[PAIR ? a : 0][r].y, valp[PAIR ? a : 0][r].z, valp[PAIR ? a : 0][r].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) sum_add(id[a == 0 ? 0 : NS - 1], 4 * r + c, o[c]);
          }
        }
      }
      FB_TICK(4);
    }
  }
  FB_TICK(0);
  {   // flush: NACC (4 | 8 | 12 | 16) adjacent lanes per entry -> the bytes of acc[p]: lane-coalesced atomics
    constexpr int EPI = 64 / NACC;      // entries per iteration (12 sums: five entries, four idle lanes)
    const int e = lane / NACC, c = lane - e * NACC;
    const float *vals = reinterpret_cast<const float *>(L.tab.vals);
    const int n = wt_compact(L.tab, lane);
    const lds_vint *list = lds_volatile(L.tab.owner);
    for (int i = e; i < n && e < EPI; i += EPI) {
      const int s = list[i];
      sum_add(L.tab.keys[s], c, vals[s * NACC + c]);
    }
  }
  FB_TICK(5);
  FB_TOUT();
}

// acc [P][S] (S = 8: (g_mu, g_a | w g_rgb), S = 4: (g_mu, g_a)) -> the gradients of what the caller passed in: verts /
// sigmas through the view's chain rule (and the sum over the batch when one Gaussian set is shared by all views),
// colours [Nattr][C].
__global__ void __launch_bounds__(256)
fragment_bwd_finish_kernel(const float *__restrict__ acc, const int S, const float *__restrict__ a_in, const int P, const int N,
                           const int B, const IsoView view, const int C, const long Nattr, float *__restrict__ g_mus,
                           float *__restrict__ g_a, float *__restrict__ g_colors) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g_colors != nullptr && g < Nattr) {
    for (int c = 0; c < C; ++c) g_colors[(size_t)g * C + c] = (g < P) ? acc[S * (size_t)g + 4 + c] : 0.0f;
  }
  const int n_out = view.shared ? N : P;
  if (g >= n_out || g_mus == nullptr) return;
  float4 v = *reinterpret_cast<const float4 *>(acc + S * (size_t)g);
  if (view.shared)
    for (int b = 1; b < B; ++b) {
      const float4 w = *reinterpret_cast<const float4 *>(acc + S * ((size_t)b * N + g));
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
  g_mus[3 * (size_t)g] = v.x; g_mus[3 * (size_t)g + 1] = v.y; g_mus[3 * (size_t)g + 2] = v.z;
  float ga = v.w;
  if (view.mode == 1) ga = 2.0f * ga;
  else if (view.mode == 2) { const float s = a_in[g]; ga = -2.0f * ga / (s * s); }
  g_a[g] = ga;
}

// General 3x3 forms: mus [P,3] + isigmas [P,9] -> 3 x float4 per Gaussian (one gather stream), acc [P][16] zeroed
__global__ void __launch_bounds__(256)
fragment_bwd_pack_kernel(const float *__restrict__ mus, const float *__restrict__ isg, const int P, float4 *__restrict__ rec,
                         float4 *__restrict__ acc, const int S4 /* float4s of sums per Gaussian: 4 | 3 */) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P) return;
  for (int q = 0; q < S4; ++q) acc[S4 * (size_t)g + q] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float *m = mus + 3 * (size_t)g, *A = isg + 9 * (size_t)g;
  rec[3 * (size_t)g + 0] = make_float4(m[0], m[1], m[2], A[0]);
  rec[3 * (size_t)g + 1] = make_float4(A[1], A[2], A[3], A[4]);
  rec[3 * (size_t)g + 2] = make_float4(A[5], A[6], A[7], A[8]);
}
// acc [P][S] (S = 16 with the colour term, 12 without) -> g_mus [P,3], g_isigmas [P,9], g_colors [Nattr,C]
__global__ void __launch_bounds__(256)
fragment_bwd_finish_general_kernel(const float *__restrict__ acc, const int S, const int P, const int C, const long Nattr,
                                   float *__restrict__ g_mus, float *__restrict__ g_isg, float *__restrict__ g_colors) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long g = t >> 4;
  const int c = (int)(t & 15);
  if ((g >= P && g >= Nattr) || c >= S) return;
  const float v = (g < P) ? acc[g * S + c] : 0.0f;
  if (c < 3) { if (g < P && g_mus != nullptr) g_mus[3 * g + c] = v; }
  else if (c < 12) { if (g < P && g_isg != nullptr) g_isg[9 * g + (c - 3)] = v; }
  else if (c - 12 < C && g < Nattr && g_colors != nullptr) g_colors[g * C + (c - 12)] = v;
}

// act / dsd of fragments that were traced without them (A = a I): the sweep epilogue's own operations per live slot,
// the trace's sentinels elsewhere.  One thread per slot.
__global__ void __launch_bounds__(256)
fragment_act_dsd_iso_kernel(const float4 *__restrict__ rec, const float *__restrict__ rays, const int32_t *__restrict__ idx,
                            const float *__restrict__ len, const int32_t *__restrict__ cnt, const long npix, const int K,
                            const int P, float *__restrict__ act, float *__restrict__ dsd) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npix * K) return;
  const long pix = t / K;
  const int k = (int)(t - pix * K);
  float a = VOGE_SENT_ACT, d = 0.0f;
  const int id = idx[t];
  if (k < cnt[pix] && id >= 0 && id < P) {
    const float4 r = rec[id];
    const float dx = rays[pix * 3], dy = rays[pix * 3 + 1], dz = rays[pix * 3 + 2];
    const PairOut o = pair_eval_iso_at(r.x, r.y, r.z, r.w, len[t], dx, dy, dz, (dx * dx + dy * dy) + dz * dz);
    a = o.act; d = o.dsd;
  }
  act[t] = a; dsd[t] = d;
}

}  // namespace voge

using namespace voge;

#ifdef VOGE_FB_TIMES
extern "C" int voge_debug_fb_times(unsigned long long *out, int reset) {      // out: [8] sums over the workgroups
  static unsigned long long host[voge::kFbTimesWaves][8];
  if (reset) {
    for (auto &row : host) for (auto &v : row) v = 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(voge::g_fb_times), host, sizeof(host));
  }
  const int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(voge::g_fb_times), sizeof(host));
  for (int i = 0; i < 8; ++i) out[i] = 0;
  for (int w = 0; w < voge::kFbTimesWaves; ++w)
    for (int i = 0; i < 8; ++i) out[i] += host[w][i];
  return rc;
}
extern "C" int voge_debug_fb_tab(unsigned long long *out /* [4] sums */, int reset) {
  static unsigned long long host[voge::kFbTimesWaves][4];
  if (reset) {
    for (auto &row : host) for (auto &v : row) v = 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(voge::g_fb_tab), host, sizeof(host));
  }
  const int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(voge::g_fb_tab), sizeof(host));
  for (int i = 0; i < 4; ++i) out[i] = 0;
  for (int w = 0; w < voge::kFbTimesWaves; ++w)
    for (int i = 0; i < 4; ++i) out[i] += host[w][i];
  return rc;
}
extern "C" int voge_debug_fb_wall(unsigned long long *out /* [n][2] */, int n, int reset) {
  static unsigned long long host[voge::kFbTimesWaves][2];
  if (reset) {
    for (auto &row : host) for (auto &v : row) v = 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(voge::g_fb_wall), host, sizeof(host));
  }
  const int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(voge::g_fb_wall), sizeof(host));
  for (int w = 0; w < n && w < voge::kFbTimesWaves; ++w) { out[2 * w] = host[w][0]; out[2 * w + 1] = host[w][1]; }
  return rc;
}
extern "C" int voge_debug_cw_stats(unsigned long long *out, int reset) {      // out: [4] sums (composite_core.h: CW_COUNT)
  static unsigned long long host[1 << 16][4];
  if (reset) {
    for (auto &row : host) for (auto &v : row) v = 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(voge::g_cw_stats), host, sizeof(host));
  }
  const int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(voge::g_cw_stats), sizeof(host));
  for (int i = 0; i < 4; ++i) out[i] = 0;
  for (int w = 0; w < (1 << 16); ++w)
    for (int i = 0; i < 4; ++i) out[i] += host[w][i];
  return rc;
}
#endif

extern "C" int voge_fragment_act_dsd_iso(const float *records, const float *rays, const int32_t *idx, const float *len,
                                         const int32_t *cnt, long npix, int K, int P, float *act, float *dsd,
                                         voge_stream_t stream) {
  if (npix < 0 || K <= 0 || P < 0) return VOGE_ERR_BAD_ARG;
  if (npix == 0) return 0;
  if ((!records && P > 0) || !rays || !idx || !len || !cnt || !act || !dsd) return VOGE_ERR_BAD_ARG;
  const long n = npix * K;
  hipLaunchKernelGGL(fragment_act_dsd_iso_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(records), rays, idx, len, cnt, npix, K, P, act, dsd);
  return launch_status();
}

// (the isotropic forms use the first 32 | 16 bytes per Gaussian; the general ones 48 of packed records + 64 | 48 of sums)
extern "C" size_t voge_fragment_bwd_workspace_bytes(int P) { return P <= 0 ? 0 : (size_t)P * 112; }

namespace {
struct FbArgs {      // what every form of the fused backward hands its kernel
  const float4 *rec; const float *rays, *colors; const int32_t *idx, *cnt; const float *weight, *act, *len, *dsd, *rgb, *wsum, *bg;
  float thr; const float *g; long gs_pix, gs_c; const float *g_hitlen; float occ; int P; long nrows; int W, K; long Nattr; float *acc;
};
template <int SRC, int C, int NS, typename OffT, bool ISO, bool NOAD>
void fb_launch(const FbArgs &a, hipStream_t st) {
  const long blocks = (long)((a.W + kFbGW - 1) / kFbGW) * ((a.nrows + kFbGH - 1) / kFbGH);
  // (the kernel's gathers are unconditional at clamped ids: row 0 of the records (P >= 1 here) and of the colour table must be
  //  readable.  An empty colour table reads its dropped row out of the records instead.)
  const float *colors = (a.Nattr > 0 && a.colors != nullptr) ? a.colors : reinterpret_cast<const float *>(a.rec);
  hipLaunchKernelGGL((fragment_bwd_kernel<SRC, C, NS, OffT, ISO, NOAD>), dim3((unsigned)blocks), dim3(64), 0, st, a.rec, a.rays,
                     colors, a.idx, a.cnt, a.weight, a.act, a.len, a.dsd, a.rgb, a.wsum, a.bg, a.thr, a.g, a.gs_pix, a.gs_c,
                     a.g_hitlen, a.occ, a.P, a.nrows, a.W, a.K, a.Nattr, a.acc);
}
template <int SRC, int C, int NS, typename OffT>
void fb_launch_diag_t(const FbArgs &a, hipStream_t st) {
  const long blocks = (long)((a.W + kFbGW - 1) / kFbGW) * ((a.nrows + kFbGH - 1) / kFbGH);
  const float *colors = (a.Nattr > 0 && a.colors != nullptr) ? a.colors : reinterpret_cast<const float *>(a.rec);
  hipLaunchKernelGGL((fragment_bwd_kernel<SRC, C, NS, OffT, false, true, true>), dim3((unsigned)blocks), dim3(64), 0, st, a.rec, a.rays,
                     colors, a.idx, a.cnt, a.weight, a.act, a.len, a.dsd, a.rgb, a.wsum, a.bg, a.thr, a.g, a.gs_pix, a.gs_c,
                     a.g_hitlen, a.occ, a.P, a.nrows, a.W, a.K, a.Nattr, a.acc);
}
template <int SRC, int C, int NS>
void fb_launch_diag(const FbArgs &a, hipStream_t st) {
  if ((double)a.nrows * a.W * a.K < (double)(1l << 30)) fb_launch_diag_t<SRC, C, NS, uint32_t>(a, st); else fb_launch_diag_t<SRC, C, NS, size_t>(a, st);
}
template <int SRC, int C, int NS, bool ISO>
void fb_launch_off(const FbArgs &a, hipStream_t st) {
  const bool small = (double)a.nrows * a.W * a.K < (double)(1l << 30);      // every byte offset fits 32 bits
  if (a.act == nullptr) { if (small) fb_launch<SRC, C, NS, uint32_t, ISO, true>(a, st); else fb_launch<SRC, C, NS, size_t, ISO, true>(a, st); }
  else { if (small) fb_launch<SRC, C, NS, uint32_t, ISO, false>(a, st); else fb_launch<SRC, C, NS, size_t, ISO, false>(a, st); }
}
template <bool ISO>
void fb_launch_shade(const FbArgs &a, int C, hipStream_t st) {
  switch (C) {
    case 1: fb_launch_off<0, 1, 2, ISO>(a, st); break;
    case 2: fb_launch_off<0, 2, 2, ISO>(a, st); break;
    case 3: fb_launch_off<0, 3, 2, ISO>(a, st); break;
    default: fb_launch_off<0, 4, 2, ISO>(a, st); break;
  }
}
template <bool ISO>
void fb_launch_gw(const FbArgs &a, hipStream_t st) {
  if (a.K <= 128) fb_launch_off<1, 0, 2, ISO>(a, st); else fb_launch_off<1, 0, 4, ISO>(a, st);
}
}  // namespace

extern "C" int voge_fragment_shade_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode,
                                           const float *rays, const float *colors, const int32_t *idx, const int32_t *cnt,
                                           const float *weight, const float *act, const float *len, const float *dsd,
                                           const float *rgb, const float *wsum, const float *bg, float thr,
                                           const float *g_img, long g_stride_pix, long g_stride_c, float occ, int B, int N,
                                           long nrows, int W, int K, int C, long Nattr, void *workspace, size_t workspace_bytes, float *g_verts,
                                           float *g_sigmas, float *g_colors, voge_stream_t stream) {
  if (B < 0 || N < 0 || nrows < 0 || W < 0 || K <= 0 || C <= 0 || C > 4 || Nattr < 0 || sigma_mode < 0 || sigma_mode > 2)
    return VOGE_ERR_BAD_ARG;
  if (K > 128) return VOGE_ERR_K_TOO_LARGE;      // a lane owns a pair of slots; a pixel's lanes fit one wave
  const int P = B * N;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0 || nrows * W == 0) {
    if (g_colors && Nattr > 0) return (int)voge_fill_async(g_colors, 0, sizeof(float) * (size_t)Nattr * C, st);
    return 0;
  }
  if (!records || !rays || !colors || !idx || !cnt || !weight || !len || !rgb || !wsum || !bg || !g_img || !workspace)
    return VOGE_ERR_BAD_ARG;
  if ((act == nullptr) != (dsd == nullptr)) return VOGE_ERR_BAD_ARG;      // both or neither (neither: re-derived from the records)
  if ((g_verts == nullptr) != (g_sigmas == nullptr) || (sigma_mode == 2 && g_sigmas && !sigmas)) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < (size_t)P * 32) return VOGE_ERR_WORKSPACE;
  if (Nattr * C >= (1l << 30) || P >= (1 << 26)) return VOGE_ERR_BAD_ARG;      // 32-bit byte offsets of the gathers
  float *acc = reinterpret_cast<float *>(workspace);
  { const hipError_t e = voge_fill_async(acc, 0, (size_t)P * 32, st); if (e != hipSuccess) return (int)e; }
  const FbArgs a{reinterpret_cast<const float4 *>(records), rays, colors, idx, cnt, weight, act, len, dsd, rgb, wsum, bg, thr, g_img,
                 g_stride_pix, g_stride_c, nullptr, occ, P, nrows, W, K, Nattr, acc};
  fb_launch_shade<true>(a, C, st);
  const long n_fin = (Nattr > P) ? Nattr : P;
  hipLaunchKernelGGL(fragment_bwd_finish_kernel, dim3((unsigned)((n_fin + 255) / 256)), dim3(256), 0, st, acc, 8, sigmas, P, N, B,
                     IsoView{nullptr, shared ? 1 : 0, sigma_mode}, C, Nattr, g_verts, g_sigmas, g_colors);
  return launch_status();
}

// Round 6, the frame's backward: voge_fragment_shade_bwd_iso / voge_fragment_merge_bwd_iso WITHOUT the fill launch in front --
// `acc` (voge_frame_bwd_acc_bytes(B * N) bytes) arrives zeroed: the frame's forward does that on its way
// (voge_frame_shade_fwd_iso's bwd_acc), and it is good for ONE backward.  The fused kernel + the finishing pass.
extern "C" size_t voge_frame_bwd_acc_bytes(int P) { return P <= 0 ? 0 : (size_t)P * 32; }

static int frame_bwd_impl(const bool merge, const float *records, const float *sigmas, int shared, int sigma_mode, const float *rays,
                          const float *colors, const int32_t *idx, const int32_t *cnt, const float *weight, const float *len,
                          const float *rgb, const float *wsum, const float *bg, float thr, const float *g, long g_stride_pix,
                          long g_stride_c, float occ, int B, int N, long nrows, int W, int K, int C, long Nattr, void *acc_zeroed,
                          size_t acc_bytes, float *g_verts, float *g_sigmas, float *g_colors, voge_stream_t stream) {
  if (B < 0 || N < 0 || nrows < 0 || W < 0 || K <= 0 || C <= 0 || C > 4 || Nattr < 0 || sigma_mode < 0 || sigma_mode > 2)
    return VOGE_ERR_BAD_ARG;
  if (K > 128) return VOGE_ERR_K_TOO_LARGE;
  const int P = B * N;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0 || nrows * W == 0) {
    if (g_colors && Nattr > 0) return (int)voge_fill_async(g_colors, 0, sizeof(float) * (size_t)Nattr * C, st);
    return 0;
  }
  if (!records || !rays || !colors || !idx || !cnt || !weight || !len || !g || !acc_zeroed) return VOGE_ERR_BAD_ARG;
  if (!merge && (!rgb || !wsum || !bg)) return VOGE_ERR_BAD_ARG;
  if ((g_verts == nullptr) != (g_sigmas == nullptr) || (sigma_mode == 2 && g_sigmas && !sigmas)) return VOGE_ERR_BAD_ARG;
  if (acc_bytes < voge_frame_bwd_acc_bytes(P)) return VOGE_ERR_WORKSPACE;
  if (Nattr * C >= (1l << 30) || P >= (1 << 26)) return VOGE_ERR_BAD_ARG;      // 32-bit byte offsets of the gathers
  float *acc = reinterpret_cast<float *>(acc_zeroed);
  // (merge form: bg = NULL selects it inside the kernel; its `wsum` operand carries g_wsum)
  // (merge form: bg = NULL selects it inside the kernel; its `wsum` operand carries g_wsum -- or, with `rgb` = the forward's weight
  //  sums, the silhouette's gradient)
  const FbArgs a{reinterpret_cast<const float4 *>(records), rays, colors, idx, cnt, weight, nullptr, len, nullptr, rgb, wsum,
                 merge ? nullptr : bg, merge ? -1.0f : thr, g, g_stride_pix, g_stride_c, nullptr, occ, P, nrows, W, K, Nattr, acc};
  fb_launch_shade<true>(a, C, st);
  const long n_fin = (Nattr > P) ? Nattr : P;
  hipLaunchKernelGGL(fragment_bwd_finish_kernel, dim3((unsigned)((n_fin + 255) / 256)), dim3(256), 0, st, acc, 8, sigmas, P, N, B,
                     IsoView{nullptr, shared ? 1 : 0, sigma_mode}, C, Nattr, g_verts, g_sigmas, g_colors);
  return launch_status();
}

extern "C" int voge_frame_shade_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode,
                                        const float *rays, const float *colors, const int32_t *idx, const int32_t *cnt,
                                        const float *weight, const float *len, const float *rgb, const float *wsum,
                                        const float *bg, float thr, const float *g_img, long g_stride_pix, long g_stride_c,
                                        float occ, int B, int N, long nrows, int W, int K, int C, long Nattr, void *acc_zeroed,
                                        size_t acc_bytes, float *g_verts, float *g_sigmas, float *g_colors, voge_stream_t stream) {
  return frame_bwd_impl(false, records, sigmas, shared, sigma_mode, rays, colors, idx, cnt, weight, len, rgb, wsum, bg, thr, g_img,
                        g_stride_pix, g_stride_c, occ, B, N, nrows, W, K, C, Nattr, acc_zeroed, acc_bytes, g_verts, g_sigmas, g_colors, stream);
}

extern "C" int voge_frame_merge_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode,
                                        const float *rays, const float *attr, const int32_t *idx, const int32_t *cnt,
                                        const float *weight, const float *len, const float *g_rgb, long g_stride_pix,
                                        long g_stride_c, const float *g_wsum, const float *wsum_fwd, float occ, int B, int N, long nrows,
                                        int W, int K, int C, long Nattr, void *acc_zeroed, size_t acc_bytes, float *g_verts,
                                        float *g_sigmas, float *g_attr, voge_stream_t stream) {
  return frame_bwd_impl(true, records, sigmas, shared, sigma_mode, rays, attr, idx, cnt, weight, len, wsum_fwd, g_wsum, nullptr, -1.0f, g_rgb,
                        g_stride_pix, g_stride_c, occ, B, N, nrows, W, K, C, Nattr, acc_zeroed, acc_bytes, g_verts, g_sigmas, g_attr, stream);
}

// ---- the general forms on the frame path: every backward route of fragments made by voge_frame_trace_fwd_gen.  The packed
// (mu, A) records come from the forward (no pack launch), `acc` [P][16 | 12] floats is either zeroed already (acc_is_zero: the
// frame's composite did it on its way) or filled here, and the finishing pass applies the renderer's own chain rule
// (Renderer.py:130-137 backwards: the sum over the views of a shared set; d A / d sigma = 2 on the diagonal for per-axis sigmas,
// 2 everywhere for [3][3] ones) -- no general_preamble_bwd launch behind it.
//   form 0: the image's gradient (to_colored_background: rgb, wsum, bg, thr, g = g_img);   form 1: merge_final's (interpolate_attr:
//   g = g_rgb, wsum = g_wsum | NULL -- or, with rgb = the forward's weight sums [pix], the gradient of get_silhouette);   form 2: the weights' own (g = g_weight with strides, g_hitlen | NULL; C = 0, no attr).
__global__ void __launch_bounds__(256)
fragment_bwd_finish_view_kernel(const float *__restrict__ acc, const int S, const int P, const int N, const int B, const int shared_v,
                                const int shared_s, const int kind, const int C, const long Nattr, float *__restrict__ g_verts,
                                float *__restrict__ g_sigmas, float *__restrict__ g_attr) {
  // acc per Gaussian -- kind 2 (S = 16 | 12): g_mu [0..2], g_A [3..11], w g_rgb [12..15]; kind 1, the per-axis kernel's entries
  // (S = 12 | 8): g_mu [0..2], g_A00 [3], g_A11 [4], g_A22 [5], w g_rgb [6..9]
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g_attr != nullptr && t < Nattr) {
    for (int c = 0; c < C; ++c) g_attr[t * C + c] = (t < P) ? acc[t * S + (kind == 1 ? 6 : 12) + c] : 0.0f;
  }
  if (g_verts != nullptr && t < (shared_v ? N : P)) {
    float v[3] = {0.f, 0.f, 0.f};
    for (int b = 0; b < (shared_v ? B : 1); ++b) {      // (a shared set: its views in a fixed order)
      const float *a = acc + ((long)b * N + t) * S;
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2];
    }
    g_verts[3 * t] = v[0]; g_verts[3 * t + 1] = v[1]; g_verts[3 * t + 2] = v[2];
  }
  if (g_sigmas != nullptr && kind == 1 && t < (shared_s ? N : P)) {
    float v[3] = {0.f, 0.f, 0.f};
    for (int b = 0; b < (shared_s ? B : 1); ++b) {
      const float *a = acc + ((long)b * N + t) * S + 3;
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2];
    }
    g_sigmas[3 * t] = 2.0f * v[0]; g_sigmas[3 * t + 1] = 2.0f * v[1]; g_sigmas[3 * t + 2] = 2.0f * v[2];
  } else if (g_sigmas != nullptr && t < (shared_s ? N : P)) {
    float v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = 0.f;
    for (int b = 0; b < (shared_s ? B : 1); ++b) {
      const float *a = acc + ((long)b * N + t) * S + 3;
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] += a[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) g_sigmas[9 * t + i] = 2.0f * v[i];
  }
}

extern "C" size_t voge_frame_bwd_gen_acc_bytes(int P) { return P <= 0 ? 0 : (size_t)P * 64; }      // (kind 1 uses 48 | 32 of them)

extern "C" int voge_frame_bwd_gen(int form, const float *records, int shared_verts, int shared_sigmas, int kind, const float *rays,
                                  const float *attr, const int32_t *idx, const int32_t *cnt, const float *weight, const float *act,
                                  const float *len, const float *dsd, const float *rgb, const float *wsum, const float *bg, float thr,
                                  const float *g, long g_stride0, long g_stride1, const float *g_hitlen, float occ, int B, int N,
                                  long nrows, int W, int K, int C, long Nattr, void *acc, size_t acc_bytes, int acc_is_zero,
                                  float *g_verts, float *g_sigmas, float *g_attr, voge_stream_t stream) {
  if (form < 0 || form > 2 || (kind != 1 && kind != 2) || B < 0 || N < 0 || nrows < 0 || W < 0 || K <= 0 || Nattr < 0) return VOGE_ERR_BAD_ARG;
  if (form != 2 && (C <= 0 || C > 4)) return VOGE_ERR_BAD_ARG;
  if (K > (form == 2 ? VOGE_MAX_K : 128)) return VOGE_ERR_K_TOO_LARGE;
  const int P = B * N;
  hipStream_t st = (hipStream_t)stream;
  const long nv = shared_verts ? N : P, ns = (long)(shared_sigmas ? N : P) * (kind == 1 ? 3 : 9);
  if (P == 0 || nrows * W == 0) {      // nothing was traced: zero gradients
    hipError_t e = hipSuccess;
    if (g_verts && nv > 0) e = voge_fill_async(g_verts, 0, sizeof(float) * 3 * (size_t)nv, st);
    if (e == hipSuccess && g_sigmas && ns > 0) e = voge_fill_async(g_sigmas, 0, sizeof(float) * (size_t)ns, st);
    if (e == hipSuccess && g_attr && Nattr > 0 && form != 2) e = voge_fill_async(g_attr, 0, sizeof(float) * (size_t)Nattr * C, st);
    return (int)e;
  }
  if (!records || !rays || !idx || !cnt || !weight || !len || !acc || (act == nullptr) != (dsd == nullptr)) return VOGE_ERR_BAD_ARG;
  if (form != 2 && (!attr || !g)) return VOGE_ERR_BAD_ARG;
  if (form == 0 && (!rgb || !wsum || !bg)) return VOGE_ERR_BAD_ARG;
  if ((g_verts == nullptr) != (g_sigmas == nullptr)) return VOGE_ERR_BAD_ARG;
  const int S = kind == 1 ? (form == 2 ? 8 : 12) : (form == 2 ? 12 : 16);
  if (kind == 1 && (act || dsd)) return VOGE_ERR_BAD_ARG;      // (the per-axis form keeps neither: re-derived from its 32-byte records)
  if (acc_bytes < (size_t)P * S * 4) return VOGE_ERR_WORKSPACE;
  if (P >= (1 << 26) || (form != 2 && Nattr * C >= (1l << 30))) return VOGE_ERR_BAD_ARG;      // 32-bit byte offsets of the gathers
  float *accf = reinterpret_cast<float *>(acc);
  if (!acc_is_zero) { const hipError_t e = voge_fill_async(accf, 0, (size_t)P * S * 4, st); if (e != hipSuccess) return (int)e; }
  const float4 *rec = reinterpret_cast<const float4 *>(records);
  if (form == 2) {
    const FbArgs a{rec, rays, nullptr, idx, cnt, weight, act, len, dsd, nullptr, nullptr, nullptr, -1.0f, g, g_stride0, g_stride1, g_hitlen,
                   occ, P, nrows, W, K, 0, accf};
    if (kind == 1) { if (K <= 128) fb_launch_diag<1, 0, 2>(a, st); else fb_launch_diag<1, 0, 4>(a, st); }
    else fb_launch_gw<false>(a, st);
  } else {
    const FbArgs a{rec, rays, attr, idx, cnt, weight, act, len, dsd, rgb /* (form 1: NULL | the forward's weight sums) */, wsum, form == 0 ? bg : nullptr,
                   form == 0 ? thr : -1.0f, g, g_stride0, g_stride1, nullptr, occ, P, nrows, W, K, Nattr, accf};
    if (kind == 1) {
      switch (C) {
        case 1: fb_launch_diag<0, 1, 2>(a, st); break;
        case 2: fb_launch_diag<0, 2, 2>(a, st); break;
        case 3: fb_launch_diag<0, 3, 2>(a, st); break;
        default: fb_launch_diag<0, 4, 2>(a, st); break;
      }
    } else {
      fb_launch_shade<false>(a, C, st);
    }
  }
  const long n_fin = (form != 2 && Nattr > P) ? Nattr : P;
  hipLaunchKernelGGL(fragment_bwd_finish_view_kernel, dim3((unsigned)((n_fin + 255) / 256)), dim3(256), 0, st, accf, S, P, N, B,
                     shared_verts ? 1 : 0, shared_sigmas ? 1 : 0, kind, form == 2 ? 0 : C, form == 2 ? 0l : Nattr, g_verts, g_sigmas,
                     form == 2 ? nullptr : g_attr);
  return launch_status();
}

extern "C" int voge_fragment_shade_bwd(const float *mus, const float *isigmas, const float *rays, const float *colors,
                                       const int32_t *idx, const int32_t *cnt, const float *weight, const float *act,
                                       const float *len, const float *dsd, const float *rgb, const float *wsum,
                                       const float *bg, float thr, const float *g_img, long g_stride_pix, long g_stride_c,
                                       float occ, int P, long nrows, int W, int K, int C, long Nattr, void *workspace,
                                       size_t workspace_bytes, float *g_mus, float *g_isigmas, float *g_colors,
                                       voge_stream_t stream) {
  if (P < 0 || nrows < 0 || W < 0 || K <= 0 || C <= 0 || C > 4 || Nattr < 0) return VOGE_ERR_BAD_ARG;
  if (K > 128) return VOGE_ERR_K_TOO_LARGE;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0 || nrows * W == 0) {
    if (g_colors && Nattr > 0) return (int)voge_fill_async(g_colors, 0, sizeof(float) * (size_t)Nattr * C, st);
    return 0;
  }
  if (!mus || !isigmas || !rays || !colors || !idx || !cnt || !weight || !len || !rgb || !wsum || !bg || !g_img || !workspace)
    return VOGE_ERR_BAD_ARG;
  if ((act == nullptr) != (dsd == nullptr)) return VOGE_ERR_BAD_ARG;      // both or neither (neither: re-derived from (mu, A))
  if ((g_mus == nullptr) != (g_isigmas == nullptr)) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < voge_fragment_bwd_workspace_bytes(P)) return VOGE_ERR_WORKSPACE;
  if (Nattr * C >= (1l << 30) || P >= (1 << 26)) return VOGE_ERR_BAD_ARG;      // 32-bit byte offsets of the gathers
  float4 *rec = reinterpret_cast<float4 *>(workspace);
  float *acc = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + (size_t)P * 48);
  hipLaunchKernelGGL(fragment_bwd_pack_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, mus, isigmas, P, rec,
                     reinterpret_cast<float4 *>(acc), 4);
  const FbArgs a{rec, rays, colors, idx, cnt, weight, act, len, dsd, rgb, wsum, bg, thr, g_img, g_stride_pix, g_stride_c, nullptr, occ,
                 P, nrows, W, K, Nattr, acc};
  fb_launch_shade<false>(a, C, st);
  const long n_fin = (Nattr > P) ? Nattr : P;
  hipLaunchKernelGGL(fragment_bwd_finish_general_kernel, dim3((unsigned)((n_fin * 16 + 255) / 256)), dim3(256), 0, st, acc, 16, P, C,
                     Nattr, g_mus, g_isigmas, g_colors);
  return launch_status();
}

// The merge form (below) for general 3x3 forms: interpolate_attr (+ get_silhouette) on the fragments of voge_trace_lean_fwd.
extern "C" int voge_fragment_merge_bwd(const float *mus, const float *isigmas, const float *rays, const float *attr,
                                       const int32_t *idx, const int32_t *cnt, const float *weight, const float *act,
                                       const float *len, const float *dsd, const float *g_rgb, long g_stride_pix,
                                       long g_stride_c, const float *g_wsum, float occ, int P, long nrows, int W, int K, int C,
                                       long Nattr, void *workspace, size_t workspace_bytes, float *g_mus, float *g_isigmas,
                                       float *g_attr, voge_stream_t stream) {
  if (P < 0 || nrows < 0 || W < 0 || K <= 0 || C <= 0 || C > 4 || Nattr < 0) return VOGE_ERR_BAD_ARG;
  if (K > 128) return VOGE_ERR_K_TOO_LARGE;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0 || nrows * W == 0) {
    if (g_attr && Nattr > 0) return (int)voge_fill_async(g_attr, 0, sizeof(float) * (size_t)Nattr * C, st);
    return 0;
  }
  if (!mus || !isigmas || !rays || !attr || !idx || !cnt || !weight || !len || !g_rgb || !workspace) return VOGE_ERR_BAD_ARG;
  if ((act == nullptr) != (dsd == nullptr)) return VOGE_ERR_BAD_ARG;
  if ((g_mus == nullptr) != (g_isigmas == nullptr)) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < voge_fragment_bwd_workspace_bytes(P)) return VOGE_ERR_WORKSPACE;
  if (Nattr * C >= (1l << 30) || P >= (1 << 26)) return VOGE_ERR_BAD_ARG;
  float4 *rec = reinterpret_cast<float4 *>(workspace);
  float *acc = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + (size_t)P * 48);
  hipLaunchKernelGGL(fragment_bwd_pack_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, mus, isigmas, P, rec,
                     reinterpret_cast<float4 *>(acc), 4);
  const FbArgs a{rec, rays, attr, idx, cnt, weight, act, len, dsd, nullptr, g_wsum, nullptr, -1.0f, g_rgb, g_stride_pix, g_stride_c,
                 nullptr, occ, P, nrows, W, K, Nattr, acc};
  fb_launch_shade<false>(a, C, st);
  const long n_fin = (Nattr > P) ? Nattr : P;
  hipLaunchKernelGGL(fragment_bwd_finish_general_kernel, dim3((unsigned)((n_fin * 16 + 255) / 256)), dim3(256), 0, st, acc, 16, P, C,
                     Nattr, g_mus, g_isigmas, g_attr);
  return launch_status();
}

// ---- interpolate_attr (+ get_silhouette) on deferred-composite fragments: merge_final's backward, the weight sum's
// gradient, the composite and the trace backward in the one pass (the shade kernel with no background stage) ----
extern "C" int voge_fragment_merge_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode,
                                           const float *rays, const float *attr, const int32_t *idx, const int32_t *cnt,
                                           const float *weight, const float *act, const float *len, const float *dsd,
                                           const float *g_rgb, long g_stride_pix, long g_stride_c, const float *g_wsum,
                                           float occ, int B, int N, long nrows, int W, int K, int C, long Nattr,
                                           void *workspace, size_t workspace_bytes, float *g_verts, float *g_sigmas,
                                           float *g_attr, voge_stream_t stream) {
  if (B < 0 || N < 0 || nrows < 0 || W < 0 || K <= 0 || C <= 0 || C > 4 || Nattr < 0 || sigma_mode < 0 || sigma_mode > 2)
    return VOGE_ERR_BAD_ARG;
  if (K > 128) return VOGE_ERR_K_TOO_LARGE;
  const int P = B * N;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0 || nrows * W == 0) {
    if (g_attr && Nattr > 0) return (int)voge_fill_async(g_attr, 0, sizeof(float) * (size_t)Nattr * C, st);
    return 0;
  }
  if (!records || !rays || !attr || !idx || !cnt || !weight || !len || !g_rgb || !workspace) return VOGE_ERR_BAD_ARG;
  if ((act == nullptr) != (dsd == nullptr)) return VOGE_ERR_BAD_ARG;
  if ((g_verts == nullptr) != (g_sigmas == nullptr) || (sigma_mode == 2 && g_sigmas && !sigmas)) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < (size_t)P * 32) return VOGE_ERR_WORKSPACE;
  if (Nattr * C >= (1l << 30) || P >= (1 << 26)) return VOGE_ERR_BAD_ARG;
  float *acc = reinterpret_cast<float *>(workspace);
  { const hipError_t e = voge_fill_async(acc, 0, (size_t)P * 32, st); if (e != hipSuccess) return (int)e; }
  // (bg = NULL selects the merge form inside the kernel; its `wsum` operand carries g_wsum)
  const FbArgs a{reinterpret_cast<const float4 *>(records), rays, attr, idx, cnt, weight, act, len, dsd, nullptr, g_wsum, nullptr, -1.0f,
                 g_rgb, g_stride_pix, g_stride_c, nullptr, occ, P, nrows, W, K, Nattr, acc};
  fb_launch_shade<true>(a, C, st);
  const long n_fin = (Nattr > P) ? Nattr : P;
  hipLaunchKernelGGL(fragment_bwd_finish_kernel, dim3((unsigned)((n_fin + 255) / 256)), dim3(256), 0, st, acc, 8, sigmas, P, N, B,
                     IsoView{nullptr, shared ? 1 : 0, sigma_mode}, C, Nattr, g_verts, g_sigmas, g_attr);
  return launch_status();
}

// ---- the same pass driven by the gradient of the weights itself (any consumer) ----
extern "C" int voge_fragment_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode, const float *rays,
                                     const int32_t *idx, const int32_t *cnt, const float *weight, const float *act,
                                     const float *len, const float *dsd, const float *g_weight, long gw_stride_pix,
                                     long gw_stride_k, const float *g_hitlen, float occ, int B, int N, long nrows, int W, int K,
                                     void *workspace, size_t workspace_bytes, float *g_verts, float *g_sigmas,
                                     voge_stream_t stream) {
  if (B < 0 || N < 0 || nrows < 0 || W < 0 || K <= 0 || sigma_mode < 0 || sigma_mode > 2) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K) return VOGE_ERR_K_TOO_LARGE;
  const int P = B * N;
  hipStream_t st = (hipStream_t)stream;
  const long n_out = shared ? N : P;
  if (n_out == 0) return 0;      // no Gaussians: nothing to write
  if (!g_verts || !g_sigmas) return VOGE_ERR_BAD_ARG;
  if (P == 0 || nrows * W == 0) {
    hipError_t e0 = voge_fill_async(g_verts, 0, sizeof(float) * 3 * (size_t)n_out, st);
    if (e0 == hipSuccess) e0 = voge_fill_async(g_sigmas, 0, sizeof(float) * (size_t)n_out, st);
    return (int)e0;
  }
  if (!records || !rays || !idx || !cnt || !weight || !len || !workspace) return VOGE_ERR_BAD_ARG;
  if ((act == nullptr) != (dsd == nullptr) || (sigma_mode == 2 && !sigmas)) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < (size_t)P * 16) return VOGE_ERR_WORKSPACE;
  if (P >= (1 << 26)) return VOGE_ERR_BAD_ARG;      // 32-bit byte offsets of the record gathers
  float *acc = reinterpret_cast<float *>(workspace);
  { const hipError_t e = voge_fill_async(acc, 0, (size_t)P * 16, st); if (e != hipSuccess) return (int)e; }
  const FbArgs a{reinterpret_cast<const float4 *>(records), rays, nullptr, idx, cnt, weight, act, len, dsd, nullptr, nullptr, nullptr,
                 -1.0f, g_weight, gw_stride_pix, gw_stride_k, g_hitlen, occ, P, nrows, W, K, 0, acc};
  fb_launch_gw<true>(a, st);
  hipLaunchKernelGGL(fragment_bwd_finish_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, acc, 4, sigmas, P, N, B,
                     IsoView{nullptr, shared ? 1 : 0, sigma_mode}, 0, 0l, g_verts, g_sigmas, (float *)nullptr);
  return launch_status();
}

extern "C" int voge_fragment_bwd(const float *mus, const float *isigmas, const float *rays, const int32_t *idx,
                                 const int32_t *cnt, const float *weight, const float *act, const float *len,
                                 const float *dsd, const float *g_weight, long gw_stride_pix, long gw_stride_k,
                                 const float *g_hitlen, float occ, int P, long nrows, int W, int K, void *workspace,
                                 size_t workspace_bytes, float *g_mus, float *g_isigmas, voge_stream_t stream) {
  if (P < 0 || nrows < 0 || W < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (K > VOGE_MAX_K) return VOGE_ERR_K_TOO_LARGE;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0) return 0;      // no Gaussians: nothing to write
  if (!g_mus || !g_isigmas) return VOGE_ERR_BAD_ARG;
  if (nrows * W == 0) {
    hipError_t e0 = voge_fill_async(g_mus, 0, sizeof(float) * 3 * (size_t)P, st);
    if (e0 == hipSuccess) e0 = voge_fill_async(g_isigmas, 0, sizeof(float) * 9 * (size_t)P, st);
    return (int)e0;
  }
  if (!mus || !isigmas || !rays || !idx || !cnt || !weight || !len || !workspace) return VOGE_ERR_BAD_ARG;
  if ((act == nullptr) != (dsd == nullptr)) return VOGE_ERR_BAD_ARG;      // both or neither (neither: re-derived from (mu, A))
  if (workspace_bytes < voge_fragment_bwd_workspace_bytes(P)) return VOGE_ERR_WORKSPACE;
  if (P >= (1 << 26)) return VOGE_ERR_BAD_ARG;
  float4 *rec = reinterpret_cast<float4 *>(workspace);
  float *acc = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + (size_t)P * 48);
  hipLaunchKernelGGL(fragment_bwd_pack_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, mus, isigmas, P, rec,
                     reinterpret_cast<float4 *>(acc), 3);
  const FbArgs a{rec, rays, nullptr, idx, cnt, weight, act, len, dsd, nullptr, nullptr, nullptr, -1.0f, g_weight, gw_stride_pix,
                 gw_stride_k, g_hitlen, occ, P, nrows, W, K, 0, acc};
  fb_launch_gw<false>(a, st);
  hipLaunchKernelGGL(fragment_bwd_finish_general_kernel, dim3((unsigned)(((long)P * 16 + 255) / 256)), dim3(256), 0, st, acc, 12, P, 0,
                     0l, g_mus, g_isigmas, (float *)nullptr);
  return launch_status();
}
