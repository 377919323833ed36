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
constexpr int kFbRowsLds = 2 * 64 + 4 * kFbG;      // a round's padded rows: 64 lanes' slots + two sentinel pairs per pixel

struct FragBwdLds {
  WaveTable<kFbNE, 2> tab;        // key = Gaussian index; values = (g_mu, g_a), (w g_rgb, -)
  float len[kFbRowsLds], sp[kFbRowsLds], E[kFbRowsLds], u[kFbRowsLds];
};

#ifndef VOGE_FB_ABL
#define VOGE_FB_ABL 0
#endif
#ifndef VOGE_FB_WPE
#define VOGE_FB_WPE 4      // capping the registers for 5 / 6 waves per SIMD spills and is slower
#endif
// C: colour channels (1..4); OffT: uint32_t when every BYTE offset into the [pix][K] arrays fits 32 bits (the loads
// then take scalar base + 32-bit lane offset and the address arithmetic leaves the vector unit), else size_t.
template <int C, typename OffT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(VOGE_FB_WPE)))
fragment_bwd_iso_kernel(const float4 *__restrict__ rec /* [P] (centred mu, a) */, const float *__restrict__ rays,
                        const float *__restrict__ colors, const int32_t *__restrict__ idx, const int32_t *__restrict__ cnt,
                        const float *__restrict__ weight, const float *__restrict__ act, const float *__restrict__ len,
                        const float *__restrict__ dsd, const float *__restrict__ rgb, const float *__restrict__ wsum,
                        const float *__restrict__ bg, const float thr, const float *__restrict__ g_img, const long gs_pix,
                        const long gs_c, const float occ,
                        const int P, const long nrows, const int W, const int K, const long Nattr,
                        float *__restrict__ acc /* [P][8], zeroed */) {
  constexpr int NS = 2;
  __shared__ __attribute__((aligned(16))) FragBwdLds L;
  float *const Llen = L.len, *const Lsp = L.sp, *const LE = L.E, *const Lu = L.u;
  const int lane = threadIdx.x;
  const int blocks_x = (W + kFbGW - 1) / kFbGW;
  const long blk = blockIdx.x;
  const int x0 = (int)(blk % blocks_x) * kFbGW;
  const long y0 = (blk / blocks_x) * kFbGH;
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
    const bool on = pk.p >= 0;
    const int q = on ? lane - pk.s0 : 0, LP = on ? pk.np : 1;
    const int lead = __shfl(lead_g, on ? pk.p : 0, 64);
    const int k0 = NS * q;
    const OffT pix = on ? (OffT)((y0 + pk.p / kFbGW) * W + x0 + (pk.p & (kFbGW - 1))) : (OffT)0;
    const OffT fb = (pix * (OffT)K + (OffT)k0) * (OffT)4;      // byte offset of the lane's pair in the [pix][K] arrays
    const int r0 = NS * pk.s0 + 4 * pk.ord, RS = NS * LP + 4;
    const int d0 = on ? r0 + 2 + k0 : 2;
    // ---- the lane's two slots ----
    int id[NS];
    float wv[NS], lm[NS], sm[NS], em[NS];
    bool live[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) { id[a] = -1; wv[a] = 0.f; lm[a] = VOGE_SENT_LEN; sm[a] = 1e-5f; em[a] = 0.f; live[a] = on && (k0 + a < lead); }
    if (live[0]) {              // 8-byte accesses (K is even: the pair is aligned and inside the pixel's row)
      const int2 i2 = at_bytes<int2>(idx, fb);
      const v2f w2 = at_bytes<v2f>(weight, fb), a2 = at_bytes<v2f>(act, fb), l2 = at_bytes<v2f>(len, fb), d2 = at_bytes<v2f>(dsd, fb);
      id[0] = i2.x; wv[0] = w2.x; lm[0] = l2.x; em[0] = FAST_EXP(-a2.x); sm[0] = FAST_SQRT(d2.x + 1e-10f);
      if (live[1]) { id[1] = i2.y; wv[1] = w2.y; lm[1] = l2.y; em[1] = FAST_EXP(-a2.y); sm[1] = FAST_SQRT(d2.y + 1e-10f); }
    }
    // gathers: the slots' colours and (mu, a); the pixel's ray, upstream gradient and forward sums
    float col[NS][4];
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      const bool ok = live[a] && id[a] >= 0 && id[a] < P;
      live[a] = ok;
      col[a][0] = col[a][1] = col[a][2] = col[a][3] = 0.0f;
      if (ok && id[a] < Nattr && !(VOGE_FB_ABL & 4)) {
        const uint32_t o = (uint32_t)id[a] * (uint32_t)(4 * C);      // (bytes; Nattr * C < 2^30: host)
        if (C == 3) {
          const float3 v = at_bytes<float3>(colors, o);
          col[a][0] = v.x; col[a][1] = v.y; col[a][2] = v.z;
        } else if (C == 4) {
          const float4 v = at_bytes<float4>(colors, o);
          col[a][0] = v.x; col[a][1] = v.y; col[a][2] = v.z; col[a][3] = v.w;
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) col[a][c] = at_bytes<float>(colors, o + 4u * c);
        }
      }
    }
    float gr[4] = {0.f, 0.f, 0.f, 0.f}, g_sum_w = 0.0f, dx = 0.f, dy = 0.f, dz = 0.f;
    if (on) {
      const OffT pb = pix * (OffT)4;
      const float ws = at_bytes<float>(wsum, pb);
      float sil = fminf(ws, 1.0f);
      const float pass_s = (thr > 0.0f) ? 0.0f : (ws < 1.0f ? 1.0f : (ws == 1.0f ? 0.5f : 0.0f));
      if (thr > 0.0f) sil = sil > thr ? 1.0f : 0.0f;
      float g_mask = 0.0f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float xc = fmaf(1.0f - sil, bg[c], at_bytes<float>(rgb, pb * (OffT)C + (OffT)(4 * c)));
        gr[c] = g_img[(long)pix * gs_pix + c * gs_c] * (xc < 1.0f ? 1.0f : (xc == 1.0f ? 0.5f : 0.0f));     // min(x, 1) passes like torch.min
        g_mask = fmaf(-gr[c], bg[c], g_mask);
      }
      g_sum_w = g_mask * pass_s;
      const float3 dv = at_bytes<float3>(rays, pb * (OffT)3);
      dx = dv.x; dy = dv.y; dz = dv.z;
    }
    // ---- shade backward: g_w of the slots; u = g_w w ----
    float um[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      const float gw = live[a] ? fmaf(gr[3], col[a][3], fmaf(gr[2], col[a][2], fmaf(gr[1], col[a][1], fmaf(gr[0], col[a][0], g_sum_w)))) : 0.0f;
      um[a] = gw * wv[a];
    }
    // ---- composite backward (closed form, composite_core.h) ----
    if (on) {
      *reinterpret_cast<v2f *>(Llen + d0) = (v2f){lm[0], lm[1]};
      *reinterpret_cast<v2f *>(Lsp + d0) = (v2f){sm[0] * kCs, sm[1] * kCs};
      *reinterpret_cast<v2f *>(LE + d0) = (v2f){em[0] * (sm[0] * kCs), em[1] * (sm[1] * kCs)};
      *reinterpret_cast<v2f *>(Lu + d0) = splat(0.0f);
      if (q < 2) {      // the sentinel pair in front of the pixel's row and the one behind it
        for (int t2 = q; t2 < 2; t2 += LP) {
          Llen[r0 + t2] = -kBig; Lsp[r0 + t2] = 1.0f; LE[r0 + t2] = 0.0f; Lu[r0 + t2] = 0.0f;
          const int eb = r0 + RS - 2 + t2;
          Llen[eb] = kBig; Lsp[eb] = 1.0f; LE[eb] = 0.0f; Lu[eb] = 0.0f;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    float ga[NS], gl[NS], gd[NS];
#if VOGE_FB_ABL & 2       // (timing experiment: no composite)
    for (int a = 0; a < NS; ++a) { ga[a] = um[a]; gl[a] = um[a] * sm[a]; gd[a] = um[a] * em[a]; }
#else
    compn_bwd_wave<NS>(lm, sm, em, um, Llen, Lsp, LE, Lu, d0, k0, NS * LP, q, LP, npm, on, on, true, pk.s0, occ, ga, gl, gd);
#endif
    __builtin_amdgcn_wave_barrier();      // the rows are rewritten by the next round
    // ---- trace backward terms (isotropic: trace_bwd.hip) + the colour term, one table entry per Gaussian.  (The
    // (mu, a) records are gathered only now: held across the composite they cost the kernel a wave per SIMD.) ----
    float4 rc[NS];
#pragma unroll
    for (int a = 0; a < NS; ++a)
      rc[a] = (live[a] && !(VOGE_FB_ABL & 8)) ? at_bytes<float4>(rec, (uint32_t)id[a] * 16u) : make_float4(0.f, 0.f, 0.f, 0.f);   // (P < 2^28: host)
    const float dn2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float idn = __builtin_amdgcn_rcpf(dn2);
#pragma unroll
    for (int a = 0; a < NS; ++a) {
      float4 val[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
      bool go = live[a];
      if (go) {
        const float mx = rc[a].x, my = rc[a].y, mz = rc[a].z, aa = rc[a].w;
        const float t = fmaf(mz, dz, fmaf(my, dy, mx * dx)) * idn;
        const float vx = fmaf(-t, dx, mx), vy = fmaf(-t, dy, my), vz = fmaf(-t, dz, mz);
        const float c1 = gl[a] * idn, c2 = 2.0f * aa * ga[a];
        val[0] = make_float4(fmaf(c1, dx, c2 * vx), fmaf(c1, dy, c2 * vy), fmaf(c1, dz, c2 * vz),
                             fmaf(ga[a], fmaf(vz, vz, fmaf(vy, vy, vx * vx)), gd[a] * dn2));
        val[1] = make_float4(wv[a] * gr[0], wv[a] * gr[1], wv[a] * gr[2], wv[a] * gr[3]);
        go = (val[0].x != 0.f || val[0].y != 0.f || val[0].z != 0.f || val[0].w != 0.f || val[1].x != 0.f || val[1].y != 0.f ||
              val[1].z != 0.f || val[1].w != 0.f);
      }
      if (!__any(go)) continue;     // uniform
#if VOGE_FB_ABL & 1       // (timing experiment: no table, nothing accumulated)
      if (go && val[0].x == 1.2345f && val[1].y == 3.21f) acc[id[a]] = val[0].w + val[1].x;
      continue;
#endif
      const int slot = wt_find(L.tab, id[a], go);
      wt_add(L.tab, slot, val, go && slot >= 0, lane);
      if (go && slot < 0) {         // table full: rare, straight to memory
        const float o[8] = {val[0].x, val[0].y, val[0].z, val[0].w, val[1].x, val[1].y, val[1].z, val[1].w};
#pragma unroll
        for (int c = 0; c < 8; ++c) unsafeAtomicAdd(acc + 8 * (size_t)id[a] + c, o[c]);
      }
    }
  }
  {   // flush: 8 adjacent lanes per entry -> the 32 bytes of acc[p]: lane-coalesced atomics
    const int c = lane & 7;
    const float *vals = reinterpret_cast<const float *>(L.tab.vals);
    const int n = wt_compact(L.tab, lane);
    const volatile int *list = L.tab.owner;
    for (int i = lane >> 3; i < n; i += 8) {
      const int s = list[i];
      unsafeAtomicAdd(acc + 8 * (size_t)L.tab.keys[s] + c, vals[s * 8 + c]);
    }
  }
}

// acc [P][8] -> the gradients of what the caller passed in: verts / sigmas through the view's chain rule (and the sum
// over the batch when one Gaussian set is shared by all views), colours [Nattr][C].
__global__ void __launch_bounds__(256)
fragment_bwd_finish_kernel(const float *__restrict__ acc, const float *__restrict__ a_in, const int P, const int N, const int B,
                           const IsoView view, const int C, const long Nattr, float *__restrict__ g_mus,
                           float *__restrict__ g_a, float *__restrict__ g_colors) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g_colors != nullptr && g < Nattr) {
    for (int c = 0; c < C; ++c) g_colors[(size_t)g * C + c] = (g < P) ? acc[8 * (size_t)g + 4 + c] : 0.0f;
  }
  const int n_out = view.shared ? N : P;
  if (g >= n_out || g_mus == nullptr) return;
  float4 v = *reinterpret_cast<const float4 *>(acc + 8 * (size_t)g);
  if (view.shared)
    for (int b = 1; b < B; ++b) {
      const float4 w = *reinterpret_cast<const float4 *>(acc + 8 * ((size_t)b * N + g));
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

extern "C" size_t voge_fragment_bwd_workspace_bytes(int P) { return P <= 0 ? 0 : (size_t)P * 32; }

extern "C" int voge_fragment_shade_bwd_iso(const float *records, const float *sigmas, int shared, int sigma_mode,
                                           const float *rays, const float *colors, const int32_t *idx, const int32_t *cnt,
                                           const float *weight, const float *act, const float *len, const float *dsd,
                                           const float *rgb, const float *wsum, const float *bg, float thr,
                                           const float *g_img, long g_stride_pix, long g_stride_c, float occ, int B, int N,
                                           long nrows, int W, int K, int C, long Nattr, void *workspace, size_t workspace_bytes, float *g_verts,
                                           float *g_sigmas, float *g_colors, voge_stream_t stream) {
  if (B < 0 || N < 0 || nrows < 0 || W < 0 || K <= 0 || C <= 0 || C > 4 || Nattr < 0 || sigma_mode < 0 || sigma_mode > 2)
    return VOGE_ERR_BAD_ARG;
  if ((K & 1) != 0 || K > 128) return VOGE_ERR_K_TOO_LARGE;      // a lane owns an aligned pair of slots; a pixel fits a wave
  const int P = B * N;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0 || nrows * W == 0) {
    if (g_colors && Nattr > 0) return (int)hipMemsetAsync(g_colors, 0, sizeof(float) * (size_t)Nattr * C, st);
    return 0;
  }
  if (!records || !rays || !colors || !idx || !cnt || !weight || !act || !len || !dsd || !rgb || !wsum || !bg || !g_img || !workspace)
    return VOGE_ERR_BAD_ARG;
  if ((g_verts == nullptr) != (g_sigmas == nullptr) || (sigma_mode == 2 && g_sigmas && !sigmas)) return VOGE_ERR_BAD_ARG;
  if (workspace_bytes < voge_fragment_bwd_workspace_bytes(P)) return VOGE_ERR_WORKSPACE;
  float *acc = reinterpret_cast<float *>(workspace);
  hipError_t e = hipMemsetAsync(acc, 0, (size_t)P * 32, st);
  if (e != hipSuccess) return (int)e;
  const long blocks = (long)((W + kFbGW - 1) / kFbGW) * ((nrows + kFbGH - 1) / kFbGH);
  if (Nattr * C >= (1l << 30) || P >= (1 << 28)) return VOGE_ERR_BAD_ARG;      // 32-bit byte offsets of the gathers
  const bool small = (double)nrows * W * K < (double)(1l << 30);
#define VOGE_LAUNCH_FB(CC, OT)                                                                                            \
  hipLaunchKernelGGL((fragment_bwd_iso_kernel<CC, OT>), dim3((unsigned)blocks), dim3(64), 0, st,                          \
                     reinterpret_cast<const float4 *>(records), rays, colors, idx, cnt, weight, act, len, dsd, rgb, wsum, bg, thr, \
                     g_img, g_stride_pix, g_stride_c, occ, P, nrows, W, K, Nattr, acc)
#define VOGE_LAUNCH_FB_C(CC) do { if (small) VOGE_LAUNCH_FB(CC, uint32_t); else VOGE_LAUNCH_FB(CC, size_t); } while (0)
  switch (C) {
    case 1: VOGE_LAUNCH_FB_C(1); break;
    case 2: VOGE_LAUNCH_FB_C(2); break;
    case 3: VOGE_LAUNCH_FB_C(3); break;
    default: VOGE_LAUNCH_FB_C(4); break;
  }
#undef VOGE_LAUNCH_FB_C
#undef VOGE_LAUNCH_FB
  const long n_fin = (Nattr > P) ? Nattr : P;
  hipLaunchKernelGGL(fragment_bwd_finish_kernel, dim3((unsigned)((n_fin + 255) / 256)), dim3(256), 0, st, acc, sigmas, P, N, B,
                     IsoView{nullptr, shared ? 1 : 0, sigma_mode}, C, Nattr, g_verts, g_sigmas, g_colors);
  return launch_status();
}
