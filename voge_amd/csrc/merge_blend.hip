// Attribute merge, background blend, and their fusion ("shade") for gfx950.
//
// Reference behaviour being reproduced: merge_final (VoGE/Aggregation.py:111-141, reached via
// interpolate_attr, VoGE/Renderer.py:153) and get_silhouette / to_colored_background
// (VoGE/Renderer.py:157-171).  The reference gathers a [B,H,W,K,C] temporary and runs ~10
// elementwise kernels; here ONE LANE OWNS ONE (pixel, slot): the per-slot arrays are read as
// coalesced runs, the per-pixel sums are segmented wave reductions, and merge + silhouette +
// blend happen in one pass (shade_fwd) / one backward pass (shade_bwd).
#include "voge_common.h"
#include <algorithm>

namespace voge {

// min(x, 1) passes the gradient where x < 1 and half of it at the tie, like torch.min.
__device__ __forceinline__ float clamp1_pass(float x) { return x < 1.0f ? 1.0f : (x == 1.0f ? 0.5f : 0.0f); }

constexpr int kRun = 8;      // pixels per wave run: their kRun*K slots are one contiguous stream
constexpr int kShadeU = 4;   // 64-slot batches whose loads are issued together

// Lane layout of the simple per-pixel kernels below: lpp = min(K, 64) lanes per pixel, ppi = 64 / lpp
// pixels per wave instruction, `sub` = which of those pixels, `kl` = slot within the chunk.
constexpr int kShadePixPerWave = 16;
struct SlotLanes {
  int lpp, ppi, sub, kl;
};
__device__ __forceinline__ SlotLanes slot_lanes(const int K, const int lane) {
  SlotLanes s;
  s.lpp = min(K, 64);
  s.ppi = 64 / s.lpp;
  s.sub = lane / s.lpp;
  s.kl = lane - s.sub * s.lpp;
  return s;
}
// broadcast the value held by the first lane of each lpp-lane segment to the whole segment
__device__ __forceinline__ float seg_bcast(const float x, const int lane, const int kl) {
  return __shfl(x, lane - kl, 64);
}

// ------------------------------------------------------------------------------------------
// forward: rgb = sum_{k < valid} attr[idx_k] w_k ; sil = min(sum_k w_k, 1) ;
//          img = min(rgb + (1 - mask(sil)) bg, 1).   Any of out_rgb / out_img / out_sil may be NULL;
//          attr == NULL skips the merge (silhouette / blend of a given rgb_in).
// A wave streams the kRun*K slots of kRun consecutive pixels, one lane per slot; per-pixel sums
// are segmented wave reductions accumulated in a few LDS words.
// ------------------------------------------------------------------------------------------
struct ShadeFwdLds {
  float acc[kRun][5];   // up to 4 channels + sum of weights
};

__global__ void __launch_bounds__(256)
shade_fwd_kernel(const float *__restrict__ attr, int32_t *__restrict__ idx, const float *__restrict__ weight,
                 const int64_t *__restrict__ valid_num, const float *__restrict__ rgb_in,
                 const float *__restrict__ bg, const float thr, const long npix, const int K, const int C,
                 const long Nattr, const int fix_idx, float *__restrict__ out_rgb, float *__restrict__ out_img,
                 float *__restrict__ out_sil, float *__restrict__ out_wsum) {
  __shared__ ShadeFwdLds Ls[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  ShadeFwdLds &L = Ls[wave];
  const long run = (long)blockIdx.x * 4 + wave;
  const long pix0 = run * kRun;
  if (pix0 >= npix) return;  // waves are independent
  const int npx = (int)min((long)kRun, npix - pix0);
  const int n_items = npx * K;
  const int nit = (n_items + 63) >> 6;
  const long base = pix0 * K;
  for (int c0 = 0; c0 < max(C, 1); c0 += 4) {
    const int nc = max(0, min(4, C - c0));
    if (lane < kRun * 5) (&L.acc[0][0])[lane] = 0.0f;
    for (int it0 = 0; it0 < nit; it0 += kShadeU) {
      float w[kShadeU];
      int p[kShadeU], lx[kShadeU];
      bool take[kShadeU];
#pragma unroll
      for (int u = 0; u < kShadeU; ++u) {
        const int j = (it0 + u) * 64 + lane;
        const bool ok = (it0 + u < nit) && (j < n_items);
        lx[u] = ok ? j / K : 64 + lane;
        w[u] = ok ? weight[base + j] : 0.0f;
        p[u] = -1;
        take[u] = false;
        if (ok && attr != nullptr) {
          const int raw = idx[base + j];
          p[u] = raw + (raw < 0);
          if (fix_idx && c0 == 0 && raw < 0) idx[base + j] = raw + 1;   // Aggregation.py:131
          const int k = j - lx[u] * K;
          const int64_t nv = valid_num[pix0 + lx[u]];
          take[u] = (k < nv) && (p[u] >= 0) && (p[u] < Nattr) && (w[u] != 0.0f);
        }
      }
      float a[kShadeU][4];
#pragma unroll
      for (int u = 0; u < kShadeU; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) a[u][c] = (take[u] && c < nc) ? attr[(size_t)p[u] * C + c0 + c] * w[u] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kShadeU; ++u) {
        if (it0 + u >= nit) break;  // uniform
        const int prev = __shfl_up(lx[u], 1, 64);
        const bool head = (lane == 0 || prev != lx[u]) && lx[u] < kRun;
        for (int c = 0; c < nc; ++c) {
          const float v = seg_sum_key(a[u][c], lx[u], lane);
          if (head) L.acc[lx[u]][c] += v;
        }
        if (c0 == 0) {
          const float v = seg_sum_key(w[u], lx[u], lane);
          if (head) L.acc[lx[u]][4] += v;
        }
      }
    }
    // finalise the run: lane = (pixel, channel)
    if (lane < npx * 4) {
      const int x = lane >> 2, c = lane & 3;
      const long pix = pix0 + x;
      if (c0 == 0 && c == 0 && out_sil != nullptr) out_sil[pix] = fminf(L.acc[x][4], 1.0f);
      if (c0 == 0 && c == 0 && out_wsum != nullptr) out_wsum[pix] = L.acc[x][4];
      if (c < nc) {
        float sil = fminf(L.acc[x][4], 1.0f);
        if (thr > 0.0f) sil = sil > thr ? 1.0f : 0.0f;
        const float v = (attr != nullptr) ? L.acc[x][c] : rgb_in[pix * C + c0 + c];
        if (out_rgb != nullptr) out_rgb[pix * C + c0 + c] = v;
        if (out_img != nullptr) out_img[pix * C + c0 + c] = fminf(fmaf(1.0f - sil, bg[c0 + c], v), 1.0f);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward, K % 4 == 0 and K <= 64 (the renderer's case).  A lane owns FOUR consecutive slots of
// one pixel (16-byte loads of idx and weight), a pixel owns an aligned row of 16 lanes, so the
// per-pixel sums are DPP row reductions: no LDS, no shuffles, a quarter of the instructions of
// the lane-per-slot kernel above.  A wave covers 4 * kSh4P pixels.
// ------------------------------------------------------------------------------------------
#ifndef VOGE_SH4_P
#define VOGE_SH4_P 1
#endif
constexpr int kSh4P = VOGE_SH4_P;

__global__ void __launch_bounds__(256)
shade_fwd4_kernel(const float *__restrict__ attr, int32_t *__restrict__ idx, const float *__restrict__ weight,
                  const int64_t *__restrict__ valid_num, const float *__restrict__ bg, const float thr,
                  const long npix, const int K, const int C, const long Nattr, const int fix_idx,
                  float *__restrict__ out_rgb, float *__restrict__ out_img, float *__restrict__ out_sil,
                  float *__restrict__ out_wsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane >> 4, q = lane & 15;
  const bool lane_on = q < (K >> 2);
  const long pixbase = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * (4 * kSh4P);
  if (pixbase >= npix) return;
  for (int c0 = 0; c0 < C; c0 += 4) {
    const int nc = min(4, C - c0);
    float4 w[kSh4P];
    int4 id[kSh4P];
    int vn[kSh4P];
    bool ok[kSh4P];
#pragma unroll
    for (int u = 0; u < kSh4P; ++u) {
      const long pix = pixbase + u * 4 + row;
      ok[u] = lane_on && pix < npix;
      w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      id[u] = make_int4(-1, -1, -1, -1);
      vn[u] = 0;
      if (ok[u]) {
        const long o = pix * K + 4 * q;
        w[u] = *reinterpret_cast<const float4 *>(weight + o);
        id[u] = *reinterpret_cast<const int4 *>(idx + o);
        vn[u] = (int)min((int64_t)K, valid_num[pix]);
      }
    }
    float acc[kSh4P][4];
    float a[kSh4P][4][4];
    bool take[kSh4P][4];
    int pp[kSh4P][4];
#pragma unroll
    for (int u = 0; u < kSh4P; ++u) {
      const int raw[4] = {id[u].x, id[u].y, id[u].z, id[u].w};
      const float wv[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        pp[u][s] = raw[s] + (raw[s] < 0);
        take[u][s] = ok[u] && (4 * q + s < vn[u]) && (pp[u][s] >= 0) && (pp[u][s] < Nattr) && (wv[s] != 0.0f);
        if (C == 3) {   // RGB: the three channels of a Gaussian as ONE 12-byte gather
          float3 v = make_float3(0.f, 0.f, 0.f);
          if (take[u][s]) v = *reinterpret_cast<const float3 *>(attr + (size_t)pp[u][s] * 3);
          a[u][s][0] = v.x; a[u][s][1] = v.y; a[u][s][2] = v.z; a[u][s][3] = 0.0f;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) a[u][s][c] = (take[u][s] && c < nc) ? attr[(size_t)pp[u][s] * C + c0 + c] : 0.0f;
        }
      }
      if (fix_idx && c0 == 0 && ok[u] && (raw[0] < 0 || raw[1] < 0 || raw[2] < 0 || raw[3] < 0))   // Aggregation.py:131
        *reinterpret_cast<int4 *>(idx + (pixbase + u * 4 + row) * K + 4 * q) =
            make_int4(pp[u][0], pp[u][1], pp[u][2], pp[u][3]);
    }
#pragma unroll
    for (int u = 0; u < kSh4P; ++u) {
      const float wv[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float x = a[u][0][c] * wv[0];
        x = fmaf(a[u][1][c], wv[1], x);
        x = fmaf(a[u][2][c], wv[2], x);
        x = fmaf(a[u][3][c], wv[3], x);
        acc[u][c] = row16_sum(x);
      }
      const float ws = row16_sum((wv[0] + wv[1]) + (wv[2] + wv[3]));
      const long pix = pixbase + u * 4 + row;
      if (pix < npix && q < 4) {   // lane q of the row finalises channel q
        const float v = q == 0 ? acc[u][0] : q == 1 ? acc[u][1] : q == 2 ? acc[u][2] : acc[u][3];
        if (c0 == 0 && q == 0) {
          if (out_sil != nullptr) out_sil[pix] = fminf(ws, 1.0f);
          if (out_wsum != nullptr) out_wsum[pix] = ws;
        }
        if (q < nc) {
          float sil = fminf(ws, 1.0f);
          if (thr > 0.0f) sil = sil > thr ? 1.0f : 0.0f;
          if (out_rgb != nullptr) out_rgb[pix * C + c0 + q] = v;
          if (out_img != nullptr) out_img[pix * C + c0 + q] = fminf(fmaf(1.0f - sil, bg[c0 + q], v), 1.0f);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward for C <= 4.  A wave owns an 8x8 pixel tile (locality for the attribute-gradient
// table) and streams it row by row, one lane per slot.
//   blend part (bg != NULL): g_rgb = g_up * [x < 1], x = rgb + (1 - mask) bg ;
//                            g_sum_w = -(sum_c g_rgb bg) * [sum w < 1]   (0 when thr > 0)
//   merge part             : g_w[k] = [k < valid] <g_rgb, attr[idx_k]> + g_sum_w ;
//                            g_attr[idx_k] += [k < valid] w_k g_rgb   (wave-private LDS table,
//                            flushed with lane-coalesced atomics)
// bg == NULL: g_up is the gradient of the merged attributes themselves (interpolate_attr).
// ------------------------------------------------------------------------------------------
#ifndef VOGE_SHADE_WAVES
#define VOGE_SHADE_WAVES 1
#endif
constexpr int kShadeWaves = VOGE_SHADE_WAVES;
#ifndef VOGE_SHADE_TH
#define VOGE_SHADE_TH 2
#endif
#ifndef VOGE_SHADE_NE
#define VOGE_SHADE_NE 128
#endif
constexpr int kShadeNE = VOGE_SHADE_NE;
constexpr int kShadeTH = VOGE_SHADE_TH;   // a wave owns an 8 x kShadeTH pixel tile

#ifndef VOGE_SHADE_BU
#define VOGE_SHADE_BU 5
#endif
constexpr int kShadeBU = VOGE_SHADE_BU;   // 64-slot batches whose loads are issued together (backward)
constexpr int kShadePix = 8 * kShadeTH;

struct ShadeBwdLds {
  WaveTable<kShadeNE, 1> tab;
  float wsum[kShadePix];
  float gr[kShadePix][4];
  float gsw[kShadePix];
  int vn[kShadePix];
};

__global__ void __launch_bounds__(64 * kShadeWaves)
shade_bwd_tile_kernel(const float *__restrict__ attr, const int32_t *__restrict__ idx,
                      const float *__restrict__ weight, const int64_t *__restrict__ valid_num,
                      const float *__restrict__ rgb, const float *__restrict__ wsum_in,
                      const float *__restrict__ bg, const float thr, const float *__restrict__ g_up,
                      const long nrows, const int W, const int K, const int C, const long Nattr,
                      float *__restrict__ g_attr, float *__restrict__ g_weight) {
  __shared__ ShadeBwdLds Ls[kShadeWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  ShadeBwdLds &L = Ls[wave];
  const int tiles_x = (W + 7) / 8;
  const long ntiles = (long)tiles_x * ((nrows + kShadeTH - 1) / kShadeTH);
  const long tile = (long)blockIdx.x * kShadeWaves + wave;
  if (tile >= ntiles) return;  // waves are independent
  const int x0 = (int)(tile % tiles_x) * 8;
  const long y0 = (tile / tiles_x) * kShadeTH;
  const int tw = min(8, W - x0);
  const int th = (int)min((long)kShadeTH, nrows - y0);
  const int n_items = tw * K;              // slots of one 8-pixel row: one contiguous run
  const int nit = (n_items + 63) >> 6;
  const int nb = th * nit;                 // 64-slot batches of the tile
  const float invK = 1.0f / (float)K;
  wt_clear(L.tab, lane);
  // ---- per-pixel silhouette sum (only when the caller did not keep the forward's) ----
  if (bg != nullptr && wsum_in == nullptr) {
    for (int i = lane; i < kShadePix; i += 64) L.wsum[i] = 0.0f;
    for (int bb = 0; bb < nb; ++bb) {
      const int r = bb / nit, it = bb - r * nit;
      const int j = it * 64 + lane;
      const bool ok = j < n_items;
      const int pl = ok ? r * 8 + __float2int_rz(((float)j + 0.5f) * invK) : 64 + lane;
      const float v = seg_sum_key(ok ? weight[((y0 + r) * W + x0) * (long)K + j] : 0.0f, pl, lane);
      const int prev = __shfl_up(pl, 1, 64);
      if ((lane == 0 || prev != pl) && pl < kShadePix) L.wsum[pl] += v;
    }
  }
  // ---- per-pixel upstream gradient -> LDS ----
  for (int i = lane; i < kShadePix; i += 64) {
    const int lx = i & 7, r = i >> 3;
    float g_sum_w = 0.0f;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    int vn = 0;
    if (lx < tw && r < th) {
      const long pix = (y0 + r) * W + x0 + lx;
      vn = (int)min((int64_t)K, valid_num[pix]);
      if (bg != nullptr) {
        const float ws = (wsum_in != nullptr) ? wsum_in[pix] : L.wsum[i];
        float sil = fminf(ws, 1.0f);
        const float pass_s = (thr > 0.0f) ? 0.0f : clamp1_pass(ws);
        if (thr > 0.0f) sil = sil > thr ? 1.0f : 0.0f;
        float g_mask = 0.0f;
        for (int c = 0; c < C; ++c) {
          const float x = fmaf(1.0f - sil, bg[c], rgb[pix * C + c]);
          g[c] = g_up[pix * C + c] * clamp1_pass(x);
          g_mask = fmaf(-g[c], bg[c], g_mask);
        }
        g_sum_w = g_mask * pass_s;
      } else {
        for (int c = 0; c < C; ++c) g[c] = g_up[pix * C + c];
      }
    }
    for (int c = 0; c < 4; ++c) L.gr[i][c] = g[c];
    L.gsw[i] = g_sum_w;
    L.vn[i] = vn;
  }
  // ---- slots of the tile: g_weight and the attribute-gradient table ----
  for (int g0 = 0; g0 < nb; g0 += kShadeBU) {
    float w[kShadeBU];
    int p[kShadeBU], pl[kShadeBU];
    long flat[kShadeBU];
    bool ok[kShadeBU];
#pragma unroll
    for (int u = 0; u < kShadeBU; ++u) {
      const int bb = g0 + u;
      const int r = bb / nit, it = bb - r * nit;
      const int j = it * 64 + lane;
      ok[u] = (bb < nb) && (j < n_items);
      const int lx = __float2int_rz(((float)j + 0.5f) * invK);
      pl[u] = ok[u] ? r * 8 + lx : 0;
      flat[u] = ((y0 + r) * W + x0) * (long)K + j;
      w[u] = 0.0f;
      p[u] = -1;
      if (ok[u] && (j - lx * K) < L.vn[pl[u]]) {
        const int raw = idx[flat[u]];
        const int q = raw + (raw < 0);
        if (q >= 0 && q < Nattr) { p[u] = q; w[u] = weight[flat[u]]; }
      }
    }
    float a[kShadeBU][4];
    if (C == 3 && attr != nullptr) {   // RGB: one 12-byte gather per slot instead of three 4-byte ones
#pragma unroll
      for (int u = 0; u < kShadeBU; ++u) {
        float3 v = make_float3(0.f, 0.f, 0.f);
        if (p[u] >= 0) v = *reinterpret_cast<const float3 *>(attr + (size_t)p[u] * 3);
        a[u][0] = v.x; a[u][1] = v.y; a[u][2] = v.z; a[u][3] = 0.0f;
      }
    } else {
#pragma unroll
      for (int u = 0; u < kShadeBU; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[u][c] = (attr != nullptr && p[u] >= 0 && c < C) ? attr[(size_t)p[u] * C + c] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kShadeBU; ++u) {
      if (g0 + u >= nb) break;  // uniform
      const float g0c = L.gr[pl[u]][0], g1 = L.gr[pl[u]][1], g2 = L.gr[pl[u]][2], g3 = L.gr[pl[u]][3];
      if (ok[u] && g_weight != nullptr)
        g_weight[flat[u]] = fmaf(g3, a[u][3], fmaf(g2, a[u][2], fmaf(g1, a[u][1], fmaf(g0c, a[u][0], L.gsw[pl[u]]))));
      if (g_attr != nullptr) {
        const float4 v[1] = {make_float4(w[u] * g0c, w[u] * g1, w[u] * g2, w[u] * g3)};
        const bool live = (p[u] >= 0) && (v[0].x != 0.0f || v[0].y != 0.0f || v[0].z != 0.0f || v[0].w != 0.0f);
        if (!__any(live)) continue;  // uniform
        const int slot = wt_find(L.tab, p[u], live);
        wt_add(L.tab, slot, v, live && slot >= 0, lane);
        if (live && slot < 0) {
          const float o[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
          for (int c = 0; c < C; ++c) unsafeAtomicAdd(g_attr + (size_t)p[u] * C + c, o[c]);
        }
      }
    }
  }
  if (g_attr != nullptr) {
    // 4 adjacent lanes per table entry -> adjacent floats of g_attr[p]: lane-coalesced atomics
    const int c = lane & 3;
    const float *vals = reinterpret_cast<const float *>(L.tab.vals);
    const int n = wt_compact(L.tab, lane);
    const lds_vint *list = lds_volatile(L.tab.owner);
    for (int i = lane >> 2; i < n; i += 16) {
      const int s = list[i];
      const int p = L.tab.keys[s];
      if (c < C) unsafeAtomicAdd(g_attr + (size_t)p * C + c, vals[s * 4 + c]);
    }
  }
}

// merge backward for C > 4 (feature maps): one (pixel, slot) item per wave iteration, lanes over
// channels: g_up / attr rows are read coalesced and the scatter is a run of C adjacent atomics.
__global__ void __launch_bounds__(256)
merge_bwd_chan_kernel(const float *__restrict__ attr, const int32_t *__restrict__ idx,
                      const float *__restrict__ weight, const int64_t *__restrict__ valid_num,
                      const float *__restrict__ g_up, const long npix, const int K, const int C,
                      const long Nattr, float *__restrict__ g_attr, float *__restrict__ g_weight) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  for (long pix = wave_id; pix < npix; pix += nwaves) {
    const int nv = (int)min((int64_t)K, max((int64_t)0, valid_num[pix]));
    for (int k = 0; k < K; ++k) {
      const long f = pix * K + k;
      float gw = 0.0f;
      if (k < nv) {
        const int raw = idx[f];
        const int p = raw + (raw < 0);
        if (p >= 0 && p < Nattr) {
          const float w = weight[f];
          for (int c = lane; c < C; c += 64) {
            const float g = g_up[pix * C + c];
            if (attr != nullptr) gw = fmaf(g, attr[(size_t)p * C + c], gw);
            if (g_attr != nullptr && w != 0.0f && g != 0.0f) unsafeAtomicAdd(g_attr + (size_t)p * C + c, w * g);
          }
          gw = wave_sum(gw);
        }
      }
      if (g_weight != nullptr && lane == 0) g_weight[f] = gw;
    }
  }
}

// ------------------------------------------------------------------------------------------
// merge alone (interpolate_attr) for what the 4-slot kernel above does not take -- more than four channels (one-hot label
// maps, feature maps up to 64 channels), K not a multiple of four or above 64: ONE LANE PER (pixel, channel).  A wave stages
// the contiguous (idx, weight) slots of its `ppw` pixels in LDS with coalesced loads (validity folded into the staged
// weight: slot >= valid_num, index out of range or weight 0 -> weight 0, index 0), then every lane walks its pixel's K
// slots: an LDS broadcast per slot, one gather of its own channel (the C lanes of a pixel read C adjacent floats), one FMA --
// no cross-lane sum at all.  shade_fwd_kernel spends ~15 shuffles per 64 slots and channel on segmented sums and re-reads
// the slots once per four channels: 256^2, K = 102, C = 6 (demo/EfficientCuboidViaOptimization.py): 99.5 us -> this kernel.
// ------------------------------------------------------------------------------------------
constexpr int kMpcSlots = 1024;      // staged slots per wave
__global__ void __launch_bounds__(256)
merge_fwd_pc_kernel(const float *__restrict__ attr, int32_t *__restrict__ idx, const float *__restrict__ weight,
                    const int64_t *__restrict__ valid_num, const long npix, const int K, const int C, const long Nattr,
                    const int fix_idx, const int ppw, float *__restrict__ out) {
  __shared__ int Li[4][kMpcSlots];
  __shared__ float Lw[4][kMpcSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long pix0 = ((long)blockIdx.x * 4 + wave) * ppw;
  if (pix0 >= npix) return;      // waves are independent
  const int npx = (int)min((long)ppw, npix - pix0);
  const int n_items = npx * K;
  const long base = pix0 * K;
  const float rK = 1.0f / (float)K;
  for (int i = lane; i < n_items; i += 64) {
    int lx = __float2int_rz(((float)i + 0.5f) * rK);      // i / K, corrected below (exact for any size)
    int k = i - lx * K;
    if (k < 0) { --lx; k += K; } else if (k >= K) { ++lx; k -= K; }
    const int raw = idx[base + i];
    const float w = weight[base + i];
    const int p = raw + (raw < 0);
    if (fix_idx && raw < 0) idx[base + i] = raw + 1;      // Aggregation.py:131
    const bool ok = (k < valid_num[pix0 + lx]) && p >= 0 && p < Nattr && w != 0.0f;
    Li[wave][i] = ok ? p : 0;
    Lw[wave][i] = ok ? w : 0.0f;
  }
  wave_lds_sync();      // (a wave's LDS operations complete in order: its own writes are there)
  const int x = lane / C, c = lane - x * C;
  if (x >= npx) return;
  const int *li = Li[wave] + x * K;
  const float *lw = Lw[wave] + x * K;
  float acc = 0.0f;
  int k = 0;
  for (; k + 4 <= K; k += 4) {      // four gathers in flight
    const float a0 = attr[(size_t)li[k] * C + c], a1 = attr[(size_t)li[k + 1] * C + c];
    const float a2 = attr[(size_t)li[k + 2] * C + c], a3 = attr[(size_t)li[k + 3] * C + c];
    acc = fmaf(a0, lw[k], acc); acc = fmaf(a1, lw[k + 1], acc); acc = fmaf(a2, lw[k + 2], acc); acc = fmaf(a3, lw[k + 3], acc);
  }
  for (; k < K; ++k) acc = fmaf(attr[(size_t)li[k] * C + c], lw[k], acc);
  out[(pix0 + x) * C + c] = acc;
}

// merge backward for 4 < C <= 64: ONE LANE PER SLOT.  The run's g_up rows sit in LDS; a lane reads its slot's attribute row
// and forms g_weight = <g_up[pixel], attr[p]> by itself -- no cross-lane sum, coalesced idx / weight / g_weight streams --
// and, when the attributes want a gradient, adds w g_up[pixel][c] to their row (float atomics).  merge_bwd_chan_kernel (one
// slot per wave iteration, lanes over channels) is built for hundreds of channels: at C = 6 it runs 6 of 64 lanes (256^2,
// K = 102: 190 us).
constexpr int kMbsRun = 8;      // pixels per wave run
__global__ void __launch_bounds__(256)
merge_bwd_slot_kernel(const float *__restrict__ attr, const int32_t *__restrict__ idx, const float *__restrict__ weight,
                      const int64_t *__restrict__ valid_num, const float *__restrict__ g_up, const long npix, const int K,
                      const int C, const long Nattr, float *__restrict__ g_attr, float *__restrict__ g_weight) {
  __shared__ float Lg[4][kMbsRun * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long pix0 = ((long)blockIdx.x * 4 + wave) * kMbsRun;
  if (pix0 >= npix) return;      // waves are independent
  const int npx = (int)min((long)kMbsRun, npix - pix0);
  for (int i = lane; i < npx * C; i += 64) Lg[wave][i] = g_up[pix0 * C + i];
  wave_lds_sync();
  const int n_items = npx * K;
  const long base = pix0 * K;
  const float rK = 1.0f / (float)K;
  for (int j = lane; j < n_items; j += 64) {
    int lx = __float2int_rz(((float)j + 0.5f) * rK);
    int k = j - lx * K;
    if (k < 0) { --lx; k += K; } else if (k >= K) { ++lx; k -= K; }
    const int raw = idx[base + j];
    const float w = weight[base + j];
    const int p = raw + (raw < 0);
    float gw = 0.0f;
    if (k < valid_num[pix0 + lx] && p >= 0 && p < Nattr) {
      const float *g = Lg[wave] + lx * C;
      for (int c = 0; c < C; ++c) {
        const float gc = g[c];
        if (attr != nullptr) gw = fmaf(gc, attr[(size_t)p * C + c], gw);
        if (g_attr != nullptr && w != 0.0f && gc != 0.0f) unsafeAtomicAdd(g_attr + (size_t)p * C + c, w * gc);
      }
    }
    if (g_weight != nullptr) g_weight[base + j] = gw;
  }
}

// blend backward alone (C > 4, or a caller that merged separately): g_rgb [npix,C] and the
// additive silhouette term g_weight_add [npix,K].
__global__ void __launch_bounds__(256)
blend_bwd_kernel(const float *__restrict__ rgb, const float *__restrict__ weight, const float *__restrict__ bg,
                 const float thr, const float *__restrict__ g_out, const long npix, const int K, const int C,
                 float *__restrict__ g_rgb, float *__restrict__ g_weight_add) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const SlotLanes sl = slot_lanes(K, lane);
  const long pix_begin = wave_id * kShadePixPerWave;
  for (int i0 = 0; i0 < kShadePixPerWave; i0 += sl.ppi) {
    const long pix = pix_begin + i0 + sl.sub;
    const bool pix_ok = (sl.sub < sl.ppi) && (i0 + sl.sub < kShadePixPerWave) && (pix < npix);
    float wsum = 0.0f;
    for (int kc = 0; kc < K; kc += 64) {
      const int k = kc + sl.kl;
      if (pix_ok && k < K) wsum += weight[pix * K + k];
    }
    wsum = seg_bcast(seg_sum(wsum, lane, sl.lpp), lane, sl.kl);
    float sil = fminf(wsum, 1.0f);
    const float pass_s = (thr > 0.0f) ? 0.0f : clamp1_pass(wsum);
    if (thr > 0.0f) sil = sil > thr ? 1.0f : 0.0f;
    float g_mask = 0.0f;
    if (pix_ok) {
      for (int c = sl.kl; c < C; c += sl.lpp) {
        const float x = fmaf(1.0f - sil, bg[c], rgb[pix * C + c]);
        const float g = g_out[pix * C + c] * clamp1_pass(x);
        if (g_rgb != nullptr) g_rgb[pix * C + c] = g;
        g_mask = fmaf(-g, bg[c], g_mask);
      }
    }
    g_mask = seg_bcast(seg_sum(g_mask, lane, sl.lpp), lane, sl.kl);
    if (g_weight_add != nullptr) {
      const float gs = g_mask * pass_s;
      for (int kc = 0; kc < K; kc += 64) {
        const int k = kc + sl.kl;
        if (pix_ok && k < K) g_weight_add[pix * K + k] = gs;
      }
    }
  }
}

// get_silhouette backward (VoGE/Renderer.py:157-159): d min(sum_k w_k, 1) / d w_k = pass(sum w), the same for every
// slot of the pixel -- ONE value per pixel; the caller hands it to autograd as a [.., K] view with stride 0.
__global__ void __launch_bounds__(256)
silhouette_bwd_kernel(const float *__restrict__ wsum, const float *__restrict__ g_sil, const long npix,
                      float *__restrict__ g_pix) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < npix) g_pix[p] = g_sil[p] * clamp1_pass(wsum[p]);
}

static inline unsigned shade_grid(long npix) {   // blend_bwd_kernel: kShadePixPerWave pixels per wave
  const long waves = (npix + kShadePixPerWave - 1) / kShadePixPerWave;
  long b = (waves + 3) / 4;
  if (b < 1) b = 1;
  return (unsigned)b;
}
static inline unsigned run_grid(long npix) {     // shade_fwd_kernel: kRun pixels per wave
  const long waves = (npix + kRun - 1) / kRun;
  long b = (waves + 3) / 4;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace voge

using namespace voge;

extern "C" int voge_shade_fwd(const float *attr, int32_t *idx, const float *weight, const int64_t *valid_num,
                              const float *bg, float thr, long npix, int K, int C, long Nattr,
                              int fix_negative_idx, float *out_rgb, float *out_img, float *out_sil,
                              float *out_wsum, voge_stream_t stream) {
  if (npix < 0 || K <= 0 || C < 0 || Nattr < 0) return VOGE_ERR_BAD_ARG;
  if (npix == 0) return 0;
  if (!weight) return VOGE_ERR_BAD_ARG;
  if (C > 0 && ((!attr && Nattr > 0) || !idx || !valid_num)) return VOGE_ERR_BAD_ARG;
  if (C > 0 && Nattr == 0 && !attr) attr = weight;      // an empty attribute table: never gathered from, but "not NULL" means "merge"
  if (out_img && !bg) return VOGE_ERR_BAD_ARG;
  const bool aligned = ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(weight)) & 15) == 0;
  if (C > 0 && (K & 3) == 0 && K <= 64 && aligned) {
#ifndef VOGE_SH4_T
#define VOGE_SH4_T 256
#endif
    const long per_wg = (VOGE_SH4_T / 64) * 4L * kSh4P;      // waves are independent: 4 * kSh4P pixels each
    hipLaunchKernelGGL(shade_fwd4_kernel, dim3((unsigned)((npix + per_wg - 1) / per_wg)), dim3(VOGE_SH4_T), 0,
                       (hipStream_t)stream, attr, idx, weight, valid_num, bg, thr, npix, K, C, Nattr, fix_negative_idx,
                       out_rgb, out_img, out_sil, out_wsum);
    return launch_status();
  }
  if (C >= 1 && C <= 64 && K <= kMpcSlots && Nattr > 0 && attr && out_rgb && !out_img && !out_sil && !out_wsum) {      // merge alone
    const int ppw = std::max(1, std::min(64 / C, kMpcSlots / K));
    const long per_wg = 4L * ppw;
    hipLaunchKernelGGL(merge_fwd_pc_kernel, dim3((unsigned)((npix + per_wg - 1) / per_wg)), dim3(256), 0, (hipStream_t)stream, attr, idx,
                       weight, valid_num, npix, K, C, Nattr, fix_negative_idx, ppw, out_rgb);
    return launch_status();
  }
  hipLaunchKernelGGL(shade_fwd_kernel, dim3(run_grid(npix)), dim3(256), 0, (hipStream_t)stream, attr, idx, weight,
                     valid_num, nullptr, bg, thr, npix, K, C, Nattr, fix_negative_idx, out_rgb, out_img, out_sil, out_wsum);
  return launch_status();
}

extern "C" int voge_shade_bwd(const float *attr, const int32_t *idx, const float *weight,
                              const int64_t *valid_num, const float *rgb, const float *wsum, const float *bg,
                              float thr, const float *g_up, long nrows, int W, int K, int C, long Nattr,
                              float *g_attr, float *g_weight, voge_stream_t stream) {
  if (nrows < 0 || W < 0 || K <= 0 || C <= 0 || C > 4 || Nattr < 0) return VOGE_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (g_attr != nullptr && Nattr > 0) {
    hipError_t e = voge_fill_async(g_attr, 0, sizeof(float) * (size_t)Nattr * C, st);
    if (e != hipSuccess) return (int)e;
  }
  if (nrows * W == 0) return 0;
  if (!idx || !weight || !valid_num || !g_up || (Nattr > 0 && !attr && g_weight) || (bg && !rgb)) return VOGE_ERR_BAD_ARG;
  const long tiles = (long)((W + 7) / 8) * ((nrows + kShadeTH - 1) / kShadeTH);
  hipLaunchKernelGGL(shade_bwd_tile_kernel, dim3((unsigned)((tiles + kShadeWaves - 1) / kShadeWaves)),
                     dim3(64 * kShadeWaves), 0, st, attr, idx, weight, valid_num, rgb, wsum, bg, thr, g_up, nrows, W, K, C,
                     Nattr, g_attr, g_weight);
  return launch_status();
}

extern "C" int voge_merge_fwd(const float *attr, int32_t *idx, const float *weight,
                              const int64_t *valid_num, long npix, int K, int C, long Nattr,
                              int fix_negative_idx, float *out, voge_stream_t stream) {
  if (C <= 0 || !out) return VOGE_ERR_BAD_ARG;
  return voge_shade_fwd(attr, idx, weight, valid_num, nullptr, -1.0f, npix, K, C, Nattr, fix_negative_idx, out, nullptr,
                        nullptr, nullptr, stream);
}

extern "C" int voge_merge_bwd(const float *attr, const int32_t *idx, const float *weight,
                              const int64_t *valid_num, const float *g_out, long nrows, int W, int K,
                              int C, long Nattr, float *g_attr, float *g_weight, voge_stream_t stream) {
  if (nrows < 0 || W < 0 || K <= 0 || C <= 0 || Nattr < 0) return VOGE_ERR_BAD_ARG;
  if (C <= 4) return voge_shade_bwd(attr, idx, weight, valid_num, nullptr, nullptr, nullptr, -1.0f, g_out, nrows, W, K, C, Nattr,
                                    g_attr, g_weight, stream);
  hipStream_t st = (hipStream_t)stream;
  if (g_attr != nullptr && Nattr > 0) {
    hipError_t e = voge_fill_async(g_attr, 0, sizeof(float) * (size_t)Nattr * C, st);
    if (e != hipSuccess) return (int)e;
  }
  const long npix = nrows * W;
  if (npix == 0) return 0;
  if (!idx || !weight || !valid_num || !g_out || (Nattr > 0 && !attr && g_weight)) return VOGE_ERR_BAD_ARG;
  if (C <= 64) {
    const long per_wg = 4L * kMbsRun;
    hipLaunchKernelGGL(merge_bwd_slot_kernel, dim3((unsigned)((npix + per_wg - 1) / per_wg)), dim3(256), 0, st, attr, idx, weight,
                       valid_num, g_out, npix, K, C, Nattr, g_attr, g_weight);
    return launch_status();
  }
  long blocks = (npix + 3) / 4;
  if (blocks > 256L * 16) blocks = 256L * 16;
  hipLaunchKernelGGL(merge_bwd_chan_kernel, dim3((unsigned)blocks), dim3(256), 0, st, attr, idx, weight, valid_num,
                     g_out, npix, K, C, Nattr, g_attr, g_weight);
  return launch_status();
}

extern "C" int voge_blend_fwd(const float *rgb, const float *weight, const float *bg, float thr,
                              long npix, int K, int C, float *out, float *sil_out,
                              voge_stream_t stream) {
  if (npix < 0 || K <= 0 || C < 0) return VOGE_ERR_BAD_ARG;
  if (npix == 0) return 0;
  if (!weight || (C > 0 && (!rgb || !bg || !out))) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(shade_fwd_kernel, dim3(run_grid(npix)), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr,
                     weight, nullptr, rgb, bg, thr, npix, K, C, 0L, 0, nullptr, out, sil_out, nullptr);
  return launch_status();
}

extern "C" int voge_blend_bwd(const float *rgb, const float *weight, const float *bg, float thr,
                              const float *g_out, long npix, int K, int C, float *g_rgb,
                              float *g_weight_add, voge_stream_t stream) {
  if (npix < 0 || K <= 0 || C <= 0) return VOGE_ERR_BAD_ARG;
  if (npix == 0) return 0;
  if (!rgb || !weight || !bg || !g_out) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(shade_grid(npix)), dim3(256), 0, (hipStream_t)stream, rgb, weight, bg,
                     thr, g_out, npix, K, C, g_rgb, g_weight_add);
  return launch_status();
}

extern "C" int voge_silhouette_fwd(const float *weight, long npix, int K, float *sil, float *wsum, voge_stream_t stream) {
  if (npix < 0 || K <= 0) return VOGE_ERR_BAD_ARG;
  if (npix == 0) return 0;
  if (!weight || !sil) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(shade_fwd_kernel, dim3(run_grid(npix)), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr,
                     weight, nullptr, nullptr, nullptr, -1.0f, npix, K, 0, 0L, 0, nullptr, nullptr, sil, wsum);
  return launch_status();
}

extern "C" int voge_silhouette_bwd(const float *wsum, const float *g_sil, long npix, float *g_pix, voge_stream_t stream) {
  if (npix < 0) return VOGE_ERR_BAD_ARG;
  if (npix == 0) return 0;
  if (!wsum || !g_sil || !g_pix) return VOGE_ERR_BAD_ARG;
  hipLaunchKernelGGL(silhouette_bwd_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wsum,
                     g_sil, npix, g_pix);
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
