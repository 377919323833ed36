/*
 * TEST INFRASTRUCTURE ONLY -- see voge_oracle.c for the header comment.
 *
 * This file is included twice by voge_oracle.c, once with REAL=double
 * (SUFFIX=f64, the parity truth) and once with REAL=float (SUFFIX=f32, the
 * "reference-order" arithmetic used only to show the reference's own fp32
 * noise floor).  Everything here is a CPU restatement of the algorithm in
 *   VoGE/csrc/ray_trace_voge/ray_trace_voge.cu   (fine trace fwd/bwd)
 *   VoGE/Aggregation.py                          (composite, merge)
 *   VoGE/Renderer.py:153-176                     (blend helpers)
 * Inputs are always the fp32/int32 arrays the reference boundary takes;
 * outputs are REAL so the f64 build keeps full precision.
 */

#define CAT2(a, b) a##_##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(name, SUFFIX)

/* a^T B c for 3-vectors a,c and a row-major 3x3 B: the 9 products are summed in
 * exactly the order of Innerdot3d (ray_trace_voge.cu:11-38). */
static inline REAL FN(form3)(const float *a, const float *Bm, const float *c) {
  const REAL a1 = a[0], a2 = a[1], a3 = a[2];
  const REAL c1 = c[0], c2 = c[1], c3 = c[2];
  const REAL b11 = Bm[0], b12 = Bm[1], b13 = Bm[2];
  const REAL b21 = Bm[3], b22 = Bm[4], b23 = Bm[5];
  const REAL b31 = Bm[6], b32 = Bm[7], b33 = Bm[8];
  return a1 * b11 * c1 + a1 * b12 * c2 + a1 * b13 * c3 + a2 * b21 * c1 +
         a2 * b22 * c2 + a2 * b23 * c3 + a3 * b31 * c1 + a3 * b32 * c2 +
         a3 * b33 * c3;
}

/*
 * Fine trace forward.  Follows RayTraceFineVogeKernel (ray_trace_voge.cu:135-217)
 * statement by statement: per pixel, walk the candidate list in order, evaluate
 * the three quadratic forms, keep a candidate when `act < thr_act && len <
 * out_len[cur]` (:197), write it at slot `cur`, bubble it towards the front with
 * a strict `<` on len (:203), advance cur up to K-1 (:210).  Outputs start at the
 * sentinels idx=-1, len=1e10, act=1e10, dsd=0 (:244-247).
 *
 * Candidate list: if bin_points == NULL the list of every pixel of batch b is
 * b*N .. b*N+N-1 in ascending order, i.e. what RayTracing.py:22-26 builds for
 * max_points_per_bin == -1.  Otherwise bin_points is [B,BH,BW,M] int32 (-1 =
 * empty) and the pixel (y,x) uses bin (y/bin_size, x/bin_size).  (The reference
 * indexes the batch offset with BH*BH*M, ray_trace_voge.cu:185, which equals
 * BH*BW*M for the square bin grids it is used with; the restatement uses BH*BW*M.)
 *
 * mus [B*N,3], isg [B*N,9], rays [B,H,W,3]; outputs [B,H,W,K].
 */
void FN(oracle_trace_fwd)(const float *mus, const float *isg, const float *rays,
                          const int32_t *bin_points, int B, int N, int H, int W,
                          int K, int BH, int BW, int M, int bin_size,
                          double thr_act, int32_t *out_idx, REAL *out_len,
                          REAL *out_act, REAL *out_dsd) {
  const REAL thr = (REAL)thr_act;
  const long npix = (long)B * H * W;
#pragma omp parallel for schedule(dynamic, 64)
  for (long pix = 0; pix < npix; ++pix) {
    const int bi = (int)(pix / ((long)H * W));
    const int rem = (int)(pix % ((long)H * W));
    const int yi = rem / W, xi = rem % W;
    int32_t *oi = out_idx + pix * K;
    REAL *ol = out_len + pix * K, *oa = out_act + pix * K, *od = out_dsd + pix * K;
    for (int k = 0; k < K; ++k) {
      oi[k] = -1;
      ol[k] = (REAL)1e10;
      oa[k] = (REAL)1e10;
      od[k] = (REAL)0;
    }
    const float *ray = rays + pix * 3;
    const int32_t *lst = NULL;
    int cnt = N;
    if (bin_points) {
      const int by = yi / bin_size, bx = xi / bin_size;
      lst = bin_points + (((long)bi * BH + by) * BW + bx) * M;
      cnt = M;
    }
    int cur = 0;
    for (int m = 0; m < cnt; ++m) {
      const int p = lst ? lst[m] : bi * N + m;
      if (p < 0) continue;
      const float *mu = mus + (long)p * 3;
      const float *Bm = isg + (long)p * 9;
      const REAL ksk = FN(form3)(ray, Bm, ray);
      const REAL msk = FN(form3)(mu, Bm, ray);
      const REAL msm = FN(form3)(mu, Bm, mu);
      const REAL len = msk / ksk;
      const REAL act = msm - msk * msk / ksk;
      if (act < thr && len < ol[cur]) {
        ol[cur] = len;
        oa[cur] = act;
        od[cur] = ksk;
        oi[cur] = p;
        for (int t = cur; t > 0 && ol[t] < ol[t - 1]; --t) {
          REAL x;
          int32_t xi32;
          x = ol[t]; ol[t] = ol[t - 1]; ol[t - 1] = x;
          x = oa[t]; oa[t] = oa[t - 1]; oa[t - 1] = x;
          x = od[t]; od[t] = od[t - 1]; od[t - 1] = x;
          xi32 = oi[t]; oi[t] = oi[t - 1]; oi[t - 1] = xi32;
        }
        if (cur < K - 1) ++cur;
      }
    }
  }
}

/*
 * Fine trace backward.  Follows RayTraceFineVogeBackwardKernel
 * (ray_trace_voge.cu:283-332) and Innerdot3dBackward (:41-91): per valid
 * (pixel,k) recompute the three forms, apply the chain rule of :324-326, then
 * scatter the three outer-product gradients.  grad_isg is the raw (not
 * symmetrised) outer-product sum.  The reference accumulates with float
 * atomicAdd in arbitrary order; here pixels run in parallel (OpenMP) and the
 * per-Gaussian sums use atomic adds as well -- in REAL = double the order
 * dependence is ~1e-16 relative.  g_ray is pixel-owned (a pixel's K slots run
 * in order on one thread).
 * Outputs (zeroed here): g_ray [B,H,W,3], g_mus [P,3], g_isg [P,9].
 */
void FN(oracle_trace_bwd)(const float *mus, const float *isg, const float *rays,
                          const int32_t *idx, const REAL *g_len,
                          const REAL *g_act, const REAL *g_dsd, long npix, int K,
                          int P, REAL *g_ray, REAL *g_mus, REAL *g_isg) {
  memset(g_ray, 0, sizeof(REAL) * npix * 3);
  memset(g_mus, 0, sizeof(REAL) * (size_t)P * 3);
  memset(g_isg, 0, sizeof(REAL) * (size_t)P * 9);
#pragma omp parallel for schedule(dynamic, 64)
  for (long r = 0; r < npix; ++r)
  for (long pid = r * K; pid < (r + 1) * K; ++pid) {
    const int p = idx[pid];
    if (p == -1) continue;
    const float *ray = rays + r * 3, *mu = mus + (long)p * 3, *Bm = isg + (long)p * 9;
    const REAL gl = g_len[pid], ga = g_act[pid], gd = g_dsd[pid];
    const REAL ksk = FN(form3)(ray, Bm, ray);
    const REAL msk = FN(form3)(mu, Bm, ray);
    const REAL g_ksk = (ga * msk - gl) * msk / (ksk * ksk) + gd;
    const REAL g_msk = (gl - 2 * ga * msk) / ksk;
    const REAL g_msm = ga;
    REAL *gr = g_ray + r * 3, *gm = g_mus + (long)p * 3, *gB = g_isg + (long)p * 9;
    /* the three (a, c, grad, a-grad target, c-grad target) triples of :328-330; this slot's
     * contributions are summed locally (same order), then added to the shared per-Gaussian sums */
    const float *av[3] = {ray, mu, mu};
    const float *cv[3] = {ray, ray, mu};
    const REAL gg[3] = {g_ksk, g_msk, g_msm};
    REAL l_ray[3] = {0, 0, 0}, l_mu[3] = {0, 0, 0}, l_B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    REAL *ga_out[3] = {l_ray, l_mu, l_mu};
    REAL *gc_out[3] = {l_ray, l_ray, l_mu};
    for (int f = 0; f < 3; ++f) {
      const float *a = av[f], *c = cv[f];
      const REAL g = gg[f];
      for (int i = 0; i < 3; ++i) {
        REAL Bc = 0, Bta = 0;
        for (int j = 0; j < 3; ++j) {
          Bc += (REAL)Bm[i * 3 + j] * (REAL)c[j];
          Bta += (REAL)Bm[j * 3 + i] * (REAL)a[j];
          l_B[i * 3 + j] += ((REAL)a[i] * (REAL)c[j]) * g;
        }
        ga_out[f][i] += Bc * g;
        gc_out[f][i] += Bta * g;
      }
    }
    for (int i = 0; i < 3; ++i) gr[i] += l_ray[i];      /* pixel-owned */
    for (int i = 0; i < 3; ++i) {
#pragma omp atomic
      gm[i] += l_mu[i];
    }
    for (int i = 0; i < 9; ++i) {
#pragma omp atomic
      gB[i] += l_B[i];
    }
  }
}

/*
 * Composite ("aggregation") forward.  Follows Aggregation.py:30-107:
 *   ca[m,k] = (len_m - len_k) * sqrt(dsd_k + 1e-10)            (:49)
 *   dd[m,k] = exp(-act_k) * (erf(ca[m,k]) + 1) / 2              (:70)
 *   T_m     = exp(-occ * sum_k dd[m,k])     (k == m included)   (:74)
 *   w_m     = T_m * exp(-act_m) / exp(-0.5)                     (:77-79)
 *   valid   = #(idx >= 0)                                       (:104)
 */
void FN(oracle_composite_fwd)(const int32_t *idx, const REAL *act, const REAL *len,
                              const REAL *dsd, double occ_, long npix, int K,
                              REAL *weight, int64_t *valid_num) {
  const REAL occ = (REAL)occ_;
  const REAL inv_norm = (REAL)(1.0 / exp(-0.5));
#pragma omp parallel for schedule(static)
  for (long pix = 0; pix < npix; ++pix) {
    const REAL *a = act + pix * K, *l = len + pix * K, *d = dsd + pix * K;
    int64_t nv = 0;
    for (int m = 0; m < K; ++m) {
      REAL s = 0;
      for (int k = 0; k < K; ++k) {
        const REAL ca = (l[m] - l[k]) * (REAL)sqrt((double)(d[k] + (REAL)1e-10));
        s += (REAL)exp((double)-a[k]) * (((REAL)erf((double)ca) + 1) / 2);
      }
      weight[pix * K + m] = (REAL)exp((double)(-s * occ)) * (REAL)exp((double)-a[m]) * inv_norm;
      nv += idx[pix * K + m] >= 0;
    }
    valid_num[pix] = nv;
  }
}

/*
 * Composite backward: the reference has no explicit code (autograd through
 * Aggregation.py:49,70,74,77).  With E_k=exp(-act_k), s_k=sqrt(dsd_k+1e-10),
 * Phi_mk=(erf(ca_mk)+1)/2, phi_mk=exp(-ca_mk^2)/sqrt(pi), u_m=g_m*w_m:
 *   dL/dact_j = -u_j + occ*E_j*sum_m u_m*Phi_mj
 *   dL/dlen_j = -occ*( u_j*sum_k E_k*phi_jk*s_k - E_j*s_j*sum_m u_m*phi_mj )
 *   dL/ddsd_j = -occ*E_j/(2*s_j) * sum_m u_m*phi_mj*(len_m-len_j)
 * These are checked against autograd of the imported reference in
 * tests/golden/make_golden.py (fixtures composite_*.npz).
 */
void FN(oracle_composite_bwd)(const REAL *act, const REAL *len, const REAL *dsd,
                              const REAL *g_weight, double occ_, long npix, int K,
                              REAL *g_act, REAL *g_len, REAL *g_dsd) {
  const REAL occ = (REAL)occ_;
  const REAL inv_norm = (REAL)(1.0 / exp(-0.5));
  const REAL rsqrt_pi = (REAL)0.56418958354775628695;
#pragma omp parallel
  {
    REAL *E = (REAL *)malloc(sizeof(REAL) * K * 3);
    REAL *s = E + K, *u = E + 2 * K;
#pragma omp for schedule(static)
    for (long pix = 0; pix < npix; ++pix) {
      const REAL *a = act + pix * K, *l = len + pix * K, *d = dsd + pix * K;
      for (int k = 0; k < K; ++k) {
        E[k] = (REAL)exp((double)-a[k]);
        s[k] = (REAL)sqrt((double)(d[k] + (REAL)1e-10));
      }
      for (int m = 0; m < K; ++m) {
        REAL sum = 0;
        for (int k = 0; k < K; ++k)
          sum += E[k] * (((REAL)erf((double)((l[m] - l[k]) * s[k])) + 1) / 2);
        const REAL w = (REAL)exp((double)(-sum * occ)) * E[m] * inv_norm;
        u[m] = g_weight[pix * K + m] * w;
      }
      for (int j = 0; j < K; ++j) {
        REAL sPhi = 0, sphi_in = 0, sphi_out = 0, sphi_len = 0;
        for (int m = 0; m < K; ++m) {
          const REAL ca_mj = (l[m] - l[j]) * s[j];
          const REAL ph_mj = (REAL)exp((double)(-ca_mj * ca_mj)) * rsqrt_pi;
          sPhi += u[m] * (((REAL)erf((double)ca_mj) + 1) / 2);
          sphi_in += u[m] * ph_mj;
          sphi_len += u[m] * ph_mj * (l[m] - l[j]);
          const REAL ca_jm = (l[j] - l[m]) * s[m];
          sphi_out += E[m] * (REAL)exp((double)(-ca_jm * ca_jm)) * rsqrt_pi * s[m];
        }
        g_act[pix * K + j] = -u[j] + occ * E[j] * sPhi;
        g_len[pix * K + j] = -occ * (u[j] * sphi_out - E[j] * s[j] * sphi_in);
        g_dsd[pix * K + j] = -occ * E[j] / (2 * s[j]) * sphi_len;
      }
    }
    free(E);
  }
}

/*
 * Attribute merge forward, Aggregation.py:111-141: only the first valid_num[pix]
 * slots contribute (:125-129,:134), negative indices are read as 0 (:131), then
 * out[pix,c] = sum_k attr[idx_k,c] * w_k (:137-140).
 */
void FN(oracle_merge_fwd)(const REAL *attr, const int32_t *idx, const REAL *weight,
                          const int64_t *valid_num, long npix, int K, int C,
                          REAL *out) {
#pragma omp parallel for schedule(static)
  for (long pix = 0; pix < npix; ++pix) {
    for (int c = 0; c < C; ++c) out[pix * C + c] = 0;
    for (int k = 0; k < K; ++k) {
      if (k >= valid_num[pix]) continue;
      int p = idx[pix * K + k];
      if (p < 0) p += 1;
      const REAL w = weight[pix * K + k];
      for (int c = 0; c < C; ++c) out[pix * C + c] += attr[(long)p * C + c] * w;
    }
  }
}

/* Merge backward (autograd of the above): g_weight[pix,k] = mask * <g_out[pix],
 * attr[idx_k]>, g_attr[idx_k] += mask * w_k * g_out[pix].  g_attr [Nattr,C] zeroed here. */
void FN(oracle_merge_bwd)(const REAL *attr, const int32_t *idx, const REAL *weight,
                          const int64_t *valid_num, const REAL *g_out, long npix,
                          int K, int C, long Nattr, REAL *g_attr, REAL *g_weight) {
  memset(g_attr, 0, sizeof(REAL) * (size_t)Nattr * C);
#pragma omp parallel for schedule(dynamic, 64)
  for (long pix = 0; pix < npix; ++pix) {
    for (int k = 0; k < K; ++k) {
      REAL gw = 0;
      if (k < valid_num[pix]) {
        int p = idx[pix * K + k];
        if (p < 0) p += 1;
        const REAL w = weight[pix * K + k];
        for (int c = 0; c < C; ++c) {
          gw += g_out[pix * C + c] * attr[(long)p * C + c];
          const REAL add = w * g_out[pix * C + c];
#pragma omp atomic
          g_attr[(long)p * C + c] += add;
        }
      }
      g_weight[pix * K + k] = gw;
    }
  }
}

/*
 * Blend onto a coloured background, Renderer.py:157-171:
 *   sil = min(sum_k w_k, 1); if thr > 0: sil = (sil > thr)
 *   out = min(rgb + (1 - sil) * bg, 1)
 */
void FN(oracle_blend_fwd)(const REAL *rgb, const REAL *weight, const REAL *bg,
                          double thr, long npix, int K, int C, REAL *out,
                          REAL *sil_out) {
#pragma omp parallel for schedule(static)
  for (long pix = 0; pix < npix; ++pix) {
    REAL s = 0;
    for (int k = 0; k < K; ++k) s += weight[pix * K + k];
    REAL sil = s < 1 ? s : 1;
    if (sil_out) sil_out[pix] = sil;
    if (thr > 0) sil = sil > (REAL)thr ? 1 : 0;
    for (int c = 0; c < C; ++c) {
      const REAL v = rgb[pix * C + c] + (1 - sil) * bg[c];
      out[pix * C + c] = v < 1 ? v : 1;
    }
  }
}

#undef FN
#undef CAT
#undef CAT2
