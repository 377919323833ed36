"""numpy restatements of the "next" rows -- TEST INFRASTRUCTURE.

dense trace / backward  VoGE/csrc/voge_ray_tracing_ray/voge_ray_tracing_ray.cu:114-188
find_nearest_k          voge_ray_tracing_ray.cu:191-239, host init :344-347
sample_voge (+backward) VoGE/csrc/sample_voge/sample_voge.cu:35-66, :173-209
scatter_max             sample_voge.cu:69-92
The reference ships no test or CPU build for any of them (CUDA only): parity unpinned beyond the
line-by-line restatement; sample_features is additionally checked against the dense formulation
its docstring quotes (Sampler.py:7-11)."""
import numpy as np


def ray_dense_fwd(mus, isg, rays):
    mu, A, d = (np.asarray(x, np.float64) for x in (mus, isg, rays))
    Ad = np.einsum("mij,nj->nmi", A, d)
    ksk = np.einsum("ni,nmi->nm", d, Ad)
    msk = np.einsum("mi,nmi->nm", mu, Ad)
    msm = np.einsum("mi,mij,mj->m", mu, A, mu)[None]
    return msk / ksk, msm - msk * msk / ksk, ksk


def ray_dense_bwd(mus, isg, rays, g_len, g_act, g_dsd):
    mu, A, d = (np.asarray(x, np.float64) for x in (mus, isg, rays))
    Ad = np.einsum("mij,nj->nmi", A, d)
    Atd = np.einsum("mji,nj->nmi", A, d)
    ksk = np.einsum("ni,nmi->nm", d, Ad)
    msk = np.einsum("mi,nmi->nm", mu, Ad)
    g_ksk = (g_act * msk - g_len) * msk / (ksk * ksk) + g_dsd     # voge_ray_tracing_ray.cu:181-183
    g_msk = (g_len - 2 * g_act * msk) / ksk
    g_msm = g_act
    Amu, Atmu = np.einsum("mij,mj->mi", A, mu), np.einsum("mji,mj->mi", A, mu)
    g_ray = np.einsum("nm,nmi->ni", g_ksk, Ad + Atd) + np.einsum("nm,mi->ni", g_msk, Atmu)
    g_mu = np.einsum("nm,nmi->mi", g_msk, Ad) + g_msm.sum(0)[:, None] * (Amu + Atmu)
    g_A = (np.einsum("nm,ni,nj->mij", g_ksk, d, d) + np.einsum("nm,mi,nj->mij", g_msk, mu, d)
           + g_msm.sum(0)[:, None, None] * mu[:, :, None] * mu[:, None, :])
    return g_ray, g_mu, g_A


def find_nearest_k(ln, act, dsd, K, thr_act):
    ln, act, dsd = (np.asarray(x, np.float64) for x in (ln, act, dsd))
    N, M = ln.shape
    idx = np.full((N, K), -1, np.int32)
    o_len, o_act, o_dsd = np.full((N, K), 1e10), np.zeros((N, K)), np.zeros((N, K))
    for n in range(N):
        cur = 0
        for m in range(M):
            if act[n, m] < thr_act and ln[n, m] < o_len[n, cur]:
                o_len[n, cur], o_act[n, cur], o_dsd[n, cur], idx[n, cur] = ln[n, m], act[n, m], dsd[n, m], m
                t = cur
                while t > 0 and o_len[n, t] < o_len[n, t - 1]:
                    for arr in (o_len, o_act, o_dsd, idx):
                        arr[n, t], arr[n, t - 1] = arr[n, t - 1], arr[n, t]
                    t -= 1
                if cur < K - 1:
                    cur += 1
    return idx, o_len, o_act, o_dsd


def sample_voge(image, weight, idx, n_vert):
    image, weight = np.asarray(image, np.float64), np.asarray(weight, np.float64)
    C, K = image.shape[-1], idx.shape[-1]
    img = image.reshape(-1, C)
    w, ix = weight.reshape(-1, K), np.asarray(idx).reshape(-1, K)
    feat, wsum = np.zeros((n_vert, C)), np.zeros(n_vert)
    for k in range(K):
        ok = ix[:, k] != -1
        np.add.at(feat, ix[ok, k], w[ok, k, None] * img[ok])
        np.add.at(wsum, ix[ok, k], w[ok, k])
    return feat, wsum


def sample_voge_bwd(image, weight, idx, g_feat, g_wsum):
    image, weight = np.asarray(image, np.float64), np.asarray(weight, np.float64)
    C, K = image.shape[-1], idx.shape[-1]
    img = image.reshape(-1, C)
    w, ix = weight.reshape(-1, K), np.asarray(idx).reshape(-1, K)
    ok = ix != -1
    safe = np.where(ok, ix, 0)
    gf = np.asarray(g_feat, np.float64)[safe]                       # [P, K, C]
    g_img = (gf * (w * ok)[..., None]).sum(1).reshape(image.shape)
    g_w = ((gf * img[:, None, :]).sum(-1) + np.asarray(g_wsum, np.float64)[safe]) * ok
    return g_img, g_w.reshape(weight.shape)


def scatter_max(weight, idx, n_vert):
    out = np.zeros(n_vert)
    w, ix = np.asarray(weight, np.float64).reshape(-1), np.asarray(idx).reshape(-1)
    ok = ix != -1
    np.maximum.at(out, ix[ok], w[ok])
    return out
