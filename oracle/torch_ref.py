"""The reference's CPU-capable TENSOR PROGRAM, restated in torch -- TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.

BASELINE.md section 3 defines "the reference's pure-PyTorch CPU path" (the reference ships no CPU ray trace) as
  * stage 3, the reference's own tensor program: the [pix, K, K] cross-activation / erf / exp chain of
    VoGE/Aggregation.py:30-107 (get_cross_activation :30-52, assign2weight :55-79, aggregation :82-107), merge_final
    (:111-141) and the silhouette / background blend of VoGE/Renderer.py:157-171, op for op on dense tensors, differentiated
    by autograd exactly as the reference differentiates it;
  * stages 1-2, a dense torch statement of the fine trace (ray_trace_voge.cu:135-217): einsum quadratic forms on
    [pixel chunk, P], the `act < thr_act` mask, topk(K, largest=False) on the hit length.
Only bench.py's cpu_baseline leg and tests/ import this module (pinned against the imported reference's outputs in
tests/test_oracle_cpu.py through the golden fixtures of tests/golden/make_golden.py).  voge_amd never does.
"""
import math

import torch

SENT = 1e10      # the trace's sentinels (ray_trace_voge.cu:244-247)


def cross_activation(sel_len, sel_dsd):
    """[P,K], [P,K] -> [P,K(m),K(k)]: (l_m - l_k) * sqrt(dsd_k + 1e-10)   (Aggregation.py:49)"""
    return (sel_len[:, :, None] - sel_len[:, None, :]) * (sel_dsd[:, None, :] + 1e-10).pow(0.5)


def weights(sel_act, cross, occ=1.0):
    """Aggregation.py:70-79: exp(-occ * sum_k exp(-act_k) (erf(ca_mk) + 1) / 2) * exp(-act_m) / exp(-0.5)"""
    density = torch.exp(-sel_act[:, None, :]) * ((torch.erf(cross) + 1) / 2)
    return torch.exp(-density.sum(2) * occ) * torch.exp(-sel_act) / math.exp(-0.5)


def aggregation(sel_idx, sel_act, sel_len, sel_dsd, occ=1.0):
    """-> weight [.., K], valid_num [..]   (Aggregation.py:82-107)"""
    K = sel_idx.shape[-1]
    w = weights(sel_act.reshape(-1, K), cross_activation(sel_len.reshape(-1, K), sel_dsd.reshape(-1, K)), occ)
    return w.reshape(sel_idx.shape), (sel_idx >= 0).sum(-1)


def merge_final(attr, weight, valid_num, idx):
    """Aggregation.py:111-141: only the first valid_num slots contribute, negative indices read row 0."""
    K = idx.shape[-1]
    live = torch.arange(K, device=idx.device).expand(idx.shape) < valid_num[..., None]
    gathered = attr[idx.clamp(min=0).long()]                      # [.., K, C]
    return (gathered * (weight * live)[..., None]).sum(-2)


def to_colored_background(rgb, weight, bg, thr=-1.0):
    """Renderer.py:157-171"""
    sil = torch.minimum(weight.sum(-1), torch.ones((), dtype=weight.dtype))
    if thr > 0:
        sil = (sil > thr).to(weight.dtype)
    return torch.minimum(rgb + (1 - sil)[..., None] * bg, torch.ones((), dtype=rgb.dtype))


def pair_forms(mus, isg, rays):
    """The three quadratic forms of ray_trace_voge.cu:184-194 on dense tensors: mus [N,3], isg [N,3,3], rays [P,3] ->
    len, act, dsd, each [P,N]."""
    Ad = torch.einsum("nij,pj->pni", isg, rays)                   # A d
    dsd = torch.einsum("pni,pi->pn", Ad, rays)                    # d^T A d
    mAd = torch.einsum("ni,pni->pn", mus, Ad)                     # mu^T A d
    mAm = torch.einsum("ni,nij,nj->n", mus, isg, mus)             # mu^T A mu
    return mAd / dsd, mAm[None] - mAd * mAd / dsd, dsd


def trace_dense(mus, isg, rays, K, thr_act, chunk=256):
    """Fine trace on dense tensors: the K smallest-len hits (act < thr_act) per ray, selection without a graph, the kept
    hits re-evaluated differentiably (as the reference differentiates only what its kernel kept).  rays [P,3] ->
    idx int64 [P,K] (-1 = empty), len, act, dsd [P,K] (sentinels in empty slots)."""
    P = rays.shape[0]
    idx = torch.full((P, K), -1, dtype=torch.int64)
    with torch.no_grad():
        for p0 in range(0, P, chunk):
            ln, act, _ = pair_forms(mus, isg, rays[p0:p0 + chunk])
            key = torch.where((act < thr_act) & (ln < SENT), ln, torch.full_like(ln, float("inf")))
            val, sel = torch.topk(key, min(K, key.shape[1]), dim=1, largest=False)
            sel = torch.where(torch.isfinite(val), sel, torch.full_like(sel, -1))
            idx[p0:p0 + chunk, : sel.shape[1]] = sel
    live = idx >= 0
    g = idx.clamp(min=0)
    m, A = mus[g], isg[g]                                         # [P,K,3], [P,K,3,3]
    Ad = torch.einsum("pkij,pj->pki", A, rays)
    dsd = torch.einsum("pki,pi->pk", Ad, rays)
    mAd = torch.einsum("pki,pki->pk", m, Ad)
    mAm = torch.einsum("pki,pkij,pkj->pk", m, A, m)
    ln = torch.where(live, mAd / dsd, torch.full_like(dsd, SENT))
    act = torch.where(live, mAm - mAd * mAd / dsd, torch.full_like(dsd, SENT))
    dsd = torch.where(live, dsd, torch.zeros_like(dsd))
    return idx, ln, act, dsd


def frame(verts, sigmas, colors, rays, origin, K, thr=0.01, occ=1.0, chunk=256):
    """One view of GaussianRenderer.forward + to_white_background on rays [P,3] (Renderer.py:124-143,174-176) for
    scalar sigmas: A = 2 sigma I.  Returns the image rows [P,3]."""
    mus = verts - origin[None]
    isg = (2 * sigmas)[:, None, None] * torch.eye(3, dtype=verts.dtype)[None]
    thr_act = -math.log(thr + 1 / 1e10)
    idx, ln, act, dsd = trace_dense(mus, isg, rays, K, thr_act, chunk)
    w, vn = aggregation(idx, act, ln, dsd, occ)
    rgb = merge_final(colors, w, vn, idx)
    return to_colored_background(rgb, w, torch.ones(3, dtype=verts.dtype))
