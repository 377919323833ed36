"""Replay the case tests/stress_parity.py saved on failure (gpurun_out/stress_fail.npz), several times,
and report per-stage errors against the oracle.  usage: python tests/stress_repro.py [file] [repeats]"""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
from voge_amd import ops

d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stress_fail.npz")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mus, isg, rays, cols, K, thr_act, occ, g_img = (d[k] for k in ("mus", "isg", "rays", "cols", "K", "thr_act", "occ", "g_img"))
K = int(K); thr_act = float(thr_act); occ = float(occ)
t = lambda x, dt=torch.float32, rg=False: torch.tensor(np.asarray(x), dtype=dt, device="cuda", requires_grad=rg)
n = lambda x: x.detach().cpu().numpy()
ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
idx_r = ref[0].reshape(-1, K)
wr, vr = oracle.composite_fwd(idx_r, ref[2].reshape(-1, K), ref[1].reshape(-1, K), ref[3].reshape(-1, K), occ)
sil = np.minimum(wr.sum(-1), 1.0)
rgb = (cols[np.maximum(idx_r, 0)] * (wr * (np.arange(K)[None] < vr[:, None]))[..., None]).sum(1)
x = rgb + (1 - sil)[:, None]
g_rgb = g_img.reshape(-1, 3) * (x < 1)
g_sil = -(g_rgb.sum(-1)) * (wr.sum(-1) < 1)
g_attr, g_w = oracle.merge_bwd(cols, idx_r, wr, vr, g_rgb)
g_w = g_w + g_sil[:, None]
print("pixels with x within 1e-4 of the clamp:", int((np.abs(x - 1) < 1e-4).sum()), " sil within 1e-4 of 1:", int((np.abs(wr.sum(-1) - 1) < 1e-4).sum()))
for r in range(reps):
    tm, tr = t(mus.reshape(-1, 3), rg=True), t(rays, rg=True)
    ta = t(isg.reshape(-1, 3, 3), rg=True)
    sel = ops.ray_trace_fine(tm, ta, tr, None, thr_act, 10, K)
    w, vn = ops.composite(sel[0], sel[2], sel[1], sel[3], occ)
    colors = t(cols, rg=True)
    w2 = w.detach().clone().requires_grad_(True)
    img = ops.shade(colors, w2, sel[0], vn, t([1.0, 1.0, 1.0]), -1.0)
    (img * t(g_img)).sum().backward()
    same = (n(sel[0]) == ref[0]).all()
    ec = np.abs(n(colors.grad) - g_attr); ew = np.abs(n(w2.grad).reshape(-1, K) - g_w)
    ei = np.abs(n(img).reshape(-1, 3) - np.minimum(x, 1))
    print(f"rep {r}: idx same {same}  img {ei.max():.2e}  g_colors {ec.max():.3e} at {np.unravel_index(ec.argmax(), ec.shape)} (ref scale {np.abs(g_attr).max():.2f})"
          f"  g_weight {ew.max():.3e} at {np.unravel_index(ew.argmax(), ew.shape)}")
