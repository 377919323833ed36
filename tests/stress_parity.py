"""Randomised parity sweep on the GPU: random (N, H, W, K, B, isotropy, occupancy) -> trace / composite / shade forward
AND backward against the fp64 oracle, through the public ops.  `run(n_cases, seed)` is what the suite calls
(tests/test_gpu_stress.py, fixed seeds, a dozen cases each); as a script it runs longer sweeps:
usage: python tests/stress_parity.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle
from oracle import camera_np
from util import TOL, compare_trace, random_scene
from voge_amd import ops

t = lambda x, dt=torch.float32, rg=False: torch.tensor(np.asarray(x), dtype=dt, device="cuda", requires_grad=rg)
n = lambda x: x.detach().cpu().numpy()


def run(n_cases=40, seed=0, verbose=True):
    """-> worst relative errors per quantity; raises AssertionError (after saving the inputs) on the first failure."""
    rng = np.random.default_rng(seed)
    worst = {}
    for case in range(n_cases):
        N = int(rng.integers(1, 1500)); H = int(rng.integers(1, 90)); W = int(rng.integers(1, 90))
        K = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 25, 31, 40, 64, 65, 100])); B = int(rng.integers(1, 3))
        aniso = bool(rng.integers(0, 2)); iso_api = (not aniso) and bool(rng.integers(0, 2))
        thr = float(rng.choice([0.0, 0.01, 0.05])); occ = float(rng.uniform(0.3, 2.0))
        verts, sig, cols = random_scene(N, seed=int(rng.integers(1 << 30)), aniso=aniso, lo=0.05, hi=0.25)
        cols = np.tile(cols, (B, 1))      # indices are packed over the batch (b * N + i), as in the reference
        if aniso:   # tame the conditioning: symmetric, moderate anisotropy
            sig = (0.5 * (sig + sig.transpose(0, 2, 1))).astype(np.float32)
        R, T = camera_np.look_at_view_transform([float(rng.uniform(2.5, 4.5))] * B, [float(rng.uniform(-40, 40))] * B,
                                                [float(rng.uniform(0, 360)) + 30 * b for b in range(B)])
        rays, origin = camera_np.pixel_rays(R, T, float(rng.uniform(0.6, 1.5)) * max(H, W), (W / 2.0, H / 2.0), (H, W))
        mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
        isg = (2 * camera_np.expand_sigma(np.asarray(sig, np.float32))).astype(np.float32)
        isg = np.ascontiguousarray(np.broadcast_to(isg[None], (B,) + isg.shape))
        thr_act = oracle.thr_act_of(thr)
        tm, tr = t(mus.reshape(-1, 3), rg=True), t(rays, rg=True)
        if iso_api:
            ta = t(np.ascontiguousarray(isg[..., 0, 0]).reshape(-1), rg=True)
            sel = ops._RayTraceVoGEIso.apply(tm, ta, tr, None, thr_act, K)
        else:
            ta = t(isg.reshape(-1, 3, 3), rg=True)
            sel = ops.ray_trace_fine(tm, ta, tr, None, thr_act, 10, K)
        ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
        tag = f"case {case}: N={N} {H}x{W} K={K} B={B} {'iso-api' if iso_api else ('aniso' if aniso else 'iso')} thr={thr}"
        try:
            compare_trace([n(x) for x in sel], ref, thr_act, min_match=max(0.0, min(0.97, 1.0 - 2.5 / (B * H * W))))
            same = (n(sel[0]) == ref[0]).all(-1)
            if same.mean() < 1.0:
                print(tag, "-> index lists differ on", int((~same).sum()), "pixels (decision-boundary flips); skipping value checks")
                continue
            w, vn = ops.composite(sel[0], sel[2], sel[1], sel[3], occ)
            wr, vr = oracle.composite_fwd(ref[0].reshape(-1, K), ref[2].reshape(-1, K), ref[1].reshape(-1, K), ref[3].reshape(-1, K), occ)
            assert (n(vn).reshape(-1) == vr).all()
            e = np.abs(n(w).reshape(-1, K) - wr).max(); worst["weight"] = max(worst.get("weight", 0), e); assert e < TOL, ("weight", e)
            colors = t(cols, rg=True)
            img = ops.shade(colors, w, sel[0], vn, t([1.0, 1.0, 1.0]), -1.0)
            g_img = rng.normal(size=tuple(img.shape))
            (img * t(g_img)).sum().backward()
            # oracle chain
            idx_r = ref[0].reshape(-1, K)
            sil = np.minimum(wr.sum(-1), 1.0)
            rgb = (cols[np.maximum(idx_r, 0)] * (wr * (np.arange(K)[None] < vr[:, None]))[..., None]).sum(1)
            x = rgb + (1 - sil)[:, None]
            e = np.abs(n(img).reshape(-1, 3) - np.minimum(x, 1)).max(); worst["img"] = max(worst.get("img", 0), e); assert e < TOL, ("img", e)
            if (((np.abs(x - 1) < 1e-5) & (x != 1)).any() or (np.abs(wr.sum(-1) - 1) < 1e-5).any()):
                print(tag, "-> a pixel sits on the min(., 1) clamp within fp32 rounding; skipping gradient checks")
                continue
            g_rgb = g_img.reshape(-1, 3) * (x < 1)
            g_sil = -(g_rgb.sum(-1)) * (wr.sum(-1) < 1)
            g_attr, g_w = oracle.merge_bwd(cols, idx_r, wr, vr, g_rgb)
            g_w = g_w + g_sil[:, None]
            g_act, g_len, g_dsd = oracle.composite_bwd(ref[2].reshape(-1, K), ref[1].reshape(-1, K), ref[3].reshape(-1, K), g_w, occ)
            shp = ref[0].shape
            g_ray, g_mu, g_A = oracle.trace_bwd(mus, isg, rays, ref[0], g_len.reshape(shp), g_act.reshape(shp), g_dsd.reshape(shp))
            checks = [("colors", colors.grad, g_attr), ("mus", tm.grad, g_mu), ("rays", tr.grad, g_ray)]
            checks.append(("a", ta.grad, np.einsum("nii->n", g_A.reshape(-1, 3, 3))) if iso_api else ("A", ta.grad, g_A))
            for name, got, want in checks:
                want = np.asarray(want, np.float64)
                err = np.abs(n(got).astype(np.float64).reshape(want.shape) - want).max() / max(1.0, np.abs(want).max())
                worst[name] = max(worst.get(name, 0), err)
                assert err <= 2 * TOL, (name, err)
        except AssertionError as ex:
            print("FAIL", tag, ex)
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/stress_fail.npz", mus=mus, isg=isg, rays=rays, cols=cols, K=K, thr_act=thr_act, occ=occ,
                     g_img=g_img if "g_img" in dir() else 0, iso_api=iso_api)
            raise
        if case % 10 == 9:
            print(f"{case + 1} cases ok; worst relative errors so far:", {k: f"{v:.1e}" for k, v in worst.items()})

    if verbose:
        print("all", n_cases, "cases ok", {k: f"{v:.1e}" for k, v in worst.items()})
    return worst


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def run_dense(n_cases=6, seed=0, verbose=True):
    """Small DENSE objects: thousands to tens of thousands of Gaussians behind a few dozen pixels, so that quads exceed the
    in-LDS sort (binB's pooled long path), binA's segments outgrow their inline part (extension chunks) or even their
    extensions (slice re-test), with random N, image shape, K, batch size and entry point.  Forward trace against the
    brute-force oracle.  -> number of cases whose quads took the long path."""
    rng = np.random.default_rng(seed)
    pooled = 0
    for case in range(n_cases):
        # (70 000 / 90 000: more than 65 536 Gaussians -- the rebuilt sweep's list entries then carry stream positions, and a
        # tile with more than 65 536 candidates takes its wide form)
        N = int(rng.choice([3000, 9000, 20000, 45000, 70000, 90000])); H = int(rng.integers(24, 80)); W = int(rng.integers(24, 80))
        K = int(rng.choice([1, 5, 12, 25, 40, 64])); B = int(rng.integers(1, 3))
        extent = float(rng.uniform(0.12, 0.5)); r_hi = float(rng.uniform(0.006, 0.03))
        iso_api = bool(rng.integers(0, 2))
        g = np.random.default_rng(int(rng.integers(1 << 30)))
        verts = g.uniform(-extent, extent, (N, 3)).astype(np.float32)
        r = g.uniform(0.5 * r_hi, r_hi, N)
        sig = (1.0 / (r * r / (2 * np.log(1 / 0.6)))).astype(np.float32)
        R, T = camera_np.look_at_view_transform([float(rng.uniform(2.5, 4.0))] * B, [float(rng.uniform(-30, 30))] * B,
                                                [float(rng.uniform(0, 360)) + 47 * b for b in range(B)])
        rays, origin = camera_np.pixel_rays(R, T, float(rng.uniform(0.8, 1.4)) * max(H, W), (W / 2.0, H / 2.0), (H, W))
        mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
        isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)
        isg = np.ascontiguousarray(np.broadcast_to(isg[None], (B,) + isg.shape))
        thr_act = oracle.thr_act_of(0.01)
        if iso_api:
            sel = ops._RayTraceVoGEIso.apply(t(mus.reshape(-1, 3)), t(np.ascontiguousarray(isg[..., 0, 0]).reshape(-1)), t(rays), None, thr_act, K)
        else:
            sel = ops.ray_trace_fine(t(mus.reshape(-1, 3)), t(isg.reshape(-1, 3, 3)), t(rays), None, thr_act, 10, K)
        used, cap = ops.trace_pool_usage("cuda:0", B, N, H, W)
        pooled += used > 0
        ref = oracle.trace_fwd(mus, isg, rays, K, thr_act)
        tag = f"dense case {case}: N={N} {H}x{W} K={K} B={B} extent={extent:.2f} r<={r_hi:.3f} {'iso-api' if iso_api else 'general-api'} pool {used}/{cap}"
        if verbose:
            print(tag, flush=True)
        compare_trace([n(x) for x in sel], ref, thr_act, min_match=0.99, label=tag)
    return pooled
