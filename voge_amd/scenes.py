"""Synthetic scenes of BASELINE.json's configs: seeded, no files needed."""
import math

import numpy as np


def random_gaussians(n, seed=0, anisotropic=False, r_lo=0.02, r_hi=0.04, extent=1.0):
    """Centres uniform in [-extent, extent]^3; isotropic "sigmas" (= Sigma^-1 scale)
    s = 1 / (r^2 / (2 ln(1/0.6))) with r ~ U(r_lo, r_hi) (the cuboid_gauss spacing rule,
    Cuboid.py:51-52); optional full 3x3 A = L L^T (EfficientCuboidViaOptimization.py:17-18) or, anisotropic="diag", (N,3) per-axis values."""
    rng = np.random.default_rng(seed)
    verts = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    r = rng.uniform(r_lo, r_hi, n)
    s = 1.0 / (r * r / (2 * math.log(1 / 0.6)))
    if anisotropic == "diag":      # (N,3): per-axis Sigma^-1 (EfficientCuboidViaOptimization.py:75-79), axis ratios up to 4
        sig = (s[:, None] * rng.uniform(0.5, 2.0, (n, 3))).astype(np.float32)
    elif anisotropic:
        L = np.tril(rng.uniform(-1, 1, (n, 3, 3)))
        d = np.arange(3)
        L[:, d, d] = np.abs(L[:, d, d]) + 0.3
        L = L * np.sqrt(s)[:, None, None]
        sig = (L @ L.transpose(0, 2, 1)).astype(np.float32)
    else:
        sig = s.astype(np.float32)
    colors = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    return verts, sig, colors


CONFIGS = {
    # name: (N, (H, W), K, focal, principal, (dist, elev, azim))
    "cfg3_50k_512": (50000, (512, 512), 40, 600.0, (256.0, 256.0), (4.0, 10.0, 70.0)),
    "cfg4_200k_1024": (200000, (1024, 1024), 40, 1200.0, (512.0, 512.0), (4.0, 10.0, 70.0)),
    "cfg5_shapefit_128": (2562, (128, 128), 25, 126.0, (64.0, 64.0), (2.7, 0.0, 0.0)),
}
