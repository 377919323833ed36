"""The reference's two remaining optimisation demos as LOOPS with a convergence check (VERDICT r4 "missing" 4):
  demo/ReasonOcclusion.py:92-112                  two cuboids, translations by Adam on an interpolate_attr MSE, 200 iterations;
  demo/EfficientCuboidViaOptimization.py:88-119   102 Gaussians of full 3x3 form fitted to a 4000-Gaussian cuboid, L1 on the
                                                   six-channel face map, Adam stepping every 10th iteration.
(Their settings at real size, one iteration against the oracle: tests/test_gpu_demo_sizes.py.)"""
import importlib.util
import os

import numpy as np
import pytest

from util import log_line

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_demo(name):
    spec = importlib.util.spec_from_file_location("demo_" + name, os.path.join(ROOT, "demo", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reason_occlusion_loop_converges(hip_lib):
    demo = load_demo("ReasonOcclusion")
    out = demo.run(iters=200, log=lambda s: log_line("[demo] ReasonOcclusion: " + s))
    # the first cuboid starts 6 units away, BEHIND the second one, and has to pass through it
    d0 = np.abs(out["v0"] - np.asarray(demo.TARGET[0])).max()
    d1 = np.abs(out["v1"] - np.asarray(demo.TARGET[1])).max()
    assert d0 < 0.05 and d1 < 0.05, (out["v0"], out["v1"])
    assert out["loss"][-1] < 5e-2 * out["loss"][0]      # (measured: 0.0978 -> 4.6e-4 ... 1.0e-3; the backward's float atomics reorder)


def test_efficient_cuboid_loop_converges(hip_lib):
    demo = load_demo("EfficientCuboidViaOptimization")
    out = demo.run(iters=800, log=lambda s: log_line("[demo] EfficientCuboid: " + s))
    head, tail = np.mean(out["loss"][:50]), np.mean(out["loss"][-50:])
    assert np.isfinite(out["loss"]).all()
    assert tail < 0.7 * head, (head, tail)      # (measured: 0.0360 -> 0.0198 after 800 iterations, 0.0168 after the demo's 3200)


def test_light_diffusion_demo(hip_lib):
    """demo/LightDiffusion.py:52-60: interpolated normals shaded by a directional light, against the oracle's merge of the
    same normals on the rendered fragments."""
    import torch
    import oracle
    demo = load_demo("LightDiffusion")
    out = demo.run(out=None, log=lambda s: log_line("[demo] LightDiffusion: " + s))
    frag, nm = out["frag"], out["normals_map"]
    want = oracle.merge_fwd(out["normals"].cpu().numpy().astype(np.float64), frag.vert_index.cpu().numpy(),
                            frag.vert_weight.cpu().numpy().astype(np.float64), frag.valid_num.cpu().numpy())
    assert np.abs(nm.cpu().numpy() - want).max() < 1e-5
    img = out["image"]
    lit = (img[0].sum(-1) > 0).float().mean().item()
    assert img.shape == (1, 256, 256, 3) and 0.15 < lit < 0.6 and 0.9 < float(img.max()) <= 1.0
    # the light term itself
    n = torch.nn.functional.normalize(nm, dim=-1, eps=1e-6)
    d = torch.nn.functional.normalize(demo.camera_position_from_spherical_angles(1, 30 + 95 * 0.5, 10, device=nm.device), dim=-1)
    assert torch.allclose(img, torch.relu((n * d).sum(-1))[..., None].expand_as(img), atol=1e-6)


def test_render_point_clouds_demo(hip_lib):
    """demo/RenderPointClouds.py:31-45 at the reference's cloud size (438 544 points, default settings: K = 20, heuristic bins)."""
    demo = load_demo("RenderPointClouds")
    out = demo.run(out=None, log=lambda s: log_line("[demo] RenderPointClouds: " + s))
    img, frag = out["image"], out["frag"]
    assert img.shape == (320, 320, 3) and bool(np.isfinite(img.cpu().numpy()).all())
    covered = (frag.valid_num > 0).float().mean().item()
    assert 0.05 < covered < 0.9
    assert float(img.min()) < 0.9 and float(img.max()) == 1.0      # the cloud's colours on the white background
