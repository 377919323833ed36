"""CPU tests of the callers / data formats either side of the path: converters
and OFF/GOFF IO against fixtures produced by the imported reference, and the `VoGE` import names."""
import os

import numpy as np
import torch

from util import GOLDEN


def test_voge_import_names_alias_the_implementation():
    import voge_amd
    import VoGE
    from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings, interpolate_attr, to_white_background  # noqa: F401
    from VoGE.Converter import Cuboid, Converters, IO  # noqa: F401
    from VoGE.Converter.IO import load_goff, load_off, to_torch  # noqa: F401
    from VoGE.Meshes import GaussianMeshes, GaussianMeshesNaive  # noqa: F401
    from VoGE.Utils import rotation_theta  # noqa: F401
    from VoGE.RayTracing import ray_tracing, ray_tracing_fine  # noqa: F401
    from VoGE.Aggregation import aggregation, merge_final, expend_sigma  # noqa: F401
    assert VoGE.Renderer is voge_amd.Renderer and VoGE.Converter.Cuboid is voge_amd.Converter.Cuboid


def test_cuboid_gauss_matches_reference():
    from voge_amd.Converter import Cuboid
    g = np.load(os.path.join(GOLDEN, "misc_api.npz"))
    v, s = Cuboid.cuboid_gauss((-1, 1), (-1, 1), (-1, 1), 1000, percentage=0.6)
    assert v.shape == (866, 3)
    assert np.abs(v - g["cuboid_verts"]).max() < 1e-6 and np.abs(s / g["cuboid_isigma"] - 1).max() < 1e-6
    v2, s2, c2 = Cuboid.cuboid_gauss((-1, 2), (0, 1), (-0.5, 0.5), 300, percentage=0.5, colors=np.arange(18.).reshape(6, 3))
    assert np.array_equal(v2, g["cuboid2_verts"]) and np.allclose(s2, g["cuboid2_isigma"]) and np.array_equal(c2, g["cuboid2_colors"])
    obj = Cuboid.cuboid_gauss((-1, 1), (-1, 1), (-1, 1), 1000, percentage=0.6, as_obj=True)
    assert obj.verts.shape == (866, 3) and obj.sigmas.dtype == torch.float32


def test_naive_vertices_converter_matches_reference_on_the_bunny():
    from voge_amd.Converter.Converters import fixed_pointcloud_converter, naive_vertices_converter
    b = np.load(os.path.join(GOLDEN, "bunny_gaussians.npz"))
    v, s, r = naive_vertices_converter(b["verts"], b["faces"], percentage=0.6)
    assert r is None and np.abs(s / b["isigma"] - 1).max() < 1e-5
    vt, st, _ = naive_vertices_converter(torch.from_numpy(b["verts"]), torch.from_numpy(b["faces"]), percentage=0.6, max_sig_rate=2)
    assert st.dtype == torch.float32 and st.max() <= 2 * np.mean(s) * (1 + 1e-5)
    _, s2, _ = fixed_pointcloud_converter(np.zeros((4, 3), np.float32), 0.1, percentage=0.5)
    assert np.allclose(s2, 1 / (0.01 / (2 * np.log(2)) + 1e-10))


def test_off_goff_round_trip(tmp_path):
    from voge_amd.Converter import IO
    rng = np.random.default_rng(0)
    v = rng.normal(size=(7, 3)).astype(np.float32)
    f = rng.integers(0, 7, (5, 3)).astype(np.int32)
    IO.save_off(str(tmp_path / "a.off"), v, f)
    v2, f2 = IO.load_off(str(tmp_path / "a.off"))
    assert np.allclose(v, v2) and np.array_equal(f, f2)
    col = rng.uniform(size=(7, 3)).astype(np.float32)
    IO.save_off(str(tmp_path / "c.off"), torch.from_numpy(v), torch.from_numpy(f), vert_color=col)
    v3, f3, c3 = IO.load_off(str(tmp_path / "c.off"), to_torch=True)
    assert np.allclose(c3.numpy(), col, atol=1e-6) and np.array_equal(f3.numpy(), f)
    for sig in (rng.uniform(1, 2, 7).astype(np.float32), rng.uniform(1, 2, (7, 3)).astype(np.float32),
                rng.uniform(1, 2, (7, 3, 3)).astype(np.float32)):
        IO.save_goff(str(tmp_path / "g.goff"), v, sig, radians=rng.uniform(size=7).astype(np.float32))
        p, s, r = IO.load_goff(str(tmp_path / "g.goff"))
        assert np.allclose(p, v) and np.allclose(np.squeeze(s), sig) and r.shape == (7,)
