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


def test_cuboid_mesh_and_normal_mesh_converter_match_reference():
    """VoGE/Converter/Cuboid.py:70-159 and Converters.py:35-71 against fixtures the imported reference produced
    (tests/golden/make_golden.py: gen_converters_more; look_at_rotation there is the oracle's restatement of the
    PyTorch3D function, here it is voge_amd.cameras.look_at_rotation)."""
    from VoGE.Converter.Converters import naive_vertices_converter, normal_mesh_converter   # the demos' import line
    from VoGE.Converter.Cuboid import cuboid_gauss, cuboid_mesh                              # noqa: F401
    g = np.load(os.path.join(GOLDEN, "converters_more.npz"))
    v, f = cuboid_mesh((-1, 1), (-1, 1), (-1, 1), 1000)
    assert np.array_equal(v, g["mesh_verts"]) and np.array_equal(f, g["mesh_faces"])
    v2, f2, c2 = cuboid_mesh((-1, 2), (0, 1), (-0.5, 0.5), 300, colors=np.arange(18.).reshape(6, 3))
    assert np.array_equal(v2, g["mesh2_verts"]) and np.array_equal(f2, g["mesh2_faces"]) and np.array_equal(c2, g["mesh2_colors"])
    assert f2.max() < len(v2) and (np.bincount(f2.ravel(), minlength=len(v2)) > 0).all()      # every vertex is in a face
    vv, isg, rad = normal_mesh_converter(v2.astype(np.float64), f2, g["nm_normals"], percentage=0.6, shape_ratio=0.3)
    assert rad is None and isg.shape == (len(v2), 3, 3)
    scale = np.abs(g["nm_isigma"]).max()
    assert np.abs(isg - g["nm_isigma"]).max() < 2e-6 * scale
    # the normal is the flattened axis: n^T A n = shape_ratio * s, any tangent t: t^T A t = s
    n = g["nm_normals"].astype(np.float64)
    s_iso = naive_vertices_converter(v2.astype(np.float64), f2, percentage=0.6)[1]
    ratio = np.einsum("ni,nij,nj->n", n, isg, n) / s_iso
    assert np.abs(np.delete(ratio, 3) / 0.3 - 1).max() < 1e-4
    # vertex 3's normal is parallel to `up`: look_at_rotation degenerates, det = 0, auto_fix makes it isotropic (:61-63)
    assert abs(ratio[3] - 1) < 1e-6
    _, capped, _ = normal_mesh_converter(v2.astype(np.float64), f2, g["nm_normals"], percentage=0.5, shape_ratio=0.5, max_sig_rate=1.5)
    assert np.abs(capped - g["nm_isigma_capped"]).max() < 2e-6 * scale
    vt, it, _ = normal_mesh_converter(torch.from_numpy(v2).float(), torch.from_numpy(f2), torch.from_numpy(g["nm_normals"]))
    assert it.dtype == torch.float32 and it.shape == (len(v2), 3, 3)
