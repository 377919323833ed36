"""CPU tests of the host-side mirror of the reference API (no kernels run here)."""
import os

import numpy as np
import pytest
import torch

from oracle import camera_np
from util import GOLDEN


def test_settings_match_reference_defaults():
    from voge_amd.Renderer import GaussianRenderSettings
    g = np.load(os.path.join(GOLDEN, "misc_api.npz"))
    st = GaussianRenderSettings(image_size=128, batch_size=-1, principal_point=(1, 2))   # unknown kwargs swallowed
    got = [st["image_size"][0], st["image_size"][1], st["max_assign"], st["thr_activation"], st["absorptivity"],
           float(st["inverse_sigma"])]
    assert np.allclose(got, g["settings_default"])
    assert st["principal"] is None and st["max_point_per_bin"] is None
    with pytest.raises(AttributeError):
        st["no_such_field"]
    st2 = GaussianRenderSettings(image_size=(64, 96), max_assign=7)
    assert st2.image_size == (64, 96) and st2["max_assign"] == 7


def test_expend_sigma_matches_reference():
    from voge_amd.Aggregation import expend_sigma
    g = np.load(os.path.join(GOLDEN, "misc_api.npz"))
    assert np.array_equal(expend_sigma(torch.tensor([1.5, 2.5])).numpy(), g["expend_1"])
    assert np.array_equal(expend_sigma(torch.tensor([[1., 2., 3.], [4., 5., 6.]])).numpy(), g["expend_2"])
    assert np.array_equal(expend_sigma(torch.arange(18.).view(2, 3, 3)).numpy(), g["expend_3"])
    with pytest.raises(Exception):
        expend_sigma(torch.zeros(2, 3, 4))


def test_fragments_container():
    from voge_amd.Renderer import Fragments
    f = Fragments(torch.zeros(2, 3, 4, 5), torch.zeros(2, 3, 4, 5, dtype=torch.int32), torch.zeros(2, 3, 4, dtype=torch.int64),
                  torch.zeros(2, 3, 4, 5))
    assert len(f) == 2 and f[1].vert_weight.shape == (3, 4, 5)
    assert f[0].unsqueeze().shape == f[0:1].shape
    assert set(f.to_dict()) == {"vert_weight", "vert_index", "valid_num", "vert_hit_length"}
    with pytest.raises(AssertionError):
        f.squeeze()
    with pytest.raises(AssertionError):
        f[0][0]


def test_meshes_containers():
    from voge_amd.Meshes import GaussianMeshes, GaussianMeshesNaive
    v, s = torch.rand(5, 3), torch.rand(5)
    m = GaussianMeshesNaive(v, s)
    assert m()[0] is v and m()[2] is None and m[1:3].verts.shape == (2, 3)
    gm = GaussianMeshes(v.clone(), s.clone(), gradianted_args=[True, False, False])
    assert len(gm.grad_parameters()) == 1 and gm()[1].requires_grad is False and gm.gradianted_args[2] is False


def test_look_at_matches_oracle_convention():
    from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
    R, T = look_at_view_transform(dist=[6.0, 2.7], elev=[10.0, -35.0], azim=[70.0, 200.0])
    R2, T2 = camera_np.look_at_view_transform([6.0, 2.7], [10.0, -35.0], [70.0, 200.0])
    assert np.abs(R.numpy() - R2).max() < 1e-6 and np.abs(T.numpy() - T2).max() < 1e-5
    cam = PerspectiveCameras(focal_length=300, principal_point=((128, 128),), R=R[:1], T=T[:1], image_size=((256, 256),))
    assert not cam.in_ndc() and cam.focal_length.shape == (1, 2)
    C = cam.get_camera_center().numpy()
    assert np.abs(C[0] - 6 * np.array([np.cos(np.deg2rad(10)) * np.sin(np.deg2rad(70)), np.sin(np.deg2rad(10)),
                                       np.cos(np.deg2rad(10)) * np.cos(np.deg2rad(70))])).max() < 1e-5


def test_bin_size_and_threshold_rules():
    import oracle
    from voge_amd.RayTracing import default_bin_size
    for size in ((256, 256), (128, 128), (512, 512), (1024, 1024), (256, 672), (400, 400)):
        assert default_bin_size(size) == oracle.bin_size_of(size)
    assert default_bin_size((256, 256)) == 10 and default_bin_size((512, 512)) == 16 and default_bin_size((1024, 1024)) == 32
    assert abs(oracle.thr_act_of(0.01) - 4.6051702) < 1e-6 and abs(oracle.thr_act_of(0.0) - 23.02585) < 1e-5


def test_renderer_refuses_cpu_tensors():
    """There is no CPU fallback: the product path raises when handed host tensors."""
    from voge_amd import _lib
    from voge_amd.Meshes import GaussianMeshesNaive
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings
    from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
    R, T = look_at_view_transform(3.0, 0.0, 0.0)
    cam = PerspectiveCameras(focal_length=30.0, principal_point=((8.0, 8.0),), image_size=((16, 16),))
    r = GaussianRenderer(cam, GaussianRenderSettings(image_size=16, max_assign=4, max_point_per_bin=-1))
    with pytest.raises(_lib.VogeHipError, match="no CPU fallback"):
        r(GaussianMeshesNaive(torch.rand(10, 3), torch.rand(10) * 50), R=R, T=T)


def test_row_bands_partition_the_image():
    from voge_amd.distributed import row_band
    for H in (1, 7, 128, 512, 1000):
        for world in (1, 2, 3, 4, 8):
            bands = [row_band(H, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == H
            assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
            sizes = [b[1] - b[0] for b in bands]
            assert max(sizes) - min(sizes) <= 1


def test_host_logic_against_the_running_reference():
    """tests/golden/host_logic.npz: GaussianRenderer.forward (Renderer.py:102-150) and ray_tracing (RayTracing.py:12-30)
    of the IMPORTED reference were run with a recording stand-in for _C.ray_trace_voge_fine.  What the reference handed
    its kernel -- bin_size, thr_act, the shape of the "-1" list, camera-centred means, 2*sigma / 2*inverse(sigma) --
    equals what this repo's host rules and its oracle produce, and the oracle's frame reproduces the Fragments the
    reference's aggregation made from the recorded call."""
    import oracle
    from oracle import camera_np
    from voge_amd.RayTracing import default_bin_size
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_logic.npz"))
    for c in ("a", "b"):
        H, W = (int(x) for x in g[c + "_size"])
        K, thr, occ = int(g[c + "_K"]), float(g[c + "_thr"]), float(g[c + "_occ"])
        assert default_bin_size((H, W)) == int(g[c + "_k_bin_size"])
        assert abs(oracle.thr_act_of(thr) - float(g[c + "_k_thr_act"])) < 1e-12
        B, N = g[c + "_R"].shape[0], g[c + "_verts"].shape[0]
        bs = int(g[c + "_k_bin_size"])
        # the reference sizes both bin axes by the image WIDTH (RayTracing.py:25) and lists all P Gaussians per bin
        assert tuple(g[c + "_k_bin_shape"]) == (B, (W - 1) // bs + 1, (W - 1) // bs + 1, N)
        rays, origin = camera_np.pixel_rays(g[c + "_R"], g[c + "_T"], float(g[c + "_focal"]), g[c + "_pp"], (H, W))
        mus = (g[c + "_verts"][None] - origin[:, None].astype(np.float32)).astype(np.float32)
        assert np.array_equal(mus.reshape(-1, 3), g[c + "_k_mus"])                       # Renderer.py:130
        sig = camera_np.expand_sigma(g[c + "_sigmas"])
        isg = 2 * np.linalg.inv(sig) if bool(g[c + "_inverse_sigma"]) else 2 * sig      # Renderer.py:133-137
        want = g[c + "_k_isigmas"].reshape(B, N, 3, 3)
        assert np.abs(isg[None] - want).max() <= 2e-5 * np.abs(want).max()
        # the frame: oracle trace (fp64) on the recorded kernel arguments -> composite == the reference's aggregation
        idx, ln, act, dsd = oracle.trace_fwd(g[c + "_k_mus"], g[c + "_k_isigmas"], rays, K, float(g[c + "_k_thr_act"]))
        w, vn = oracle.composite_fwd(idx, act, ln, dsd, occ)
        same = (idx == g[c + "_index"]).all(-1)
        assert same.mean() > 0.995 and (vn[same] == g[c + "_valid_num"][same]).all()
        assert np.abs(w[same] - g[c + "_weight"][same]).max() < 2e-4
        hit = (idx >= 0) & same[..., None]
        assert np.abs(ln[hit] - g[c + "_hit_length"][hit]).max() < 1e-4 * 4
