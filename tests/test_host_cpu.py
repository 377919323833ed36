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
