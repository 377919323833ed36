"""world_size-2 test of the pixel-row sharding on CPU (gloo): the host logic that the 8-GPU run
uses with RCCL.  The per-rank "renderer" here is the oracle (test infrastructure), because the
product kernels need a GPU; what is under test is row_band / gather_rows / allreduce_grads."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import camera_np
from util import cuboid_scene


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from voge_amd.distributed import FlatGrads, allreduce_grads, gather_rows, gather_rows_async, row_band
    sc = cuboid_scene()
    R, T = camera_np.look_at_view_transform(sc["dist"], sc["elev"], sc["azim"])
    r0, r1 = row_band(H, rank, world)
    rays, origin = camera_np.pixel_rays(R, T, 60.0, (W / 2, H / 2), (H, W))
    mus = (sc["verts"][None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(sc["sigmas"])).astype(np.float32)[None]
    idx, ln, act, dsd = oracle.trace_fwd(mus, isg, rays[:, r0:r1], 8, oracle.thr_act_of(0.01))
    w, vn = oracle.composite_fwd(idx, act, ln, dsd, 1.0)
    colors = torch.tensor(sc["colors"], dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w)
    valid = torch.tensor(np.arange(8)[None, None, None] < vn[..., None])
    gathered = colors[torch.tensor(np.maximum(idx, 0)).long()]                 # [1,h,W,K,3]
    band = (gathered * (wt * valid)[..., None]).sum(-2)                         # merge of the band
    img = gather_rows(band, H)
    assert img.shape == (1, H, W, 3)
    finish = gather_rows_async(band.detach(), H)          # the overlapped form bench.py uses
    assert torch.equal(finish(), img.detach())
    # explicit (work-balanced) boundaries: same image from very uneven bands
    bounds = [0, 3, H] if world == 2 else None
    if bounds is not None:
        b0, b1 = bounds[rank], bounds[rank + 1]
        full = img.detach()
        assert torch.equal(gather_rows(full[:, b0:b1].contiguous(), H, bounds=bounds), full)
        assert torch.equal(gather_rows_async(full[:, b0:b1].contiguous(), H, bounds=bounds)(), full)
    img[:, r0:r1].sum().backward()          # each rank owns the loss of its band
    local = colors.grad.clone()
    allreduce_grads([colors])
    # the persistent flat buffer bench.py uses: gradients accumulate into views of it, one collective, no copies
    extra = torch.ones(5, dtype=torch.float64, requires_grad=True)
    fg = FlatGrads([colors, extra])
    assert colors.grad.data_ptr() == fg.flat.data_ptr() and float(fg.flat.abs().sum()) == 0.0
    for _ in range(2):          # two steps: zero() really resets the accumulation
        fg.zero()
        band2 = (colors[torch.tensor(np.maximum(idx, 0)).long()] * (wt * valid)[..., None]).sum(-2)
        (band2.sum() + (extra * (rank + 1)).sum()).backward()
        assert torch.equal(colors.grad, local)
        fg.allreduce()
    assert torch.equal(extra.grad, torch.full((5,), 3.0, dtype=torch.float64))      # 1 + 2 over the two ranks
    if rank == 0:
        torch.save({"img": img.detach(), "g": colors.grad.clone(), "g_list": local * 0 + colors.grad}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_frame_equals_single_rank(tmp_path):
    H, W, world = 21, 16, 2          # odd height: uneven bands
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, _free_port(), H, W, out), nprocs=world, join=True)
    got = torch.load(out)
    sc = cuboid_scene()
    R, T = camera_np.look_at_view_transform(sc["dist"], sc["elev"], sc["azim"])
    ref = oracle.render(sc["verts"], sc["sigmas"], sc["colors"], R, T, 60.0, (W / 2, H / 2), (H, W), K=8)
    assert np.abs(got["img"].numpy() - ref["rgb"]).max() < 1e-12
    g_attr, _ = oracle.merge_bwd(sc["colors"], ref["idx"], ref["weight"], ref["valid_num"], np.ones_like(ref["rgb"]))
    assert np.abs(got["g"].numpy() - g_attr).max() < 1e-10


def test_balanced_row_bounds():
    """Work-balanced contiguous bands: equal weight per band, every band non-empty, deterministic."""
    from voge_amd.distributed import balanced_row_bounds, projected_row_weight
    w = torch.zeros(100)
    w[40:60] = 1.0
    b = balanced_row_bounds(w, 4, floor=0.0)
    assert b[0] == 0 and b[-1] == 100 and all(b[i] < b[i + 1] for i in range(4))
    assert b[1:4] == [45, 50, 55]                       # the mass sits in rows 40..59
    assert balanced_row_bounds(torch.zeros(5), 5) == [0, 1, 2, 3, 4, 5]
    assert balanced_row_bounds(torch.tensor([0.0, 0, 0, 100, 0, 0]), 3)[-1] == 6
    uni = balanced_row_bounds(torch.ones(64), 8)
    assert uni == list(range(0, 65, 8))
    # a camera looking down -z at points spread in y: rows of the projected centres
    verts = torch.tensor([[0.0, 0.5, 0.0], [0.0, -0.5, 0.0], [0.0, 0.0, 0.0]])
    R, T = torch.eye(3), torch.tensor([0.0, 0.0, 2.0])
    pw = projected_row_weight(verts, R, T, 20.0, 16.0, 32, smooth=1)
    assert pw.shape == (32,) and float(pw.sum()) == 3.0
    assert pw[10] == 1 and pw[15] == 1 and pw[20] == 1     # row = py - fy*Y/Z - 0.5


def test_measured_rebalancing_converges():
    """distributed.rebalance_bounds: boundaries follow MEASURED band times.  Simulated frame: a per-row cost profile (a
    dense centre, empty borders) plus a fixed cost per band; a few measure -> rebalance rounds level the bands."""
    import numpy as np
    from voge_amd.distributed import rebalance_bounds, row_band
    H, n, fixed = 512, 8, 90.0
    y = np.arange(H)
    row_cost = 0.05 + 2.4 * np.exp(-((y - 270) / 130.0) ** 2)          # us per row
    measure = lambda b: [fixed + row_cost[b[r]:b[r + 1]].sum() for r in range(n)]
    bounds = [row_band(H, r, n)[0] for r in range(n)] + [H]
    t0 = measure(bounds)
    for _ in range(5):
        bounds = rebalance_bounds(bounds, measure(bounds), fixed=0.5 * min(t0), damping=0.8, min_rows=8)
        assert bounds[0] == 0 and bounds[-1] == H and all(b1 - b0 >= 8 for b0, b1 in zip(bounds, bounds[1:]))
    t = measure(bounds)
    assert max(t) < 0.9 * max(t0) and max(t) / (sum(t) / n) < 1.05


# ------------------------------------------------------------------------------- stacked (view, row) sharding of a batch
def test_stacked_bounds_and_segments():
    from voge_amd.distributed import rebalance_bounds, stacked_bounds, stacked_segments
    assert stacked_bounds(8, 512, 8) == [i * 512 for i in range(9)]                  # whole views
    assert stacked_bounds(8, 512, 4) == [i * 1024 for i in range(5)]
    b = stacked_bounds(5, 128, 2)                                                        # 5 views on 2 ranks: 320 rows each
    assert b == [0, 320, 640]
    assert stacked_segments(0, 320, 128) == [(0, 2, 0, 128), (2, 3, 0, 64)]
    assert stacked_segments(320, 640, 128) == [(2, 3, 64, 128), (3, 5, 0, 128)]
    assert stacked_segments(10, 20, 128) == [(0, 1, 10, 20)] and stacked_segments(7, 7, 128) == []
    assert stacked_segments(100, 300, 128) == [(0, 1, 100, 128), (1, 2, 0, 128), (2, 3, 0, 44)]
    for B, H, world in ((3, 7, 2), (5, 16, 8), (1, 64, 8), (8, 4, 3)):
        bd = stacked_bounds(B, H, world)
        rows = sum((b1 - b0) * (r1 - r0) for r in range(world) for b0, b1, r0, r1 in stacked_segments(bd[r], bd[r + 1], H))
        assert bd[0] == 0 and bd[-1] == B * H and rows == B * H
    # min_rows larger than H / n must not produce crossing bands
    out = rebalance_bounds([0, 2, 4, 6, 9], [5.0, 1.0, 1.0, 9.0], min_rows=8)
    assert out[0] == 0 and out[-1] == 9 and all(b1 > b0 for b0, b1 in zip(out, out[1:]))


def _stacked_worker(rank, world, port, B, H, W, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from voge_amd.distributed import gather_stacked, render_stacked, stacked_bounds
    sc = cuboid_scene()
    R, T = camera_np.look_at_view_transform([sc["dist"]] * B, [10.0 + 15 * b for b in range(B)], [70.0 + 40 * b for b in range(B)])
    colors = torch.tensor(sc["colors"], dtype=torch.float64, requires_grad=True)
    isg = (2 * camera_np.expand_sigma(sc["sigmas"])).astype(np.float32)

    def render_views(b0, b1, r0, r1):      # the oracle stands in for the renderer (product kernels need a GPU)
        rays, origin = camera_np.pixel_rays(R[b0:b1], T[b0:b1], 60.0, (W / 2, H / 2), (H, W))
        mus = (sc["verts"][None] - origin[:, None].astype(np.float32)).astype(np.float32)
        idx, ln, act, dsd = oracle.trace_fwd(mus, np.broadcast_to(isg[None], (b1 - b0,) + isg.shape), rays[:, r0:r1], 8, oracle.thr_act_of(0.01))
        w, vn = oracle.composite_fwd(idx, act, ln, dsd, 1.0)
        valid = torch.tensor(np.arange(8)[None, None, None] < vn[..., None])
        n_g = sc["verts"].shape[0]
        return (colors[torch.tensor(np.maximum(idx, 0) % n_g).long()] * (torch.tensor(w) * valid)[..., None]).sum(-2)
    bounds = stacked_bounds(B, H, world)
    rows = render_stacked(render_views, bounds[rank], bounds[rank + 1], H)
    assert rows.shape == (bounds[rank + 1] - bounds[rank], W, 3)
    img = gather_stacked(rows, B, H)
    assert img.shape == (B, H, W, 3)
    img.reshape(B * H, W, 3)[bounds[rank]:bounds[rank + 1]].sum().backward()      # each rank owns the loss of its range
    from voge_amd.distributed import allreduce_grads
    allreduce_grads([colors])
    if rank == 0:
        torch.save({"img": img.detach(), "g": colors.grad.clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_stacked_view_row_sharding_equals_single_rank(tmp_path):
    """A 3-view batch over 2 ranks on the stacked (view, row) axis: rank 0 renders view 0 and the top of view 1, rank 1
    the rest; the gathered batch and the all-reduced gradient equal the single-process render."""
    B, H, W, world = 3, 7, 12, 2
    out = str(tmp_path / "r0.pt")
    mp.spawn(_stacked_worker, args=(world, _free_port(), B, H, W, out), nprocs=world, join=True)
    got = torch.load(out)
    sc = cuboid_scene()
    R, T = camera_np.look_at_view_transform([sc["dist"]] * B, [10.0 + 15 * b for b in range(B)], [70.0 + 40 * b for b in range(B)])
    ref = oracle.render(sc["verts"], sc["sigmas"], sc["colors"], R, T, 60.0, (W / 2, H / 2), (H, W), K=8)
    assert np.abs(got["img"].numpy() - ref["rgb"]).max() < 1e-12
    n_g = sc["verts"].shape[0]
    g_attr, _ = oracle.merge_bwd(np.tile(sc["colors"], (B, 1)), ref["idx"], ref["weight"], ref["valid_num"], np.ones_like(ref["rgb"]))
    assert np.abs(got["g"].numpy() - g_attr.reshape(B, n_g, 3).sum(0)).max() < 1e-10


# ------------------------------------------------------------------ round 4: one frame dealt in interleaved stripes
def _stripe_worker(rank, world, port, H, W, stripe_h, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from voge_amd.distributed import FlatGrads, Stripes, gather_stripes, gather_stripes_async
    sc = cuboid_scene()
    R, T = camera_np.look_at_view_transform(sc["dist"], sc["elev"], sc["azim"])
    mine = Stripes(H, rank, world, stripe_h)
    rows = mine.image_rows().numpy()
    rays, origin = camera_np.pixel_rays(R, T, 60.0, (W / 2, H / 2), (H, W))
    mus = (sc["verts"][None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(sc["sigmas"])).astype(np.float32)[None]
    stacked = np.ascontiguousarray(rays[:, rows])                              # the rank's stripes as ONE image of mine.h rows
    idx, ln, act, dsd = oracle.trace_fwd(mus, isg, stacked, 8, oracle.thr_act_of(0.01))
    w, vn = oracle.composite_fwd(idx, act, ln, dsd, 1.0)
    colors = torch.tensor(sc["colors"], dtype=torch.float64, requires_grad=True)
    fg = FlatGrads([colors])
    wt = torch.tensor(w)
    valid = torch.tensor(np.arange(8)[None, None, None] < vn[..., None])
    band = (colors[torch.tensor(np.maximum(idx, 0)).long()] * (wt * valid)[..., None]).sum(-2)      # [1, h, W, 3]
    assert band.shape[1] == mine.h
    img = gather_stripes(band, H, stripe_h)                                    # ONE all_gather + a local row permutation
    assert img.shape == (1, H, W, 3)
    assert torch.equal(gather_stripes_async(band.detach(), H, stripe_h)(), img.detach())
    (img * torch.linspace(0.5, 1.5, H, dtype=torch.float64)[None, :, None, None]).sum().backward()   # a loss on the WHOLE frame
    fg.allreduce()                                                             # ONE all_reduce
    if rank == 0:
        torch.save({"img": img.detach(), "g": colors.grad.clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W,world,stripe_h", [(21, 16, 2, 4), (23, 12, 3, 4)])
def test_stripe_dealt_frame_equals_single_rank(tmp_path, H, W, world, stripe_h):
    """Stripes(H, rank, world, stripe_h): every rank renders its interleaved stripes stacked into one image; the gathered,
    re-ordered frame and the all-reduced gradient of a full-frame loss equal the single-process render.  (H = 21 with
    stripes of 4 rows over 2 ranks: rank 0 gets rows 0-3, 8-11, 16-19, rank 1 rows 4-7, 12-15, 20 -- a cut last stripe and
    unequal stacked heights.  H = 23 over 3 ranks: H is not a multiple of world * stripe_h, the last rank's second stripe is
    cut to 3 rows -- ADVICE r4.)"""
    from voge_amd.distributed import Stripes, stripe_height
    if world == 2:
        s0, s1 = Stripes(H, 0, world, stripe_h), Stripes(H, 1, world, stripe_h)
        assert s0.image_rows().tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 16, 17, 18, 19] and s1.image_rows().tolist() == [4, 5, 6, 7, 12, 13, 14, 15, 20]
    else:
        assert Stripes(H, 2, world, stripe_h).image_rows().tolist() == [8, 9, 10, 11, 20, 21, 22]
    assert sorted(sum((Stripes(H, r, world, stripe_h).image_rows().tolist() for r in range(world)), [])) == list(range(H))
    assert stripe_height(1024, 8) == 32 and stripe_height(64, 8) == 8 and stripe_height(21, 2, 4) == 4 and stripe_height(5, 8) == 1
    out = str(tmp_path / "s0.pt")
    mp.spawn(_stripe_worker, args=(world, _free_port(), H, W, stripe_h, out), nprocs=world, join=True)
    got = torch.load(out)
    sc = cuboid_scene()
    R, T = camera_np.look_at_view_transform(sc["dist"], sc["elev"], sc["azim"])
    rays, origin = camera_np.pixel_rays(R, T, 60.0, (W / 2, H / 2), (H, W))
    mus = (sc["verts"][None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(sc["sigmas"])).astype(np.float32)[None]
    idx, ln, act, dsd = oracle.trace_fwd(mus, isg, rays, 8, oracle.thr_act_of(0.01))
    w, vn = oracle.composite_fwd(idx, act, ln, dsd, 1.0)
    colors = torch.tensor(sc["colors"], dtype=torch.float64, requires_grad=True)
    valid = torch.tensor(np.arange(8)[None, None, None] < vn[..., None])
    img = (colors[torch.tensor(np.maximum(idx, 0)).long()] * (torch.tensor(w) * valid)[..., None]).sum(-2)
    (img * torch.linspace(0.5, 1.5, H, dtype=torch.float64)[None, :, None, None]).sum().backward()
    assert torch.equal(got["img"], img.detach())                               # pixels are independent: the same bits
    assert torch.allclose(got["g"], colors.grad, rtol=1e-12, atol=1e-14)


def _bad_stripe_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from voge_amd.distributed import Stripes, gather_stripes
    H, W, stripe_h = 4, 8, 4                      # one stripe in all: rank 1 owns no row
    mine = Stripes(H, rank, world, stripe_h)
    band = torch.zeros((1, mine.h, W, 3), dtype=torch.float64)
    msg = ""
    try:
        gather_stripes(band, H, stripe_h)         # (the local check fires on every rank before any collective is entered)
    except AssertionError as e:
        msg = str(e)
    if rank == 0:
        torch.save({"msg": msg}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_stripes_that_leave_a_rank_empty_are_refused(tmp_path):
    """A stripe height that gives some rank no row at all (Stripes.h == 0) must not reach the renderer or the collective:
    gather_stripes names distributed.stripe_height's answer instead."""
    out = str(tmp_path / "bad.pt")
    mp.spawn(_bad_stripe_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    msg = torch.load(out)["msg"]
    assert "without a row" in msg and "stripe_height" in msg
