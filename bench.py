#!/usr/bin/env python3
"""Benchmark of the VoGE hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

A "step" is one forward+backward frame of BASELINE.json's metric config (config 3: 50 000
synthetic Gaussians, 512x512, K=40, max_point_per_bin=-1):
    frag = renderer(gaussians, R=R, T=T); img = to_white_background(frag, colors); img.sum().backward(one)      # one = a 1.0 on the device, made once
with gradients to verts [N,3], sigmas [N] and colours [N,3].  Inputs are resident in HBM before
the timed region.  With --gpus N > 1 (one process per GPU) the frame's pixel rows are sharded over the
ranks -- dealt in interleaved 32-row stripes (voge_amd.distributed.Stripes), so that every rank gets a sample of
the whole image; --row-bands: one contiguous band per rank with measured rebalancing, rounds 1-2's scheme --: each
rank traces / composites its own rows, the image is assembled by ONE all-gather (RCCL) and the per-Gaussian
gradients by ONE all-reduce -> total work is fixed, "scaling": "strong".  --views runs the weak-scaling batch of
N views instead (one whole view per rank); both alternatives and config 4 are also reported under `variants`.
`python bench.py --gpus N` from a bare shell starts its own N ranks (torch.distributed.run as a child
process, before this process touches a GPU); under torchrun it is one of the ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import gc
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L3_BYTES = 256 << 20   # Infinity Cache: stage timings rotate over buffer sets larger than this
TRACE_KERNELS = "voge_trace_topk_fwd(_iso) = binA + binB + sweep_iso_kernel<false> (scalar sigmas; <true> for 3x3 forms); the cone hierarchy (super-tiles, quads, tiles) comes with the rays"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3_50k_512")
    ap.add_argument("--anisotropic", action="store_true", help="full [N,3,3] Sigma^-1 (L L^T) instead of scalar sigmas")
    ap.add_argument("--default-bins", action="store_true", help="max_point_per_bin=None (the demos' default) instead of -1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--diagonal", action="store_true", help="(N,3) per-axis Sigma^-1 instead of scalar sigmas")
    ap.add_argument("--no-variants", action="store_true", help="skip the anisotropic / default-bins variant frames")
    ap.add_argument("--balance-rounds", type=int, default=4,
                    help="multi-GPU: measure -> rebalance rounds of the row bands before the timed region (0: equal-height bands)")
    ap.add_argument("--only-stage", default=None,
                    help="profiling aid: launch nothing but this stage's entry point (no frame), e.g. trace_fwd for the PMC passes; "
                         "'frame': the frame's steps alone")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a HIP graph")
    ap.add_argument("--graph-steps", type=int, default=0, help="consecutive steps captured per HIP graph (0: the largest of 30, 20, 15, 10, 6, 5, 4, 3, 2, 1 that divides --steps)")
    ap.add_argument("--no-launch-probe", action="store_true", help="time the HIP graph replay whatever the eager launches of the same step would do")
    ap.add_argument("--split-graph", action="store_true",
                    help="single GPU: use the multi-GPU launch scheme (forward graph / eager exchange / backward graph)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target seconds of CPU oracle work for cpu_baseline")
    ap.add_argument("--row-bands", action="store_true",
                    help="multi-GPU: ONE frame as one CONTIGUOUS pixel-row band per rank with measured rebalancing (rounds 1-2's "
                         "scheme) instead of the default interleaved stripes; strong scaling either way")
    ap.add_argument("--views", action="store_true",
                    help="multi-GPU: WEAK scaling -- a batch of N views of the same Gaussians (a multi-view training iteration), "
                         "one whole view per rank.  Default for --gpus N > 1: STRONG scaling -- ONE frame of the metric config, "
                         "its pixel rows dealt to the ranks in interleaved stripes")
    ap.add_argument("--loop", action="store_true",
                    help="BASELINE config 5 as a loop: a step = ONE ShapeFitting iteration (5 views batched, interpolate_attr + "
                         "get_silhouette MSE losses, backward, SGD step); implies --config cfg5_shapefit_128")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: run N ranks of this script under torch.distributed.run as a
    CHILD process and pass its output through.  Nothing here has touched a GPU yet (no torch.cuda call), and the
    parent only waits: no process that initialised HIP is ever replaced."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def stage_bytes(P, npix, K, C=3, iso=False):
    """ALGORITHMIC HBM bytes per launch of each stage, fp32/int32.  iso: the Gaussians
    enter as (mu, a) = 16 bytes instead of (mu, A) = 48."""
    g = 16 if iso else 48
    return {
        "trace_fwd": P * g + npix * 12 + npix * K * 16,
        "composite_fwd": npix * K * 12 + npix * 4 + npix * K * 4 + npix * 8,     # act,len,dsd + hit count -> weight, valid_num
        # fused merge+silhouette+blend: read idx,w + colours, write rgb,img / read idx,w,rgb,g_img, write g_w,g_col
        "shade_fwd": npix * K * 8 + P * 4 * C + 2 * npix * 4 * C,
        "shade_bwd": npix * K * 8 + 2 * npix * 4 * C + P * 4 * C + npix * K * 4 + P * 4 * C,
        "composite_bwd": npix * K * (12 + 4 + 4) + npix * K * 12,      # act,len,dsd + weight + g_weight -> 3 grads
        "trace_bwd": npix * K * 16 + npix * 12 + P * g + npix * 12 + P * g,
        # the three backward stages as ONE kernel: idx, weight, act, len, dsd once; g_img, rgb, wsum, rays per pixel;
        # colours + (mu, a) records in, (g_verts, g_sigmas, g_colors) out
        "fragment_bwd": npix * K * (12 if iso else 20) + npix * (2 * 4 * C + 4 + 12) + P * (4 * C + g) + P * (4 * C + g),
        # scalar sigmas, trace + composite in one entry point: Gaussians and rays in; idx, len, weight, valid_num, cnt out
        "fragments_fwd": P * 16 + npix * 12 + npix * K * 12 + npix * 12,
        # the renderer's forward with the composite deferred: Gaussians and rays in; idx, len, cnt, records out
        "trace_lean_fwd": P * 16 + npix * 12 + npix * K * 8 + npix * 4 + P * 16,
        # composite + shade in one pass: idx, len, cnt, rays, records, colours in; weight, valid_num, rgb, img, wsum out
        "composite_shade_fwd": npix * K * 8 + npix * 4 + npix * 12 + P * 16 + P * 4 * C + npix * K * 4 + npix * 8 + 2 * npix * 4 * C + npix * 4,
        # round 6, the frame path (ABI 7).  The trace from the camera: Gaussians (verts 12 + sigma 4) in; idx, len, cnt, records AND the
        # ray bundle out (no rays in: they are made in registers)
        "frame_trace_fwd": P * 16 + npix * K * 8 + npix * 4 + P * 16 + npix * 12,
        # composite_shade_fwd + the zeroed accumulator of the backward (32 bytes per Gaussian)
        "frame_shade_fwd": (npix * K * 8 + npix * 4 + npix * 12 + P * 16 + P * 4 * C + npix * K * 4 + npix * 8 + 2 * npix * 4 * C + npix * 4
                            + P * 32),
        # fragment_bwd without its fill launch: the same algorithmic bytes
        "frame_shade_bwd": npix * K * 12 + npix * (2 * 4 * C + 4 + 12) + P * (4 * C + 16) + P * (4 * C + 16),
    }


def valu_table():
    """Per dominant kernel {VALU wave-instructions (M), VALU utilisation, share of wave cycles in s_waitcnt} from the newest
    profiles/r*_valu_utilisation.txt (tools/valu_util.py over rocprofv3 --pmc SQ passes of this command; counters cannot be read
    from inside the run).  The LAST block of the file is the round's final build."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_utilisation.txt")), key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
    if not files:
        return None
    rows, out = [], {}
    for line in open(files[-1]):
        if line.startswith("=="):
            rows = []
        else:
            rows.append(line)
    for line in rows:
        m = re.match(r"(\S.*?)\s+VALU\s+([\d.]+) M.*?busy\s+([\d.]+) us.*?VALU utilisation ([\d.]+).*?waiting \(s_waitcnt\) ([\d.]+)", line)
        if m:
            out[m.group(1).strip()] = {"valu_Minst": float(m.group(2)), "busy_us": float(m.group(3)), "valu_utilisation": float(m.group(4)),
                                       "s_waitcnt_share": float(m.group(5))}
    return {"source": "profiles/" + os.path.basename(files[-1]), "kernels": out}


def loop_bench(args):
    """BASELINE.json config 5 ("ShapeFitting.py end-to-end optimisation loop ... wall-clock vs CPU baseline") as a
    measured loop.  A step = one iteration of demo/ShapeFitting.py (reference loop: demo/ShapeFitting.py:250-296): five of
    twenty views, rendered as one batched call, image = interpolate_attr(frag, colours), silhouette = get_silhouette(frag),
    two MSE losses, backward (one kernel), SGD(lr 0.8, momentum 0.9) step on the vertices and the colours.
    The cost of an iteration is NOT constant along the optimisation: the sphere contracts onto the target, its Gaussians
    (fixed sigmas) overlap more and more, every pixel's list fills to K and the tiles' candidate lists grow -- a late
    iteration costs ~5x an early one.  `value` is therefore taken over the reference's WHOLE loop (Niter = 2000, rgb loss from
    iteration 400, ShapeFitting.py:237,276; `--steps` other than the default 30 overrides the length), replayed as a HIP
    graph; the first 300 iterations and the eager / one-view-at-a-time forms are reported beside it."""
    import importlib.util
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the VoGE hot path has no CPU fallback")
    spec = importlib.util.spec_from_file_location("shape_fitting_demo", os.path.join(ROOT, "demo", "ShapeFitting.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    B, V, K, size = 5, 20, 25, 128
    niter = 2000 if args.steps == 30 else max(args.steps, 1)
    rgb_on = min(400, niter // 5)
    out, first, final = {}, {}, None
    for name, kw in (("graph", dict(graph=True)), ("eager", dict()), ("per_view_eager", dict(per_view=True))):
        if name == "graph" and args.no_graph:
            continue
        demo.fit(iters=200, quiet=True, rgb_on=0, **kw)                           # settle: pools, code objects, clocks
        h = demo.fit(iters=min(300, niter), quiet=True, rgb_on=rgb_on, **kw)
        first[name] = h["sec_per_iter"] * 1e3
        h = demo.fit(iters=niter, quiet=True, rgb_on=rgb_on, **kw)
        out[name] = h["sec_per_iter"] * 1e3
        sil = np.asarray(h["silhouette"])
        assert np.isfinite(sil).all() and sil[-10:].mean() < 0.2 * sil[:10].mean(), "the loop must run and descend"
        if final is None:
            final = h
    best = "graph" if "graph" in out else "eager"
    ms = out[best]
    N = 2562
    result = {
        "metric": "ShapeFitting iterations/sec (BASELINE config 5: 5 views fwd+bwd + SGD step per iteration)",
        "value": 1e3 / ms, "unit": "iterations/s", "n_gpus": 1, "steps": niter, "warmup": 200, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "frames_per_s": B * 1e3 / ms, "loop_wall_clock_s": round(ms * niter / 1e3, 3),
        "config": {"workload": f"cfg5_shapefit_128 loop: {N} Gaussians (ico-sphere 4), {size}x{size}, K={K}, {B} of {V} views per "
                               f"iteration as one batch, interpolate_attr + get_silhouette, MSE losses (rgb from iteration {rgb_on}), "
                               f"SGD momentum step; {niter} iterations from the unit sphere",
                   "launch": {"graph": "hip graph replay of the whole iteration (views gathered on the device)",
                              "eager": "eager"}[best], "parallelism": "1 gpu"},
        "ms_per_iteration": {k: round(v, 4) for k, v in out.items()},
        "ms_per_iteration_first_300": {k: round(v, 4) for k, v in first.items()},
        "ms_per_iteration_note": "graph / eager: the batched iteration (demo/ShapeFitting.py BatchedIteration); per_view_eager: one "
                                 "renderer call per view as the reference's loop is written (ShapeFitting.py:258-259).  Whole loop vs "
                                 "its first 300 iterations: the iteration gets costlier as the shape contracts (see docstring)",
        "final_losses": {"silhouette": float(np.mean(final["silhouette"][-20:])), "rgb": float(np.mean(final["rgb"][-20:]))},
    }
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_loop(demo, B, K, size, final)
    print(json.dumps(result))


def cpu_baseline_loop(demo, B, K, size, final):
    """One ShapeFitting iteration (5 views: trace, composite, merge, silhouette, two MSE losses, the whole backward
    chain, the SGD update) in the CPU oracle (C / OpenMP fp64 port), at BOTH ends of the optimisation -- the unit sphere it
    starts from and the shape the GPU loop ended with -- because an iteration's cost grows along the way; value = 1 / the
    mean of the two."""
    import numpy as np
    import oracle
    from oracle import camera_np
    oracle.build()
    v, _ = demo.ico_sphere(4)
    N = v.shape[0]
    sig = np.asarray(final["sigmas"], np.float32)
    elev, azim = np.linspace(0, 360, 20)[:B], np.linspace(-180, 180, 20)[:B]
    R, T = camera_np.look_at_view_transform([2.7] * B, list(elev), list(azim))
    focal, pp = 126.0 * size / 128.0, (size / 2.0, size / 2.0)
    tgt_rgb = np.zeros((B, size, size, 3))
    tgt_sil = np.zeros((B, size, size))
    thr_act = oracle.thr_act_of(0.01)
    isg = np.ascontiguousarray(np.broadcast_to((2 * camera_np.expand_sigma(sig)).astype(np.float32)[None], (B, N, 3, 3)))

    def iteration(verts, cols):
        rays, origin = camera_np.pixel_rays(R, T, focal, pp, (size, size))
        mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
        idx, ln, act, dsd = oracle.trace_fwd(mus, isg, rays, K, thr_act)
        w, vn = oracle.composite_fwd(idx, act, ln, dsd, 1.0)
        colsB = np.tile(cols, (B, 1))
        rgb = oracle.merge_fwd(colsB, idx, w, vn)
        wsum = w.sum(-1)
        sil = np.minimum(wsum, 1)
        g_rgb = 2 * (rgb - tgt_rgb) / rgb.size
        g_sil = 2 * (sil - tgt_sil) / sil.size * (wsum < 1)
        g_attr, g_w = oracle.merge_bwd(colsB, idx, w, vn, g_rgb)
        live = np.arange(K)[None, None, None] < vn[..., None]
        g_act, g_len, g_dsd = oracle.composite_bwd(act, ln, dsd, g_w + g_sil[..., None] * live, 1.0)
        _, g_mu, _ = oracle.trace_bwd(mus, isg, rays, idx, g_len, g_act, g_dsd)
        return verts - 0.8 * g_mu.reshape(B, N, 3).sum(0).astype(np.float32), cols - 0.8 * g_attr.reshape(B, N, 3).sum(0)

    times = {}
    for label, verts, cols in (("start (unit sphere)", np.asarray(v, np.float32), np.full((N, 3), 0.5, np.float32)),
                               ("end (fitted shape)", np.asarray(final["final_verts"], np.float32), np.asarray(final["final_colors"], np.float32))):
        iteration(verts, cols)
        reps, t0 = 0, time.perf_counter()
        while reps < 2 or time.perf_counter() - t0 < 5.0:
            iteration(verts, cols)
            reps += 1
        times[label] = (time.perf_counter() - t0) / reps
    dt = sum(times.values()) / len(times)
    return {"value": 1.0 / dt, "unit": "iterations/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"the whole iteration ({B} views of {size}x{size}, {N} Gaussians, K={K}: trace, composite, merge, silhouette, "
                      f"losses, backward chain, SGD update), oracle/voge_oracle.c fp64 with OpenMP ({os.cpu_count()} threads), at the "
                      "two ends of the optimisation: " + ", ".join(f"{k} {v * 1e3:.0f} ms" for k, v in times.items()) + "; value = 1 / mean"}


def main():
    args = parse()
    if args.loop:
        return loop_bench(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # `multi`: the N > 1 code path.  VOGE_BENCH_FORCE_DIST=1 takes it with ONE rank too -- process group, split HIP graphs, the
    # all_gather / all_reduce through the real backend (RCCL) -- which is what a single-GPU box can check of it
    # (tests/test_gpu_parity.py::test_bench_one_rank_through_rccl).
    multi = world > 1 or os.environ.get("VOGE_BENCH_FORCE_DIST", "0") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the VoGE hot path has no CPU fallback")
    # VOGE_BENCH_BACKEND=gloo lets several ranks share one GPU (a functional check of the multi-GPU
    # launch scheme on a single-GPU box); the real runs use RCCL, one rank per GPU.
    backend = os.environ.get("VOGE_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from voge_amd import _lib, ops, scenes
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, get_silhouette, interpolate_attr, to_white_background
    from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
    from types import SimpleNamespace
    from voge_amd.distributed import (FlatGrads, Stripes, gather_rows, gather_rows_async, gather_stripes, gather_stripes_async,
                                      rebalance_bounds, row_band, stripe_height)
    _lib.load()

    N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[args.config]
    def measure(mode, primary=True, cfg=None):
        """Set up `mode` (see below) on config `cfg` (default: --config), build the step (HIP graph(s)), settle, and time
        args.steps steps.  Returns every local (the frame factory, the cameras, the partition, the timings) as a namespace."""
        N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[cfg or args.config]
        # Multi-GPU modes (`mode`):
        #  "stripes" (default): north_star's split -- ONE frame of the metric config, its pixel rows dealt to the ranks in
        #     interleaved 32-row stripes (distributed.Stripes: balanced by construction), ONE all_gather for the image, ONE
        #     all_reduce for the gradients; total work fixed: "scaling": "strong".
        #  "row_bands": the same frame as one contiguous band per rank with measured rebalancing (rounds 1-2's scheme).
        #  "views": a batch of `world` views of the SAME Gaussians -- what a multi-view training iteration renders -- one whole
        #     view per rank (distributed.stacked_bounds); per-GPU work fixed: "scaling": "weak", value = all ranks' frames/s.
        by_views = multi and mode == "views"
        striped = multi and mode == "stripes"
        stripe_h = stripe_height(H, world) if striped else None
        if by_views:
            Rv, Tv = look_at_view_transform(dist=[dd] * world, elev=[el] * world, azim=[az + 360.0 / world * r for r in range(world)], device=dev)
            R, T = Rv[rank:rank + 1].contiguous(), Tv[rank:rank + 1].contiguous()
        else:
            R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
        # (the cameras carry the view too: the stand-alone stage timings below build their rays from `cams` alone)
        cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), R=R, T=T, device=dev)
        # row bands of the ranks: band r = rows [bounds[r], bounds[r + 1]); equal heights to start with
        bands = {"bounds": [row_band(H, r, world)[0] for r in range(world)] + [H]} if (multi and mode == "row_bands") else None
        rows = (bands["bounds"][rank], bands["bounds"][rank + 1]) if bands is not None else None
        my_stripes = Stripes(H, rank, world, stripe_h) if striped else None

        def rows_kw():
            if striped:
                return {"rows": my_stripes}
            return {} if bands is None else {"rows": (bands["bounds"][rank], bands["bounds"][rank + 1])}
        # what the gather sees: the rank's rows of the stacked image (views mode: view r = rows [r H, (r + 1) H) of world x H)
        H_all = world * H if by_views else H
        stack_bounds = [r * H for r in range(world + 1)] if by_views else None

        renderer_of = [None]      # (the renderer of the frame made last)

        def make_frame(anisotropic, default_bins, pattern="white_background"):
            """(step function, parameters) of one forward+backward frame of the config.  pattern: the metric's
            to_white_background image, or the reference training loops' interpolate_attr + get_silhouette pair."""
            verts, sig, cols = scenes.random_gaussians(N, seed=0, anisotropic=anisotropic)
            gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
            colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
            settings = GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1,
                                              max_point_per_bin=None if default_bins else -1)
            renderer = GaussianRenderer(cams, settings).to(dev)
            renderer_of[0] = renderer
            params = [gm.verts, gm.sigmas, colors]

            def fwd():
                frag = renderer(gm, R=R, T=T, **rows_kw())
                if pattern == "white_background":
                    return to_white_background(frag, colors)
                # (one tensor for the step's single .sum(): two separate reductions measured slower than the concatenation,
                # 2866 against 2957 frames/s)
                return torch.cat((interpolate_attr(frag, colors), get_silhouette(frag).unsqueeze(-1)), dim=-1)
            return fwd, params, gm, colors, (verts, sig, cols)

        def warm_side_stream(fn):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()

        graph_checked = []      # (one entry per captured frame whose replayed gradients were compared with the eager step's)

        def graphed_step(fwd, params):
            """One frame as a HIP graph replay (every replay runs exactly the kernels of an eager step on the same static
            tensors); eager on capture failure or --no-graph.  Returns (callable, launch description)."""
            one = torch.ones((), dtype=torch.float32, device=dev)      # d loss / d loss, made once: backward() without it fills a fresh
                                                                       # one-element tensor every step (a 4.8 us launch at its floor)
            def step():
                for p in params:
                    p.grad = None
                fwd().sum().backward(one)
            if args.no_graph:
                for _ in range(60):      # allocator pools, lazily loaded code objects, clocks: what graph capture warms on the way
                    step()
                torch.cuda.synchronize()
                return step, "eager"
            try:
                warm_side_stream(step)
                torch.cuda.synchronize()
                want = [p.grad.detach().clone() for p in params]      # (the eager step's gradients: what a replay must reproduce)
                # U consecutive steps per graph: two graph launches are 9 us apart on a fast host and 25 us on a slow one (a launch
                # waits for the one before it through a completion signal the host's runtime thread forwards -- tools/frame_gaps.sh,
                # profiles/r6_frame_gaps.txt), kernels inside a graph follow each other without a gap.  `replay` below is called once
                # per STEP and launches the graph on every U-th call: any K calls with K % U == 0 run exactly K steps on the GPU.
                # (a graph of at most ~8 ms: thirty steps of the 1.2 ms cfg4 frame in one graph replayed at 1.6 ms per step -- its
                #  intermediates no longer stay in the memory-side cache from one step to the next)
                torch.cuda.synchronize()
                t_e = time.perf_counter()
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                u_cap = max(1, int(8e-3 / max((time.perf_counter() - t_e) / 5, 1e-6)))
                U = (args.graph_steps if args.graph_steps > 0 and args.steps % args.graph_steps == 0 else
                     max(u for u in (30, 20, 15, 10, 6, 5, 4, 3, 2, 1) if args.steps % u == 0 and (u <= u_cap or u == 1)))
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(U):
                        step()
                for _ in range(3):                                     # back-to-back replays, no host synchronisation between them
                    graph.replay()
                torch.cuda.synchronize()
                worst = 0.0
                for p, w in zip(params, want):
                    err = float((p.grad - w).abs().max()) / max(float(w.abs().max()), 1e-30)
                    worst = max(worst, err)
                    # (the only difference between two runs of the same step is the order of the backward's float atomics:
                    #  measured 1e-7 .. 2e-6 of the largest entry; north_star's tolerance is the bound)
                    assert err < 1e-4, f"HIP graph replay does not reproduce the eager step's gradients (rel. error {err:.2e})"
                graph_checked.append(worst)
                graphed_step.eager = step      # (the same step launched eagerly: see the launch probe in front of the timed region)
                pending = [0]

                def replay():
                    pending[0] += 1
                    if pending[0] == U:
                        pending[0] = 0
                        graph.replay()
                replay.U = U
                return replay, ("hip graph replay" if U == 1 else f"hip graph replay, {U} consecutive steps per graph launch")
            except Exception as e:  # pragma: no cover - depends on the runtime
                print(f"[bench] HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                torch.cuda.synchronize()
                return step, "eager"

        def barrier():
            if multi:
                dist.barrier()
            torch.cuda.synchronize()

        def timed(run, steps, warmup):
            # (a graph of U steps per launch: the region is made of whole launches -- `steps` is rounded up to a multiple of U, which
            #  the contract's K already is; timed.last_steps says how many steps the region ran)
            steps = -(-steps // getattr(run, "U", 1)) * getattr(run, "U", 1)
            timed.last_steps = steps
            gc.collect()      # (a collection walks the whole heap: tens of milliseconds in which the GPU idles and drops its clocks, and
            gc.disable()      #  the host's caches are cold afterwards -- a region timed right behind one read 0.264 ms where the same
            for _ in range(100):      #  callable behind 30 steps of work read 0.252.  So: collect, switch the collector off as timeit
                run()         #  does, put the GPU and the host back under load with 100 untimed steps, THEN the contract's W warm-up
            for _ in range(warmup):      # steps, the barrier and the K timed steps.)
                run()
            barrier()
            try:
                t0 = time.perf_counter()
                for _ in range(steps):
                    run()
                barrier()
                dt = time.perf_counter() - t0
            finally:
                gc.enable()
            timed.last_local_dt = dt      # (this rank's own clock; the line reports min / max over the ranks)
            if multi:
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            return dt

        fwd, params, gm, colors, host_scene = make_frame("diag" if args.diagonal else args.anisotropic, args.default_bins)
        if not primary:
            del host_scene
            host_scene = None
        balance_log = []
        if primary and args.only_stage and args.only_stage != "frame":      # ("frame": the frame alone, no stand-alone stage calls)
            run, launch = (lambda: None), "none (--only-stage)"
        elif not multi and not args.split_graph:
            run, launch = graphed_step(fwd, params)
        else:
            # Multi-GPU: the local compute is two HIP graphs -- band forward, band backward -- and the two exchanges
            # (all_gather of image rows, asynchronous, overlapping the backward graph; all_reduce of the gradients) run
            # eagerly, so no collective is ever captured.  The loss is sum(image): every rank owns the loss of its band,
            # whose upstream gradient is a constant tensor of ones.  The gradients live in ONE persistent flat buffer
            # (FlatGrads): the backward graph accumulates into views of it, the all_reduce runs on it in place.
            flat = FlatGrads(params)

            def eager_once():
                flat.zero()
                b = fwd()
                torch.autograd.backward(b, torch.ones_like(b))
            # Measured load balancing (stationary scene): every rank times its band's local compute, the times are exchanged
            # once per round and every rank moves the boundaries the same way (distributed.rebalance_bounds).  Setup, untimed.
            if multi and bands is not None and args.balance_rounds > 0:
                fixed = None
                side = torch.cuda.Stream()      # (like every eager run in front of a capture: never on the default stream)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for it in range(args.balance_rounds + 1):
                        for _w in range(3):
                            eager_once()
                        side.synchronize()
                        t0 = time.perf_counter()
                        for _w in range(8):
                            eager_once()
                        side.synchronize()
                        mine = (time.perf_counter() - t0) / 8 * 1e6
                        times = [None] * world
                        dist.all_gather_object(times, float(mine))
                        balance_log.append({"bounds": list(bands["bounds"]), "band_us": [round(t) for t in times]})
                        if it == args.balance_rounds:
                            break
                        if fixed is None:
                            fixed = 0.5 * min(times)      # what a nearly empty band still costs: latency, not work
                        bands["bounds"] = rebalance_bounds(bands["bounds"], times, fixed=fixed, damping=0.8, min_rows=8)
                # the best partition actually measured (the equal-height start included): noise cannot make things worse
                best = min(balance_log, key=lambda e: max(e["band_us"]))
                bands["bounds"] = list(best["bounds"])
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                rows = (bands["bounds"][rank], bands["bounds"][rank + 1])
            gb = stack_bounds if by_views else (None if bands is None else bands["bounds"])
            run, launch = None, "eager"
            if not args.no_graph:
                try:
                    warm_side_stream(eager_once)
                    g_fwd, g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    # thread_local: RCCL's watchdog thread polls events while we capture
                    with torch.cuda.graph(g_fwd, capture_error_mode="thread_local"):
                        band_static = fwd()
                    ones_static = torch.ones_like(band_static)
                    with torch.cuda.graph(g_bwd, pool=g_fwd.pool(), capture_error_mode="thread_local"):
                        flat.zero()
                        torch.autograd.backward(band_static, ones_static)

                    def run_split():
                        g_fwd.replay()
                        finish = (gather_stripes_async(band_static.detach(), H, stripe_h) if striped else      # all_gather starts (no-op on one GPU) ...
                                  gather_rows_async(band_static.detach(), H_all, bounds=gb))
                        g_bwd.replay()                                          # ... and overlaps the band's backward
                        flat.allreduce()                                        # one eager all_reduce, in place
                        return finish().sum()                                   # the full-image loss every rank holds
                    run_split()
                    torch.cuda.synchronize()
                    run = run_split
                    launch = "hip graphs (band forward, band backward) + eager all_gather_into_tensor / all_reduce (flat buffer)"
                except Exception as e:  # pragma: no cover - depends on the runtime
                    print(f"[bench] split HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                    torch.cuda.synchronize()
            if run is None:
                def run():
                    flat.zero()
                    band = fwd()
                    if striped:
                        img = gather_stripes(band, H, stripe_h)
                        img.index_select(1, my_stripes.image_rows(dev)).sum().backward()
                    else:
                        img = gather_rows(band, H_all, bounds=gb)
                        r0, r1 = (rank * H, (rank + 1) * H) if by_views else (rows if rows is not None else (0, H))
                        img[:, r0:r1].sum().backward()      # each rank owns the loss of its band / view; grads are summed below
                    flat.allreduce()
                    return img

        # settle (untimed, before the contract's W warmup steps): a fresh box ramps its clocks and pools over the first
        # tenths of a second of load; K steps of a 0.4 ms frame would otherwise be timed on the ramp
        if not multi:
            t_settle = time.perf_counter()
            while time.perf_counter() - t_settle < 0.3:
                for _ in range(20):
                    run()
                torch.cuda.synchronize()
        else:                                  # (collectives inside: every rank must run the SAME number of steps)
            for _ in range(300):
                run()
            torch.cuda.synchronize()
        # Launch probe (untimed, one GPU): a HIP graph replay leaves the GPU idle for ~9 us between two replays (the dependency of
        # a graph launch on the one before it: tools/frame_gaps.sh, profiles/r6_frame_gaps.txt), eager launches follow each other
        # within 0.2 us -- but only if the host enqueues a frame (~165 us of Python and C) faster than the GPU renders it (~250 us),
        # which depends on the box.  Both are the same step on the same tensors; the contract's region runs the faster one.
        launch_probe = None
        if primary and not multi and launch.startswith("hip graph replay") and getattr(graphed_step, "eager", None) is not None and not args.no_launch_probe:
            launch_probe = {}
            np_ = 30 * ((max(args.steps, 30) + 29) // 30)      # (a multiple of every U)
            for name, fn in (("hip graph replay", run), ("eager", graphed_step.eager)):
                launch_probe[name] = round(min(timed(fn, np_, 30) for _ in range(2)) / np_ * 1e3, 5)
            if launch_probe["eager"] < 0.985 * launch_probe["hip graph replay"]:
                run, launch = graphed_step.eager, "eager (same step; faster than its hip graph replay on this host, see launch_probe_ms_per_step)"
                for _ in range(30):
                    run()
                torch.cuda.synchronize()
        # Once, untimed, N > 1 only: the frame the split assembles must be the frame ONE GPU renders -- the same bits (pixels
        # are independent in every stage; stripes / bands / views only choose who computes which rows).  Every rank takes part
        # in the gather; rank 0 renders the whole frame (views mode: every view) alone and compares with torch.equal.
        split_exact = None
        if multi and primary and not (args.only_stage and args.only_stage != "frame"):
            with torch.no_grad():
                band = fwd()
                whole = gather_stripes(band, H, stripe_h) if striped else gather_rows(band, H_all, bounds=gb)
                if rank == 0:
                    rr = renderer_of[0]
                    if by_views:
                        alone = torch.cat([to_white_background(rr(gm, R=Rv[v:v + 1].contiguous(), T=Tv[v:v + 1].contiguous()), colors)
                                           for v in range(world)], dim=1)
                    else:
                        alone = to_white_background(rr(gm, R=R, T=T), colors)
                    split_exact = bool(torch.equal(whole, alone))
                    assert split_exact, (f"the {mode} split over {world} ranks does not reproduce the single-GPU frame: "
                                         f"max |diff| {float((whole - alone).abs().max()):.3e}")
            torch.cuda.synchronize()
        dt = timed(run, args.steps, args.warmup)
        ms = dt / args.steps * 1e3
        # (not part of the contract's number: four more regions of the same K steps, so that the line shows the spread of its
        #  headline -- at cfg3 a region is 20 x 0.26 ms)
        repeats = None
        if primary and not multi:
            repeats = [round(timed(run, args.steps, 0) / args.steps * 1e3, 5) for _ in range(4)]
        fps = (world if by_views else 1) * args.steps / dt      # (views mode: every step renders `world` frames, one per rank)
        # what the line says about the job it ran on: the process group's own view of the world, and every rank's local clock
        # over the same timed region (before the max-reduction above)
        dist_info = None
        if multi:
            mine = [None] * world
            dist.all_gather_object(mine, float(timed.last_local_dt / args.steps * 1e3))
            dist_info = {"world_size": int(dist.get_world_size()), "backend": str(dist.get_backend()),
                         "rank_ms_per_step": {"min": round(min(mine), 4), "max": round(max(mine), 4),
                                              "per_rank": [round(x, 4) for x in mine]},
                         "devices": int(torch.cuda.device_count()),
                         "gathered_frame_equals_single_gpu_render": split_exact}
        return SimpleNamespace(**{k: v for k, v in locals().items() if k != "mode"})


    mode = "views" if args.views else ("row_bands" if args.row_bands else "stripes")
    M = measure(mode if multi else "single")
    (by_views, striped, stripe_h, bands, rows, my_stripes, cams, R, T, make_frame, graphed_step, timed, barrier, renderer_of, fwd, params, gm,
     colors, host_scene, balance_log, launch, dt, ms, fps, rows_kw, graph_checked) = (
        M.by_views, M.striped, M.stripe_h, M.bands, M.rows, M.my_stripes, M.cams, M.R, M.T, M.make_frame, M.graphed_step, M.timed, M.barrier,
        M.renderer_of, M.fwd, M.params, M.gm, M.colors, M.host_scene, M.balance_log, M.launch, M.dt, M.ms, M.fps, M.rows_kw, M.graph_checked)

    sig_kind = "(N,3) per-axis sigmas" if args.diagonal else ("[N,3,3] L L^T sigmas" if args.anisotropic else "scalar sigmas")
    bins_kind = "None (default)" if args.default_bins else "-1"
    result = {
        "metric": "forward+backward frames/sec at 512^2, 50k Gaussians; ray-trace HBM GB/s vs peak",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak" if by_views else "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: {N} random Gaussians ({sig_kind}), {H}x{W}, K={K}, max_point_per_bin={bins_kind}, "
                               f"fwd+bwd (grads to verts, sigmas, colors)",
                   "launch": launch, "graph_replay_gradients_checked_against_eager": bool(graph_checked),
                   "graph_replay_gradient_rel_err": (max(graph_checked) if graph_checked else None),
                   "parallelism": ("1 gpu" if not multi else
                                   f"a batch of {world} views on the stacked (view, row) axis, view first: one whole view per rank; "
                                   f"all_gather(images) + all_reduce(gradients); a step = {world} frames" if by_views else
                                   f"ONE frame, pixel rows dealt to {world} ranks in interleaved stripes of {stripe_h} rows (each rank renders "
                                   f"its stripes stacked into one image); all_gather(image) + all_reduce(grads)" if striped else
                                   f"ONE frame, one contiguous pixel-row band per rank x{world} (measured rebalancing), "
                                   f"all_gather(image)+all_reduce(grads)")},
    }
    if M.launch_probe is not None:
        result["config"]["launch_probe_ms_per_step"] = M.launch_probe      # (untimed, in front of the contract's region: both launch modes of the same step)
    if M.repeats is not None:
        result["repeat_ms_per_step"] = M.repeats      # four more K-step regions behind the contract's one: the headline's spread
    if multi and rank == 0:
        result["config"]["multi_gpu_mode"] = mode
        result["config"].update(M.dist_info)      # world_size / backend as torch.distributed reports them, per-rank ms, the exactness check

    if multi and bands is not None:
        result["config"]["bands"] = list(bands["bounds"])      # rows [b[r], b[r+1]) of rank r, after the measured balancing
        result["band_balance"] = balance_log                      # (setup, untimed) what every round measured
    lib = _lib.load()
    P_ = lambda x: None if x is None else x.data_ptr()
    thr_act = -np.log(0.01 + 1e-10)

    def rotating(make_set, fn_of_set, bytes_per_call, iters=20, warm=3):
        """Average duration (ms) of one call with HIP events on torch's current stream -- the stream every voge_*
        entry point is launched on -- cycling over enough independent buffer sets that no call finds its operands in the
        256 MB Infinity Cache (a replay on ONE set of buffers would: at cfg3 every stage's working set fits)."""
        nset = int(min(8, max(2, -(-3 * L3_BYTES // max(bytes_per_call, 1)))))
        sets = [make_set() for _ in range(nset)]
        for i in range(warm):
            fn_of_set(sets[i % nset])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn_of_set(sets[i % nset])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters, nset

    if rank == 0 and not multi:
        # ---- per-stage kernel timings on the same inputs (HIP events, current stream) --------
        with torch.no_grad():
            from voge_amd.cameras import pixel_rays
            from voge_amd.Aggregation import expend_sigma
            rays, origin = pixel_rays(cams, (H, W))
            cones = ops.cones_of(rays, 1, H, W)
            mus = (gm.verts[None] - origin[:, None]).reshape(-1, 3).contiguous()
            iso = gm.sigmas.dim() == 1     # the renderer keeps (N,) sigmas in scalar form (A = a I)
            isg = (2 * gm.sigmas).contiguous() if iso else (2 * expend_sigma(gm.sigmas)).contiguous()
            if iso:
                sel = ops._RayTraceVoGEIso.apply(mus, isg, rays, None, thr_act, K)
            else:
                sel = ops.ray_trace_fine(mus, isg, rays, None, thr_act, 16, K)
            cnt = ops.hit_count_of(sel[0])
            w, vn = ops.composite(sel[0], sel[2], sel[1], sel[3], 1.0)
            idx = sel[0].clone()
            rgb = ops.merge(colors.detach(), w, idx, vn)
            bg = torch.ones(3, device=dev)
            wsum = w.sum(-1)
            st = torch.cuda.current_stream().cuda_stream
            npix = H * W
            trace_fwd_fn = lib.voge_trace_topk_fwd_iso if iso else lib.voge_trace_topk_fwd
            trace_bwd_fn = lib.voge_trace_bwd_iso if iso else lib.voge_trace_bwd
            nws = lib.voge_trace_workspace_bytes(1, N, H, W)
            nws_b = lib.voge_trace_bwd_iso_workspace_bytes(N) if iso else lib.voge_trace_bwd_workspace_bytes(N)
            E = lambda ref: torch.empty_like(ref)
            C_ = lambda ref: ref.clone()
            g_rand = torch.rand_like(w)
            stage_defs = {
                # name: (make_set, call)
                "trace_fwd": (lambda: dict(ws=torch.empty(nws, dtype=torch.uint8, device=dev), o=[E(x) for x in sel],
                                           c=torch.empty((1, H, W), dtype=torch.int32, device=dev)),
                              lambda s: trace_fwd_fn(P_(mus), P_(isg), P_(rays), None, P_(cones), 1, N, H, W, K, thr_act, P_(s["ws"]), nws,
                                                     P_(s["o"][0]), P_(s["o"][1]), P_(s["o"][2]), P_(s["o"][3]), P_(s["c"]), st)),
                "composite_fwd": (lambda: dict(a=C_(sel[2]), l=C_(sel[1]), d=C_(sel[3]), c=C_(cnt), w=E(w), v=E(vn)),
                                  lambda s: lib.voge_composite_fwd(None, P_(s["c"]), P_(s["a"]), P_(s["l"]), P_(s["d"]), 1.0, npix, K,
                                                                   P_(s["w"]), P_(s["v"]), st)),
                "shade_fwd": (lambda: dict(i=C_(idx), w=C_(w), v=C_(vn), rgb=E(rgb), img=E(rgb), ws=E(wsum)),
                              lambda s: lib.voge_shade_fwd(P_(colors), P_(s["i"]), P_(s["w"]), P_(s["v"]), P_(bg), -1.0, npix, K, 3, N, 1,
                                                           P_(s["rgb"]), P_(s["img"]), None, P_(s["ws"]), st)),
                "shade_bwd": (lambda: dict(i=C_(idx), w=C_(w), v=C_(vn), rgb=C_(rgb), ws=C_(wsum), g=torch.ones_like(rgb),
                                           ga=torch.empty_like(colors), gw=E(w)),
                              lambda s: lib.voge_shade_bwd(P_(colors), P_(s["i"]), P_(s["w"]), P_(s["v"]), P_(s["rgb"]), P_(s["ws"]), P_(bg),
                                                           -1.0, P_(s["g"]), H, W, K, 3, N, P_(s["ga"]), P_(s["gw"]), st)),
                "composite_bwd": (lambda: dict(a=C_(sel[2]), l=C_(sel[1]), d=C_(sel[3]), w=C_(w), c=C_(cnt), g=C_(g_rand),
                                               o=[E(w) for _ in range(3)]),
                                  lambda s: lib.voge_composite_bwd(P_(s["a"]), P_(s["l"]), P_(s["d"]), P_(s["w"]), P_(s["c"]), P_(s["g"]), 1.0,
                                                                   npix, K, P_(s["o"][0]), P_(s["o"][1]), P_(s["o"][2]), st)),
                "trace_bwd": (lambda: dict(i=C_(sel[0]), c=C_(cnt), g=[C_(g_rand) for _ in range(3)],
                                           ws=torch.empty(nws_b, dtype=torch.uint8, device=dev), gm=E(mus), gA=E(isg)),
                              lambda s: trace_bwd_fn(P_(mus), P_(isg), P_(rays), P_(s["i"]), P_(s["c"]), P_(s["g"][0]), P_(s["g"][1]),
                                                     P_(s["g"][2]), N, H, W, K, P_(s["ws"]), nws_b, None, P_(s["gm"]), P_(s["gA"]), st)),
            }
            if iso:
                # the frame's actual backward: shade -> composite -> trace in one kernel (voge_fragment_shade_bwd_iso)
                # (scalar sigmas keep no act / dsd: both entry points derive them from the (mu, a) records)
                recs = torch.cat([mus, isg[:, None]], dim=1).contiguous()
                nfb = lib.voge_fragment_bwd_workspace_bytes(N)
                stage_defs["fragment_bwd"] = (
                    lambda: dict(i=C_(idx), c=C_(cnt), w=C_(w), l=C_(sel[1]), rgb=C_(rgb), ws=C_(wsum),
                                 g=torch.ones_like(rgb), wk=torch.empty(nfb, dtype=torch.uint8, device=dev), gv=E(mus), gs=E(isg),
                                 gc=torch.empty_like(colors)),
                    lambda s: lib.voge_fragment_shade_bwd_iso(P_(recs), P_(isg), 0, 0, P_(rays), P_(colors), P_(s["i"]), P_(s["c"]), P_(s["w"]),
                                                              None, P_(s["l"]), None, P_(s["rgb"]), P_(s["ws"]), P_(bg), -1.0,
                                                              P_(s["g"]), 3, 1, 1.0, 1, N, H, W, K, 3, N, P_(s["wk"]), nfb, P_(s["gv"]),
                                                              P_(s["gs"]), P_(s["gc"]), st))
                # the frame's actual forward: trace + composite behind one entry point, fragments = (weight, idx, len, valid_num)
                stage_defs["fragments_fwd"] = (
                    lambda: dict(ws=torch.empty(nws, dtype=torch.uint8, device=dev), i=E(idx), l=E(w), w=E(w), v=E(vn),
                                 c=torch.empty((1, H, W), dtype=torch.int32, device=dev), r=torch.empty((N, 4), device=dev)),
                    lambda s: lib.voge_fragments_fwd_iso(P_(mus), P_(isg), P_(rays), None, P_(cones), 1, N, H, W, K, thr_act, 1.0,
                                                         P_(s["ws"]), nws, P_(s["i"]), P_(s["l"]), None, None, P_(s["c"]), P_(s["w"]),
                                                         P_(s["v"]), P_(s["r"]), st))
                # the frame as the renderer runs it since round 3: the trace alone (composite deferred) ...
                stage_defs["trace_lean_fwd"] = (
                    lambda: dict(ws=torch.empty(nws, dtype=torch.uint8, device=dev), i=E(idx), l=E(w),
                                 c=torch.empty((1, H, W), dtype=torch.int32, device=dev), r=torch.empty((N, 4), device=dev)),
                    lambda s: lib.voge_fragments_fwd_iso(P_(mus), P_(isg), P_(rays), None, P_(cones), 1, N, H, W, K, thr_act, 1.0,
                                                         P_(s["ws"]), nws, P_(s["i"]), P_(s["l"]), None, None, P_(s["c"]), None,
                                                         None, P_(s["r"]), st))
                # ... and composite + shade in one pass when to_white_background asks for the image
                if True:
                    stage_defs["composite_shade_fwd"] = (
                        lambda: dict(i=C_(sel[0]), c=C_(cnt), l=C_(sel[1]), w=E(w), v=E(vn), rgb=E(rgb), img=E(rgb), ws=E(wsum)),
                        lambda s: lib.voge_composite_shade_fwd_iso(P_(s["i"]), P_(s["c"]), P_(s["l"]), P_(recs), P_(rays), 1.0, P_(colors),
                                                                   P_(bg), -1.0, npix, K, 3, N, P_(s["w"]), P_(s["v"]), P_(s["rgb"]),
                                                                   P_(s["img"]), P_(s["ws"]), st))
                # round 6: the frame path's three entry points (what the renderer launches for scalar sigmas and fixed cameras)
                from voge_amd.cameras import camera_tensors
                cam = camera_tensors(cams, (H, W))
                if cam is not None:
                    Rc, Tc, fc, pc = (x.contiguous() for x in cam[:4])
                    vts, sgs = gm.verts.detach().contiguous(), gm.sigmas.detach().contiguous()
                    stage_defs["frame_trace_fwd"] = (
                        lambda: dict(ws=torch.empty(nws, dtype=torch.uint8, device=dev), i=E(idx), l=E(w),
                                     c=torch.empty((1, H, W), dtype=torch.int32, device=dev), r=torch.empty((N, 4), device=dev),
                                     ry=E(rays), o=torch.empty((1, 3), device=dev)),
                        lambda s: lib.voge_frame_trace_fwd_iso(P_(vts), P_(sgs), 1, 1, P_(Rc), P_(Tc), P_(fc), P_(pc), 0, H, 0, 0, 1, N, H, W, K,
                                                               thr_act, P_(s["ws"]), nws, P_(s["i"]), P_(s["l"]), P_(s["c"]), P_(s["r"]),
                                                               P_(s["ry"]), P_(s["o"]), st))
                    stage_defs["frame_shade_fwd"] = (
                        lambda: dict(i=C_(sel[0]), c=C_(cnt), l=C_(sel[1]), w=E(w), v=E(vn), rgb=E(rgb), img=E(rgb), ws=E(wsum),
                                     acc=torch.empty(N * 32, dtype=torch.uint8, device=dev)),
                        lambda s: lib.voge_frame_shade_fwd_iso(P_(s["i"]), P_(s["c"]), P_(s["l"]), P_(recs), P_(rays), 1.0, P_(colors),
                                                               P_(bg), -1.0, npix, K, 3, N, P_(s["w"]), P_(s["v"]), P_(s["rgb"]),
                                                               P_(s["img"]), P_(s["ws"]), None, P_(s["acc"]), N * 32, st))
                    # (the accumulator is not re-zeroed between the timed calls: the kernel adds to whatever is there -- same work)
                    stage_defs["frame_shade_bwd"] = (
                        lambda: dict(i=C_(idx), c=C_(cnt), w=C_(w), l=C_(sel[1]), rgb=C_(rgb), ws=C_(wsum), g=torch.ones_like(rgb),
                                     acc=torch.zeros(N * 32, dtype=torch.uint8, device=dev), gv=E(mus), gs=E(isg), gc=torch.empty_like(colors)),
                        lambda s: lib.voge_frame_shade_bwd_iso(P_(recs), P_(sgs), 1, 1, P_(rays), P_(colors), P_(s["i"]), P_(s["c"]), P_(s["w"]),
                                                               P_(s["l"]), P_(s["rgb"]), P_(s["ws"]), P_(bg), -1.0, P_(s["g"]), 3, 1, 1.0,
                                                               1, N, H, W, K, 3, N, P_(s["acc"]), N * 32, P_(s["gv"]), P_(s["gs"]), P_(s["gc"]), st))
            else:
                # the general path's fused backward (voge_fragment_shade_bwd): full 3x3 forms
                nfb = lib.voge_fragment_bwd_workspace_bytes(N)
                stage_defs["fragment_bwd"] = (
                    lambda: dict(i=C_(idx), c=C_(cnt), w=C_(w), a=C_(sel[2]), l=C_(sel[1]), d=C_(sel[3]), rgb=C_(rgb), ws=C_(wsum),
                                 g=torch.ones_like(rgb), wk=torch.empty(nfb, dtype=torch.uint8, device=dev), gv=E(mus), gs=E(isg),
                                 gc=torch.empty_like(colors)),
                    lambda s: lib.voge_fragment_shade_bwd(P_(mus), P_(isg), P_(rays), P_(colors), P_(s["i"]), P_(s["c"]), P_(s["w"]),
                                                          P_(s["a"]), P_(s["l"]), P_(s["d"]), P_(s["rgb"]), P_(s["ws"]), P_(bg), -1.0,
                                                          P_(s["g"]), 3, 1, 1.0, N, H, W, K, 3, N, P_(s["wk"]), nfb, P_(s["gv"]),
                                                          P_(s["gs"]), P_(s["gc"]), st))
            nbytes = stage_bytes(N, npix, K, iso=iso)
            stages = {}
            for name, (mk, call) in stage_defs.items():
                if args.only_stage and name != args.only_stage:
                    continue
                t_ms, nset = rotating(mk, call, nbytes[name])
                one = mk()
                t_same, _ = rotating(lambda one=one: one, call, 1 << 40)      # the same call replayed on ONE buffer set
                stages[name] = {"ms": round(t_ms, 4), "algo_MB": round(nbytes[name] / 1e6, 1),
                                "GBps": round(nbytes[name] / 1e9 / (t_ms / 1e3), 1), "buffer_sets": nset,
                                "ms_same_buffers": round(t_same, 4)}
                del one
            hits = int((sel[0] >= 0).sum().item())
            del stage_defs
            torch.cuda.empty_cache()
        if args.only_stage:
            print(json.dumps({"only_stage": args.only_stage, "stages": stages}))
            return
        dom = "trace_fwd"  # the sweep BASELINE.json's metric names
        a = stages[dom]["GBps"]
        traffic, traffic_src = None, None
        tname = next((t for t in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", t))), None)
        tcfg = {}
        if tname is not None and not args.anisotropic:
            # HBM bytes per launch from rocprofv3 PMC passes of this same command (not collectable from inside the run)
            tcfg = json.load(open(os.path.join(ROOT, "profiles", tname))).get(args.config, {})
            traffic = tcfg.get("voge_trace_topk_fwd_bytes")
            if traffic is not None:
                traffic_src = (f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2: "
                               "tools/refresh_profiles.sh)")
        result["roofline"] = {"kernel": TRACE_KERNELS, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(a / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                              "algorithmic_bytes": nbytes[dom], "avg_launch_ms": stages[dom]["ms"]}
        result["stages"] = stages
        result["stages_note"] = ("ms / GBps: the call cycling over `buffer_sets` independent sets of operands (> 3 x the 256 MB "
                                 "Infinity Cache in total), i.e. served from HBM; ms_same_buffers: replayed on one set (L3-assisted)")
        on_frame = (("frame_trace_fwd", "frame_shade_fwd", "frame_shade_bwd") if "frame_shade_bwd" in stages and ops.FRAME_PATH
                    else ("trace_lean_fwd", "composite_shade_fwd", "fragment_bwd") if "composite_shade_fwd" in stages
                    else ("fragments_fwd", "shade_fwd", "fragment_bwd") if "fragments_fwd" in stages
                    else ("trace_fwd", "composite_fwd", "shade_fwd", "fragment_bwd"))
        result["stages_on_frame"] = list(on_frame)      # (the stand-alone entry points are timed for reference, not launched)
        # SURVEY 8d's second fraction and the variant the frame actually runs: the trace as the renderer
        # launches it (no act / dsd), and the SQ counters of the dominant kernels from the committed rocprofv3 --pmc passes
        tr = on_frame[0]
        result["roofline"]["on_frame"] = {"entry": tr, "algorithmic_bytes": nbytes[tr], "avg_launch_ms": stages[tr]["ms"],
                                          "achieved": stages[tr]["GBps"], "frac": round(stages[tr]["GBps"] / HBM_PEAK_GBS, 4),
                                          "note": "the renderer keeps no act / dsd: about half of the stand-alone entry's bytes"}
        result["roofline"]["valu"] = valu_table()
        result["frame_kernel_ms_sum"] = round(sum(stages[k]["ms"] for k in on_frame), 4)
        result["hits_per_pixel"] = round(hits / npix, 2)
        # the whole frame against the same roofline: algorithmic bytes of the entry points it launches, and the HBM bytes
        # the PMC counters saw for one frame of this config (committed passes), both over the measured ms_per_step
        f_algo = sum(nbytes[k] for k in on_frame) + (0 if on_frame[0] == "frame_trace_fwd" else npix * 12)
        f_traffic = tcfg.get("frame_hbm_bytes")
        result["frame_roofline"] = {
            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes": f_algo,
            "achieved": round(f_algo / 1e9 / (ms / 1e3), 1), "frac": round(f_algo / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4),
            "traffic": f_traffic,
            "traffic_GBps": None if f_traffic is None else round(f_traffic / 1e9 / (ms / 1e3), 1),
            "traffic_frac": None if f_traffic is None else round(f_traffic / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4),
            "note": "entry points of stages_on_frame + the rays; divided by ms_per_step (graph replay: includes the loss' reduction)"}
        if not args.no_variants and not args.anisotropic and not args.default_bins:
            # the same step on the metric's stated variants: full 3x3 forms; the demos' default bins
            variants = {}
            for vname, (an, db, pat) in (("anisotropic_3x3", (True, False, "white_background")),
                                         ("diagonal_Nx3", ("diag", False, "white_background")),      # (EfficientCuboidViaOptimization's form)
                                         ("max_point_per_bin_None", (False, True, "white_background")),
                                         # what the reference's training loops differentiate (ShapeFitting.py:217,295)
                                         ("interpolate_attr_and_silhouette", (False, False, "attr_and_silhouette"))):
                del fwd, params, gm, colors
                torch.cuda.empty_cache()
                fwd, params, gm, colors, _ = make_frame(an, db, pat)
                vrun, vlaunch = graphed_step(fwd, params)
                vdt = timed(vrun, max(10, args.steps // 2), 3)
                variants[vname] = {"value": round(timed.last_steps / vdt, 1), "unit": "frames/s", "launch": vlaunch}
            # the honest companion of the graph-replay headline: launched eagerly, and the camera MOVES every step (a new
            # R, T on the device each frame -- nothing about a frame can be reused from the last one)
            del fwd, params, gm, colors
            torch.cuda.empty_cache()
            fwd, params, gm, colors, _ = make_frame(False, False)
            nview = 64
            Rm, Tm = look_at_view_transform(dist=[dd] * nview, elev=[el] * nview, azim=[az + 0.5 * i for i in range(nview)], device=dev)
            state = {"i": 0}

            def moving():
                i = state["i"] = (state["i"] + 1) % nview
                for p_ in params:
                    p_.grad = None
                to_white_background(renderer_of[0](gm, R=Rm[i:i + 1], T=Tm[i:i + 1], **rows_kw()), colors).sum().backward()
            for _ in range(60):
                moving()
            torch.cuda.synchronize()
            nm = 10 * max(20, args.steps)      # (an eager loop's first frames behind a synchronisation run on a cold host: 30 steps read 0.284 ms
                                               #  where 300 read 0.249 -- this variant is about the loop's steady state)
            vdt = timed(moving, nm, 5)
            variants["eager_moving_camera"] = {"value": round(nm / vdt, 1), "unit": "frames/s", "launch": "eager, a different camera every step"}
            result["variants"] = variants
        if not args.no_cpu_baseline:
            verts, sig, cols = host_scene
            result["cpu_baseline"] = cpu_baseline(verts, sig, cols, H, W, K, focal, pp, (dd, el, az), args.cpu_seconds)
            # ... and the reference's own CPU-capable tensor program beside the C port
            result["cpu_baseline_torch"] = cpu_baseline_torch(verts, sig, cols, H, W, K, focal, pp, (dd, el, az), args.cpu_seconds)
    if multi and not args.no_variants:
        # the other ways to use the same N GPUs, same scene (every rank takes part): the contiguous bands of rounds 1-2, the
        # weak-scaling batch of views, and north_star's config 4 (200k Gaussians, 1024^2) in the default mode
        variants = {}
        for vname, vmode, vcfg in (("row_bands_contiguous", "row_bands", None), ("views_weak_scaling", "views", None),
                                   ("stripes", "stripes", None), ("cfg4_200k_1024_" + mode, mode, "cfg4_200k_1024")):
            if (vmode == mode and vcfg is None) or (vcfg is not None and vcfg == args.config):
                continue
            try:
                V = measure(vmode, primary=False, cfg=vcfg)
                variants[vname] = {"value": round(V.fps, 1), "unit": "frames/s", "ms_per_step": round(V.ms, 4),
                                   "scaling": "weak" if vmode == "views" else "strong", "launch": V.launch}
                del V
            except Exception as e:  # pragma: no cover - a variant must never take the headline line down with it
                variants[vname] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        result["variants"] = variants
    if rank == 0 and multi:
        # the dominant kernel on rank 0's rows (same entry point, shorter image), timed live
        with torch.no_grad():
            from voge_amd.cameras import pixel_rays
            r0, r1 = rows if rows is not None else (0, H)       # (views mode: rank 0's whole view)
            rays_b, origin = pixel_rays(cams, (H, W), rows=my_stripes if striped else rows)
            h = my_stripes.h if striped else r1 - r0
            cones = ops.cones_of(rays_b, 1, h, W)
            iso = gm.sigmas.dim() == 1
            st = torch.cuda.current_stream().cuda_stream
            nws = lib.voge_trace_workspace_bytes(1, N, h, W)
            mk = lambda: dict(ws=torch.empty(nws, dtype=torch.uint8, device=dev),
                              o=[torch.empty((1, h, W, K), dtype=d, device=dev) for d in (torch.int32, torch.float32, torch.float32, torch.float32)],
                              c=torch.empty((1, h, W), dtype=torch.int32, device=dev))
            if iso:
                call = lambda s: lib.voge_trace_topk_fwd_iso_view(P_(gm.verts), P_(gm.sigmas), P_(origin), 1, 1, P_(rays_b), None, P_(cones), 1, N, h, W,
                                                                  K, thr_act, P_(s["ws"]), nws, *[P_(o) for o in s["o"]], P_(s["c"]), st)
            else:
                from voge_amd.Aggregation import expend_sigma
                mus = (gm.verts[None] - origin[:, None]).reshape(-1, 3).contiguous()
                isg = (2 * expend_sigma(gm.sigmas)).contiguous()
                call = lambda s: lib.voge_trace_topk_fwd(P_(mus), P_(isg), P_(rays_b), None, P_(cones), 1, N, h, W, K, thr_act, P_(s["ws"]), nws,
                                                         *[P_(o) for o in s["o"]], P_(s["c"]), st)
            nb = stage_bytes(N, h * W, K, iso=iso)["trace_fwd"]
            t_ms, _ = rotating(mk, call, nb)
            a = round(nb / 1e9 / (t_ms / 1e3), 1)
            result["roofline"] = {"kernel": TRACE_KERNELS + (" on rank 0's view" if by_views else f" on rank 0's {h} rows (interleaved stripes)" if striped
                                                             else f" on rank 0's band of rows [{r0}, {r1})"),
                                  "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(a / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                                  "algorithmic_bytes": nb, "avg_launch_ms": round(t_ms, 4)}
    # the ONE line, and the LAST one: RCCL writes a version banner to the C library's stdout, which sits in its buffer until the
    # process ends -- behind the line -- when stdout is a pipe or a file; every rank sends it out first
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if multi:
        dist.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(verts, sig, cols, H, W, K, focal, pp, view, target_s):
    """The CPU oracle (a C/OpenMP port of the reference algorithm, fp64) timed on this host on a
    bounded sample: `nrows` pixel rows around the image centre, forward + backward, scaled to a
    whole frame by H / nrows.  nrows is calibrated on a 2-row probe to give ~target_s of work."""
    import numpy as np
    import oracle
    from oracle import camera_np
    oracle.build()
    R, T = camera_np.look_at_view_transform(*view)
    rays_all, origin = camera_np.pixel_rays(R, T, focal, pp, (H, W))
    mus = (verts[None] - origin[:, None].astype(np.float32)).astype(np.float32)
    isg = (2 * camera_np.expand_sigma(sig)).astype(np.float32)[None]
    thr_act = oracle.thr_act_of(0.01)

    def run(nrows):
        r0 = H // 2 - nrows // 2
        rays = np.ascontiguousarray(rays_all[:, r0:r0 + nrows])
        t0 = time.perf_counter()
        idx, ln, act, dsd = oracle.trace_fwd(mus, isg, rays, K, thr_act)
        w, vn = oracle.composite_fwd(idx, act, ln, dsd, 1.0)
        rgb = oracle.merge_fwd(cols, idx, w, vn)
        img, sil = oracle.blend_fwd(rgb, w)
        t_fwd = time.perf_counter() - t0
        g_rgb = np.ones_like(img) * (rgb + (1 - sil)[..., None] < 1)
        g_attr, g_w = oracle.merge_bwd(cols, idx, w, vn, g_rgb)
        g_w = g_w - (g_rgb.sum(-1) * (w.sum(-1) < 1))[..., None]
        g_act, g_len, g_dsd = oracle.composite_bwd(act, ln, dsd, g_w, 1.0)
        oracle.trace_bwd(mus, isg, rays, idx, g_len, g_act, g_dsd)
        return time.perf_counter() - t0, t_fwd, r0

    t_probe, _, _ = run(2)
    nrows = int(max(2, min(H, round(2 * target_s / max(t_probe, 1e-3)))))
    t_all, t_fwd, r0 = run(nrows)
    scale = H / nrows
    return {"value": 1.0 / (t_all * scale), "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{nrows} of {H} pixel rows (rows {r0}..{r0 + nrows - 1}) of the same frame, fwd+bwd, "
                      f"oracle/voge_oracle.c fp64, OpenMP over pixels in every stage ({os.cpu_count()} threads; the two backward "
                      f"scatters use atomic adds); {t_all:.2f} s measured (fwd {t_fwd:.2f} s), scaled x{scale:.1f}"}


def cpu_baseline_torch(verts, sig, cols, H, W, K, focal, pp, view, target_s):
    """BASELINE.md section 3's "reference's pure-PyTorch CPU path", timed on this host: the reference's dense tensor program
    for aggregation / merge / blend (VoGE/Aggregation.py:30-141, VoGE/Renderer.py:157-176: [pix, K, K] erf / exp tensors,
    autograd backward) behind a dense einsum + topk statement of the fine trace (ray_trace_voge.cu:135-217) -- the
    restatement in oracle/torch_ref.py, fp32, torch.set_num_threads(all cores) -- on a bounded sample: `nrows` pixel rows
    around the image centre, forward and forward+backward, scaled to a whole frame by H / nrows."""
    import numpy as np
    import torch
    from oracle import camera_np, torch_ref
    ncore = os.cpu_count()
    torch.set_num_threads(ncore)
    R, T = camera_np.look_at_view_transform(*view)
    rays_all, origin = camera_np.pixel_rays(R, T, focal, pp, (H, W))
    origin_t = torch.tensor(origin[0], dtype=torch.float32)

    def run(nrows):
        r0 = H // 2 - nrows // 2
        rays = torch.tensor(np.ascontiguousarray(rays_all[0, r0:r0 + nrows]).reshape(-1, 3), dtype=torch.float32)
        v = torch.tensor(verts, dtype=torch.float32, requires_grad=True)
        s = torch.tensor(sig, dtype=torch.float32, requires_grad=True)
        c = torch.tensor(cols, dtype=torch.float32, requires_grad=True)
        t0 = time.perf_counter()
        img = torch_ref.frame(v, s, c, rays, origin_t, K)
        t_fwd = time.perf_counter() - t0
        img.sum().backward()
        assert torch.isfinite(v.grad).all() and float(v.grad.abs().max()) > 0
        return time.perf_counter() - t0, t_fwd, r0

    run(1)                                  # (first call: thread pool, allocator)
    t_probe, f_probe, r_probe = run(2)
    nrows = int(max(2, min(H, round(2 * target_s / max(t_probe, 1e-3)))))
    t_all, t_fwd, r0 = (t_probe, f_probe, r_probe) if nrows == 2 else run(nrows)      # (the probe IS the sample when two rows fill the budget)
    scale = H / nrows
    return {"value": 1.0 / (t_all * scale), "unit": "frames/s", "cores": ncore, "kind": "port",
            "forward_only_frames_per_s": 1.0 / (t_fwd * scale),
            "sample": f"{nrows} of {H} pixel rows (rows {r0}..{r0 + nrows - 1}) of the same frame, fwd+bwd in torch {torch.__version__} CPU fp32 "
                      f"({ncore} threads): dense einsum + topk trace over all {len(verts)} Gaussians, then the reference's [pix,K,K] "
                      f"aggregation / merge_final / to_white_background tensor program with autograd (oracle/torch_ref.py, pinned to the "
                      f"imported reference by tests/test_oracle_cpu.py); {t_all:.2f} s measured (fwd {t_fwd:.2f} s), scaled x{scale:.1f}"}


if __name__ == "__main__":
    main()
