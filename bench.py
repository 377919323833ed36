#!/usr/bin/env python3
"""Benchmark of the VoGE hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

A "step" is one forward+backward frame of BASELINE.json's metric config (config 3: 50 000
synthetic Gaussians, 512x512, K=40, max_point_per_bin=-1):
    frag = renderer(gaussians, R=R, T=T); img = to_white_background(frag, colors); img.sum().backward()
with gradients to verts [N,3], sigmas [N] and colours [N,3].  Inputs are resident in HBM before
the timed region.  With --gpus N > 1 (one process per GPU, torchrun) the frame's pixel rows are
sharded over the ranks: each rank traces / composites its own row band, the image is assembled
by ONE all-gather (RCCL) and the per-Gaussian gradients by ONE all-reduce -> total work is
fixed, "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3_50k_512")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a HIP graph")
    ap.add_argument("--split-graph", action="store_true",
                    help="single GPU: use the multi-GPU launch scheme (forward graph / eager exchange / backward graph)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target seconds of CPU oracle work for cpu_baseline")
    return ap.parse_args()


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
    }


def time_kernel(fn, iters=20, warm=3):
    """Average duration (ms) of `fn` with HIP events on torch's current stream -- the stream
    every voge_* entry point is launched on."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the VoGE hot path has no CPU fallback")
    # VOGE_BENCH_BACKEND=gloo lets several ranks share one GPU (a functional check of the multi-GPU
    # launch scheme on a single-GPU box); the real runs use RCCL, one rank per GPU.
    backend = os.environ.get("VOGE_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from voge_amd import _lib, ops, scenes
    from voge_amd.Meshes import GaussianMeshes
    from voge_amd.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background
    from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
    from voge_amd.distributed import allreduce_grads, gather_rows, gather_rows_async, row_band
    _lib.load()

    N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[args.config]
    verts, sig, cols = scenes.random_gaussians(N, seed=0)
    gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
    colors = torch.from_numpy(cols).to(dev).requires_grad_(True)
    R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
    cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
    settings = GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1,
                                      max_point_per_bin=-1)
    renderer = GaussianRenderer(cams, settings).to(dev)
    rows = row_band(H, rank, world) if world > 1 else None
    params = [gm.verts, gm.sigmas, colors]

    def step():
        for p in params:
            p.grad = None
        frag = renderer(gm, R=R, T=T, **({} if rows is None else {"rows": rows}))
        band = to_white_background(frag, colors)
        img = gather_rows(band, H)
        if world > 1:
            r0, r1 = rows
            img[:, r0:r1].sum().backward()   # each rank owns the loss of its band; grads are summed below
            allreduce_grads(params)
        else:
            img.sum().backward()
        return img

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The frame is launch-bound on the host (~35 kernels of 5-220 us): capture one forward+backward
    # into a HIP graph and replay it per step.  Every replay runs exactly the kernels of an eager
    # step on the same (static) tensors; eager launches remain available with --no-graph.
    run = step
    graphed = False
    launch = "eager"

    def warm_side_stream(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    if not args.no_graph and world == 1 and not args.split_graph:
        try:
            warm_side_stream(step)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            graph.replay()
            torch.cuda.synchronize()
            run, graphed, launch = graph.replay, True, "hip graph replay"
        except Exception as e:  # pragma: no cover - depends on the runtime
            print(f"[bench] HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            run = step
    elif not args.no_graph:
        # Multi-GPU: the local compute is two HIP graphs -- band forward, band backward -- and the
        # two exchanges (all_gather of image rows -- asynchronous, overlapping the backward graph --
        # and all_reduce of the gradients) run eagerly, so no collective is ever captured.  The loss is sum(image): every rank
        # owns the loss of its band, whose upstream gradient is a constant tensor of ones.
        try:
            def fwd_only():
                frag = renderer(gm, R=R, T=T, **({} if rows is None else {"rows": rows}))
                return to_white_background(frag, colors)

            def eager_once():
                b = fwd_only()
                torch.autograd.grad(b, params, torch.ones_like(b))
            warm_side_stream(eager_once)
            g_fwd, g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread polls events while we capture
            with torch.cuda.graph(g_fwd, capture_error_mode="thread_local"):
                band_static = fwd_only()
            ones_static = torch.ones_like(band_static)
            with torch.cuda.graph(g_bwd, pool=g_fwd.pool(), capture_error_mode="thread_local"):
                grads_static = torch.autograd.grad(band_static, params, ones_static)
            for p_, g_ in zip(params, grads_static):
                p_.grad = g_

            def run_split():
                g_fwd.replay()
                finish = gather_rows_async(band_static.detach(), H)   # all_gather starts (no-op on one GPU) ...
                g_bwd.replay()                                          # ... and overlaps the band's backward
                if world > 1:
                    allreduce_grads(params)                             # eager all_reduce, in place on .grad
                return finish().sum()                                   # the full-image loss every rank holds
            run_split()
            torch.cuda.synchronize()
            run, graphed = run_split, True
            launch = "hip graphs (band forward, band backward) + eager all_gather / all_reduce"
        except Exception as e:  # pragma: no cover - depends on the runtime
            print(f"[bench] split HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            for p_ in params:
                p_.grad = None
            run = step

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    fps = args.steps / dt

    result = {
        "metric": "forward+backward frames/sec at 512^2, 50k Gaussians; ray-trace HBM GB/s vs peak",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: {N} random Gaussians, {H}x{W}, K={K}, max_point_per_bin=-1, "
                               f"fwd+bwd (grads to verts, sigmas, colors)",
                   "launch": launch,
                   "parallelism": "1 gpu" if world == 1 else f"pixel-row bands x{world}, all_gather(image)+all_reduce(grads)"},
    }

    if rank == 0 and world == 1:
        # ---- per-stage kernel timings on the same inputs (HIP events, current stream) --------
        with torch.no_grad():
            from voge_amd.cameras import pixel_rays
            from voge_amd.Aggregation import expend_sigma
            rays, origin = pixel_rays(cams, (H, W))
            mus = (gm.verts[None] - origin[:, None]).reshape(-1, 3).contiguous()
            iso = gm.sigmas.dim() == 1     # the renderer keeps (N,) sigmas in scalar form (A = a I)
            isg = (2 * gm.sigmas).contiguous() if iso else (2 * expend_sigma(gm.sigmas)).contiguous()
            thr_act = -np.log(0.01 + 1e-10)
            if iso:
                sel = ops._RayTraceVoGEIso.apply(mus, isg, rays, None, thr_act, K)
            else:
                sel = ops.ray_trace_fine(mus, isg, rays, None, thr_act, 16, K)
            w, vn = ops.composite(sel[0], sel[2], sel[1], sel[3], 1.0)
            idx = sel[0].clone()
            vn32 = vn.to(torch.int32).contiguous()
            rgb = ops.merge(colors.detach(), w, idx, vn)
            bg = torch.ones(3, device=dev)
            g_img = torch.ones_like(rgb)
            lib = _lib.load()
            st = torch.cuda.current_stream().cuda_stream
            npix = H * W
            ws = torch.empty(lib.voge_trace_workspace_bytes(1, N, H, W), dtype=torch.uint8, device=dev)
            ws_b = torch.empty(lib.voge_trace_bwd_workspace_bytes(N), dtype=torch.uint8, device=dev)
            trace_fwd_fn = lib.voge_trace_topk_fwd_iso if iso else lib.voge_trace_topk_fwd
            trace_bwd_fn = lib.voge_trace_bwd_iso if iso else lib.voge_trace_bwd
            o_i, o_l, o_a, o_d = (torch.empty_like(x) for x in sel)
            o_c = torch.empty((1, H, W), dtype=torch.int32, device=dev)
            g3 = [torch.empty_like(w) for _ in range(3)]
            g_w = torch.rand_like(w)
            g_ray, g_mu, g_A = torch.empty_like(rays), torch.empty_like(mus), torch.empty_like(isg)
            out3, g_attr = torch.empty_like(rgb), torch.empty_like(colors)
            wsum = torch.empty(rgb.shape[:-1], dtype=torch.float32, device=rgb.device)
            P = lambda x: x.data_ptr()
            calls = {
                "trace_fwd": lambda: trace_fwd_fn(P(mus), P(isg), P(rays), None, P(ops.cones_of(rays, 1, H, W)), 1, N, H, W, K, thr_act, P(ws),
                                                             ws.numel(), P(o_i), P(o_l), P(o_a), P(o_d), P(o_c), st),
                "composite_fwd": lambda: lib.voge_composite_fwd(None, P(ops.hit_count_of(sel[0])), P(sel[2]), P(sel[1]), P(sel[3]), 1.0, npix, K,
                                                                P(g3[0]), P(vn), st),
                "shade_fwd": lambda: lib.voge_shade_fwd(P(colors), P(idx), P(w), P(vn), P(bg), -1.0, npix, K, 3, N, 1, P(rgb),
                                                        P(out3), None, P(wsum), st),
                "shade_bwd": lambda: lib.voge_shade_bwd(P(colors), P(idx), P(w), P(vn), P(rgb), P(wsum), P(bg), -1.0, P(g_img), H, W, K,
                                                        3, N, P(g_attr), P(g3[0]), st),
                "composite_bwd": lambda: lib.voge_composite_bwd(P(sel[2]), P(sel[1]), P(sel[3]), P(w), P(ops.hit_count_of(sel[0])), P(g_w), 1.0, npix, K, P(g3[0]),
                                                                P(g3[1]), P(g3[2]), st),
                "trace_bwd": lambda: trace_bwd_fn(P(mus), P(isg), P(rays), P(sel[0]), P(vn32), P(w), P(w), P(w), N, H, W, K,
                                                        P(ws_b), ws_b.numel(), None, P(g_mu), P(g_A), st),
            }
            nbytes = stage_bytes(N, npix, K, iso=iso)
            stages = {}
            for name, fn in calls.items():
                t_ms = time_kernel(fn)
                stages[name] = {"ms": round(t_ms, 4), "algo_MB": round(nbytes[name] / 1e6, 1),
                                "GBps": round(nbytes[name] / 1e9 / (t_ms / 1e3), 1)}
            hits = int((sel[0] >= 0).sum().item())
        dom = "trace_fwd"  # the sweep BASELINE.json's metric names
        a = stages[dom]["GBps"]
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", "r1_traffic.json")
        if args.config == "cfg3_50k_512" and os.path.exists(tfile):
            # HBM bytes per launch from rocprofv3 PMC passes of this same command (not collectable live)
            traffic = json.load(open(tfile)).get("voge_trace_topk_fwd_bytes")
            traffic_src = "profiles/r1_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2)"
        result["roofline"] = {"kernel": "voge_trace_topk_fwd(_iso) = prep_cone + bin0 + bin + bin2 + tile_order + trace_fwd_kernel",
                              "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(a / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                              "algorithmic_bytes": nbytes[dom], "avg_launch_ms": stages[dom]["ms"]}
        result["stages"] = stages
        result["frame_kernel_ms_sum"] = round(sum(s["ms"] for s in stages.values()), 4)
        result["hits_per_pixel"] = round(hits / npix, 2)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(verts, sig, cols, H, W, K, focal, pp, (dd, el, az), args.cpu_seconds)
    if rank == 0 and world > 1:
        # the dominant kernel on rank 0's band of rows (same entry point, shorter image), timed live
        with torch.no_grad():
            from voge_amd.cameras import pixel_rays
            lib = _lib.load()
            r0, r1 = rows
            rays_b, origin = pixel_rays(cams, (H, W), rows=rows)
            h = r1 - r0
            iso = gm.sigmas.dim() == 1
            thr_act = -np.log(0.01 + 1e-10)
            outs = [torch.empty((1, h, W, K), dtype=d, device=dev) for d in (torch.int32, torch.float32, torch.float32, torch.float32)]
            o_c = torch.empty((1, h, W), dtype=torch.int32, device=dev)
            ws = torch.empty(lib.voge_trace_workspace_bytes(1, N, h, W), dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            P = lambda x: x.data_ptr()
            if iso:
                fn = lambda: lib.voge_trace_topk_fwd_iso_view(P(gm.verts), P(gm.sigmas), P(origin), 1, 1, P(rays_b), None, P(ops.cones_of(rays_b, 1, h, W)), 1, N, h, W, K,
                                                              thr_act, P(ws), ws.numel(), *[P(o) for o in outs], P(o_c), st)
            else:
                from voge_amd.Aggregation import expend_sigma
                mus = (gm.verts[None] - origin[:, None]).reshape(-1, 3).contiguous()
                isg = (2 * expend_sigma(gm.sigmas)).contiguous()
                fn = lambda: lib.voge_trace_topk_fwd(P(mus), P(isg), P(rays_b), None, P(ops.cones_of(rays_b, 1, h, W)), 1, N, h, W, K, thr_act, P(ws), ws.numel(),
                                                     *[P(o) for o in outs], P(o_c), st)
            t_ms = time_kernel(fn)
            nb = stage_bytes(N, h * W, K, iso=iso)["trace_fwd"]
            a = round(nb / 1e9 / (t_ms / 1e3), 1)
            result["roofline"] = {"kernel": "voge_trace_topk_fwd(_iso) on rank 0's band of rows "
                                            f"[{r0}, {r1}) = prep_cone + bin0 + bin + bin2 + tile_order + trace_fwd_kernel",
                                  "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(a / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
                                  "algorithmic_bytes": nb, "avg_launch_ms": round(t_ms, 4)}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(verts, sig, cols, H, W, K, focal, pp, view, target_s):
    """The CPU oracle (a C/OpenMP port of the reference algorithm, fp64) timed on this host on a
    bounded sample: `nrows` pixel rows around the image centre, forward + backward, scaled to a
    whole frame by H / nrows.  nrows is calibrated on a 2-row probe to give ~target_s of work."""
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
                      f"oracle/voge_oracle.c fp64, OpenMP over pixels ({os.cpu_count()} threads; the two backward "
                      f"scatter stages are serial); {t_all:.2f} s measured (fwd {t_fwd:.2f} s), scaled x{scale:.1f}"}


if __name__ == "__main__":
    main()
