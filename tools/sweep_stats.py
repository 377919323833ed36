"""Counters of the forward sweep (debug build: tools/tune_variants.sh stats:"-DVOGE_SWEEP_STATS").
usage on the GPU box: VOGE_HIP_LIB=build/variants/stats.so python tools/sweep_stats.py [config]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from voge_amd import _lib, scenes  # noqa: E402
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform  # noqa: E402
from voge_amd.Renderer import GaussianRenderSettings, GaussianRenderer  # noqa: E402
from voge_amd.Meshes import GaussianMeshes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_50k_512"
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
import os
verts, sig, colors = scenes.random_gaussians(N, seed=0, anisotropic=bool(os.environ.get("ANISO")))
dev = torch.device("cuda", 0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
settings = GaussianRenderSettings(image_size=(H, W), max_assign=K, thr_activation=0.01, absorptivity=1,
                                  max_point_per_bin=-1)
renderer = GaussianRenderer(cams, settings).to(dev)
lib = _lib.load()
out = (ctypes.c_ulonglong * 16)()
ctypes.CDLL(_lib.LIB_PATH).voge_debug_sweep_stats(out)
FULL = bool(os.environ.get("FULL"))      # FULL=1: the stand-alone trace entry point (idx, len, act, dsd) instead of the renderer's
if FULL:
    import math
    from voge_amd import ops
    from oracle import camera_np
    Rn, Tn = camera_np.look_at_view_transform([dd], [el], [az])
    rays_np, origin = camera_np.pixel_rays(Rn, Tn, focal, pp, (H, W))
    mus_t = torch.from_numpy((verts - origin[0].astype("float32")).astype("float32")).to(dev)
    a_t = torch.from_numpy((2 * sig).astype("float32")).to(dev)
    rays_t = torch.from_numpy(rays_np.astype("float32")).to(dev)
    thr_act = -math.log(0.01 + 1e-10)
    run = lambda: ops._RayTraceVoGEIso.apply(mus_t, a_t, rays_t, None, thr_act, K)
else:
    run = lambda: renderer(gm, R=R, T=T)
with torch.no_grad():
    for _ in range(3):
        frag = run()
    torch.cuda.synchronize()
    ctypes.CDLL(_lib.LIB_PATH).voge_debug_sweep_stats(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    frag = run()
    e1.record()
torch.cuda.synchronize()
print('renderer forward (events) us', e0.elapsed_time(e1) * 1000)
ctypes.CDLL(_lib.LIB_PATH).voge_debug_sweep_stats(out)
names = ["waves", "staged(per WG)", "evaluated(per wave sum)", "trips", "slow_entries", "slow_shift_steps(wave max sum)",
         "hits(lane sum)", "list_len(per WG sum)", "list_consumed", "batches"]
v = list(out)
for n, x in zip(names, v):
    print(f"{n:36s} {x:14d}   per wave {x / max(v[0], 1):10.1f}")
print("rays", H * W, "hits per ray", v[6] / (H * W), "evals per ray (wave evals)", v[2] * 64 / (H * W))

import numpy as np
nwg = ((W + 7) // 8) * ((H + 7) // 8)
nwg = min(nwg, 8192)
buf = (ctypes.c_ulonglong * (8 * nwg))()
ctypes.CDLL(_lib.LIB_PATH).voge_debug_sweep_times(buf, nwg)
t = np.array(list(buf), dtype=np.float64).reshape(nwg, 8)
t0 = t[:, 0].min()
tick = 0.01  # us per tick (100 MHz)
start, cones, fill, cons, loop_end, end, ev = [(t[:, i]) for i in range(7)]
ran = start > 0                      # tiles the sweep actually worked on (binB wrote the others; they never stamp)
t0 = start[ran].min()
print("tiles swept", int(ran.sum()), "of", nwg, " kernel span us", (end[ran].max() - t0) * tick)
tot = (end - start)[ran] * tick
print(f"per tile (us): prologue {((cones - start)[ran] * tick).mean():.1f}  fill {(fill[ran] * tick).mean():.1f}  consume {(cons[ran] * tick).mean():.1f}"
      f"  epilogue {((end - loop_end)[ran] * tick).mean():.1f}  total mean {tot.mean():.1f}  max {tot.max():.1f}")
o = np.argsort(-tot)[:8]
idx = np.nonzero(ran)[0]
for i in o:
    j = idx[i]
    print(f"  tile {j}: start +{(start[j] - t0) * tick:.1f}  prologue {(cones[j] - start[j]) * tick:.1f} fill {fill[j] * tick:.1f} consume {cons[j] * tick:.1f}"
          f" epilogue {(end[j] - loop_end[j]) * tick:.1f} total {tot[i]:.1f} end +{(end[j] - t0) * tick:.1f}")
h, edges = np.histogram((start[ran] - t0) * tick, bins=10)
print("start-time histogram (us):", [(round(float(a), 1), int(b)) for a, b in zip(edges[:-1], h)])
h, edges = np.histogram((end[ran] - t0) * tick, bins=10)
print("end-time histogram (us):", [(round(float(a), 1), int(b)) for a, b in zip(edges[:-1], h)])
h, edges = np.histogram(tot, bins=10)
print("tile-duration histogram (us):", [(round(float(a), 1), int(b)) for a, b in zip(edges[:-1], h)])
# candidates evaluated per tile (the low word of stamp 6) against its duration and the hits its rays kept
evs = (ev.astype(np.uint64) & np.uint64(0xffffffff)).astype(np.float64)
vn = (frag.valid_num if not FULL else (frag[0] >= 0).sum(-1)).reshape(H, W).float().cpu().numpy()
ty, tx = (H + 7) // 8, (W + 7) // 8
padv = np.zeros((ty * 8, tx * 8)); padv[:H, :W] = vn
kept = padv.reshape(ty, 8, tx, 8).transpose(0, 2, 1, 3).reshape(ty * tx, 64).sum(1)[:nwg]
e, k = evs[ran], kept[ran]
cons_us = cons[ran] * tick
print(f"evaluated candidates per swept tile: mean {e.mean():.0f}, max {e.max():.0f}; kept hits per tile: mean {k.mean():.0f} (64 K = {64 * K})")
print(f"consume time per evaluated candidate: mean {1e3 * cons_us.sum() / e.sum():.0f} ns; correlation duration~evaluated {np.corrcoef(tot, e)[0, 1]:.3f}")
print(f"kept hits / (64 x evaluated): {k.sum() / (64 * e.sum()):.2f} overall; the 200 longest tiles {k[np.argsort(-tot)[:200]].sum() / (64 * e[np.argsort(-tot)[:200]].sum()):.2f}")
for i in np.argsort(-tot)[:6]:
    print(f"   tile total {tot[i]:.1f} us: evaluated {e[i]:.0f}, kept {k[i]:.0f}, consume {cons_us[i]:.1f} us = {1e3 * cons_us[i] / max(e[i], 1):.0f} ns per candidate")
if os.environ.get("SLOW"):      # build with -DVOGE_SWEEP_SLOW: stamp 7 = ticks inside the list insertions << 32 | candidates that needed one
    s7 = t[:, 7].astype(np.uint64)
    nslow = (s7 & np.uint64(0xffffffff)).astype(np.float64)[ran]
    tslow = (s7 >> np.uint64(32)).astype(np.float64)[ran] * tick
    print(f"candidates with an insertion (any lane): {nslow.sum() / e.sum():.2f} of the evaluated; time inside: {tslow.sum() / cons_us.sum():.2f} of the consume phase")
    for i in np.argsort(-tot)[:6]:
        print(f"   tile total {tot[i]:.1f} us: evaluated {e[i]:.0f}, with an insertion {nslow[i]:.0f}, inside them {tslow[i]:.1f} us of {cons_us[i]:.1f}")
    s2 = t[:, 2].astype(np.uint64)[ran]
    lanes, moved, far = (s2 & np.uint64(0xffffffff)).astype(float), ((s2 >> np.uint64(32)) & np.uint64(0xffff)).astype(float), (s2 >> np.uint64(48)).astype(float)
    print(f"per insertion event: {lanes.sum() / max(nslow.sum(), 1):.1f} lanes insert, the longest walk among them {moved.sum() / max(nslow.sum(), 1):.1f} rows on average; "
          f"events with a walk of 4+ rows: {far.sum() / max(nslow.sum(), 1):.2f}")
    for i in np.argsort(-tot)[:6]:
        print(f"   tile total {tot[i]:.1f} us: events {nslow[i]:.0f}, lanes per event {lanes[i] / max(nslow[i], 1):.1f}, longest walk per event {moved[i] / max(nslow[i], 1):.1f}, far events {far[i]:.0f}")
