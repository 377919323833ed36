"""Where the fused backward's time goes: per-section cycle counters of fragment_bwd_kernel (build with
tools/tune_variants.sh fbt:"-DVOGE_FB_TIMES"; sections in fragment_bwd.hip).  Cycles are wall cycles of a wave (its
SIMD is shared with up to three others), summed over all waves.
usage on the GPU box: VOGE_HIP_LIB=build/variants/fbt.so python tools/fb_sections.py [config] [pattern]
pattern: white (to_white_background: shade-through kernel) | attr (interpolate_attr + get_silhouette: weight-gradient form)"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from voge_amd import _lib, scenes  # noqa: E402
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform  # noqa: E402
from voge_amd.Renderer import GaussianRenderSettings, GaussianRenderer, get_silhouette, interpolate_attr, to_white_background  # noqa: E402
from voge_amd.Meshes import GaussianMeshes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_50k_512"
pattern = sys.argv[2] if len(sys.argv) > 2 else "white"
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
import os
verts, sig, colors = scenes.random_gaussians(N, seed=0, anisotropic=bool(os.environ.get("ANISO")))      # ANISO=1: [N,3,3] sigmas
dev = torch.device("cuda", 0)
gm = GaussianMeshes(torch.from_numpy(verts), torch.from_numpy(sig)).to(dev)
cols = torch.from_numpy(colors).to(dev).requires_grad_(True)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev)
renderer = GaussianRenderer(cams, GaussianRenderSettings(image_size=(H, W), max_assign=K, max_point_per_bin=-1)).to(dev)
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()


def frame():
    frag = renderer(gm, R=R, T=T)
    if pattern == "white":
        return to_white_background(frag, cols).sum()
    return interpolate_attr(frag, cols).sum() + get_silhouette(frag).sum()


for _ in range(3):
    frame().backward()
torch.cuda.synchronize()
raw.voge_debug_fb_times(out, 1)
cw = (ctypes.c_ulonglong * 4)()
raw.voge_debug_cw_stats(cw, 1)
tab = (ctypes.c_ulonglong * 4)()
raw.voge_debug_fb_tab(tab, 1)
loss = frame()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
loss.backward()
e1.record()
torch.cuda.synchronize()
raw.voge_debug_fb_times(out, 0)
names = ["set-up + lane packing", "streams + gathers + shade terms", "LDS rows + composite backward", "trace terms",
         "table (find + add)", "flush"]
tot = sum(out[i] for i in range(6))
print(f"{name} {pattern}: backward {e0.elapsed_time(e1) * 1e3:.1f} us (events, whole backward); {out[6]} waves, "
      f"{tot / max(out[6], 1):.0f} cycles per wave")
for i, nm in enumerate(names):
    print(f"  {nm:34s} {100.0 * out[i] / max(tot, 1):5.1f} %   {out[i] / max(out[6], 1):8.0f} cycles per wave")
raw.voge_debug_cw_stats(cw, 0)
print(f"  window loops: row loops {cw[0] // 64} wave-iterations, {100.0 * cw[1] / max(cw[0], 1):.1f} % of lanes active; "
      f"column loops {cw[2] // 64} wave-iterations, {100.0 * cw[3] / max(cw[2], 1):.1f} % of lanes active")
raw.voge_debug_fb_tab(tab, 0)
print(f"  table: {tab[0]} wave-wide accumulations, {tab[1] / max(tab[0], 1):.2f} election rounds each, {tab[2] / max(tab[0], 1):.1f} lanes taking part "
      f"({tab[2] / max(tab[1], 1):.1f} lanes served per round)")
