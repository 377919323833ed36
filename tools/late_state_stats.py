"""Sweep counters on the fitted ShapeFitting state (stats build: tools/tune_variants.sh stats:"-DVOGE_SWEEP_STATS -DVOGE_SWEEP_TIMES").
usage: VOGE_HIP_LIB=build/variants/stats.so python tools/late_state_stats.py [B]"""
import ctypes, importlib.util, os, sys, numpy as np, torch
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("sf", os.path.join("demo", "ShapeFitting.py"))
sf = importlib.util.module_from_spec(spec); spec.loader.exec_module(sf)
from VoGE.Converter import Converters
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings
from voge_amd.cameras import PerspectiveCameras
from voge_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
R, T = sf.make_views(20, 2.7, dev)
cam = PerspectiveCameras(device=dev, R=R[None, 1], T=T[None, 1], image_size=((128, 128),), principal_point=((64.0, 64.0),), focal_length=126.0)
render = GaussianRenderer(cam, GaussianRenderSettings(image_size=(128, 128), max_assign=25, max_point_per_bin=-1)).to(dev)
sv, sff = sf.ico_sphere(4)
g = Converters.to_gaussian_meshes(Converters.naive_vertices_converter)(torch.from_numpy(sv), torch.from_numpy(sff), device=dev, gradianted_args=[False, False, False])
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 16)()
names = ["waves", "staged(per WG)", "evaluated(per wave sum)", "trips", "slow_entries", "slow_shift_steps(wave max sum)",
         "hits(lane sum)", "list_len(per WG sum)", "list_consumed", "batches"]
for label, verts in (("unit sphere", torch.from_numpy(sv)), ("fitted", torch.from_numpy(np.load("gpurun_out/late_verts.npy")))):
    with torch.no_grad():
        g.verts.copy_(verts.to(dev))
        for _ in range(3):
            f = render(g, R=R[:B], T=T[:B])
        torch.cuda.synchronize()
        raw.voge_debug_sweep_stats(out)          # read + reset
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f = render(g, R=R[:B], T=T[:B]); e1.record(); torch.cuda.synchronize()
        raw.voge_debug_sweep_stats(out)
    v = list(out)
    print(f"== {label}: forward {e0.elapsed_time(e1) * 1e3:.0f} us, hits per pixel {float(f.valid_num.float().mean()):.1f}")
    for nme, x in zip(names, v):
        print(f"   {nme:36s} {x:12d}   per wave {x / max(v[0], 1):10.1f}")
