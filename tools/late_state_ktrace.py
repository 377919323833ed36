"""Render the fitted ShapeFitting state (gpurun_out/late_verts.npy, written with `save`) N times, synchronising after each
call, so that a rocprofv3 kernel trace shows clean per-kernel durations.  usage: python tools/late_state_ktrace.py save|run"""
import importlib.util, os, sys, numpy as np, torch
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("sf", os.path.join("demo", "ShapeFitting.py"))
sf = importlib.util.module_from_spec(spec); spec.loader.exec_module(sf)
from VoGE.Converter import Converters
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings
from voge_amd.cameras import PerspectiveCameras
dev = torch.device("cuda", 0)
if sys.argv[1] == "save":
    h = sf.fit(iters=1500, quiet=True, rgb_on=300)
    np.save("gpurun_out/late_verts.npy", h["final_verts"])
    sys.exit(0)
R, T = sf.make_views(20, 2.7, dev)
cam = PerspectiveCameras(device=dev, R=R[None, 1], T=T[None, 1], image_size=((128, 128),), principal_point=((64.0, 64.0),), focal_length=126.0)
render = GaussianRenderer(cam, GaussianRenderSettings(image_size=(128, 128), max_assign=25, max_point_per_bin=-1)).to(dev)
sv, sff = sf.ico_sphere(4)
g = Converters.to_gaussian_meshes(Converters.naive_vertices_converter)(torch.from_numpy(sv), torch.from_numpy(sff), device=dev, gradianted_args=[False, False, False])
with torch.no_grad():
    g.verts.copy_(torch.from_numpy(np.load("gpurun_out/late_verts.npy")).to(dev))
    for _ in range(30):
        f = render(g, R=R[:5], T=T[:5])
        torch.cuda.synchronize()
    print("hits per pixel", float(f.valid_num.float().mean()), "max", int(f.valid_num.max()))
