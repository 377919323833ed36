"""cfg5 (ShapeFitting scene: ico-sphere 4, 128^2, K = 25): event time of the renderer's forward (binA + binB + sweep) for
B views in one call, eager, after warm-up.  usage: python tools/cfg5_fwd_time.py [B] [K]"""
import importlib.util, os, sys, torch
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("sf", os.path.join("demo", "ShapeFitting.py"))
sf = importlib.util.module_from_spec(spec); spec.loader.exec_module(sf)
from VoGE.Converter import Converters
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings
from voge_amd.cameras import PerspectiveCameras
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device("cuda", 0)
R, T = sf.make_views(20, 2.7, dev)
cam = PerspectiveCameras(device=dev, R=R[None, 1], T=T[None, 1], image_size=((128, 128),), principal_point=((64.0, 64.0),), focal_length=126.0)
render = GaussianRenderer(cam, GaussianRenderSettings(image_size=(128, 128), max_assign=K, max_point_per_bin=-1)).to(dev)
sv, sf_ = sf.ico_sphere(4)
g = Converters.to_gaussian_meshes(Converters.naive_vertices_converter)(torch.from_numpy(sv), torch.from_numpy(sf_), device=dev, gradianted_args=[True, False, False])
print("sigma range", float(g.sigmas.min()), float(g.sigmas.max()))
with torch.no_grad():
    for _ in range(20):
        f = render(g, R=R[:B], T=T[:B])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f = render(g, R=R[:B], T=T[:B])
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} K={K}: renderer forward (rays + binA + binB + sweep) {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call; hits per pixel {float(f.valid_num.float().mean()):.1f}")
