"""Where a LATE ShapeFitting iteration spends its time: run the loop for `iters` iterations, then time the stages of one
more iteration on that state with HIP events (synchronised between stages).  usage: python tools/loop_state_profile.py [iters]"""
import importlib.util, os, sys, time, torch
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("sf", os.path.join("demo", "ShapeFitting.py"))
sf = importlib.util.module_from_spec(spec); spec.loader.exec_module(sf)
from VoGE.Converter import Converters
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings, get_silhouette, interpolate_attr
from voge_amd.cameras import PerspectiveCameras
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda", 0)
h = sf.fit(iters=iters, quiet=True, rgb_on=min(400, iters // 5), graph=True)
import numpy as np
fv = h["final_verts"]
print("final verts: finite", np.isfinite(fv).all(), "radius min/mean/max", np.linalg.norm(fv, axis=1).min(), np.linalg.norm(fv, axis=1).mean(),
      np.linalg.norm(fv, axis=1).max(), "losses", h["silhouette"][-1], h["rgb"][-1], "sigmas", h["sigmas"].min(), h["sigmas"].max())
R, T = sf.make_views(20, 2.7, dev)
cam = PerspectiveCameras(device=dev, R=R[None, 1], T=T[None, 1], image_size=((128, 128),), principal_point=((64.0, 64.0),), focal_length=126.0)
render = GaussianRenderer(cam, GaussianRenderSettings(image_size=(128, 128), max_assign=25, max_point_per_bin=-1)).to(dev)
for label, verts in (("unit sphere", torch.from_numpy(sf.ico_sphere(4)[0])), (f"after {iters} iterations", torch.from_numpy(h["final_verts"]))):
    sv, sff = sf.ico_sphere(4)
    g = Converters.to_gaussian_meshes(Converters.naive_vertices_converter)(torch.from_numpy(sv), torch.from_numpy(sff), device=dev, gradianted_args=[True, False, False])
    with torch.no_grad():
        g.verts.copy_(verts.to(dev))
    col = torch.full((2562, 3), 0.5, device=dev, requires_grad=True)
    def ev():
        e = torch.cuda.Event(enable_timing=True); e.record(); return e
    acc = {}
    for rep in range(12):
        t = [ev()]
        frag = render(g, R=R[:5], T=T[:5]); idx = frag.vert_index; t.append(ev())
        rgb = interpolate_attr(frag, col.repeat(5, 1)); sil = get_silhouette(frag); t.append(ev())
        loss = (rgb ** 2).mean() + ((sil - 1) ** 2).mean(); t.append(ev())
        loss.backward(); t.append(ev())
        torch.cuda.synchronize()
        if rep >= 2:
            for i, nm in enumerate(("trace (rays + binA + binB + sweep)", "composite + merge", "losses", "backward")):
                acc[nm] = acc.get(nm, 0.0) + t[i].elapsed_time(t[i + 1]) * 1e3 / 10
        g.verts.grad = None; col.grad = None
    print(label, "hits per pixel", float(frag.valid_num.float().mean()), {k: round(v, 1) for k, v in acc.items()}, "us")
