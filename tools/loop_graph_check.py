"""BatchedIteration: a replayed iteration must do exactly what the eager one does (debug aid / regression check).
usage: python tools/loop_graph_check.py"""
import importlib.util, os, sys, numpy as np, torch
sys.path.insert(0, ".")
spec = importlib.util.spec_from_file_location("sf", os.path.join("demo", "ShapeFitting.py"))
sf = importlib.util.module_from_spec(spec); spec.loader.exec_module(sf)
from VoGE.Converter import Converters
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings
from voge_amd.cameras import PerspectiveCameras
dev = torch.device("cuda", 0)
R, T = sf.make_views(20, 2.7, dev)
cam = PerspectiveCameras(device=dev, R=R[None, 1], T=T[None, 1], image_size=((128, 128),), principal_point=((64.0, 64.0),), focal_length=126.0)
render = GaussianRenderer(cam, GaussianRenderSettings(image_size=(128, 128), max_assign=25, max_point_per_bin=-1)).to(dev)
gv, gf, gc = sf.ground_truth_shape(4)
gt = Converters.to_gaussian_meshes(Converters.naive_vertices_converter)(torch.from_numpy(gv), torch.from_numpy(gf), device=dev, gradianted_args=[False, False, False])
with torch.no_grad():
    targets = [sf.gauss_renderer(render, gt, R[None, j], T[None, j], torch.from_numpy(gc).to(dev)) for j in range(20)]
trgb, tsil = [t[..., :3] * t[..., 3:4] for t in targets], [t[..., 3] for t in targets]
out = {}
for graph in (False, True):
    sv, sff = sf.ico_sphere(4)
    g = Converters.to_gaussian_meshes(Converters.naive_vertices_converter)(torch.from_numpy(sv), torch.from_numpy(sff), device=dev, gradianted_args=[True, False, False])
    col = torch.nn.Parameter(torch.full((2562, 3), 0.5, device=dev))
    opt = torch.optim.SGD(list(g.grad_parameters()) + [col], lr=0.8, momentum=0.9)
    step = sf.BatchedIteration(render, g, col, opt, R, T, trgb, tsil, 5, graph=graph)
    traj = []
    NIT = int(os.environ.get("NIT", "6"))
    WR = float(os.environ.get("WRGB", "1.0"))
    for i in range(NIT):
        views = torch.tensor([(3 * i + k) % 20 for k in range(5)], device=dev)
        l = step(views, WR).clone()
        if i < 6 or i % 20 == 0:
            traj.append((g.verts.detach().clone(), col.detach().clone(), l, i))
    out[graph] = traj
for i in range(len(out[True])):
    dv = (out[True][i][0] - out[False][i][0]).abs().max().item()
    dc = (out[True][i][1] - out[False][i][1]).abs().max().item()
    print(f"iteration {out[True][i][3]}: |verts graph - eager| max {dv:.3e}, |colors| {dc:.3e}, losses eager {out[False][i][2].tolist()} graph {out[True][i][2].tolist()}")
