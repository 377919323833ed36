#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the IMPORTED reference.

Run in the build container only (needs /root/reference, read-only).  The GPU box never
sees /root/reference; it only sees the .npz files this script wrote.  A fixture is data:
seeded inputs plus the outputs the reference's own Python produced for them.

Import recipe: `import VoGE` fails because VoGE/__init__.py pulls in the
CUDA extension `VoGE._C` and pytorch3d.  Registering a bare namespace package `VoGE`
whose __path__ is the reference directory, an EMPTY placeholder `VoGE._C`, and empty
placeholder `pytorch3d.*` modules makes `VoGE.Aggregation`, `VoGE.Renderer` (helpers),
`VoGE.Converter.Cuboid/Converters/IO` importable.  Nothing in the placeholders is ever
called by the functions exercised here.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    pkg = types.ModuleType("VoGE")
    pkg.__path__ = [os.path.join(REF, "VoGE")]
    sys.modules["VoGE"] = pkg
    sys.modules["VoGE._C"] = types.ModuleType("VoGE._C")
    pkg._C = sys.modules["VoGE._C"]
    for name in ("pytorch3d", "pytorch3d.renderer", "pytorch3d.renderer.implicit",
                 "pytorch3d.renderer.implicit.raysampling", "pytorch3d.structures",
                 "pytorch3d.renderer.cameras"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["pytorch3d.renderer.implicit.raysampling"].NDCMultinomialRaysampler = None
    sys.modules["pytorch3d.structures"].Meshes = None
    sys.modules["pytorch3d.renderer"].look_at_rotation = None
    import VoGE.Aggregation as Agg
    import VoGE.Renderer as Ren
    import VoGE.Converter.Cuboid as Cub
    import VoGE.Converter.Converters as Conv
    import VoGE.Converter.IO as IO
    return Agg, Ren, Cub, Conv, IO


def synth_sel(rng, npix, K, fill_lo=0, dtype=np.float64, nmax=500):
    """Seeded stand-ins for the fine kernel's outputs, incl. sentinel tails
    (idx -1, len 1e10, act 1e10, dsd 0; ray_trace_voge.cu:244-247)."""
    idx = np.full((npix, K), -1, np.int32)
    ln = np.full((npix, K), 1e10, dtype)
    act = np.full((npix, K), 1e10, dtype)
    dsd = np.zeros((npix, K), dtype)
    for p in range(npix):
        n = int(rng.integers(fill_lo, K + 1))
        ln[p, :n] = np.sort(rng.uniform(2.0, 6.0, n))
        # a few near-coincident depths so the erf term is exercised around 0
        if n > 3:
            ln[p, 1] = ln[p, 0] + 1e-3
            ln[p, :n] = np.sort(ln[p, :n])
        act[p, :n] = rng.uniform(0.0, 4.6, n)
        dsd[p, :n] = rng.uniform(5.0, 4000.0, n)
        idx[p, :n] = rng.integers(0, nmax, n)
    return idx, ln, act, dsd


def gen_composite(Agg, Ren):
    rng = np.random.default_rng(20240521)
    for name, (npix, K, occ) in dict(k5=(48, 5, 1.0), k25=(40, 25, 1.3), k40=(32, 40, 1.0)).items():
        idx, ln, act, dsd = synth_sel(rng, npix, K)
        g_w = rng.normal(size=(npix, K))
        shape = (1, 4, npix // 4, K)
        t = lambda a, rg=False: torch.tensor(a.reshape(shape), dtype=torch.float64, requires_grad=rg)
        t_idx = torch.tensor(idx.reshape(shape))
        t_act, t_len, t_dsd = t(act, True), t(ln, True), t(dsd, True)
        w, idx_o, vn, hl = Agg.aggregation(t_idx, t_act, t_len, t_dsd, occupation_weight=occ)
        assert hl is t_len and idx_o is t_idx
        (w * torch.tensor(g_w.reshape(shape))).sum().backward()
        # fp32 run of the same program (what the reference actually executes)
        w32, _, _, _ = Agg.aggregation(t_idx, t(act).float(), t(ln).float(), t(dsd).float(), occupation_weight=occ)
        np.savez_compressed(os.path.join(OUT, f"composite_{name}.npz"), idx=idx.reshape(shape), act=act.reshape(shape),
                            len=ln.reshape(shape), dsd=dsd.reshape(shape), occ=occ, g_weight=g_w.reshape(shape),
                            weight=w.detach().numpy(), weight_f32=w32.numpy(), valid_num=vn.numpy(),
                            g_act=t_act.grad.numpy(), g_len=t_len.grad.numpy(), g_dsd=t_dsd.grad.numpy())
        print("composite", name, "w max", float(w.detach().max()), "valid", vn.min().item(), vn.max().item())


def gen_merge_blend(Agg, Ren):
    rng = np.random.default_rng(7)
    npix, K, N, C = 60, 12, 300, 3
    idx, ln, act, dsd = synth_sel(rng, npix, K, nmax=N)
    shape = (1, 6, 10, K)
    t_idx = torch.tensor(idx.reshape(shape))
    t64 = lambda a: torch.tensor(a.reshape(shape), dtype=torch.float64)
    w, _, vn, hl = Agg.aggregation(t_idx, t64(act), t64(ln), t64(dsd), occupation_weight=1.0)
    # boost some weights so the silhouette clamp (min(sum w, 1)) and the output clamp trigger
    w = (w * torch.tensor(rng.uniform(0.5, 3.0, size=shape))).detach().requires_grad_(True)
    colors = torch.tensor(rng.uniform(0, 1, (N, C)), dtype=torch.float64, requires_grad=True)
    feat = torch.tensor(rng.normal(size=(N, 7)), dtype=torch.float64, requires_grad=True)
    g_img = torch.tensor(rng.normal(size=(1, 6, 10, C)))
    g_feat = torch.tensor(rng.normal(size=(1, 6, 10, 7)))
    out = {}
    frag = Ren.Fragments(vert_weight=w, vert_index=t_idx.clone(), valid_num=vn, vert_hit_length=hl)
    idx_before = frag.vert_index.clone()
    rgb = Ren.interpolate_attr(frag, colors)
    out["idx_after_merge"] = frag.vert_index.numpy().copy()  # merge_final mutates -1 -> 0 (Aggregation.py:131)
    img_white = Ren.to_white_background(frag, colors)
    img_col = Ren.to_colored_background(frag, colors, background_color=(0.2, 0.5, 0.9), thr=0.3)
    sil = Ren.get_silhouette(frag)
    fm = Ren.interpolate_attr(frag, feat)
    ((img_white * g_img).sum() + (fm * g_feat).sum()).backward()
    out.update(idx=idx_before.numpy(), weight=w.detach().numpy(), valid_num=vn.numpy(), colors=colors.detach().numpy(),
               feat=feat.detach().numpy(), rgb=rgb.detach().numpy(), img_white=img_white.detach().numpy(),
               img_colored_thr=img_col.detach().numpy(), bg=np.array([0.2, 0.5, 0.9]), thr=0.3,
               silhouette=sil.detach().numpy(), feat_map=fm.detach().numpy(), g_img=g_img.numpy(), g_feat=g_feat.numpy(),
               g_weight=w.grad.numpy(), g_colors=colors.grad.numpy(), g_feat_attr=feat.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "merge_blend.npz"), **out)
    print("merge/blend: img max", float(img_white.max()), "sil max", float(sil.max()))


def gen_misc(Agg, Ren, Cub, Conv, IO):
    out = {}
    s1 = torch.tensor([1.5, 2.5])
    s2 = torch.tensor([[1., 2., 3.], [4., 5., 6.]])
    s3 = torch.arange(18.).view(2, 3, 3)
    out["expend_1"] = Agg.expend_sigma(s1).numpy()
    out["expend_2"] = Agg.expend_sigma(s2).numpy()
    out["expend_3"] = Agg.expend_sigma(s3).numpy()
    st = Ren.GaussianRenderSettings(image_size=128, batch_size=-1, principal_point=(1, 2))
    out["settings_default"] = np.array([st["image_size"][0], st["image_size"][1], st["max_assign"],
                                        st["thr_activation"], st["absorptivity"], float(st["inverse_sigma"])])
    assert st["principal"] is None and st["max_point_per_bin"] is None
    verts, isig = Cub.cuboid_gauss((-1, 1), (-1, 1), (-1, 1), 1000, percentage=0.6)
    out["cuboid_verts"] = verts.astype(np.float32)
    out["cuboid_isigma"] = isig.astype(np.float32)
    v2, i2, c2 = Cub.cuboid_gauss((-1, 2), (0, 1), (-0.5, 0.5), 300, percentage=0.5,
                                  colors=np.arange(18.).reshape(6, 3))
    out["cuboid2_verts"], out["cuboid2_isigma"], out["cuboid2_colors"] = v2, i2, c2
    np.savez_compressed(os.path.join(OUT, "misc_api.npz"), **out)
    print("cuboid:", verts.shape, isig[0])

    # config 2 input: the bunny through the reference's own loader + converter
    bv, bf = IO.load_off(os.path.join(REF, "demo/data/bunny.off"))
    cv, cs, _ = Conv.naive_vertices_converter(bv, bf, percentage=0.6)
    # per-vertex normals (area-weighted), standing in for pytorch3d's verts_normals_packed (RenderBunny.py:30)
    fn = np.cross(bv[bf[:, 1]] - bv[bf[:, 0]], bv[bf[:, 2]] - bv[bf[:, 0]])
    vn = np.zeros_like(bv)
    for k in range(3):
        np.add.at(vn, bf[:, k], fn)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-12)
    np.savez_compressed(os.path.join(OUT, "bunny_gaussians.npz"), verts=cv.astype(np.float32),
                        isigma=cs.astype(np.float32), faces=bf.astype(np.int32), colors=(vn * 0.4 + 0.4).astype(np.float32))
    print("bunny:", cv.shape, float(cs.mean()), float(cs.max()))


def gen_car(Conv, IO):
    """demo/ExtractTexture.py:40-42's input at its real size: data/car.off through the reference's own loader,
    pre_process_pascal and naive_vertices_converter(percentage=0.5, max_sig_rate=2) -- 25 662 Gaussians -- plus the pose of
    data/car_annotation.npz (theta, azimuth, elevation) the demo renders it with."""
    cv, cs, _ = Conv.naive_vertices_converter(*IO.pre_process_pascal(*IO.load_off(os.path.join(REF, "demo/data/car.off"))),
                                              percentage=0.5, max_sig_rate=2)
    an = np.load(os.path.join(REF, "demo/data/car_annotation.npz"))
    np.savez_compressed(os.path.join(OUT, "car_gaussians.npz"), verts=np.asarray(cv, np.float32), isigma=np.asarray(cs, np.float32),
                        theta=float(an["theta"]), azimuth=float(an["azimuth"]), elevation=float(an["elevation"]))
    print("car:", np.asarray(cv).shape, float(np.asarray(cs).mean()), float(np.asarray(cs).max()))


def gen_trace_known_answer():
    """The reference's embedded backward proof (ray_trace_voge.cu:381-448): same inputs, the
    same three forms, loss = len + act, gradients by torch autograd."""
    isg = torch.tensor([[2., 1., 0.], [1., 1.2, 0.], [0., 0., 1.]], requires_grad=True)
    mu = torch.tensor([1.2, 0.2, 0.], requires_grad=True)
    ray = torch.tensor([1., 0., 0.], requires_grad=True)
    msk, ksk, msm = mu @ isg @ ray, ray @ isg @ ray, mu @ isg @ mu
    ln, act = msk / ksk, msm - msk * msk / ksk
    (ln + act).backward()
    np.savez(os.path.join(OUT, "trace_bwd_known_answer.npz"), isigma=isg.detach().numpy(), mu=mu.detach().numpy(),
             ray=ray.detach().numpy(), len=ln.item(), act=act.item(), dsd=ksk.item(), g_mu=mu.grad.numpy(),
             g_isigma=isg.grad.numpy(), g_ray=ray.grad.numpy())
    print("known answer:", mu.grad.tolist(), isg.grad.tolist(), ray.grad.tolist())


def gen_converters_more(Cub, Conv):
    """cuboid_mesh (Cuboid.py:70-159, arrays form) and normal_mesh_converter (Converters.py:35-71).  The latter
    calls pytorch3d's look_at_rotation, which is absent here: the generator hands the reference the oracle's
    restatement of it (oracle/camera_np.look_at_rotation), so the fixture pins the converter's own arithmetic
    (edge lengths, sigma scale, R diag(1,1,shape_ratio) R^T, auto_fix, max_sig_rate) and leaves the PyTorch3D
    convention itself "unpinned", as everywhere."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import camera_np
    out = {}
    v, f = Cub.cuboid_mesh((-1, 1), (-1, 1), (-1, 1), 1000)
    out["mesh_verts"], out["mesh_faces"] = v, f.astype(np.int32)
    v2, f2, c2 = Cub.cuboid_mesh((-1, 2), (0, 1), (-0.5, 0.5), 300, colors=np.arange(18.).reshape(6, 3))
    out["mesh2_verts"], out["mesh2_faces"], out["mesh2_colors"] = v2, f2.astype(np.int32), c2
    Conv.look_at_rotation = lambda pos: torch.from_numpy(camera_np.look_at_rotation(pos.numpy()).astype(np.float32))
    rng = np.random.default_rng(11)
    nrm = rng.normal(size=v2.shape)
    nrm[3] = [0.0, 1.0, 0.0]          # up-parallel normal: the degenerate branch of look_at_rotation
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    vv, isg, rad = Conv.normal_mesh_converter(v2.astype(np.float64), f2, nrm.astype(np.float32), percentage=0.6, shape_ratio=0.3)
    assert rad is None
    out["nm_normals"], out["nm_isigma"] = nrm.astype(np.float32), isg
    _, isg2, _ = Conv.normal_mesh_converter(v2.astype(np.float64), f2, nrm.astype(np.float32), percentage=0.5, shape_ratio=0.5,
                                            max_sig_rate=1.5)
    out["nm_isigma_capped"] = isg2
    np.savez_compressed(os.path.join(OUT, "converters_more.npz"), **out)
    print("cuboid_mesh", v.shape, f.shape, "normal_mesh isigma", isg.shape, float(np.abs(isg).max()))


def gen_host_logic(Ren):
    """The reference's own host logic around the fine kernel, RUN: GaussianRenderer.forward (Renderer.py:102-150) and
    ray_tracing (RayTracing.py:12-30) execute here with
      * VoGE._C.ray_trace_voge_fine replaced by a recorder that keeps the arguments the reference passes to its
        kernel (centred means, 2*sigma / 2*inverse, rays, the "-1" candidate list, thr_act, bin_size, K) and answers
        with the oracle's trace on exactly those arguments (explicit bin list), and
      * a stand-in ray sampler (oracle/camera_np.pixel_rays: the PyTorch3D convention itself stays unpinned).
    The fixture holds the inputs, the recorded kernel arguments and the Fragments the reference's aggregation made.
    (Image heights do not exceed the widths: the "-1" list is built as [B, BW, BW, P] -- point_idx_size[1] twice,
    RayTracing.py:25 -- so a taller-than-wide image would make the reference's own kernel read past its rows.)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import oracle
    from oracle import camera_np
    import VoGE.RayTracing as RT
    rec = {}

    def fake_fine(mus, isigmas, rays, bin_points, thr_act, bin_size, K):
        rec.update(mus=mus.numpy().copy(), isigmas=isigmas.numpy().copy(), rays=rays.numpy().copy(), thr_act=float(thr_act),
                   bin_size=int(bin_size), K=int(K), bin_shape=np.array(bin_points.shape), bin_dtype=str(bin_points.dtype))
        B, P = rays.shape[0], mus.shape[0]
        want = (torch.arange(P // B).view(1, 1, 1, -1) + torch.arange(B).view(-1, 1, 1, 1) * (P // B)).expand(bin_points.shape)
        rec["bin_is_arange_plus_bP"] = bool(torch.equal(bin_points.long(), want))
        o = oracle.trace_fwd(mus.numpy(), isigmas.numpy(), rays.numpy(), K, thr_act, bin_points=bin_points.numpy(),
                             bin_size=bin_size, precision="f32")
        return tuple(torch.from_numpy(np.ascontiguousarray(x)) for x in o)
    RT._C.ray_trace_voge_fine = fake_fine

    class Bundle:
        pass

    class Sampler:
        def __init__(self, image_width, image_height, unit_directions, n_pts_per_ray, min_depth, max_depth):
            assert unit_directions and n_pts_per_ray == 1
            self.size = (image_height, image_width)

        def __call__(self, cameras):
            d, o = camera_np.pixel_rays(cameras.R.numpy(), cameras.T.numpy(), cameras.focal_length.numpy(),
                                        cameras.principal_point.numpy(), self.size)
            b = Bundle()
            b.directions = torch.from_numpy(d)
            b.origins = torch.from_numpy(o.astype(np.float32))[:, None, None, :].expand(-1, self.size[0], self.size[1], -1)
            return b
    Ren.NDCMultinomialRaysampler = Sampler

    class Cams:
        device = "cpu"

        def __init__(self, focal, pp):
            self.focal_length = torch.tensor(focal, dtype=torch.float32).reshape(1, -1)
            self.principal_point = torch.tensor(pp, dtype=torch.float32).reshape(1, 2)
            self.R = self.T = None

        def in_ndc(self):
            return False

        def to(self, device):
            return self

    rng = np.random.default_rng(5)
    out = {}
    cases = {
        "a": dict(N=300, size=(40, 48), K=8, B=1, sig="scalar", inv=False, thr=0.01, occ=1.0),
        "b": dict(N=200, size=(30, 44), K=5, B=2, sig="full", inv=True, thr=0.0, occ=1.3),
    }
    for name, c in cases.items():
        N, (H, W), B = c["N"], c["size"], c["B"]
        verts = rng.uniform(-1, 1, (N, 3)).astype(np.float32)
        r = rng.uniform(0.1, 0.25, N)
        s = (1.0 / (r * r / (2 * np.log(1 / 0.6)))).astype(np.float32)
        if c["sig"] == "full":      # inverse_sigma=True: the user passes covariances, the renderer inverts
            L = np.tril(rng.uniform(-1, 1, (N, 3, 3)))
            L[:, [0, 1, 2], [0, 1, 2]] = np.abs(L[:, [0, 1, 2], [0, 1, 2]]) + 0.5
            sig = ((L @ L.transpose(0, 2, 1)) / s[:, None, None]).astype(np.float32)
        else:
            sig = s
        R, T = camera_np.look_at_view_transform([3.0] * B, [10.0, -20.0][:B], [30.0, 140.0][:B])
        focal, pp = 0.9 * max(H, W), (W / 2.0 + 1.5, H / 2.0 - 0.5)
        st = Ren.GaussianRenderSettings(image_size=(H, W), max_assign=c["K"], thr_activation=c["thr"], absorptivity=c["occ"],
                                        inverse_sigma=c["inv"], max_point_per_bin=-1)
        renderer = Ren.GaussianRenderer(Cams([focal], pp), st)
        gm = lambda v=verts, g=sig: (torch.from_numpy(v), torch.from_numpy(g), None)
        rec.clear()
        frag = renderer(gm, R=torch.from_numpy(R), T=torch.from_numpy(T))
        assert rec["bin_is_arange_plus_bP"]
        out.update({f"{name}_verts": verts, f"{name}_sigmas": sig, f"{name}_R": R, f"{name}_T": T, f"{name}_focal": focal,
                    f"{name}_pp": np.array(pp), f"{name}_size": np.array([H, W]), f"{name}_K": c["K"], f"{name}_thr": c["thr"],
                    f"{name}_occ": c["occ"], f"{name}_inverse_sigma": c["inv"],
                    f"{name}_k_mus": rec["mus"], f"{name}_k_isigmas": rec["isigmas"], f"{name}_k_thr_act": rec["thr_act"],
                    f"{name}_k_bin_size": rec["bin_size"], f"{name}_k_bin_shape": rec["bin_shape"],
                    f"{name}_weight": frag.vert_weight.numpy(), f"{name}_index": frag.vert_index.numpy(),
                    f"{name}_valid_num": frag.valid_num.numpy(), f"{name}_hit_length": frag.vert_hit_length.numpy()})
        print("host logic", name, "bin_size", rec["bin_size"], "thr_act", rec["thr_act"], "bin list", rec["bin_shape"],
              rec["bin_dtype"], "hits", int((frag.vert_index >= 0).sum()))
    np.savez_compressed(os.path.join(OUT, "host_logic.npz"), **out)


def gen_ray_camera_space(Agg):
    """The one piece of ray-generation arithmetic the reference carries itself: get_ray_camera_space
    (VoGE/Aggregation.py:11-27), RUN here.  It states the view-space sign convention of the path's rays
    (x = -(col - px) / fx, y = -(row - py) / fy, z = 1, normalised; `principle` is (row, col) ordered) with pixel
    CORNERS as sample points; PyTorch3D's sampler, which the renderer uses, takes pixel CENTRES (a half-pixel shift).
    The fixture pins oracle/camera_np.pixel_rays and voge_rays_fwd to it: same directions once the principal point is
    shifted by half a pixel."""
    out = {}
    for name, (hw, principle, focal) in dict(a=((5, 7), (2.0, 3.0), 9.0), b=((32, 48), (15.5, 23.25), torch.tensor([40.0, 44.0])),
                                             c=((16, 16), (8.0, 8.0), torch.tensor([[21.0, 21.0]]))).items():
        d = Agg.get_ray_camera_space(hw, principle, focal if not isinstance(focal, float) else float(focal))
        out[name + "_dirs"] = d.numpy()
        out[name + "_size"] = np.array(hw)
        out[name + "_principle_row_col"] = np.array(principle, np.float64)
        f = np.array([focal, focal], np.float64) if isinstance(focal, float) else focal.numpy().astype(np.float64).reshape(-1)
        out[name + "_focal_row_col"] = f
    np.savez_compressed(os.path.join(OUT, "ray_camera_space.npz"), **out)
    print("ray_camera_space", {k: v.shape for k, v in out.items() if k.endswith("dirs")})


if __name__ == "__main__":
    torch.manual_seed(0)
    mods = import_reference()
    gen_composite(mods[0], mods[1])
    gen_merge_blend(mods[0], mods[1])
    gen_misc(*mods)
    gen_trace_known_answer()
    gen_converters_more(mods[2], mods[3])
    gen_host_logic(mods[1])
    gen_ray_camera_space(mods[0])
    gen_car(mods[3], mods[4])
