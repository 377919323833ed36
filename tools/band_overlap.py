"""does the trace entry gain from running independent image bands as concurrent chains (binA -> binB ->
sweep per band, each band on its own stream, event dependencies only)?  No kernel changes: the entry point is called per band
with the band's rays / cones / output slices.  usage (GPU box): python tools/band_overlap.py [config]"""
import sys, torch
sys.path.insert(0, ".")
from voge_amd import scenes, ops, _lib
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform, pixel_rays
import math

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_50k_512"
dev = torch.device("cuda", 0)
lib = _lib.load()
N, (H, W), K, focal, pp, (dd, el, az) = scenes.CONFIGS[name]
verts, sig, cols = scenes.random_gaussians(N, seed=0)
verts, sig = torch.from_numpy(verts).to(dev), torch.from_numpy(sig).to(dev)
R, T = look_at_view_transform(dist=dd, elev=el, azim=az, device=dev)
cams = PerspectiveCameras(focal_length=focal, principal_point=(pp,), image_size=((H, W),), device=dev, R=R, T=T)
thr_act = -math.log(0.01 + 1e-10)
P_ = lambda t: None if t is None else t.data_ptr()
with torch.no_grad():
    rays, origin = pixel_rays(cams, (H, W))
    mus = (verts[None] - origin[:, None]).reshape(-1, 3).contiguous()
    isg = (2 * sig).contiguous()


def make_bands(nb):
    hb = H // nb
    bands = []
    for i in range(nb):
        with torch.no_grad():
            rb, _ = pixel_rays(cams, (H, W), rows=(i * hb, (i + 1) * hb))
        cb = ops.cones_of(rb, 1, hb, W)
        nws = lib.voge_trace_workspace_bytes(1, N, hb, W)
        bands.append(dict(rays=rb, cones=cb, h=hb, r0=i * hb, nws=nws, stream=torch.cuda.Stream(device=dev)))
    return bands


def make_set(bands):
    out = dict(idx=torch.empty((1, H, W, K), dtype=torch.int32, device=dev), len=torch.empty((1, H, W, K), device=dev),
               act=torch.empty((1, H, W, K), device=dev), dsd=torch.empty((1, H, W, K), device=dev),
               cnt=torch.empty((1, H, W), dtype=torch.int32, device=dev), ws=[torch.empty(b["nws"], dtype=torch.uint8, device=dev) for b in bands])
    return out


def run(bands, s, main):
    if len(bands) == 1:
        b = bands[0]
        rc = lib.voge_trace_topk_fwd_iso(P_(mus), P_(isg), P_(b["rays"]), None, P_(b["cones"]), 1, N, H, W, K, thr_act, P_(s["ws"][0]), b["nws"],
                                         P_(s["idx"]), P_(s["len"]), P_(s["act"]), P_(s["dsd"]), P_(s["cnt"]), main.cuda_stream)
        assert rc == 0
        return
    e0 = torch.cuda.Event()
    e0.record(main)
    for i, b in enumerate(bands):
        st = b["stream"]
        st.wait_event(e0)
        r0, h = b["r0"], b["h"]
        rc = lib.voge_trace_topk_fwd_iso(P_(mus), P_(isg), P_(b["rays"]), None, P_(b["cones"]), 1, N, h, W, K, thr_act, P_(s["ws"][i]), b["nws"],
                                         P_(s["idx"][:, r0:r0 + h]), P_(s["len"][:, r0:r0 + h]), P_(s["act"][:, r0:r0 + h]), P_(s["dsd"][:, r0:r0 + h]),
                                         P_(s["cnt"][:, r0:r0 + h]), st.cuda_stream)
        assert rc == 0
        e = torch.cuda.Event()
        e.record(st)
        main.wait_event(e)


main = torch.cuda.current_stream()
ref = None
for nb in (1, 2, 4, 1, 2, 4):
    if H % (nb * 128):
        continue
    bands = make_bands(nb)
    sets = [make_set(bands) for _ in range(4)]
    for i in range(12):
        run(bands, sets[i % 4], main)
    torch.cuda.synchronize()
    if ref is None:
        ref = [sets[0][k].clone() for k in ("idx", "len", "act", "dsd")]
    else:
        same = all(torch.equal(a, sets[0][k]) for a, k in zip(ref, ("idx", "len", "act", "dsd")))
        assert same, "bands differ from the whole frame"
    iters = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        run(bands, sets[i % 4], main)
    e1.record()
    torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) / iters * 1e3
    # the same fork / join captured ONCE into a HIP graph (4 frames, rotating sets) and replayed: no host in the loop
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream(device=dev)
    cs.wait_stream(main)
    with torch.cuda.graph(g, stream=cs):
        for i in range(4):
            run(bands, sets[i], torch.cuda.current_stream())
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {nb} band(s) concurrently: {eager:.1f} us per frame's entry launched eagerly, {e0.elapsed_time(e1) / 200 * 1e3:.1f} us replayed as a graph")
