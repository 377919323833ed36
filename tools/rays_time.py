import sys, torch
sys.path.insert(0, ".")
from voge_amd import _lib
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform, pixel_rays
dev = torch.device("cuda", 0)
R, T = look_at_view_transform(dist=4.0, elev=10.0, azim=70.0, device=dev)
for H in (512, 1024):
    cams = PerspectiveCameras(focal_length=600.0, principal_point=((H/2, H/2),), image_size=((H, H),), device=dev, R=R, T=T)
    for _ in range(3): rays, o = pixel_rays(cams, (H, H))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): rays, o = pixel_rays(cams, (H, H))
    e1.record(); torch.cuda.synchronize()
    print(H, "pixel_rays us", e0.elapsed_time(e1) * 1000 / 20)
    lib = _lib.load()
    Rc, Tc = cams.R.contiguous(), cams.T.contiguous()
    f = torch.tensor([[600.0, 600.0]], device=dev); pp = torch.tensor([[H/2, H/2]], device=dev)
    st = torch.cuda.current_stream().cuda_stream
    e0.record()
    for _ in range(20): lib.voge_rays_fwd(Rc.data_ptr(), Tc.data_ptr(), f.data_ptr(), pp.data_ptr(), 1, 0, H, H, rays.data_ptr(), o.data_ptr(), None, st)
    e1.record(); torch.cuda.synchronize()
    print(H, "voge_rays_fwd us", e0.elapsed_time(e1) * 1000 / 20)
