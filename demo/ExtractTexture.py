"""Counterpart of the reference's demo/ExtractTexture.py (structure of :29-59): render the object, pull the
colours of an IMAGE back onto the Gaussians with `sample_features` (per-Gaussian weighted sums over the
pixels it covers, Sampler.py:17-42), then re-render from a view rotated by 30 degrees with the extracted
texture.  The reference samples a photograph of a car aligned with a CAD model (data/car_image.JPEG,
car_annotation.npz, car.off) -- not available here; the "photograph" is the bunny of demo/RenderBunny.py
rendered with its reference colours, so the extraction can also be scored: the texture re-rendered from
the ORIGINAL view should resemble the image it was sampled from (not exactly: the sampled colours are
composited a second time, which is what the reference's factor 0.7 compensates).

usage: python demo/ExtractTexture.py [--out PREFIX]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from VoGE.Meshes import GaussianMeshesNaive                                            # noqa: E402
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background   # noqa: E402
from VoGE.Sampler import sample_features                                               # noqa: E402
from voge_amd.cameras import PerspectiveCameras, look_at_view_transform               # noqa: E402


def extract(device="cuda:0", size=256, max_assign=40):
    g = np.load(os.path.join(ROOT, "tests", "golden", "bunny_gaussians.npz"))
    verts, sigmas, color = (torch.from_numpy(g[k]) for k in ("verts", "isigma", "colors"))
    meshes = GaussianMeshesNaive(verts, sigmas, None).to(device)
    color = color.to(device)
    f = 2000.0 * size / 256.0
    render_settings = GaussianRenderSettings(batch_size=-1, image_size=(size, size), max_assign=max_assign)
    cameras = PerspectiveCameras(focal_length=f, principal_point=((size / 2, size / 2),),
                                 image_size=(render_settings['image_size'],), device=device, in_ndc=False)
    render = GaussianRenderer(cameras=cameras, render_settings=render_settings)
    R, T = look_at_view_transform([6], [0], [10], degrees=True)
    with torch.no_grad():
        frag = render(meshes, R=R, T=T)
        im = to_white_background(frag.copy(), color) * 255.0            # the "photograph", [1,H,W,3] in 0..255
        # ---- the reference's extraction (ExtractTexture.py:47-49), including its 0.7: the weights of a pixel sum
        # to more than 1 inside the object (absorptivity 1), so re-compositing the sampled colours brightens them
        get, get_sum = sample_features(frag, im, meshes.verts.shape[0])
        texture = get / (1e-8 + get_sum[:, None]) / 255
        texture = texture * 0.7
        seen = get_sum > 1e-6
        back = to_white_background(render(meshes, R=R, T=T), texture)           # same view: must match the photograph
        R2, T2 = look_at_view_transform([6], [0], [10 + 30], degrees=True)      # :53 rotates by pi/6
        novel = to_white_background(render(meshes, R=R2, T=T2), texture)
    err = (back - im / 255.0).abs()
    return dict(image=im[0] / 255.0, back=back[0], novel=novel[0], texture=texture, seen=seen,
                mean_abs_err=float(err.mean()), max_abs_err=float(err.max()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="texture")
    a = ap.parse_args()
    r = extract()
    print(f"Gaussians that received texture: {int(r['seen'].sum())} of {r['seen'].numel()}; "
          f"re-render of the sampled view: mean |err| {r['mean_abs_err']:.4f}, max {r['max_abs_err']:.3f}")
    np.savez(a.out + ".npz", **{k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in r.items()})
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        strip = torch.cat([r["image"], r["back"], r["novel"]], dim=1).clamp(0, 1).cpu().numpy()
        plt.imsave(a.out + ".png", strip)
        print("->", a.out + ".png  (photograph | texture from the same view | texture from +30 degrees)")
    except ImportError:
        pass
