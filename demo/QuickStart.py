"""The reference's Quick Start (Readme.md:70-101), unchanged except for the camera import: PyTorch3D's
PerspectiveCameras / look_at_view_transform are used when installed, this package's stand-ins
(voge_amd/cameras.py, same conventions) otherwise.  Renders the 866-Gaussian cuboid of BASELINE config 1
and writes the image (PNG through matplotlib when available, .npy always).

usage: python demo/QuickStart.py [out_prefix]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:
    from pytorch3d.renderer import PerspectiveCameras, look_at_view_transform
except ImportError:
    from voge_amd.cameras import PerspectiveCameras, look_at_view_transform
from VoGE.Converter import Cuboid
from VoGE.Renderer import GaussianRenderer, GaussianRenderSettings, to_white_background

device = 'cuda'

# Create gaussians
gaussians = Cuboid.cuboid_gauss((-1, 1), (-1, 1), (-1, 1), 1000, percentage=0.6, as_obj=True).to(device)

# Create a camera
camera = PerspectiveCameras(focal_length=300, image_size=((256, 256), ), principal_point=((128, 128), ), device=device)

# Create the renderer
render_settings = GaussianRenderSettings(image_size=(256, 256), principal=(128, 128), )
renderer = GaussianRenderer(cameras=camera, render_settings=render_settings)

# Compute camera pose
R, T = look_at_view_transform(dist=6, elev=10, azim=70, device=device)

# Render the Gaussians
frag = renderer(gaussians, R=R, T=T)

# Convert into a image
img = to_white_background(frag, (gaussians.verts + 1) / 3).clamp(0, 1)

out = sys.argv[1] if len(sys.argv) > 1 else "quick_start"
arr = img.squeeze(0).detach().cpu().numpy()
import numpy as np
np.save(out + ".npy", arr)
print("image", arr.shape, "mean", float(arr.mean()), "covered pixels", int((arr.min(-1) < 0.999).sum()), "->", out + ".npy")
try:
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.imsave(out + ".png", arr)
    print("->", out + ".png")
except ImportError:
    pass
