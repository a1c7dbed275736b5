"""scene2 in photon mode at the reference's own settings vs the committed render/rgb.ppm."""
import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import importlib
rrt = importlib.import_module("rust-raytracer_amd")
from PIL import Image
nph = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
params = rrt.default_params(photons_per_light=nph, caustic_photons_per_light=10000)
sg, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/scene2.txt"), params=params)
print("mode", sg.mode, cam.width, cam.height)
t = time.time(); sg.pre_render(1); print("pre_render %.3fs" % (time.time() - t), sg.stats())
t = time.time(); rgb, depth, ids = sg.render(cam); print("render %.3fs" % (time.time() - t), sg.stats())
q = rrt.quantise(rgb)
ref = np.asarray(Image.open(os.path.join(REPO, "tests/golden/render/rgb.ppm")))
dref = np.asarray(Image.open(os.path.join(REPO, "tests/golden/render/depth.ppm")))
d = np.abs(q.astype(int) - ref.astype(int))
mse = (d.astype(float) ** 2).mean()
print("vs golden rgb.ppm: MAE %.3f/255 PSNR %.2f dB max %d  per-channel MAE %s" % (d.mean(), 10 * np.log10(255 ** 2 / mse), d.max(), d.mean(axis=(0, 1))))
dq = rrt.quantise(depth)
print("depth quantised equal to golden depth.ppm:", np.array_equal(dq, dref), "golden all 255:", (dref == 255).all())
Image.fromarray(q).save(os.path.join(REPO, "gpurun_out", "scene2_photon_gpu.png"))
