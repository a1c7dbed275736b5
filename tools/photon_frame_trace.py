"""One warm photon-mapped frame of scene2 (1 M photons, 1024^2) for a rocprofv3 kernel trace."""
import importlib, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
params = rrt.default_params(photons_per_light=int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, caustic_photons_per_light=10_000,
                            gather_heavy_threshold=int(os.environ.get("THR", "0")))
scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/scene2.txt"), params=params)
scene.pre_render(1)
W, H = cam.width, cam.height
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); depth = torch.empty((H, W), dtype=torch.float32, device="cuda"); ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
for _ in range(4):
    scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(scene.stats())
