"""The fixed cost of a wave: frames of the C4 scene through a camera that sees only sky (argument 'sky'), only the floor
('floor'), or the benchmark's view ('full').  Run under rocprofv3 --pmc to count the instructions a wave spends outside the walk."""
import importlib, math, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
import bench_extra as bx
which = sys.argv[1] if len(sys.argv) > 1 else "sky"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
scene, cam, ntri = bx.build_mesh_scene(rrt)
scene.pre_render(1)
if which == "sky":
    cam = rrt.make_camera(3840, 2160, math.radians(40.0), (0.0, 260.0, -420.0), (0.0, -1.0, 0.2))
elif which == "floor":  # straight down at the floor far away from the teapots
    cam = rrt.make_camera(3840, 2160, math.radians(40.0), (50000.0, 260.0, 50000.0), (0.0, 1.0, 0.001))
W, H = cam.width, cam.height
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); depth = torch.empty((H, W), dtype=torch.float32, device="cuda"); ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(frames):
    scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
torch.cuda.synchronize()
s = scene.stats()
print(which, "kernel_ms", round(s["ms_trace_kernels"], 3), "rays", s["rays_primary"], s["rays_shadow"], "hit pixels", int((ids >= 0).sum()))
