"""Profiling target: 5 frames of one variant of the C4 mesh scene (bench_extra.build_mesh_scene).  usage: python3 tools/r2_c4_pmc_target.py <variant>"""
import importlib, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
import bench_extra as bx
variant = sys.argv[1] if len(sys.argv) > 1 else "phong"
scene, cam, _ = bx.build_mesh_scene(rrt, variant=variant)
scene.pre_render(1)
W, H = cam.width, cam.height
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); depth = torch.empty((H, W), dtype=torch.float32, device="cuda"); ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
torch.cuda.synchronize()
s = scene.stats()
print({k: s[k] for k in ("rays_primary", "rays_secondary", "rays_shadow", "bvh_nodes_visited", "bvh_tris_tested", "ms_trace_kernels")})
