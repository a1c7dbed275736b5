"""Where the C4 frame's time goes: the full frame, the same without the light (no shadow rays, no per-light shading), and
a camera that sees only sky (fixed cost of the launches)."""
import importlib, os, sys, time, math
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
import bench_extra as bx

def timed(scene, cam, steps=5):
    W, H = cam.width, cam.height
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); depth = torch.empty((H, W), dtype=torch.float32, device="cuda"); ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
    torch.cuda.synchronize()
    k = 0.0; t0 = time.perf_counter()
    for _ in range(steps):
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
        s = scene.stats(); k += s["ms_trace_kernels"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    rays = s["rays_primary"] + s["rays_secondary"] + s["rays_shadow"]
    return dict(ms=round(dt, 3), kernel_ms=round(k / steps, 3), rays=rays, grays=round(rays / dt / 1e6, 2), nodes_per_ray=round(s["bvh_nodes_visited"] / rays, 2),
                tris_per_ray=round(s["bvh_tris_tested"] / rays, 2), shadow=s["rays_shadow"], secondary=s["rays_secondary"])

os.environ["RRT_BVH_INFO"] = "1"
scene, cam, ntri = bx.build_mesh_scene(rrt)
scene.pre_render(1)
print("full      ", timed(scene, cam))
sky = rrt.make_camera(3840, 2160, math.radians(40.0), (0.0, 260.0, -420.0), (0.0, -1.0, 0.2))
print("sky only  ", timed(scene, sky))
for name in ("nolight", "mirror", "glass"):
    sc, cm, _ = bx.build_mesh_scene(rrt, variant=name)
    sc.pre_render(1)
    print("%-10s" % name, timed(sc, cm))
