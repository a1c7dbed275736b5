"""Device LBVH builder, warm: builds the 1 068 080-triangle mesh scene three times in one process with RRT_LBVH_TIMING=1
(phases on stderr) and prints ms_bvh_build / pre_render wall time of each.   usage (GPU box): python tools/lbvh_build_timing.py"""
import importlib, json, os, sys, time
os.environ["RRT_LBVH_TIMING"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: F401  (same HIP runtime as the library)
rrt = importlib.import_module("rust-raytracer_amd")
import bench_extra
out = []
for builder in (1, 1, 1, 2, 2, 2, 0):
    scene, cam, ntri = bench_extra.build_mesh_scene(rrt, bvh_builder=builder)
    t0 = time.perf_counter()
    scene.pre_render(1)
    dt = (time.perf_counter() - t0) * 1e3
    out.append({"builder": ("host SAH", "device LBVH", "device SAH")[builder], "triangles": ntri, "ms_bvh_build": scene.stats()["ms_bvh_build"], "ms_pre_render": dt})
    sys.stderr.write("----\n")
print(json.dumps(out, indent=1))
