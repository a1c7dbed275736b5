"""Lane occupancy of the BVH walk on the C4 frame, from instrumented builds of librrt (tools/build_variant.sh <name> -DRRT_WALK_STATS=1|2,
RRT_LIB=variants/librrt_<name>.so).  The three sums travel in the photon statistics' slots of rrt_stats.

  RRT_WALK_STATS=1  time (s_memtime) a lane of a primary ray spends inside mesh_closest before its walk is over, against the time its wave
                    spends there: the rest is a finished lane waiting for the wave (what refilling lanes could win back)
  RRT_WALK_STATS=2  wave-level iteration counts of the node loop and the triangle loop against the per-lane counts
"""
import importlib, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
import bench_extra as bx

which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variants = sys.argv[2:] or ["phong"]
for variant in variants:
    scene, cam, ntri = bx.build_mesh_scene(rrt, variant="phong" if variant == "sky" else variant)
    if variant == "sky":  # a camera that sees nothing: the fixed cost of a wave
        import math
        cam = rrt.make_camera(3840, 2160, math.radians(40.0), (0.0, 260.0, -420.0), (0.0, -1.0, 0.2))
    scene.pre_render(1)
    W, H = cam.width, cam.height
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); depth = torch.empty((H, W), dtype=torch.float32, device="cuda"); ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
    torch.cuda.synchronize()
    s = scene.stats()
    a, b, c = s["photons_gathered"], s["gather_queries"], s["gather_candidates"]
    rays = s["rays_primary"] + s["rays_secondary"] + s["rays_shadow"]
    print(variant, "rays", rays, "kernel_ms", round(s["ms_trace_kernels"], 3))
    if which == 1:
        print("  primary walks: %d lanes; lane busy-or-waiting-for-its-phase %.1f %% of the wave's time in the walk, finished and idle %.1f %%; mean wave walk %.0f cycles"
              % (c, 100.0 * a / max(1, b), 100.0 * (1 - a / max(1, b)), b / max(1, c)))
    elif which == 3:
        print("  wave time: whole kernel body %d cycles per wave (%d waves); trace() %.1f %%, shadowtrace() %.1f %%" % (c / (W * H / 64), W * H / 64, 100.0 * a / c, 100.0 * b / c))
    elif which == 4:
        print("  wave time: whole kernel body %d cycles per wave; ray set-up %.1f %%, eval_hit %.1f %%" % (c / (W * H / 64), 100.0 * a / c, 100.0 * b / c))
    elif which == 8:
        print("  node iterations %d: all active lanes on one node in %.1f %% of them; lanes on the first lane's node per iteration %.1f of %.1f active"
              % (a, 100.0 * b / max(1, a), c / max(1, a), s["bvh_nodes_visited"] / 4 / max(1, a)))
    elif which == 5:
        nb = int(os.environ.get("RRT_POOL_NB", "8")); pools = W * H / 64 / nb
        print("  pool kernel, cycles per pool of %d batches: rays + other records %d, primary walks %d, shadow set-up (2a) %d" % (nb, a / pools, b / pools, c / pools))
    elif which == 6:
        nb = int(os.environ.get("RRT_POOL_NB", "8")); pools = W * H / 64 / nb
        print("  pool kernel, cycles per pool of %d batches: shadow walks %d, shading (2c) %d, whole pool %d (= %d per batch)" % (nb, a / pools, b / pools, c / pools, c / pools / nb))
    elif which == 7:
        nb = int(os.environ.get("RRT_POOL_NB", "8")); pools = W * H / 64 / nb
        print("  pool kernel, rounds per pool of %d batches: primary walks %.1f, shadow walks %.1f" % (nb, a / pools, b / pools))
    else:
        nodes, tris = s["bvh_nodes_visited"] / 4, s["bvh_tris_tested"]
        print("  node loop: %d wave iterations, lanes active %.1f %%; triangle loop: %d wave iterations, lanes active %.1f %%; %d outer rounds (%.2f node iterations per round)"
              % (a, 100.0 * nodes / max(1, 64 * a), b, 100.0 * tris / max(1, 64 * b), c, a / max(1, c)))
