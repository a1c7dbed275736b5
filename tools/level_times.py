"""Per-level timings of one frame of a mesh variant (RRT_LEVEL_SYNC=1 RRT_LEVEL_STATS=1 python3 tools/level_times.py mirror glass)."""
import importlib, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
import bench_extra as bx
for variant in sys.argv[1:] or ["mirror"]:
    if variant == "photon":  # BASELINE configs[2]: scene2 in photon mode, 1 M + 10 k photons
        scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/scene2.txt"), params=rrt.default_params(photons_per_light=1_000_000, caustic_photons_per_light=10_000))
    elif variant == "csg":  # BASELINE configs[1]
        scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/c2_quadric_csg_fresnel.txt"))
    else:
        scene, cam, ntri = bx.build_mesh_scene(rrt, variant=variant)
    scene.pre_render(1)
    W, H = cam.width, cam.height
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); depth = torch.empty((H, W), dtype=torch.float32, device="cuda"); ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    print("==", variant, flush=True)
    for _ in range(2):
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
        torch.cuda.synchronize()
        print("-- frame kernel_ms", round(scene.stats()["ms_trace_kernels"], 3), flush=True)
