"""Photon-mapped scene2 (1024^2): rrt_params.gather_heavy_threshold sweep, one frame at a time and three in flight.
   usage: python tools/heavy_threshold_probe.py [photons]"""
import importlib, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench_extra
rrt = importlib.import_module("rust-raytracer_amd")
photons = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for thr in (0xFFFFFFFF, 0, 16384, 8192, 4096, 2048, 1024):
    params = rrt.default_params(photons_per_light=photons, caustic_photons_per_light=10_000, gather_heavy_threshold=thr)
    scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/scene2.txt"), params=params)
    scene.pre_render(1)
    W, H = cam.width, cam.height
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); depth = torch.empty((H, W), dtype=torch.float32, device="cuda"); ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / 6 * 1e3
    three = bench_extra.frames_in_flight_ms(torch, scene, cam, 12)
    print("threshold %10d: %.2f ms/frame alone, %.2f ms/frame with 3 in flight" % (thr, one, three), flush=True)
