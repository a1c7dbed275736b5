"""C5 per-rank frame time under variants of the fine photon grid (development helper): RRT_DENSE_DIV and the heavy-query
threshold.   usage: python tools/c5_frame_probe.py [photons=100000000]"""
import importlib, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
photons = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
W, H = 3840, 2160
rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
depth = torch.zeros((H, W), dtype=torch.float32, device="cuda")
ids = torch.zeros((H, W), dtype=torch.int32, device="cuda")
for div, heavy, factor in ((3, 0, 104), (3, 0, 208), (3, 0, 416), (3, 0, 832), (3, 0, 1664), (2, 0, 208), (4, 0, 208)):
    os.environ["RRT_DENSE_DIV"] = str(div)
    os.environ["RRT_HEAVY_FACTOR"] = str(factor)
    params = rrt.default_params(photons_per_light=photons, caustic_photons_per_light=max(10_000, photons // 500), gather_heavy_threshold=heavy)
    scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/scene2.txt"), params=params, mode_override=rrt._abi.MODE_PHOTON)
    cam.width, cam.height = W, H
    scene.pre_render(1)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr())
        torch.cuda.synchronize(); ts.append(time.time() - t0)
    st = scene.stats()
    print("factor %d div %d heavy %s: frame %.3f s, kernels %.1f ms, build %.1f ms, gathered %d candidates %d" % (
        factor, div, "auto" if heavy == 0 else "off", min(ts), st["ms_trace_kernels"], st["ms_photon_build"], st["photons_gathered"], st["gather_candidates"]), flush=True)
    del scene
    torch.cuda.empty_cache()
