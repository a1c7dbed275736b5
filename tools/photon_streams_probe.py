"""Photon-mapped scene2 (1 M photons, 1024^2), ONE frame at a time: rrt_params.render_streams = 1, 2, 4, 8 and chunk sizes."""
import importlib, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
for streams, chunk in ((1, 0), (2, 0), (4, 0), (8, 0), (1, 1 << 18), (1, 1 << 16)):
    params = rrt.default_params(photons_per_light=1_000_000, caustic_photons_per_light=10_000, render_streams=streams, chunk_pixels=chunk)
    scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/scene2.txt"), params=params)
    scene.pre_render(1)
    W, H = cam.width, cam.height
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); depth = torch.empty((H, W), dtype=torch.float32, device="cuda"); ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
    torch.cuda.synchronize()
    print("render_streams %d chunk %d: %.2f ms/frame (kernels %.2f, repeated %d)" % (streams, chunk, (time.perf_counter() - t0) / 5 * 1e3, scene.stats()["ms_trace_kernels"], scene.stats()["frames_repeated"]), flush=True)
