"""One frame at a time, cut into B row bands rendered by B threads on one scene handle (the reference's render_rows split).
   usage: python tools/bands_probe.py [scene] [frames]"""
import importlib, os, sys, threading, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
rrt = importlib.import_module("rust-raytracer_amd")
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "assets/scenes/c2_quadric_csg_fresnel.txt")
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
scene, cam0 = rrt.Scene.load(path)
scene.pre_render(seed=1)
W, H = cam0.width, cam0.height
rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
depth = torch.empty((H, W), dtype=torch.float32, device="cuda")
ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
for B in (1, 2, 3, 4):
    rows = [(H * j // B) // 8 * 8 for j in range(B)] + [H]
    streams = [torch.cuda.Stream() for _ in range(B)]
    start = threading.Barrier(B + 1)
    done = threading.Barrier(B + 1)
    stop = False
    cam_box = [cam0]

    def work(j):
        while True:
            start.wait()
            if stop:
                return
            y0, y1 = rows[j], rows[j + 1]
            scene.render_device(cam_box[0], y0, y1, rgb[y0:].data_ptr(), depth[y0:].data_ptr(), ids[y0:].data_ptr(), streams[j].cuda_stream)
            done.wait()

    th = [threading.Thread(target=work, args=(j,)) for j in range(B)]
    for t in th: t.start()
    def frame(i):
        cam_box[0] = bench.orbit_camera(rrt, cam0, 0, i)
        start.wait(); done.wait()
    for i in range(4): frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames): frame(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("bands %d: %.3f ms/frame" % (B, dt / frames * 1e3), flush=True)
    stop = True
    start.wait()
    for t in th: t.join()
