"""Throughput of F independent frames in flight on one GPU: F host threads and F streams on ONE scene handle, or with
SEPARATE_HANDLES=1 on F handles (ctypes releases the GIL during rrt_render_device).   usage: python tools/frames_in_flight_probe.py [scene] [frames]"""
import importlib, os, sys, threading, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
rrt = importlib.import_module("rust-raytracer_amd")
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "assets/scenes/c2_quadric_csg_fresnel.txt")
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 48


class Worker:
    shared = None

    def __init__(self, k):
        self.k = k
        if os.environ.get("SEPARATE_HANDLES") or Worker.shared is None:
            scene, cam0 = rrt.Scene.load(path)
            scene.pre_render(seed=1)
            Worker.shared = (scene, cam0)
        self.scene, self.cam0 = Worker.shared
        W, H = self.cam0.width, self.cam0.height
        self.rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        self.depth = torch.empty((H, W), dtype=torch.float32, device="cuda")
        self.ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
        self.stream = torch.cuda.Stream()
        self.rays = 0

    def run(self, first, count, stride):
        H = self.cam0.height
        for i in range(first, count, stride):
            cam = bench.orbit_camera(rrt, self.cam0, 0, i)
            self.scene.render_device(cam, 0, H, self.rgb.data_ptr(), self.depth.data_ptr(), self.ids.data_ptr(), self.stream.cuda_stream)
            st = self.scene.stats()
            self.rays += st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]


for F in (1, 2, 3, 4, 6):
    ws = [Worker(k) for k in range(F)]
    for w in ws:
        w.run(w.k, 2 * F, F)   # warm-up: level-size predictions
        w.rays = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=w.run, args=(w.k, frames, F)) for w in ws]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("frames in flight %d: %.3f ms/frame, %.0f Mrays/s" % (F, dt / frames * 1e3, sum(w.rays for w in ws) / dt / 1e6), flush=True)
    del ws
    Worker.shared = None
