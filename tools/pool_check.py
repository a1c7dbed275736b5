"""k_trace_shade_pool against k_trace_shade on the C4 frame and its mirror / glass variants: the frames must be the same bits, the ray
counts equal; prints the time of both.  RRT_POOL_NB is read when a scene is flattened (runtime.cpp), so one process can build both."""
import importlib, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
import bench_extra as bx


def frame(variant, nb, steps=8):
    os.environ["RRT_POOL_NB"] = str(nb)
    scene, cam, _ = bx.build_mesh_scene(rrt, variant=variant)
    scene.pre_render(1)
    W, H = cam.width, cam.height
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); depth = torch.empty((H, W), dtype=torch.float32, device="cuda"); ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
    torch.cuda.synchronize()
    k = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st)
        s = scene.stats(); k += s["ms_trace_kernels"]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    rays = (s["rays_primary"], s["rays_secondary"], s["rays_shadow"])
    return rgb, depth, ids, ms, k / steps, rays, s


variants = sys.argv[1:] or ["phong", "mirror", "glass"]
ok = True
for v in variants:
    ref = frame(v, 0)
    line = "%-7s classic %.3f ms (kernels %.3f)" % (v, ref[3], ref[4])
    for nb in (2, 4, 8):
        got = frame(v, nb)
        same = bool(torch.equal(ref[0].view(torch.int32), got[0].view(torch.int32)) and torch.equal(ref[1].view(torch.int32), got[1].view(torch.int32)) and torch.equal(ref[2], got[2]))
        same_rays = ref[5] == got[5]
        ok &= same and same_rays
        line += " | pool %d: %.3f ms (kernels %.3f) %s%s" % (nb, got[3], got[4], "same bits" if same else "FRAMES DIFFER", "" if same_rays else " RAY COUNTS DIFFER %s vs %s" % (ref[5], got[5]))
        if not same:
            d = (ref[0] != got[0]).any(dim=2) | (ref[1] != got[1]) | (ref[2] != got[2])
            ys, xs = torch.nonzero(d, as_tuple=True)
            line += " (%d pixels, first at x=%d y=%d)" % (int(d.sum()), int(xs[0]), int(ys[0]))
    print(line, flush=True)
    print("        box tests per ray classic %.2f, pool %.2f; %d rays" % (ref[6]["bvh_nodes_visited"] / sum(ref[5]), got[6]["bvh_nodes_visited"] / sum(got[5]), sum(ref[5])), flush=True)
sys.exit(0 if ok else 1)
