"""Fine photon grids against the per-record gather (development helper / evidence for DESIGN.md): the same maps indexed with
dense cells of edge r / div for div = 1 (every record of the cube's 3 x 3 x 3 cells is read) and div = 2, 4 (cells inside the
cube are added as a whole), the same frame rendered over each.  Counts must be equal, colours differ by f32 rounding only.
   usage: python tools/fine_grid_ab.py [photons=20000000] [width=3840] [height=2160] [divs=1,2,4]"""
import importlib, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
photons = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
divs = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,2,4").split(",")]
params = rrt.default_params(photons_per_light=photons, caustic_photons_per_light=max(10_000, photons // 500))
scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/scene2.txt"), params=params, mode_override=rrt._abi.MODE_PHOTON)
cam.width, cam.height = W, H
ref = None
out = []
for div in divs:
    os.environ["RRT_DENSE_DIV"] = str(div)
    scene.pre_render(1)
    build = scene.stats()["ms_photon_build"]
    scene.pre_render(1)
    build = min(build, scene.stats()["ms_photon_build"])
    scene.render(cam)  # sizes the queues
    t0 = time.perf_counter()
    rgb, depth, ids = scene.render(cam)
    dt = time.perf_counter() - t0
    st = scene.stats()
    q = rrt.quantise(rgb).astype(np.int32)
    row = {"photons": photons, "div": div, "records": st["photons_stored_regular"] + st["photons_stored_caustic"], "index_build_ms": build,
           "frame_s": dt, "kernel_ms": st["ms_trace_kernels"], "gathered": st["photons_gathered"], "candidates": st["gather_candidates"], "queries": st["gather_queries"]}
    if ref is None:
        ref = (q, st["photons_gathered"], ids.copy(), depth.copy())
    else:
        d = np.abs(q - ref[0])
        row.update({"gathered_equal": st["photons_gathered"] == ref[1], "rgb_mae_255": float(d.mean()), "rgb_max_255": int(d.max()),
                    "ids_equal": bool(np.array_equal(ids, ref[2])), "depth_equal": bool(np.array_equal(depth.view(np.uint32), ref[3].view(np.uint32)))})
    print(json.dumps(row), flush=True)
