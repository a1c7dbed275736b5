"""Index-build A/B (development helper): the photon-mapped frame of scene2 and the build time, for the library named by
RRT_LIB (default: the in-tree build).  Prints a hash of the frame per photon count, so that two builds can be compared
(tools/index_ab.sh runs the saved start-of-round library and the current one).
   usage: python tools/index_ab.py [photon counts ...]"""
import hashlib, importlib, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
counts = [int(a) for a in sys.argv[1:]] or [1_000_000]
for photons in counts:
    params = rrt.default_params(photons_per_light=photons, caustic_photons_per_light=max(10_000, photons // 500))
    scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/scene2.txt"), params=params)
    scene.pre_render(1)  # warm-up: allocations
    best = 1e9
    for it in range(3):
        scene.pre_render(1)
        best = min(best, scene.stats()["ms_photon_build"])
    cam.width = cam.height = 512
    t0 = time.perf_counter()
    rgb, depth, ids = scene.render(cam)
    dt = time.perf_counter() - t0
    st = scene.stats()
    h = hashlib.sha256(np.ascontiguousarray(rgb).tobytes() + np.ascontiguousarray(depth).tobytes() + np.ascontiguousarray(ids).tobytes()).hexdigest()[:16]
    print("photons %d records %d index_build_ms %.3f shoot_ms %.3f frame_s %.4f gathered %d candidates %d frame_hash %s" % (
        photons, st["photons_stored_regular"] + st["photons_stored_caustic"], best, st["ms_photon_shoot"], dt, st["photons_gathered"],
        st["gather_candidates"], h), flush=True)
