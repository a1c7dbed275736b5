"""Per-primitive cost probe: k_batch_trace on sub-scenes of C2, 64K rays (one wave per SIMD: latency) and 2M rays
(throughput).  Run under `rocprofv3 --kernel-trace --output-format csv` and read the k_batch_trace durations in order."""
import importlib, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
HEAD = "scene Scene\n\ncamera Camera\n    width 64\n    height 64\n    position 0.0 1.0 -7.0\n    lookat 0.0 -0.1 1.0\n\nlight Point\n    position -1.5 4.0 -4.5\n    colour 1.6\n\n"
PLANE = "object Plane\n    point 0.0 -1.5 0.0\n    up 0.0 0.0 1.0\n    normal 0.0 1.0 0.0\n    material Monochrome\n        colour 0.8 0.8 0.8\n        shininess 100.0\n\n"
QUAD = "object Quadratic\n    a 2.0\n    e 2.0\n    h 2.0\n    j -1.0\n    translate %f 0.0 0.0\n    material Monochrome\n        colour 1.0 0.0 0.0\n        shininess 100.0\n\n"
SPH = "object Sphere\n    centre %f -0.6 -2.5\n    radius 0.9\n    material Monochrome\n        colour 0.2 0.2 0.3\n        shininess 100.0\n\n"
PIPE = "object MarioPipe\n    translate 0.6 -1.5 1.0\n    scale 1.0\n    material Monochrome\n        colour 0.0 1.0 0.0\n        shininess 100.0\n\n"
COIN = "object MarioCoin\n    translate -0.6 0.9 0.5\n    scale 1.2\n    material Monochrome\n        colour 1.0 1.0 0.0\n        shininess 100.0\n\n"
BOX = "object Cuboid\n    corner -1.0 -1.0 0.0\n    size 2.0 2.0 2.0\n    material Monochrome\n        colour 1.0 1.0 0.0\n        shininess 100.0\n\n"
SCENES = [("plane", PLANE), ("plane+3quad", PLANE + QUAD % 2.3 + QUAD % 0.65 + QUAD % -2.5), ("plane+2sph", PLANE + SPH % 0.4 + SPH % -1.0),
          ("plane+pipe", PLANE + PIPE), ("plane+coin", PLANE + COIN), ("plane+3box", PLANE + BOX * 3), ("c2", None)]
rng = np.random.default_rng(3)
def rays(n):
    o = (rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32(2.0) + np.array([0, 1, -7], np.float32)
    tgt = (rng.random((n, 3), dtype=np.float32) - 0.5) * np.array([3.0, 4.0, 3.0], np.float32) + np.array([0, 0.5, 0.5], np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)
small, big = rays(65536), rays(1 << 21)
for name, body in SCENES:
    if body is None:
        sc, _ = rrt.Scene.load(os.path.join(REPO, "assets/scenes/c2_quadric_csg_fresnel.txt"))
    else:
        sc, _ = rrt.Scene.from_text(HEAD + body)
    sc.pre_render(1)
    sc.trace(*small)
    r = sc.trace(*small)
    sc.trace(*big)
    print(name, "hit fraction %.2f" % (r["object_id"] != 0xFFFFFFFF).mean(), flush=True)
