"""Where does level 0 of the C2 frame go?  Renders 1080p variants of the C2 scene with all materials made
non-recursive (Monochrome), objects removed step by step, and prints the trace-kernel time of each."""
import importlib, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
rrt = importlib.import_module("rust-raytracer_amd")
text = open(os.path.join(REPO, "assets/scenes/c2_quadric_csg_fresnel.txt")).read()
paras = []
for p in re.split(r"\n\s*\n", text):
    p = "\n".join(l for l in p.split("\n") if not l.startswith("#")).strip("\n")
    if p.strip():
        paras.append(p)
def flat_material(p):
    if not p.startswith("object"):
        return p
    head = p.split("    material")[0].rstrip("\n")
    return head + "\n    material Monochrome\n        colour 0.7 0.7 0.7\n        ambient 0.1\n        shininess 100.0"
objs = [flat_material(p) for p in paras if p.startswith("object")]
rest = [p for p in paras if not p.startswith("object")]
names = [o.split("\n")[0] for o in objs]
def run(label, keep, recursive=False):
    chosen = [(paras_o if recursive else o) for o, paras_o, k in zip(objs, [p for p in paras if p.startswith("object")], keep) if k]
    sc, cam = rrt.Scene.from_text("\n\n".join(rest + chosen) + "\n")
    sc.pre_render(1)
    W, H = cam.width, cam.height
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda"); dep = torch.empty((H, W), dtype=torch.float32, device="cuda"); ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    ms = []
    for i in range(6):
        sc.render_device(cam, 0, H, rgb.data_ptr(), dep.data_ptr(), ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
        st = sc.stats(); ms.append(st["ms_trace_kernels"])
    print("%-46s trace %.3f ms  rays %d (shadow %d)" % (label, min(ms[1:]), st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"], st["rays_shadow"]), flush=True)
n = len(objs)
print(names)
run("all objects, original materials (full recursion)", [True] * n, recursive=True)
run("all objects, flat materials (level 0 only)", [True] * n)
run("without pipe and coin", [not ("Mario" in x) for x in names])
run("plane + 3 quadrics", [("Plane" in x or "Quadratic" in x) for x in names])
run("plane only", ["Plane" in x for x in names])
run("pipe + coin + plane", [("Plane" in x or "Mario" in x) for x in names])
