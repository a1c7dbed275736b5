"""Quick GPU-vs-oracle comparison over the reference scenes (development helper)."""
import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob
rrt = ob.rrt
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["quadratics", "reflection_refraction", "textures", "scene", "pm_diffuse", "pm_caustics", "scene2"]
for name in names:
    path = os.path.join(REPO, "assets/scenes", name + ".txt")
    so, cam = ob.OracleScene.load(path, mode_override=0)
    sg, cam2 = rrt.Scene.load(path, mode_override=0)
    cam.width = cam.height = res
    t = time.time(); rgb_o, dep_o, id_o = so.render(cam); t_o = time.time() - t
    sg.pre_render(1)
    t = time.time(); rgb_g, dep_g, id_g = sg.render(cam); t_g = time.time() - t
    st = sg.stats()
    qo, qg = ob.quantise(rgb_o).astype(int), rrt.quantise(rgb_g).astype(int)
    print("%-22s id_mismatch=%d depth_mismatch=%d rgb_max_q_diff=%d rgb_exact=%.4f oracle=%.2fs gpu=%.3fs kernel_ms=%.3f sec=%d/%d shadow=%d/%d" % (
        name, (id_o != id_g).sum(), (dep_o.view(np.uint32) != dep_g.view(np.uint32)).sum(), np.abs(qo - qg).max(),
        (rgb_o.view(np.uint32) == rgb_g.view(np.uint32)).mean(), t_o, t_g, st["ms_trace_kernels"],
        st["rays_secondary"], so.stats()["rays_secondary"], st["rays_shadow"], so.stats()["rays_shadow"]), flush=True)
