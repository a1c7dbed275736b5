"""GPU-vs-oracle photon pass comparison (development helper)."""
import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob
rrt = ob.rrt
name = sys.argv[1] if len(sys.argv) > 1 else "pm_caustics"
nph = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ncau = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
res = int(sys.argv[4]) if len(sys.argv) > 4 else 128
path = os.path.join(REPO, "assets/scenes", name + ".txt")
params = rrt.default_params(photons_per_light=nph, caustic_photons_per_light=ncau)
so, cam = ob.OracleScene.load(path, params=params, mode_override=1)
sg, _ = rrt.Scene.load(path, params=params, mode_override=1)
t = time.time(); so.pre_render(7); t_o = time.time() - t
t = time.time(); sg.pre_render(7); t_g = time.time() - t
print("pre_render oracle %.2fs gpu %.3fs" % (t_o, t_g), sg.stats())
for which in (0, 1):
    po, pg = so.photon_read(which), sg.photon_read(which)
    print("map", which, "count oracle", len(po["key"]), "gpu", len(pg["key"]))
    common, io, ig = np.intersect1d(po["key"], pg["key"], return_indices=True)
    same_pos = (po["position"][io].view(np.uint32) == pg["position"][ig].view(np.uint32)).all(axis=1)
    same_all = same_pos & (po["incident"][io].view(np.uint32) == pg["incident"][ig].view(np.uint32)).all(axis=1) & \
        (po["intensity"][io].view(np.uint32) == pg["intensity"][ig].view(np.uint32)).all(axis=1) & (po["type"][io] == pg["type"][ig])
    print("   common keys", len(common), "bit-identical records", same_all.sum(), "pos-identical", same_pos.sum())
cam.width = cam.height = res
t = time.time(); rgb_o, dep_o, id_o = so.render(cam); t_o = time.time() - t
t = time.time(); rgb_g, dep_g, id_g = sg.render(cam); t_g = time.time() - t
qo, qg = ob.quantise(rgb_o).astype(int), rrt.quantise(rgb_g).astype(int)
d = np.abs(qo - qg)
print("render oracle %.2fs gpu %.3fs  id_mismatch=%d depth_mismatch=%d  rgb MAE=%.4f max=%d frac>1=%.5f" % (
    t_o, t_g, (id_o != id_g).sum(), (dep_o.view(np.uint32) != dep_g.view(np.uint32)).sum(), d.mean(), d.max(), (d > 1).mean()))
print(sg.stats()); print(so.stats())
from PIL import Image
Image.fromarray(qg.astype(np.uint8)).save(os.path.join(REPO, "gpurun_out", "photon_%s_gpu.png" % name))
Image.fromarray(qo.astype(np.uint8)).save(os.path.join(REPO, "gpurun_out", "photon_%s_cpu.png" % name))
