"""CPU pre-flight for kernel changes: random scenes (tests/scene_kit.random_scene) through the DEVICE trace code in the
wave emulator (tools/cpu_emu/emu, ASan + LDS ownership check) against the oracle.  No GPU needed.
   usage: python tools/cpu_emu/check_random_scenes.py [first_seed [count]]      (EMU_TEAPOTS=1: also a 25 k-triangle mesh)"""
import os, subprocess, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob, scene_kit as kit
rrt = ob.rrt; abi = rrt._abi
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
emu = os.path.join(REPO, "tools/cpu_emu/emu")
tmp = tempfile.mkdtemp()
bad = 0
def teapots(scene):
    """2 x 2 teapots (25 280 triangles: a tree deep enough for the four-wide collapse and the threaded SAH build to matter),
    smooth, plus a floor; rays come from around the grid."""
    import bench_extra
    g = int(os.environ.get("EMU_TEAPOT_GRID", "2"))  # 4: 101 120 triangles, enough nodes for the threaded collapse (bvh_build.cpp)
    pos, nrm = bench_extra.teapot_grid(g, g)
    m = kit.monochrome(scene, (0.7, 0.5, 0.3))
    scene.add_object(kit.mesh(scene, pos, nrm, m, True))
    scene.add_object(kit.plane(scene, (0.0, -0.5, 0.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), m))
    return scene

seeds = list(range(first, first + count))
if os.environ.get("EMU_TEAPOTS"):
    seeds.append("teapots")
for seed in seeds:
    os.environ["RRT_DUMP_SCENE"] = os.path.join(tmp, "scene.bin")
    if seed == "teapots":
        gpu = teapots(rrt.Scene())
        try:
            gpu.pre_render(1)
        except rrt.RrtError as e:
            assert e.code == abi.ERR_DEVICE, e
        cpu = teapots(ob.OracleScene())
        o, d = kit.coherent_rays(int(os.environ.get("EMU_RAYS", "3000")), 5, origin_scale=60.0, towards=(0.0, 5.0, 0.0), spread=25.0)
        np.concatenate([o, d], axis=1).astype(np.float32).tofile(os.path.join(tmp, "rays.f32"))
        r = subprocess.run([emu, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "rays.f32"), os.path.join(tmp, "out.bin")], capture_output=True, text=True,
                           env=dict(os.environ, EMU_SHADOW_LIMIT="40"))
        if r.returncode:
            print("teapots EMULATOR FAILED\n", r.stderr[-2000:]); bad += 1; continue
        out = np.fromfile(os.path.join(tmp, "out.bin"), np.uint32).reshape(-1, 4)
        c = cpu.trace(o, d)
        sh = cpu.shadowtrace(o, d, np.full(o.shape[0], 40.0, np.float32))
        mism = [(out[:, 0] != c["object_id"]).sum(), (out[:, 1] != c["distance"].view(np.uint32)).sum(), (out[:, 2] != c["prim_id"]).sum(), (out[:, 3] != sh).sum()]
        print("teapots: mesh hits %.2f mismatches (id, t, prim, shadow) %s" % ((c["object_id"] == 0).mean(), mism))
        bad += any(mism)
        continue
    make_scene = kit.csg_mesh_scene if os.environ.get("EMU_SCENE") == "csgmesh" else kit.random_scene   # csgmesh: CSG trees with mesh leaves
    gpu = make_scene(rrt.Scene(), seed)
    try:
        gpu.pre_render(1)   # dumps the arrays, then fails for want of a GPU (or succeeds on a GPU box)
    except rrt.RrtError as e:
        if e.code != abi.ERR_DEVICE:
            print("seed", seed, "flatten refused:", e); continue
    cpu = make_scene(ob.OracleScene(), seed)
    make_rays = kit.coherent_rays if os.environ.get("EMU_COHERENT") else kit.random_rays   # coherent: the wave cull bites
    o, d = make_rays(int(os.environ.get("EMU_RAYS", "2000")), seed, origin_scale=5.0, towards=(0.0, 0.0, 0.0), spread=3.0)
    np.concatenate([o, d], axis=1).astype(np.float32).tofile(os.path.join(tmp, "rays.f32"))
    r = subprocess.run([emu, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "rays.f32"), os.path.join(tmp, "out.bin")], capture_output=True, text=True)
    if r.returncode:
        print("seed", seed, "EMULATOR FAILED\n", r.stderr[-2000:]); bad += 1; continue
    emu_overflow = "overflow 1" in r.stderr
    out = np.fromfile(os.path.join(tmp, "out.bin"), np.uint32).reshape(-1, 4)
    try:
        c = cpu.trace(o, d)
        sh = cpu.shadowtrace(o, d, np.full(o.shape[0], 6.0, np.float32))
    except ob.OracleError as e:
        print("seed", seed, "oracle:", e, "| emulator overflow flag:", emu_overflow)
        bad += 0 if emu_overflow else 1
        continue
    mism = [(out[:, 0] != c["object_id"]).sum(), (out[:, 1] != c["distance"].view(np.uint32)).sum(), (out[:, 2] != c["prim_id"]).sum(), (out[:, 3] != sh).sum()]
    hitf = (c["object_id"] != abi.MISS).mean()
    print("seed %d: objects %d hit %.2f mismatches (id, t, prim, shadow) %s%s" % (seed, gpu.object_count(), hitf, mism, " OVERFLOW FLAG WITHOUT ORACLE ERROR" if emu_overflow else ""))
    bad += any(mism) or emu_overflow
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
