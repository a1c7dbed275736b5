"""CPU pre-flight for the sphere / ellipsoid boxes: bundles of 64 rays that share an origin (near and very far) and
graze the silhouettes of stand-alone spheres and ellipsoids within a few ulps.  The reference decides such rays by the
sign of a rounded discriminant; the device code must agree bit for bit, also when an experiment culls such objects by
(inflated) boxes per ray or per wave (tools/experiments/wave_cull.patch; RRT_BOUND_G=0 there makes this check fail, as
it should).  Runs the DEVICE code in the wave emulator against the oracle.  No GPU needed.
   usage: python tools/cpu_emu/check_grazing.py [seeds]"""
import os, subprocess, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob, scene_kit as kit
rrt = ob.rrt; abi = rrt._abi
emu = os.path.join(REPO, "tools/cpu_emu/emu")
tmp = tempfile.mkdtemp()


def build(scene, rng_seed):
    rng = np.random.default_rng(rng_seed)
    m = kit.monochrome(scene)
    shapes = []
    for i in range(6):
        c = rng.uniform(-3, 3, 3).astype(np.float32)
        if i % 2 == 0:
            r = np.float32(rng.uniform(0.2, 1.5))
            scene.add_object(kit.sphere(scene, tuple(c), float(r), m))
            shapes.append((c, np.array([r, r, r], np.float32)))
        else:
            a = rng.uniform(0.3, 4.0, 3).astype(np.float32)   # a x^2 + b y^2 + c z^2 = k
            k = np.float32(rng.uniform(0.3, 1.5))
            q = np.zeros(10, np.float32); q[0], q[4], q[7], q[9] = a[0], a[1], a[2], -k
            h = kit.quadric(scene, tuple(q), m)
            scene.object_transform(h, kit.translation(tuple(c)))
            scene.add_object(h)
            shapes.append((c, np.sqrt(k / a).astype(np.float32)))
    kit.light(scene, abi.LIGHT_POINT, position=(0.0, 8.0, 0.0))
    return shapes


def grazing_rays(shapes, seed):
    rng = np.random.default_rng(seed + 1000)
    O, D = [], []
    for _ in range(96):
        c, semi = shapes[rng.integers(0, len(shapes))]
        dist = 10.0 ** rng.uniform(0.3, 3.5)                 # 2 ... 3000 units away
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        o = (c + u * dist).astype(np.float32)
        # silhouette points of the ellipsoid seen from o (exact for a sphere, close for an ellipsoid), then offsets of a
        # few 1e-7 relative either side, so the bundle straddles the limb
        w = rng.normal(size=(64, 3)); w -= (w @ u)[:, None] * u; w /= np.linalg.norm(w, axis=1, keepdims=True)
        r_eff = np.linalg.norm(w * semi, axis=1)
        sin_a = np.clip(r_eff / dist, 0, 0.999)
        limb = c + (w * r_eff[:, None]) * np.sqrt(1 - sin_a ** 2)[:, None] + u * (r_eff * sin_a)[:, None]
        scale = 1.0 + rng.choice([0.0, 1e-7, -1e-7, 3e-7, -3e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-3, -1e-3], 64)
        t = c + (limb - c) * scale[:, None]
        d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
        O.append(np.repeat(o[None], 64, 0)); D.append(d.astype(np.float32))
    return np.concatenate(O).astype(np.float32), np.concatenate(D).astype(np.float32)


bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    os.environ["RRT_DUMP_SCENE"] = os.path.join(tmp, "scene.bin")
    gpu = rrt.Scene(); shapes = build(gpu, seed)
    try:
        gpu.pre_render(1)
    except rrt.RrtError as e:
        if e.code != abi.ERR_DEVICE: raise
    cpu = ob.OracleScene(); build(cpu, seed)
    o, d = grazing_rays(shapes, seed)
    np.concatenate([o, d], axis=1).astype(np.float32).tofile(os.path.join(tmp, "rays.f32"))
    os.environ["EMU_SHADOW_LIMIT"] = "1e9"
    r = subprocess.run([emu, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "rays.f32"), os.path.join(tmp, "out.bin")], capture_output=True, text=True)
    if r.returncode:
        print("seed", seed, "EMULATOR FAILED\n", r.stderr[-2000:]); bad += 1; continue
    out = np.fromfile(os.path.join(tmp, "out.bin"), np.uint32).reshape(-1, 4)
    c = cpu.trace(o, d)
    sh = cpu.shadowtrace(o, d, np.full(o.shape[0], 1e9, np.float32))
    mism = [int((out[:, 0] != c["object_id"]).sum()), int((out[:, 1] != c["distance"].view(np.uint32)).sum()), int((out[:, 3] != sh).sum())]
    print("seed %d: hit %.2f mismatches (id, t, shadow) %s | %s" % (seed, (c["object_id"] != abi.MISS).mean(), mism, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ""))
    bad += any(mism)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
