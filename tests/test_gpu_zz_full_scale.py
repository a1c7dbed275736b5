"""BASELINE configs[4] at full size on one GPU (collected last: the big one).

A rank of the 100 M-photon job ends up holding the complete maps and rendering a band of the 4K frame.  The oracle cannot
follow to this size, so the checks are the size-independent ones of tools/c5_scale_check.py: the library's own record
accounting over the photon-index ranges, rrt_photon_query against a brute-force closed-cube search over the exported
records (counts exact, k smallest squared distances bit-exact) and a finite frame."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_c5_per_rank_share_100m_photons_4k(repo):
    spec = importlib.util.spec_from_file_location("c5_scale_check", os.path.join(repo, "tools/c5_scale_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    photons = int(os.environ.get("RRT_TEST_C5_PHOTONS", "100000000"))
    rep = mod.run(photons, 3840, 2160)
    assert rep["photons_emitted"] == photons + max(10_000, photons // 500)
    assert rep["records_regular"] >= photons
    assert rep["gather_queries"] >= 3840 * 2160          # every pixel's first hit queries both maps, plus the vueons' hits
    assert rep["query_map0"]["largest_count"] > 1000      # the queries really were where the photons are


def test_c4_million_triangle_mesh_4k_both_builders_and_oracle_sample(repo):
    """BASELINE configs[3] at full size: 1 068 080 triangles at 3840x2160.  The reference's O(N) loop per ray is out of reach
    for whole frames, so: (1) the frames of the host-SAH tree and of the device-LBVH tree are identical bit for bit (ids,
    depth, rgb); (2) a sample of the 4K frame's camera rays goes through the oracle's brute force over all the triangles:
    object, triangle, distance, position and normal bit-exact; (3) the frame is band-independent."""
    import sys

    import numpy as np

    sys.path.insert(0, repo)
    sys.path.insert(0, os.path.join(repo, "tests"))
    import bench_extra
    import oracle_binding as ob
    import scene_kit as kit

    rrt = ob.rrt
    abi = rrt._abi
    pos, nrm = bench_extra.teapot_grid()
    cam = rrt.make_camera(3840, 2160, 0.6981317, (0.0, 260.0, -420.0), (0.0, 0.55, 1.0))

    def build(scene):
        scene.add_object(kit.mesh(scene, pos, nrm, kit.monochrome(scene, (0.8, 0.6, 0.2), 0.1, 50.0), True))
        scene.add_object(kit.plane(scene, (0, 0, 0), (0, 0, 1), (0, 1, 0), kit.monochrome(scene, (0.3, 0.5, 0.3), 0.1, 100.0)))
        kit.light(scene, abi.LIGHT_POINT, position=(0.0, 400.0, -300.0), colour=(1.2, 1.2, 1.2))
        return scene

    frames = []
    scenes = []
    for builder in (abi.BVH_HOST_SAH, abi.BVH_DEVICE_LBVH, abi.BVH_DEVICE_SAH):
        s = build(rrt.Scene(params=rrt.default_params(bvh_builder=builder)))
        s.pre_render(1)
        frames.append(s.render(cam))
        scenes.append(s)
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
    for other in frames[1:]:
        for a, b in zip(frames[0], other):
            assert np.array_equal(bits(a), bits(b))
    rgb, depth, ids = frames[0]
    assert (ids == 0).mean() > 0.1 and (ids == 1).mean() > 0.05          # teapots and floor both in view
    top, bottom = scenes[1].render(cam, 0, 1000), scenes[1].render(cam, 1000, 2160)
    assert np.array_equal(bits(np.concatenate([top[0], bottom[0]])), bits(rgb))
    # oracle sample: 4096 camera rays — a third on teapot silhouettes (pixels whose neighbour shows another object), a third on
    # the teapots, the rest anywhere — and the shadow rays of their hits, all through the oracle's brute force over every triangle
    rng = np.random.default_rng(4)
    edge = np.zeros(ids.shape, bool)
    edge[:, 1:] |= ids[:, 1:] != ids[:, :-1]
    edge[1:, :] |= ids[1:, :] != ids[:-1, :]
    on_edge, on_mesh = np.argwhere(edge), np.argwhere(ids == 0)
    assert len(on_edge) > 10000
    px = np.concatenate([on_edge[rng.choice(len(on_edge), 1400, replace=False)][:, ::-1], on_mesh[rng.choice(len(on_mesh), 1400, replace=False)][:, ::-1],
                         np.stack([rng.integers(0, 3840, 1296), rng.integers(0, 2160, 1296)], axis=1)])
    rays = [rrt.camera_ray(cam, int(x), int(y)) for x, y in px]
    o = np.ascontiguousarray([r[0] for r in rays], np.float32)
    d = np.ascontiguousarray([r[1] for r in rays], np.float32)
    cpu = build(ob.OracleScene())
    cpu.pre_render(1, threads=1)
    c = threaded(lambda idx: cpu.trace(o[idx], d[idx]), len(o))
    for s in scenes:
        g = s.trace(o, d)
        for k in ("object_id", "prim_id", "distance", "position", "normal"):
            assert np.array_equal(bits(g[k]), bits(c[k])), k
    assert np.array_equal(c["object_id"], ids[px[:, 1], px[:, 0]])       # and they are the frame's primary hits
    assert np.array_equal(bits(c["distance"]), bits(depth[px[:, 1], px[:, 0]]))
    # Scene::shadowtrace from every hit towards the light, limit |ldir| = 1 (scene.rs:106-111): grazing rays that start ON the mesh
    hit = c["object_id"] != 0xFFFFFFFF
    pos = c["position"][hit]
    ldir = pos - np.array([0.0, 400.0, -300.0], np.float32)
    ldir = (ldir / np.sqrt((ldir * ldir).sum(axis=1, dtype=np.float32))[:, None]).astype(np.float32)
    sd = -ldir
    so = (pos + sd * np.float32(0.0001)).astype(np.float32)
    limits = np.ones(len(so), np.float32)
    occ_c = threaded(lambda idx: {"o": cpu.shadowtrace(so[idx], sd[idx], limits[idx])}, len(so))["o"]
    assert 0.02 < occ_c.mean() < 0.9                                      # some hits are in shadow, some are lit
    for s in scenes:
        assert np.array_equal(s.shadowtrace(so, sd, limits), occ_c)


def threaded(fn, n, threads=None):
    """fn(index array) -> dict of arrays, over min(16, cpus) host threads (the oracle's calls release the GIL); results concatenated"""
    import threading

    import numpy as np

    threads = threads or min(16, os.cpu_count() or 1)
    parts = [p for p in np.array_split(np.arange(n), threads) if len(p)]
    out = [None] * len(parts)

    def work(j):
        out[j] = fn(parts[j])

    th = [threading.Thread(target=work, args=(j,)) for j in range(len(parts))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return {k: np.concatenate([p[k] for p in out]) for k in out[0]}


@pytest.mark.parametrize("variant", ["mirror", "glass"])
def test_c4_secondary_rays_through_the_million_triangle_bvh(repo, variant):
    """BASELINE configs[3] with reflective / refractive teapots (bench_extra.build_mesh_scene variants): reflection and
    refraction rays bounce incoherently through the 1 068 080-triangle BVH, five levels deep.  The whole recursion of a
    sample of the 4K frame's camera rays goes through the oracle (brute force over every triangle for every bounce and
    shadow ray): object ids and depth bit-exact, RGB within 1/255; and the sample equals the frame's pixels."""
    import sys

    import numpy as np

    sys.path.insert(0, repo)
    sys.path.insert(0, os.path.join(repo, "tests"))
    import bench_extra
    import oracle_binding as ob

    rrt = ob.rrt
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
    gpu, cam, ntri = bench_extra.build_mesh_scene(rrt, variant=variant)
    gpu.pre_render(1)
    rgb, depth, ids = gpu.render(cam)
    st = gpu.stats()
    assert ntri == 1068080 and st["rays_secondary"] > 2_000_000           # secondary rays really were traced through the BVH
    assert st["bvh_tris_tested"] > st["rays_primary"]
    cpu = ob.OracleScene()
    bench_extra.build_mesh_scene(rrt, variant=variant, scene=cpu)
    cpu.pre_render(1, threads=1)
    rng = np.random.default_rng(11)
    # 1024 rays: pixels on teapots (object 0) are the ones that spawn secondary rays, so most of the sample lies there, a third of
    # it on silhouettes (pixels whose neighbour shows another object), the rest anywhere
    edge = np.zeros(ids.shape, bool)
    edge[:, 1:] |= ids[:, 1:] != ids[:, :-1]
    edge[1:, :] |= ids[1:, :] != ids[:-1, :]
    on_edge, on_mesh = np.argwhere(edge), np.argwhere(ids == 0)
    px = np.concatenate([on_edge[rng.choice(len(on_edge), 340, replace=False)][:, ::-1], on_mesh[rng.choice(len(on_mesh), 500, replace=False)][:, ::-1],
                         np.stack([rng.integers(0, 3840, 184), rng.integers(0, 2160, 184)], axis=1)])
    rays = [rrt.camera_ray(cam, int(x), int(y)) for x, y in px]
    o = np.ascontiguousarray([r[0] for r in rays], np.float32)
    d = np.ascontiguousarray([r[1] for r in rays], np.float32)
    res = threaded(lambda idx: dict(zip(("rgb", "depth", "id"), cpu.raytrace(o[idx], d[idx]))), len(o))
    rgb_c, depth_c, id_c = res["rgb"], res["depth"], res["id"]
    rgb_g, depth_g, id_g = gpu.raytrace(o, d)
    assert np.array_equal(id_g, id_c) and np.array_equal(bits(depth_g), bits(depth_c))
    assert np.abs(rrt.quantise(rgb_g).astype(int) - ob.quantise(rgb_c).astype(int)).max() <= 1
    assert (bits(rgb_g) == bits(rgb_c)).mean() > 0.9                      # and nearly every f32 value is the same bits
    assert np.array_equal(id_g, ids[px[:, 1], px[:, 0]]) and np.array_equal(bits(rgb_g), bits(rgb[px[:, 1], px[:, 0]]))
