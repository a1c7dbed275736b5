"""GPU parity tests: librrt (HIP, through the C ABI) against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): object ids and depth bit-exact; hit position/distance bit-exact;
RGB within 1/255 in raytracing mode; photon maps compared record by record under a fixed seed and
photon-mapped frames within a stated MAE.  The libm calls on the path (atan2 / acos of the sphere
UVs, pow of the specular term) are evaluated in f64 and rounded to f32 once on both sides, so no
scene needs a libm tolerance.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle_binding as ob
import scene_kit as kit

rrt = ob.rrt
abi = rrt._abi
pytestmark = pytest.mark.gpu

REFERENCE_SCENES = ["quadratics", "reflection_refraction", "textures", "scene", "pm_diffuse", "pm_caustics", "scene2", "c2_quadric_csg_fresnel"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def load_pair(repo, name, mode=abi.MODE_SCENE, params=None):
    path = os.path.join(repo, "assets/scenes", name + ".txt")
    cpu, cam = ob.OracleScene.load(path, params=params, mode_override=mode)
    gpu, _ = rrt.Scene.load(path, params=params, mode_override=mode)
    return cpu, gpu, cam


@pytest.mark.parametrize("name", REFERENCE_SCENES)
def test_frames_match_oracle_in_raytracing_mode(repo, name):
    cpu, gpu, cam = load_pair(repo, name)
    cam.width, cam.height = 320, 200  # non-square, not a multiple of the block size
    gpu.pre_render(1)
    rgb_g, depth_g, id_g = gpu.render(cam)
    rgb_c, depth_c, id_c = cpu.render(cam)
    assert np.array_equal(id_g, id_c)                      # bit-exact primary-hit object ids
    assert np.array_equal(bits(depth_g), bits(depth_c))    # bit-exact depth buffer
    dq = np.abs(rrt.quantise(rgb_g).astype(int) - ob.quantise(rgb_c).astype(int))
    assert dq.max() <= 1                                   # RGB within 1/255 (north_star), every scene
    # identical work: the same number of secondary and shadow rays were traced
    sg, sc = gpu.stats(), cpu.stats()
    for k in ("rays_primary", "rays_secondary", "rays_shadow"):
        assert sg[k] == sc[k], k


@pytest.mark.parametrize("size", [(37, 21), (8, 8), (5, 3), (67, 9)])
def test_ragged_frame_sizes_bands_and_chunks(repo, size):
    """The band is walked in 8x8 tiles with ragged right / bottom edges and split into chunks: every pixel must
    still land where the reference's row-major loop (camera.rs:14-42) puts it."""
    cpu, gpu, cam = load_pair(repo, "c2_quadric_csg_fresnel")
    cam.width, cam.height = size
    gpu.pre_render(1)
    rgb_c, depth_c, id_c = cpu.render(cam)
    rgb_g, depth_g, id_g = gpu.render(cam)
    assert np.array_equal(id_g, id_c) and np.array_equal(bits(depth_g), bits(depth_c))
    assert np.abs(rrt.quantise(rgb_g).astype(int) - ob.quantise(rgb_c).astype(int)).max() <= 1
    # a band of rows (what one rank of a multi-GPU render does) and a chunk size that is no multiple of a tile
    y0, y1 = size[1] // 3, size[1]
    p = rrt.default_params()
    p.chunk_pixels = 100
    gpu.set_params(p)
    gpu.pre_render(1)
    rgb_b, depth_b, id_b = gpu.render(cam, y0, y1)
    assert np.array_equal(id_b, id_c[y0:y1]) and np.array_equal(bits(depth_b), bits(depth_c[y0:y1]))
    assert np.array_equal(bits(rgb_b), bits(rgb_g[y0:y1]))


def test_predicted_level_sizes_and_their_fallback(repo):
    """Frames after the first are launched for the previous frame's level sizes without per-level readbacks; a frame
    that outgrows the prediction is repeated level by level.  Both must give the first frame's bits."""
    cpu, gpu, cam = load_pair(repo, "c2_quadric_csg_fresnel")
    cam.width, cam.height = 256, 160
    gpu.pre_render(1)
    first = gpu.render(cam)            # sizes found level by level
    s1 = gpu.stats()
    second = gpu.render(cam)           # predicted sizes
    s2 = gpu.stats()
    for a, b in zip(first, second):
        assert np.array_equal(bits(a), bits(b))
    for k in ("rays_primary", "rays_secondary", "rays_shadow"):
        assert s1[k] == s2[k], k
    # a camera that sees only sky leaves a prediction of zero secondary rays ...
    away = rrt.make_camera(256, 160, 0.5, (0.0, 50.0, 0.0), (0.0, -1.0, 0.0), (0.0, 0.0, 1.0))  # file-space -y looks up
    _, _, ids = gpu.render(away)
    ids2 = gpu.render(away)[2]
    assert (ids == abi.MISS).all() and (ids2 == abi.MISS).all() and gpu.stats()["rays_secondary"] == 0
    # ... which the real view overflows by far: repeated transparently
    third = gpu.render(cam)
    s3 = gpu.stats()
    for a, b in zip(first, third):
        assert np.array_equal(bits(a), bits(b))
    assert s3["rays_secondary"] == s1["rays_secondary"] and s3["rays_shadow"] == s1["rays_shadow"]
    rgb_c, depth_c, id_c = cpu.render(cam)
    assert np.array_equal(first[2], id_c) and np.array_equal(bits(first[1]), bits(depth_c))


def test_frames_of_several_chunk_rounds_keep_their_predictions(repo):
    """A frame larger than the chunk size goes through several rounds of chunks, each over another part of the image with
    its own level sizes: every round keeps its own prediction, so steady-state frames are not repeated."""
    _, gpu, cam = load_pair(repo, "c2_quadric_csg_fresnel", params=rrt.default_params(chunk_pixels=8192))
    cam.width, cam.height = 256, 160    # 5 rounds of 8192 pixels
    gpu.pre_render(1)
    first = gpu.render(cam)
    for _ in range(3):
        again = gpu.render(cam)
        for a, b in zip(first, again):
            assert np.array_equal(bits(a), bits(b))
    assert gpu.stats()["frames_repeated"] == 0
    whole, _ = rrt.Scene.load(os.path.join(repo, "assets/scenes/c2_quadric_csg_fresnel.txt"), mode_override=abi.MODE_SCENE)
    whole.pre_render(1)
    for a, b in zip(first, whole.render(cam)):
        assert np.array_equal(bits(a), bits(b))


def test_chunks_on_several_streams_match_single_stream_and_oracle(repo):
    """With rrt_params.render_streams > 1, frames of 2^18 pixels and more are cut into one chunk per stream so that deep
    recursion levels of one chunk may overlap shallow levels of another: same bits as on one stream, same as the oracle."""
    cpu, gpu, cam = load_pair(repo, "c2_quadric_csg_fresnel", params=rrt.default_params(render_streams=4))
    cam.width, cam.height = 648, 420  # 272160 pixels: 4 chunks of 68096 (whole 256-blocks) with a ragged last one
    gpu.pre_render(1)
    a1 = gpu.render(cam)  # first frame: level sizes by readback, chunk after chunk
    a2 = gpu.render(cam)  # second frame: predicted sizes, chunks in flight together
    s2 = gpu.stats()
    one, _ = rrt.Scene.load(os.path.join(repo, "assets/scenes/c2_quadric_csg_fresnel.txt"), params=rrt.default_params(render_streams=1),
                            mode_override=abi.MODE_SCENE)
    one.pre_render(1)
    b1 = one.render(cam)
    b2 = one.render(cam)
    s1 = one.stats()
    for x in (a2, b1, b2):
        for u, v in zip(a1, x):
            assert np.array_equal(bits(u), bits(v))
    for k in ("rays_primary", "rays_secondary", "rays_shadow"):
        assert s1[k] == s2[k], k
    rgb_c, depth_c, id_c = cpu.render(cam)
    assert np.array_equal(a1[2], id_c) and np.array_equal(bits(a1[1]), bits(depth_c))
    assert np.abs(rrt.quantise(a1[0]).astype(int) - ob.quantise(rgb_c).astype(int)).max() <= 1
    assert cpu.stats()["rays_secondary"] == s2["rays_secondary"] and cpu.stats()["rays_shadow"] == s2["rays_shadow"]


@pytest.mark.parametrize("name", REFERENCE_SCENES)
def test_trace_and_shadowtrace_match_oracle_on_random_rays(repo, name):
    cpu, gpu, cam = load_pair(repo, name)
    gpu.pre_render(1)
    towards = (0.0, 2.0, 5.0) if name not in ("quadratics", "reflection_refraction", "c2_quadric_csg_fresnel") else (0.0, 0.0, 1.0)
    o, d = kit.random_rays(20000, 42, origin_scale=6.0, towards=towards, spread=5.0)
    g, c = gpu.trace(o, d), cpu.trace(o, d)
    assert (c["object_id"] != abi.MISS).mean() > 0.15
    # uv of sphere hits: atan2 / acos in f64 rounded once to f32 on both sides -> bit-exact too, and with them the
    # normal-map texels and the perturbed normals
    for k in ("object_id", "prim_id", "distance", "position", "uv", "normal"):
        assert np.array_equal(bits(g[k]), bits(c[k])), k
    limits = np.random.default_rng(1).random(o.shape[0]).astype(np.float32) * 12.0
    assert np.array_equal(gpu.shadowtrace(o, d, limits), cpu.shadowtrace(o, d, limits))


def test_primitives_known_answers_on_gpu():
    """The known answers of SURVEY.md §8c straight through the C ABI (no oracle involved)."""
    s = rrt.Scene()
    m = kit.monochrome(s)
    s.add_object(kit.sphere(s, (0, 0, 0), 1.0, m))
    s.add_object(kit.cuboid(s, (10, 0, 0), (1, 1, 1), m))
    s.add_object(kit.quadric(s, (1, 0, 0, 0, 1, 0, 0, 1, 0, -1), m))  # unit sphere as a quadric, hidden behind object 0 on ties
    s.add_object(kit.plane(s, (0, -50, 0), (0, 0, 1), (0, 1, 0), m))
    s.pre_render(1)
    r = s.trace([(0, 0, -5), (10.5, 0.5, -2), (0, 3, 0), (30, 3, 0)], [(0, 0, 1), (0, 0, 1), (0, 1, 0), (0, -1, 0)])
    assert list(r["object_id"]) == [0, 1, abi.MISS, 3]       # strict `<`: the first object wins the sphere/quadric tie
    assert list(r["distance"]) == [4.0, 2.0, 0.0, 53.0]
    assert np.array_equal(r["normal"][0], [0, 0, -1]) and np.array_equal(r["normal"][1], [0, 0, -1]) and np.array_equal(r["normal"][3], [0, 1, 0])
    assert r["uv"][0][0] == pytest.approx(1.0, abs=1e-6) and r["uv"][0][1] == pytest.approx(0.5, abs=1e-6)


@pytest.mark.parametrize("mode", [abi.CSG_UNION, abi.CSG_INTERSECTION, abi.CSG_DIFFERENCE])
def test_csg_interval_lists_match_oracle(mode):
    def build(s):
        m1, m2, m3 = kit.monochrome(s, (1, 0, 0)), kit.monochrome(s, (0, 1, 0)), kit.monochrome(s, (0, 0, 1))
        a = kit.csg(s, mode, kit.sphere(s, (0, 0, 0), 1.0, m1), kit.cuboid(s, (-0.3, -0.3, -2), (0.9, 0.9, 2.5), m2))
        b = kit.csg(s, abi.CSG_DIFFERENCE, a, kit.quadric(s, (4, 0, 0, 0, 0, 0, 0, 4, 0, -1), m3))
        s.add_object(b)
        s.add_object(kit.csg(s, mode, kit.plane(s, (0, -1.5, 0), (0, 0, 1), (0, 1, 0), m2), kit.sphere(s, (3, -1.5, 0), 1.0, m1)))
        kit.light(s, abi.LIGHT_POINT, position=(2, 4, -3), colour=(1, 1, 1))

    cpu, gpu = ob.OracleScene(), rrt.Scene()
    build(cpu)
    build(gpu)
    gpu.pre_render(1)
    o, d = kit.random_rays(30000, 7, origin_scale=4.0, towards=(0.5, -0.3, 0.0), spread=2.5)
    g, c = gpu.trace(o, d), cpu.trace(o, d)
    for k in ("object_id", "prim_id", "distance", "position", "normal"):
        assert np.array_equal(bits(g[k]), bits(c[k])), k
    rg, rc = gpu.raytrace(o[:4000], d[:4000]), cpu.raytrace(o[:4000], d[:4000])
    assert np.array_equal(rg[2], rc[2]) and np.array_equal(bits(rg[1]), bits(rc[1]))
    assert np.abs(rrt.quantise(rg[0]).astype(int) - ob.quantise(rc[0]).astype(int)).max() <= 1


def test_hitvec_overflow_is_reported_from_the_device():
    s = rrt.Scene()
    m = kit.monochrome(s)
    h = kit.sphere(s, (0, 0, 0), 1.0, m)
    for k in range(1, 4):
        h = kit.csg(s, abi.CSG_UNION, h, kit.sphere(s, (0, 0, 3.0 * k), 1.0, m))
    s.add_object(h)
    s.pre_render(1)
    with pytest.raises(rrt.RrtError) as e:
        s.trace([(0, 0, -5)], [(0, 0, 1)])
    assert e.value.code == abi.ERR_OVERFLOW  # core/hit.rs:70-73 panics here
    assert s.trace([(5, 5, -5)], [(0, 0, 1)])["object_id"][0] == abi.MISS


# ---- meshes: BVH traversal vs the reference's brute-force loop ---------------------------------------------------
def adversarial_rays(pos, n, seed):
    """Random rays plus the cases a BVH gets wrong first: axis-parallel rays through vertices, origins on box
    planes, grazing rays along edges, far-away origins (cancellation in o + d*t)."""
    rng = np.random.default_rng(seed)
    v = pos.reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    c, ext = (lo + hi) / 2, (hi - lo).max()
    o, d = kit.random_rays(n, seed, origin_scale=ext * 1.5, towards=c, spread=ext * 0.6)
    o += c.astype(np.float32)
    k = n // 8
    pick = v[rng.integers(0, len(v), k)]
    axis = rng.integers(0, 3, k)
    o[:k] = pick
    o[np.arange(k), axis] = (lo - ext)[axis]
    d[:k] = np.eye(3, dtype=np.float32)[axis]            # axis-parallel through a vertex
    e0 = v[rng.integers(0, len(v) // 3, k) * 3]
    e1 = v[rng.integers(0, len(v) // 3, k) * 3 + 1]
    dd = (e1 - e0) + (rng.random((k, 3)) - 0.5) * 1e-3 * ext
    nn = np.linalg.norm(dd, axis=1, keepdims=True)
    ok = nn[:, 0] > 0
    d[k:2 * k][ok] = (dd / np.where(nn > 0, nn, 1))[ok]
    o[k:2 * k] = e0 - d[k:2 * k] * ext                      # grazing along an edge
    far = slice(2 * k, 3 * k)
    o[far] = c + (o[far] - c) * 400.0                      # origins hundreds of extents away
    t = c + (rng.random((k, 3)) - 0.5) * ext * 0.8
    dv = t - o[far]
    d[far] = dv / np.linalg.norm(dv, axis=1, keepdims=True)
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)


def mesh_pair(pos, nrm, smooth, builder=abi.BVH_HOST_SAH):
    cpu, gpu = ob.OracleScene(), rrt.Scene(params=rrt.default_params(bvh_builder=builder))
    for s in (cpu, gpu):
        s.add_object(kit.mesh(s, pos, nrm, kit.falsecolour(s), smooth))
    gpu.pre_render(1)
    return cpu, gpu


def check_mesh(cpu, gpu, o, d):
    g, c = gpu.trace(o, d), cpu.trace(o, d)
    assert (c["object_id"] != abi.MISS).sum() > 50
    for k in ("object_id", "prim_id", "distance", "position", "normal"):
        assert np.array_equal(bits(g[k]), bits(c[k])), k
    lim = np.abs(o).max(axis=1).astype(np.float32)
    assert np.array_equal(gpu.shadowtrace(o, d, lim), cpu.shadowtrace(o, d, lim))


def load_teapot(repo, name="teapot.obj"):
    import test_host_side

    return test_host_side.python_obj_loader(os.path.join(repo, "assets/models", name))


BUILDERS = [abi.BVH_HOST_SAH, abi.BVH_DEVICE_LBVH, abi.BVH_DEVICE_SAH]


@pytest.mark.parametrize("builder", BUILDERS)
def test_bvh_equals_brute_force_teapot(repo, builder):
    pos, nrm = load_teapot(repo)
    cpu, gpu = mesh_pair(pos, nrm, True, builder)
    o, d = adversarial_rays(pos, 24000, 3)
    check_mesh(cpu, gpu, o, d)
    assert gpu.stats()["bvh_tris_tested"] < 0.05 * o.shape[0] * pos.shape[0]  # the BVH actually prunes


@pytest.mark.parametrize("builder", BUILDERS)
def test_bvh_equals_brute_force_triangle_soup(builder):
    rng = np.random.default_rng(5)
    n = 30000
    centre = (rng.random((n, 1, 3)) - 0.5) * 20
    pos = (centre + (rng.random((n, 3, 3)) - 0.5) * rng.choice([0.05, 0.5, 3.0], (n, 1, 1))).astype(np.float32)
    pos[:50] = pos[50:100]            # exact duplicates: ties must go to the lowest index
    pos[100:110, 2] = pos[100:110, 1]  # degenerate (zero-area) triangles never hit
    cpu, gpu = mesh_pair(pos, None, False, builder)
    o, d = adversarial_rays(pos, 6000, 8)
    check_mesh(cpu, gpu, o, d)


def test_device_lbvh_degenerate_meshes_and_errors():
    """The device builder on inputs that stress its tie-breaks: all triangles identical (every Morton key equal), a flat
    grid (one axis without extent), sizes around the leaf size; and its error behaviour."""
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    grid = np.array([[[x, y, 0], [x + 1, y, 0], [x, y + 1, 0]] for y in range(20) for x in range(20)], np.float32)
    cases = [np.repeat(tri[None], 37, 0), grid] + [grid[:n] for n in (5, 6, 9, 17)]
    for pos, builder in [(p, b) for p in cases for b in (abi.BVH_DEVICE_LBVH, abi.BVH_DEVICE_SAH)]:
        cpu, gpu = mesh_pair(pos, None, False, builder)
        rng = np.random.default_rng(len(pos))
        hi = pos.reshape(-1, 3)[:, :2].max(0)
        # from above: the triangles face +z, and only entering hits are visible (environment.rs:39-44)
        o = np.concatenate([rng.uniform(-0.3, 1.0, (600, 2)) * (hi + 0.3), np.full((600, 1), 3.0)], axis=1).astype(np.float32)
        d = np.tile(np.array([0, 0, -1], np.float32), (600, 1))
        d[300:] += rng.normal(0, 0.2, (300, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        g, c = gpu.trace(o, d), cpu.trace(o, d)
        assert (c["object_id"] != abi.MISS).sum() > 5
        for k in ("object_id", "prim_id", "distance"):
            assert np.array_equal(bits(g[k]), bits(c[k])), (len(pos), k)
        assert gpu.stats()["ms_bvh_build"] > 0
    with pytest.raises(rrt.RrtError) as e:
        mesh_pair(grid, None, False, 7)
    assert e.value.code == abi.ERR_INVALID


def _flattened_arrays(make_scene, builder, path):
    """The flattened scene as RRT_DUMP_SCENE writes it (debugging aid of the flattener: element size, count, bytes per array)."""
    import struct

    os.environ["RRT_DUMP_SCENE"] = path
    try:
        s = make_scene(rrt.Scene(params=rrt.default_params(bvh_builder=builder)))
        s.pre_render(1)
    finally:
        del os.environ["RRT_DUMP_SCENE"]
    b, off, out = open(path, "rb").read(), 0, []
    while off < len(b):
        e, n = struct.unpack_from("<QQ", b, off)
        out.append(b[off + 16:off + 16 + e * n])
        off += 16 + e * n
    return out, s.stats()["ms_bvh_build"]


def test_device_sah_builds_the_host_builders_tree(repo, tmp_path):
    """RRT_BVH_DEVICE_SAH runs bvh_build.cpp's algorithm level by level on the GPU with the same f32 operations: the
    flattened scene (triangle order, quantised four-wide nodes, grid, stack need) is the host builder's, byte for byte —
    on the teapot and on a soup with duplicates and degenerate triangles.  (Where every centroid of a node coincides both
    builders split by position, and positions depend on the host's unstable std::partition: such meshes get an equally valid
    tree with another triangle order; test_device_lbvh_degenerate_meshes_and_errors traces them.)  A mesh with a non-finite
    coordinate goes to the host builder (its 'wild mesh' rules)."""
    pos_t, nrm_t = load_teapot(repo)
    rng = np.random.default_rng(11)
    n = 40000
    centre = (rng.random((n, 1, 3)) - 0.5) * 20
    soup = (centre + (rng.random((n, 3, 3)) - 0.5) * rng.choice([0.05, 0.5, 3.0], (n, 1, 1))).astype(np.float32)
    soup[:50] = soup[50:100]
    soup[100:110, 2] = soup[100:110, 1]
    wild = soup[:300].copy()
    wild[7, 1, 2] = np.inf
    for name, pos, nrm in (("teapot", pos_t, nrm_t), ("soup", soup, None), ("wild", wild, None)):
        make = lambda s: (s.add_object(kit.mesh(s, pos, nrm, kit.falsecolour(s), nrm is not None)), s)[1]
        host, _ = _flattened_arrays(make, abi.BVH_HOST_SAH, str(tmp_path / "host.bin"))
        dev, ms = _flattened_arrays(make, abi.BVH_DEVICE_SAH, str(tmp_path / "dev.bin"))
        assert len(host) == len(dev) and ms > 0
        for k, (a, b) in enumerate(zip(host, dev)):
            assert a == b, (name, "array", k, len(a), len(b))


def test_device_lbvh_frames_equal_host_sah_frames(repo):
    """Whole frames (mesh + plane + light, shadows, FalseColour and Phong) are bit-identical whichever builder made the tree."""
    frames = []
    for builder in BUILDERS:
        gpu, cam = rrt.Scene.load(os.path.join(repo, "assets/scenes/scene2.txt"), params=rrt.default_params(bvh_builder=builder),
                                  mode_override=abi.MODE_SCENE)
        gpu.pre_render(1)
        cam.width, cam.height = 320, 240
        frames.append(gpu.render(cam))
    for other in frames[1:]:
        for a, b in zip(frames[0], other):
            assert np.array_equal(bits(a), bits(b))


def test_bvh_tiny_meshes():
    for n in (1, 2, 4, 5):
        pos = np.array([[[0, 0, k], [1, 0, k], [0, 1, k]] for k in range(n)], np.float32)
        cpu, gpu = mesh_pair(pos, None, False)
        o = np.array([(0.2, 0.2, -3), (0.2, 0.2, n + 3), (5, 5, -3), (0.2, 0.2, 0.5)], np.float32)
        d = np.array([(0, 0, 1), (0, 0, -1), (0, 0, 1), (0, 0, 1)], np.float32)
        g, c = gpu.trace(o, d), cpu.trace(o, d)
        for k in ("object_id", "prim_id", "distance"):
            assert np.array_equal(bits(g[k]), bits(c[k])), (n, k)


def test_scene2_transformed_teapot_and_normals(repo):
    # Model rules of scene_file.rs:361-402: rotate_teapot, then scale + translate in one matrix, smooth normals
    cpu, gpu, cam = load_pair(repo, "scene2")
    gpu.pre_render(1)
    rng = np.random.default_rng(12)
    centre = np.array([1.5, 4.8, 5.1], np.float32)          # the teapot after scale 0.04 + translate
    off = rng.normal(size=(20000, 3)).astype(np.float32)
    o = centre + 2.0 * off / np.linalg.norm(off, axis=1, keepdims=True)
    t = centre + (rng.random((20000, 3)).astype(np.float32) - 0.5) * np.array([1.4, 0.7, 1.0], np.float32)
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o, d = np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)
    g, c = gpu.trace(o, d), cpu.trace(o, d)
    assert (c["object_id"] == 8).sum() > 2000
    for k in ("object_id", "prim_id", "distance", "position", "normal"):
        assert np.array_equal(bits(g[k]), bits(c[k])), k


# ---- full-size, size-independent properties ------------------------------------------------------------------------
def test_full_size_frame_properties(repo):
    """BASELINE configs[1] at its full 1920x1080: the oracle would take a while, so check what does not need it:
    band independence (the multi-GPU shard unit), chunk independence, run-to-run determinism, the batched
    raytrace entry, and the oracle on a sample of rows."""
    path = os.path.join(repo, "assets/scenes/c2_quadric_csg_fresnel.txt")
    gpu, cam = rrt.Scene.load(path)
    gpu.pre_render(1)
    rgb, depth, ids = gpu.render(cam)
    assert rgb.shape == (1080, 1920, 3)
    st_full = gpu.stats()
    again = gpu.render(cam)
    assert all(np.array_equal(bits(a), bits(b)) for a, b in zip((rgb, depth, ids), again))
    parts = [gpu.render(cam, y0, y1) for (y0, y1) in ((0, 135), (135, 700), (700, 1080))]
    assert np.array_equal(bits(np.concatenate([p[0] for p in parts])), bits(rgb))
    assert np.array_equal(bits(np.concatenate([p[1] for p in parts])), bits(depth))
    small = rrt.Scene.load(path, params=rrt.default_params(chunk_pixels=100_003))[0]
    small.pre_render(1)
    chunked = small.render(cam)
    assert all(np.array_equal(bits(a), bits(b)) for a, b in zip((rgb, depth, ids), chunked))
    assert small.stats()["rays_secondary"] == st_full["rays_secondary"] and small.stats()["rays_shadow"] == st_full["rays_shadow"]
    rows = [7, 333, 540, 901, 1079]
    cpu, _ = ob.OracleScene.load(path)
    for y in rows:
        rc, dc, ic = cpu.render(cam, y, y + 1)
        assert np.array_equal(ic[0], ids[y]) and np.array_equal(bits(dc[0]), bits(depth[y]))
        assert np.abs(ob.quantise(rc[0]).astype(int) - rrt.quantise(rgb[y]).astype(int)).max() <= 1
    o = np.zeros((1920, 3), np.float32)
    d = np.zeros((1920, 3), np.float32)
    for x in range(1920):
        o[x], d[x] = rrt.camera_ray(cam, x, 540)
    r_rgb, r_depth, r_ids = gpu.raytrace(o, d)
    assert np.array_equal(r_ids, ids[540]) and np.array_equal(bits(r_depth), bits(depth[540])) and np.array_equal(bits(r_rgb), bits(rgb[540]))


def test_scene2_raytracing_mode_at_its_full_size(repo):
    """BASELINE configs[0]: assets/scenes/scene2.txt in raytracing mode at its own 1024 x 1024 (the committed file selects
    PhotonScene: the mode is overridden).  The whole frame through the oracle: object ids and depth buffer
    the same bits, RGB within 1/255, the same numbers of rays; and the depth buffer quantises to the reference's committed
    render/depth.ppm (every byte 255: a weak check, but the only stored output of the depth path)."""
    cpu, gpu, cam = load_pair(repo, "scene2")
    assert (cam.width, cam.height) == (1024, 1024)
    gpu.pre_render(1)
    rgb_g, depth_g, id_g = gpu.render(cam)
    rgb_c, depth_c, id_c = cpu.render(cam)
    assert np.array_equal(id_g, id_c) and np.array_equal(bits(depth_g), bits(depth_c))
    assert np.abs(rrt.quantise(rgb_g).astype(int) - ob.quantise(rgb_c).astype(int)).max() <= 1
    sg, sc = gpu.stats(), cpu.stats()
    for k in ("rays_primary", "rays_secondary", "rays_shadow"):
        assert sg[k] == sc[k], k
    golden = open(os.path.join(repo, "tests/golden/render/depth.ppm"), "rb").read()
    assert rrt.quantise(depth_g).tobytes() == golden[-1024 * 1024:]


def test_render_rejects_bad_input(repo):
    gpu, cam = rrt.Scene.load(os.path.join(repo, "assets/scenes/quadratics.txt"))
    with pytest.raises(rrt.RrtError) as e:
        gpu.render(cam)
    assert e.value.code == abi.ERR_STATE  # render before pre_render
    gpu.pre_render(1)
    with pytest.raises(rrt.RrtError):
        gpu.render(cam, 10, 5000)
    rgb, depth, ids = gpu.render(cam, 5, 5)  # empty band
    assert rgb.shape == (0, 1024, 3)
    r = gpu.trace(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert r["object_id"].shape == (0,)
    empty = rrt.Scene()
    empty.pre_render(1)
    cam.width = cam.height = 8
    rgb, depth, ids = empty.render(cam)  # a scene with no objects: scene.rs:85-90
    assert (ids == abi.MISS).all() and not rgb.any() and not depth.any()


def test_concurrent_render_calls_on_one_handle(repo):
    """Environment is Sync in the reference (raytrace(&self) runs on every render thread, camera.rs:22-40): several threads
    may render on ONE scene handle at once.  Each call gets its own queues (a worker that shares the scene by pointer);
    frames and per-thread statistics equal the sequential ones bit for bit, for row bands and for whole frames."""
    import threading

    gpu, cam = rrt.Scene.load(os.path.join(repo, "assets/scenes/c2_quadric_csg_fresnel.txt"))
    gpu.pre_render(1)
    cam.width, cam.height = 256, 144
    cams = []
    for k in range(6):
        c = type(cam)()
        C.memmove(C.byref(c), C.byref(cam), C.sizeof(cam))
        c.position[0] = cam.position[0] + 0.05 * k
        cams.append(c)
    want, want_rays = [], []
    for c in cams:
        want.append(gpu.render(c))
        st = gpu.stats()
        want_rays.append((st["rays_primary"], st["rays_secondary"], st["rays_shadow"]))
    assert len(set(want_rays)) > 1                      # the frames really differ
    got, got_rays, errors = [None] * len(cams), [None] * len(cams), []

    def work(tid, n_threads):
        try:
            for rep in range(3):
                for k in range(tid, len(cams), n_threads):
                    got[k] = gpu.render(cams[k])
                    st = gpu.stats()                    # this thread's own last frame
                    got_rays[k] = (st["rays_primary"], st["rays_secondary"], st["rays_shadow"])
        except Exception as e:
            errors.append(e)

    th = [threading.Thread(target=work, args=(t, 3)) for t in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for k in range(len(cams)):
        for a, b in zip(want[k], got[k]):
            assert np.array_equal(bits(a), bits(b)), k
        assert got_rays[k] == want_rays[k], k
    # bands of one frame from two threads at once (the reference's render_rows split)
    half = cam.height // 2
    parts = [None, None]

    def band(j):
        try:
            parts[j] = gpu.render(cams[0], j * half, (j + 1) * half)
        except Exception as e:
            errors.append(e)

    th = [threading.Thread(target=band, args=(j,)) for j in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for a, b0, b1 in zip(want[0], parts[0], parts[1]):
        assert np.array_equal(bits(a), bits(np.concatenate([b0, b1])))
    # a rebuild drops the workers; the handle keeps working
    gpu.pre_render(1)
    for a, b in zip(want[2], gpu.render(cams[2])):
        assert np.array_equal(bits(a), bits(b))


def test_set_params_reaches_every_worker_of_concurrent_renders(repo):
    """Workers leased by concurrent render calls copy mode and params when they are made.  rrt_scene_set_params /
    rrt_scene_set_mode must drop them: after max_depth changes, every band of a concurrently rendered frame must use the
    new depth, whichever worker renders it (a stale worker would still bounce five times)."""
    import threading

    gpu, cam = rrt.Scene.load(os.path.join(repo, "assets/scenes/c2_quadric_csg_fresnel.txt"))
    gpu.pre_render(1)
    cam.width, cam.height = 256, 144
    bands = [(0, 48), (48, 96), (96, 144)]

    def concurrent_frame():
        parts, errors = [None] * len(bands), []

        def work(j):
            try:
                parts[j] = gpu.render(cam, *bands[j])
            except Exception as e:
                errors.append(e)

        th = [threading.Thread(target=work, args=(j,)) for j in range(len(bands))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        return [np.concatenate([p[k] for p in parts]) for k in range(3)]

    deep = concurrent_frame()                       # creates the workers, max_depth 5
    p = rrt.default_params(max_depth=1)
    gpu.set_params(p)                               # no pre_render in between: scene mode renders with the new values
    shallow = concurrent_frame()
    fresh, _ = rrt.Scene.load(os.path.join(repo, "assets/scenes/c2_quadric_csg_fresnel.txt"), params=p)
    fresh.pre_render(1)
    want = fresh.render(cam)
    for a, b in zip(shallow, want):
        assert np.array_equal(bits(a), bits(b))
    assert not np.array_equal(bits(deep[0]), bits(shallow[0]))   # the depth really matters in this scene


def test_interleaved_row_groups_give_the_same_frame(repo):
    """rrt_render_interleaved_device (the load-balanced multi-GPU split): the shares of 3 'ranks' in groups of 8 rows, put
    back in image order, equal the whole frame bit for bit; ragged height (not a multiple of 8 or of 24)."""
    import torch

    gpu, cam = rrt.Scene.load(os.path.join(repo, "assets/scenes/scene2.txt"), mode_override=abi.MODE_SCENE)
    gpu.pre_render(1)
    cam.width, cam.height = 200, 117
    want = gpu.render(cam)
    got = [np.zeros_like(w) for w in want]
    seen = np.zeros(cam.height, int)
    for world, group in ((3, 8), (2, 16), (1, 8)):
        seen[:] = 0
        for r in range(world):
            rows = rrt.interleaved_rows(cam.height, group, r, world)
            n = len(rows)
            rgb = torch.full((max(n, 1), cam.width, 3), -7.0, dtype=torch.float32, device="cuda")
            depth = torch.full((max(n, 1), cam.width), -7.0, dtype=torch.float32, device="cuda")
            ids = torch.full((max(n, 1), cam.width), 123, dtype=torch.int32, device="cuda")
            gpu.render_interleaved_device(cam, group, r, world, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), None)
            torch.cuda.synchronize()
            got[0][rows], got[1][rows], got[2][rows] = rgb[:n].cpu().numpy(), depth[:n].cpu().numpy(), ids[:n].cpu().numpy().view(np.uint32)
            seen[rows] += 1
        assert (seen == 1).all()
        for a, b in zip(got, want):
            assert np.array_equal(bits(a), bits(b)), (world, group)
    with pytest.raises(rrt.RrtError):
        gpu.render_interleaved_device(cam, 12, 0, 2, 0, 0, 0, None)     # groups are whole 8-row tiles
    with pytest.raises(rrt.RrtError):
        gpu.render_interleaved_device(cam, 8, 2, 2, 0, 0, 0, None)      # first group must be below the stride
