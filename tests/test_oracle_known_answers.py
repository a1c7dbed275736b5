"""Pins the CPU oracle: hand-derived known answers, the published Philox vectors and
the reference's committed render/depth.ppm.  The reference has no tests of its own; these are the
only anchors, so the oracle's parity status stays "pinned only weakly" (oracle/oracle.h).
No GPU needed.
"""
import math
import os

import numpy as np
import pytest

import oracle_binding as ob
import scene_kit as kit

abi = ob.abi
INF = float("inf")


def single(defn):
    """Scene with one object built by `defn(scene, material)`."""
    s = ob.OracleScene()
    m = kit.monochrome(s)
    s.add_object(defn(s, m))
    return s


def test_sphere_known_answer():
    # sphere_object.rs:35-91: unit sphere at the origin, ray (0,0,-5) -> +z
    s = single(lambda s, m: kit.sphere(s, (0, 0, 0), 1.0, m))
    hits = s.intersect(0, (0, 0, -5), (0, 0, 1))
    assert [h["distance"] for h in hits] == [4.0, 6.0]
    assert [h["entering"] for h in hits] == [True, False]
    assert np.array_equal(hits[0]["normal"], [0, 0, -1])
    # u = (atan2(0,-1)+pi)/2pi = 1, v = (pi - acos(-0))/pi = 0.5
    assert hits[0]["uv"][0] == pytest.approx(1.0, abs=1e-6) and hits[0]["uv"][1] == pytest.approx(0.5, abs=1e-6)
    # the exit hit's normal is flipped to face the ray as well (sphere_object.rs:58-60)
    assert np.array_equal(hits[1]["normal"], [0, 0, -1])
    assert s.intersect(0, (0, 2, -5), (0, 0, 1)) == []


def test_sphere_divides_by_two_not_2a():
    # sphere_object.rs:52-53 assumes |d| = 1: with |d| = 2 the roots are those of t^2... /2.0
    s = single(lambda s, m: kit.sphere(s, (0, 0, 0), 1.0, m))
    hits = s.intersect(0, (0, 0, -5), (0, 0, 2))
    # a=4, b=-20, c=24, disc=400-384=16 -> (20 -+ 4)/2.0 = 8, 12 (the true roots /(2a) would be 2, 3)
    assert [h["distance"] for h in hits] == [8.0, 12.0]


def test_plane_known_answer():
    # plane_object.rs:54-113: y = 0, n = +y
    s = single(lambda s, m: kit.plane(s, (0, 0, 0), (0, 0, 1), (0, 1, 0), m))
    hits = s.intersect(0, (0, 3, 0), (0, -1, 0))
    assert [(h["distance"], h["entering"]) for h in hits] == [(3.0, True), (INF, False)]
    assert np.array_equal(hits[0]["normal"], [0, 1, 0])
    hits = s.intersect(0, (0, 3, 0), (0, 1, 0))
    assert [(h["distance"], h["entering"]) for h in hits] == [(-INF, True), (-3.0, False)]
    assert np.array_equal(hits[1]["normal"], [0, -1, 0])
    # parallel: inside the half space (U >= 0) -> (-inf, +inf); outside -> nothing
    hits = s.intersect(0, (0, 3, 0), (1, 0, 0))
    assert [(h["distance"], h["entering"]) for h in hits] == [(-INF, True), (INF, False)]
    assert s.intersect(0, (0, -3, 0), (1, 0, 0)) == []
    # infinite hits carry position = normal = 0 (core/hit.rs:37-52)
    assert np.array_equal(hits[0]["position"], [0, 0, 0]) and np.array_equal(hits[0]["normal"], [0, 0, 0])


def test_plane_uv_frame():
    # plane_object.rs:95-99: v along `up`, u along up x normal
    s = single(lambda s, m: kit.plane(s, (1, 0, 2), (0, 0, 2), (0, 1, 0), m))
    h = s.intersect(0, (4, 3, 7), (0, -1, 0))[0]
    # from_centre = (3,0,5); v_unit = (0,0,1); u_unit = (0,0,1)x(0,1,0) = (-1,0,0)
    assert h["uv"][0] == -3.0 and h["uv"][1] == 5.0


def test_cuboid_known_answer():
    # cuboid_object.rs:94-148
    s = single(lambda s, m: kit.cuboid(s, (0, 0, 0), (1, 1, 1), m))
    hits = s.intersect(0, (0.5, 0.5, -2), (0, 0, 1))
    assert [(h["distance"], h["entering"]) for h in hits] == [(2.0, True), (3.0, False)]
    assert np.array_equal(hits[0]["normal"], [0, 0, -1]) and np.array_equal(hits[1]["normal"], [0, 0, -1])
    # starting inside: only the exit survives (t < 0 entering hits are skipped)
    hits = s.intersect(0, (0.5, 0.5, 0.5), (0, 0, 1))
    assert [(h["distance"], h["entering"]) for h in hits] == [(0.5, False)]
    assert s.intersect(0, (2.5, 0.5, -2), (0, 0, 1)) == []


def test_cuboid_epsilon_slack():
    # +-1e-4 slack on the inside test (cuboid_object.rs:107-112)
    s = single(lambda s, m: kit.cuboid(s, (0, 0, 0), (1, 1, 1), m))
    assert len(s.intersect(0, (1.00005, 0.5, -2), (0, 0, 1))) == 2
    assert s.intersect(0, (1.0003, 0.5, -2), (0, 0, 1)) == []


def test_quadric_unit_sphere_and_translate():
    # quadratic_object.rs:43-124: a=e=h=1, j=-1 is the unit sphere, roots /(2 Aq)
    s = single(lambda s, m: kit.quadric(s, (1, 0, 0, 0, 1, 0, 0, 1, 0, -1), m))
    hits = s.intersect(0, (0, 0, -5), (0, 0, 1))
    assert [(h["distance"], h["entering"]) for h in hits] == [(4.0, True), (6.0, False)]
    assert np.array_equal(hits[0]["normal"], [0, 0, -1])

    # quadratics.txt:15-24: a=e=h=2, j=-1 translated by (2.3, 0, 0) -> d = -4.6, j = 9.58
    def defn(s, m):
        h = kit.quadric(s, (2, 0, 0, 0, 2, 0, 0, 2, 0, -1), m)
        s.object_transform(h, kit.translation((2.3, 0, 0)))
        return h

    s = single(defn)
    r = 1.0 / math.sqrt(2.0)
    hits = s.intersect(0, (2.3, 0, -5), (0, 0, 1))
    assert hits[0]["distance"] == pytest.approx(5 - r, abs=1e-5) and hits[1]["distance"] == pytest.approx(5 + r, abs=1e-5)
    hits = s.intersect(0, (-5, 0, 0), (1, 0, 0))
    assert hits[0]["distance"] == pytest.approx(7.3 - r, abs=1e-5)


def test_transform_inverse():
    # transform.rs:46-177
    t = kit.translation((2.3, -1.0, 4.0))
    inv = ob.mat4_inverse(t)
    assert np.array_equal(inv, kit.translation((-2.3, 1.0, -4.0)))
    rng = np.random.default_rng(3)
    m = rng.random((4, 4)).astype(np.float32) + np.eye(4, dtype=np.float32) * 2
    assert np.allclose(ob.mat4_inverse(m), np.linalg.inv(m.astype(np.float64)), rtol=1e-4, atol=1e-5)
    with pytest.raises(ob.OracleError):
        ob.mat4_inverse(np.zeros((4, 4), np.float32))


def test_transform_rules_per_type():
    # sphere_object.rs:93-100: radius follows uniform scaling only; cuboid_object.rs:150-160: diagonal only
    def sph(scale_m):
        def defn(s, m):
            h = kit.sphere(s, (1, 0, 0), 1.0, m)
            s.object_transform(h, scale_m)
            return h
        return single(defn)

    s = sph(kit.scaling(2.0))
    hits = s.intersect(0, (2, 0, -10), (0, 0, 1))  # centre (2,0,0), radius 2
    assert [h["distance"] for h in hits] == [8.0, 12.0]
    nonuni = np.diag([2.0, 3.0, 2.0, 1.0]).astype(np.float32)
    s = sph(nonuni)
    hits = s.intersect(0, (2, 0, -10), (0, 0, 1))  # radius unchanged
    assert [h["distance"] for h in hits] == [9.0, 11.0]

    def cub(s, m):
        h = kit.cuboid(s, (1, 1, 1), (1, 2, 3), m)
        t = kit.scaling(2.0)
        t[:3, 3] = (1, 0, 0)
        s.object_transform(h, t)
        return h

    s = single(cub)  # corner (3,2,2), size (2,4,6)
    hits = s.intersect(0, (4, 3, -2), (0, 0, 1))
    assert [h["distance"] for h in hits] == [4.0, 10.0]


def csg_reference(mode, L, R):
    """Independent restatement of csg_object.rs:39-108 on (distance, entering) lists."""
    tables = {
        abi.CSG_UNION: ["ADrop", "BDrop", "AExit", "BDrop", "ADrop", "BExit", "AEnter", "BEnter"],
        abi.CSG_INTERSECTION: ["AExit", "BExit", "ADrop", "BEnter", "AEnter", "BDrop", "ADrop", "BDrop"],
        abi.CSG_DIFFERENCE: ["ADrop", "BEnter", "AExit", "BExit", "ADrop", "BDrop", "AEnter", "BDrop"],
    }
    actions = tables[mode]
    L, R, out = list(L), list(R), []
    while L and R:
        state = 4 * L[0][1] + 2 * R[0][1] + (L[0][0] > R[0][0])
        a = actions[state]
        if a in ("AEnter", "AExit"):
            out.append((L.pop(0)[0], a == "AEnter"))
        elif a == "ADrop":
            L.pop(0)
        elif a in ("BEnter", "BExit"):
            out.append((R.pop(0)[0], a == "BEnter"))
        else:
            R.pop(0)
    if mode == abi.CSG_DIFFERENCE:
        out += L
    elif mode == abi.CSG_UNION:
        out += L + R
    return out


@pytest.mark.parametrize("mode", [abi.CSG_UNION, abi.CSG_INTERSECTION, abi.CSG_DIFFERENCE])
@pytest.mark.parametrize("offset", [1.0, 2.5, -1.0, 0.0, 0.3])
def test_csg_tables(mode, offset):
    s = ob.OracleScene()
    m = kit.monochrome(s)
    a = kit.sphere(s, (0, 0, 0), 1.0, m)
    b = kit.sphere(s, (0, 0, offset), 1.0, m)
    s.add_object(kit.csg(s, mode, a, b))
    for x in (0.0, 0.4, 0.9):
        o = (x, 0, -5)
        got = [(float(h["distance"]), h["entering"]) for h in s.intersect(0, o, (0, 0, 1))]
        sa, sb = ob.OracleScene(), ob.OracleScene()
        sa.add_object(kit.sphere(sa, (0, 0, 0), 1.0, kit.monochrome(sa)))
        sb.add_object(kit.sphere(sb, (0, 0, offset), 1.0, kit.monochrome(sb)))
        L = [(float(h["distance"]), h["entering"]) for h in sa.intersect(0, o, (0, 0, 1))]
        R = [(float(h["distance"]), h["entering"]) for h in sb.intersect(0, o, (0, 0, 1))]
        assert got == csg_reference(mode, L, R)


def test_csg_union_known_sequence():
    # two unit spheres at z=0 and z=1 along the axis: A = [4 in, 6 out], B = [5 in, 7 out]
    s = ob.OracleScene()
    m = kit.monochrome(s)
    s.add_object(kit.csg(s, abi.CSG_UNION, kit.sphere(s, (0, 0, 0), 1.0, m), kit.sphere(s, (0, 0, 1), 1.0, m)))
    got = [(float(h["distance"]), h["entering"]) for h in s.intersect(0, (0, 0, -5), (0, 0, 1))]
    assert got == [(4.0, True), (7.0, False)]


def test_hitvec_overflow_is_an_error():
    # core/hit.rs:70-73: more than 6 hits panics; here RRT_ERR_OVERFLOW
    s = ob.OracleScene()
    m = kit.monochrome(s)
    h = kit.sphere(s, (0, 0, 0), 1.0, m)
    for k in range(1, 4):
        h = kit.csg(s, abi.CSG_UNION, h, kit.sphere(s, (0, 0, 3.0 * k), 1.0, m))
    s.add_object(h)
    with pytest.raises(ob.OracleError) as e:
        s.intersect(0, (0, 0, -5), (0, 0, 1))
    assert e.value.code == abi.ERR_OVERFLOW


def test_triangle_same_side_and_facing():
    # triangle_object.rs:129-160 via a one-triangle mesh; normal = (ab x bc).normalised() = +z
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)
    s = ob.OracleScene()
    s.add_object(kit.mesh(s, tri, None, kit.monochrome(s), False))
    hits = s.intersect(0, (0.25, 0.25, 5), (0, 0, -1))  # against the normal: entering
    assert [(h["distance"], h["entering"]) for h in hits] == [(5.0, True)]
    assert np.array_equal(hits[0]["normal"], [0, 0, 1])
    hits = s.intersect(0, (0.25, 0.25, -5), (0, 0, 1))  # along the normal: the plane's exit hit
    assert [(h["distance"], h["entering"]) for h in hits] == [(5.0, False)]
    assert np.array_equal(hits[0]["normal"], [0, 0, -1])
    # outside the triangle: only an infinite plane hit could pass (position (0,0,0) is a corner here: it does not)
    assert [h for h in s.intersect(0, (0.9, 0.9, 5), (0, 0, -1)) if math.isfinite(h["distance"])] == []


def test_mesh_nearer_backface_hides_frontface():
    # polymesh_object.rs:216-242 + environment.rs:42-44
    back = [[0, 0, 1], [0, 1, 1], [1, 0, 1]]   # normal -z... (ab x bc) = (0,1,0)x(1,-1,0) = (0,0,-1): faces -z
    front = [[0, 0, 2], [1, 0, 2], [0, 1, 2]]  # normal +z
    s = ob.OracleScene()
    s.add_object(kit.mesh(s, np.array([front, back], np.float32), None, kit.monochrome(s), False))
    # ray from z=5 going -z meets `front` (z=2, entering) first
    r = s.trace([(0.2, 0.2, 5)], [(0, 0, -1)])
    assert r["object_id"][0] == 0 and r["distance"][0] == 3.0 and r["prim_id"][0] == 0
    # ray from z=-5 going +z meets `back` (z=1) first and it faces the ray -> entering -> visible
    r = s.trace([(0.2, 0.2, -5)], [(0, 0, 1)])
    assert r["distance"][0] == 6.0 and r["prim_id"][0] == 1
    # flip the ray: from z=1.5 going +z the first hit is `front` seen from behind (exit) -> whole mesh invisible
    r = s.trace([(0.2, 0.2, 1.5)], [(0, 0, 1)])
    assert r["object_id"][0] == abi.MISS


def test_smooth_normal_is_barycentric_blend():
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)
    nrm = np.array([[[1, 0, 0], [0, 1, 0], [0, 0, 1]]], np.float32)
    s = ob.OracleScene()
    s.add_object(kit.mesh(s, tri, nrm, kit.monochrome(s), True))
    h = s.intersect(0, (0.25, 0.25, 5), (0, 0, -1))[0]
    # weights: alpha=|bi x bc|, beta=|ci x ca|, gamma=|ai x ab| -> (0.5, 0.25, 0.25) normalised
    w = np.array([0.5, 0.25, 0.25])
    assert np.allclose(h["normal"], w / np.linalg.norm(w), atol=1e-6)


def test_refraction_normal_incidence():
    # global_material.rs:28-67 at normal incidence, ior 1.5: kr = 0.04, T = D.  A glass sphere in front of an
    # ambient-only wall, no lights: colour = black*kr + (C*ambient*refract_w)*(1-kr)
    s = ob.OracleScene()
    glass = kit.global_material(s, 1.0, 1.0, 1.5)
    wall = kit.monochrome(s, (0.5, 0.25, 1.0), ambient=0.5)
    s.add_object(kit.sphere(s, (0, 0, 0), 1.0, glass))
    s.add_object(kit.plane(s, (0, 0, 10), (0, 1, 0), (0, 0, -1), wall))
    rgb, depth, ids = s.raytrace([(0, 0, -5)], [(0, 0, 1)])
    assert ids[0] == 0 and depth[0] == 4.0
    assert np.allclose(rgb[0], np.array([0.25, 0.125, 0.5]) * 0.96, rtol=1e-5)


def test_depth_cap_five():
    # global_material.rs:72: two facing mirrors; recursion stops at depth 5 -> 6 hits traced
    s = ob.OracleScene()
    mirror = kit.global_material(s, 1.0, 0.0, 1.0)
    s.add_object(kit.plane(s, (0, 0, 5), (0, 1, 0), (0, 0, -1), mirror))
    s.add_object(kit.plane(s, (0, 0, -5), (0, 1, 0), (0, 0, 1), mirror))
    s.raytrace([(0, 0, 0)], [(0, 0, 1)])
    st = s.stats()
    assert st["rays_primary"] == 1 and st["rays_secondary"] == 5


def test_shadow_range_is_the_light_direction_length():
    # scene.rs:111: limit = |ldir| (= 1 for point lights): only occluders within one unit shadow
    def build(gap):
        s = ob.OracleScene()
        m = kit.monochrome(s, (1, 1, 1))
        s.add_object(kit.plane(s, (0, 0, 0), (0, 0, 1), (0, 1, 0), m))
        s.add_object(kit.sphere(s, (0, gap + 0.25, 0), 0.25, m))
        kit.light(s, abi.LIGHT_POINT, position=(0, 10, 0), colour=(1, 1, 1))
        return s

    near, far = build(0.5), build(2.0)
    o, d = [(3, 3, 0)], [(-1, -1, 0)]
    d = np.array(d, np.float32) / np.sqrt(2)
    lit_far = far.raytrace(o, d)[0][0]
    lit_near = near.raytrace(o, d)[0][0]
    assert lit_far.sum() > lit_near.sum() > 0  # the near sphere shadows (0,0,0), the far one does not


def test_camera_known_answer():
    # scene2 camera; after FullCamera::new up=(0,-1,0), right=(1,0,0)
    cam = ob.make_camera(1024, 1024, math.radians(40.0), (0, 3, 0), (0, 0, 1))
    assert list(cam.up) == [0.0, -1.0, 0.0] and list(cam.right) == [1.0, 0.0, 0.0] and list(cam.lookat)[2] == 1.0
    o, d = ob.camera_ray(cam, 511, 511)
    raw = np.array([-0.5 / 1024, 0.5 / 1024, np.float32(40.0) * np.float32(math.pi / 180)], np.float64)
    assert np.allclose(d, raw / np.linalg.norm(raw), atol=1e-7)
    assert list(o) == [0.0, 3.0, 0.0]
    with pytest.raises(Exception):
        ob.make_camera(16, 16, 0.7, (0, 0, 0), (0, 0, 1), up=(0, 0.5, 1))


def test_quantiser():
    # framebuffer.rs:105-107: `(c * 255.0) as u8`, saturating, NaN -> 0, truncation
    v = np.array([1.0, 0.999, -0.1, 2.0, np.nan, 0.5, 0.0], np.float32)
    assert list(ob.quantise(v)) == [255, 254, 0, 255, 0, 127, 0]


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert [hex(x) for x in ob.philox([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = 0xFFFFFFFF
    assert [hex(x) for x in ob.philox([f, f, f, f], [f, f])] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in ob.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])] == [
        "0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_texture_lookup_and_16bit():
    # texture.rs:105-119: nearest texel with the (-u).rem_euclid(1) flip; 16-bit samples big-endian (:83-87)
    s = ob.OracleScene()
    px = bytes([255, 0, 0, 0, 255, 0, 0, 0, 255, 255, 255, 255])  # 2x2: R G / B W
    t8 = s.add_texture(2, 2, 255, px)
    mat = kit.texture_material(s, t8, ambient=1.0)
    s.add_object(kit.plane(s, (0, 0, 0), (0, 0, 1), (0, 1, 0), mat))
    # u = -x (u_unit = (-1,0,0)), v = z; get_uv(u,v): x = floor(rem_euclid(-u,1)*(W-1)), y likewise
    rgb, _, _ = s.raytrace([(0.25, 1, 0.25)], [(0, -1, 0)])   # u=-0.25 -> x=floor(.25)=0 ; v=.25 -> -v rem 1 = .75 -> y=0
    assert list(rgb[0]) == [1.0, 0.0, 0.0]
    rgb, _, _ = s.raytrace([(-0.25, 1, 0.25)], [(0, -1, 0)])  # u=.25 -> rem(-.25)=.75 -> x=0
    assert list(rgb[0]) == [1.0, 0.0, 0.0]
    s2 = ob.OracleScene()
    t16 = s2.add_texture(1, 1, 65535, bytes([0x80, 0x00, 0xFF, 0xFF, 0x00, 0x01]))
    s2.add_object(kit.plane(s2, (0, 0, 0), (0, 0, 1), (0, 1, 0), kit.texture_material(s2, t16, ambient=1.0)))
    rgb, _, _ = s2.raytrace([(0.1, 1, 0.1)], [(0, -1, 0)])
    assert list(rgb[0]) == [np.float32(32768) / np.float32(65535), 1.0, np.float32(1) / np.float32(65535)]


def test_golden_depth_ppm_scene2(repo):
    """render/depth.ppm of the reference is all 255: every primary ray of scene2 hits at distance >= 1.

    The oracle must reproduce it (every 8th row here to keep the CPU suite short); depth is the same in
    raytracing and photon mode, both come from trace()."""
    golden = open(os.path.join(repo, "tests/golden/render/depth.ppm"), "rb").read()
    assert golden.startswith(b"P5\n1024 1024\n255\n")
    body = np.frombuffer(golden[len(b"P5\n1024 1024\n255\n"):], np.uint8).reshape(1024, 1024)
    assert (body == 255).all()
    s, cam = ob.OracleScene.load(os.path.join(repo, "assets/scenes/scene2.txt"), mode_override=abi.MODE_SCENE)
    assert (cam.width, cam.height) == (1024, 1024)
    for y in range(0, 1024, 8):
        _, depth, ids = s.render(cam, y, y + 1)
        assert (ids != abi.MISS).all()
        assert np.array_equal(ob.quantise(depth)[0], body[y])


def test_photon_pass_is_independent_of_the_split(repo):
    """Counter-based streams: shooting [0,n) at once or as two ranges gives the same record set."""
    params = ob.rrt.default_params(photons_per_light=3000, caustic_photons_per_light=300)
    path = os.path.join(repo, "assets/scenes/pm_caustics.txt")
    a, _ = ob.OracleScene.load(path, params=params, mode_override=abi.MODE_PHOTON)
    a.pre_render(5, threads=3)
    b, _ = ob.OracleScene.load(path, params=params, mode_override=abi.MODE_PHOTON)
    b.photon_begin(5)
    b.photon_shoot(0, 0, 0, 1000, threads=2)
    b.photon_shoot(0, 0, 1000, 2000, threads=1)
    b.photon_shoot(0, 1, 0, 300, threads=4)
    b.photon_build()
    for which in (0, 1):
        pa, pb = a.photon_read(which), b.photon_read(which)
        assert len(pa["key"]) > 0
        for k in pa:
            assert np.array_equal(pa[k].view(np.uint8), pb[k].view(np.uint8)), k
    # the kd-tree box query returns exactly the brute-force set
    rgb, depth, ids = a.raytrace([(0, 3, 0.5)], [(0, -0.5, 0.8)])
    pos = a.trace([(0, 3, 0.5)], [(0, -0.5, 0.8)])["position"][0]
    k, _, _ = a.photon_query(0, pos, 0.1)
    st = a.stats()
    kc, _, _ = a.photon_query(1, pos, 0.1)
    assert st["photons_gathered"] == k + kc


def test_gather_k_is_get_n_within_radius(repo):
    """rrt_params.gather_k in the oracle: the sorted list of get_within_distance truncated to k (photon_tree.rs:41-50).  A k
    no cube reaches changes nothing; k = 1 changes the frame but not the number of photons found in the cubes."""
    path = os.path.join(repo, "assets/scenes/pm_caustics.txt")
    frames = {}
    for k in (0, 4096, 1):
        params = ob.rrt.default_params(photons_per_light=20000, caustic_photons_per_light=1500, gather_k=k)
        s, cam = ob.OracleScene.load(path, params=params, mode_override=abi.MODE_PHOTON)
        s.pre_render(3)
        cam.width, cam.height = 64, 48
        frames[k] = (s, s.render(cam)[0])
    assert np.array_equal(frames[0][1].view(np.uint32), frames[4096][1].view(np.uint32))
    assert not np.array_equal(frames[0][1].view(np.uint32), frames[1][1].view(np.uint32))
    # every query of the k = 1 frame used at most one photon per map; the statistics still count the cube's photons
    assert frames[1][0].stats()["photons_gathered"] == frames[0][0].stats()["photons_gathered"]
