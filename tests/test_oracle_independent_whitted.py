"""The oracle's raytracing-mode RGB against a second restatement of the reference (tests/independent_whitted.py, numpy float32
scalars, written from the Rust sources and sharing no code with oracle/): the reference holds no stored frame of `Scene` mode
("raytracing-mode RGB is parity unpinned"), so this is what stands between the oracle and a misreading of
scene.rs:83-128, phong_material.rs:23-34, global_material.rs:28-108 and compound_material.rs:79-102 — the same treatment the CSG
tables and the scene-file parser got."""
import numpy as np

import independent_whitted as iw
import oracle_binding as ob
import scene_kit as kit

abi = ob.abi


def build_both():
    """A floor, a back wall, a matt sphere, a mirror sphere (`Simple`), a glass sphere (`Transparent`, ior 1.5) and a mirror that
    is also refractive with ior 1 (the shape `TransparentTexture` takes); a point light and a directional light whose direction is
    not of unit length (directional_light.rs:20-22 hands it out as it is)."""
    o = ob.OracleScene()
    s = iw.Scene()
    f = np.float32

    def mono(colour, ambient, shininess):
        return kit.monochrome(o, colour, ambient, shininess), iw.Monochrome(iw.vec(*colour), ambient, shininess)

    def comp(colour, shininess, reflect, refract, ior):
        om, im = mono(colour, 0.1, shininess)
        return kit.compound(o, [om, kit.global_material(o, reflect, refract, ior)]), iw.Compound([im, iw.Global(reflect, refract, ior)])

    floor = mono((0.3, 0.5, 0.3), 0.1, 100.0)
    wall = comp((0.2, 0.2, 0.6), 50.0, 0.3, 0.0, 1.0)
    matt = mono((0.8, 0.3, 0.2), 0.2, 20.0)
    mirror = comp((0.9, 0.9, 0.9), 100.0, 0.8, 0.0, 1.0)            # compound_material.rs:39-49
    t = 0.7
    glass = comp((0.8 * (1 - t), 0.9 * (1 - t), 1.0 * (1 - t)), 100.0, t, t, 1.5)  # compound_material.rs:51-63
    thin = comp((0.5, 0.5, 0.1), 10.0, 0.4, 0.4, 1.0)

    def add(handle, obj):
        o.add_object(handle)
        s.objects.append(obj)

    add(kit.plane(o, (0, -1, 0), (0, 0, 1), (0, 1, 0), floor[0]), iw.Plane(iw.vec(0, -1, 0), iw.vec(0, 0, 1), iw.vec(0, 1, 0), floor[1]))
    add(kit.plane(o, (0, 0, 9), (0, 1, 0), (0, 0, -1), wall[0]), iw.Plane(iw.vec(0, 0, 9), iw.vec(0, 1, 0), iw.vec(0, 0, -1), wall[1]))
    add(kit.sphere(o, (-2.2, 0.0, 5.0), 1.0, matt[0]), iw.Sphere(iw.vec(-2.2, 0.0, 5.0), 1.0, matt[1]))
    add(kit.sphere(o, (0.0, 0.2, 6.0), 1.2, mirror[0]), iw.Sphere(iw.vec(0.0, 0.2, 6.0), 1.2, mirror[1]))
    add(kit.sphere(o, (1.6, -0.3, 3.6), 0.7, glass[0]), iw.Sphere(iw.vec(1.6, -0.3, 3.6), 0.7, glass[1]))
    add(kit.sphere(o, (-0.6, -0.55, 3.2), 0.45, thin[0]), iw.Sphere(iw.vec(-0.6, -0.55, 3.2), 0.45, thin[1]))
    kit.light(o, abi.LIGHT_POINT, position=(3.0, 5.0, 0.0), colour=(0.9, 0.8, 0.7))
    s.lights.append(iw.PointLight(iw.vec(3.0, 5.0, 0.0), iw.vec(0.9, 0.8, 0.7)))
    kit.light(o, abi.LIGHT_DIRECTIONAL, direction=(0.5, -1.0, 0.3), colour=(0.3, 0.3, 0.4))
    s.lights.append(iw.DirectionalLight(iw.vec(0.5, -1.0, 0.3), iw.vec(0.3, 0.3, 0.4)))
    return o, s


def test_raytraced_frame_of_the_oracle_equals_an_independent_restatement():
    o, s = build_both()
    W, H, fov = 56, 40, 0.7
    pos, lookat, up = (0.0, 0.4, -1.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0)
    cam_o = ob.make_camera(W, H, fov, pos, lookat, up)
    cam_i = iw.Camera(W, H, fov, iw.vec(*pos), iw.vec(*lookat), iw.vec(*up))
    # the camera rays themselves
    for (x, y) in ((0, 0), (W - 1, H - 1), (17, 5)):
        oo, od = ob.camera_ray(cam_o, x, y)
        io, idr = cam_i.ray(x, y)
        assert np.array_equal(np.asarray(od, np.float32).view(np.uint32), np.asarray(idr, np.float32).view(np.uint32))
    o.pre_render(1)
    rgb_o, depth_o, id_o = o.render(cam_o)
    rgb_i, depth_i, id_i = iw.render(s, cam_i)
    assert np.array_equal(id_o, id_i)
    assert np.array_equal(depth_o.view(np.uint32), depth_i.view(np.uint32))          # hit distances: the same bits
    st = o.stats()
    assert st["rays_secondary"] == s.rays["secondary"] and st["rays_shadow"] == s.rays["shadow"]
    assert st["rays_secondary"] > 2 * W * H // 3                                     # the recursion is exercised (mirrors facing mirrors)
    # every operation is IEEE float32 in the same order on both sides and powf goes through double on both: the same bits
    same = rgb_o.view(np.uint32) == rgb_i.view(np.uint32)
    assert same.all(), "RGB differs at %d of %d values, max |diff| %g" % ((~same).sum(), same.size, np.abs(rgb_o - rgb_i).max())
    q = np.vectorize(iw.quantise)(rgb_i).astype(np.uint8)
    assert np.array_equal(q, ob.quantise(rgb_o))
