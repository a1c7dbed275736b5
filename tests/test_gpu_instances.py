"""Kernel instances per scene feature set (k_trace_shade is compiled per mode and per what the scene holds; scenes of planes and
meshes only take the FEAT_NO_SOLIDS instances, whose sphere / quadric / cuboid cases are not compiled at all).  Adding a
sphere that no ray can see must switch the instance and change nothing else: frames, ray counts and the statistics of the
mesh walk stay the same bit for bit -- for the single-level (fused) kernel and for the one with child rays."""
import importlib
import math
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(rrt, bench_extra, variant, with_sphere):
    abi = rrt._abi
    scene, cam, ntri = bench_extra.build_mesh_scene(rrt, variant=variant)
    if with_sphere:  # far behind the camera and below the floor: no camera, shadow or child ray of this view can reach it
        d = abi.ObjectDesc()
        d.kind = abi.OBJ_SPHERE
        d.material = 0
        d.v[:4] = (0.0, 50000.0, -900000.0, 1.0)
        scene.add_object(scene.object_define(d))
    cam = rrt.make_camera(480, 272, math.radians(40.0), (0.0, 260.0, -420.0), (0.0, 0.55, 1.0))
    scene.pre_render(1)
    return scene, cam


@pytest.mark.parametrize("variant", ["phong", "mirror", "glass"])
def test_unseen_sphere_switches_the_instance_and_nothing_else(repo, variant):
    sys.path.insert(0, repo)
    rrt = importlib.import_module("rust-raytracer_amd")
    import bench_extra

    plain, cam = _scene(rrt, bench_extra, variant, False)
    solid, _ = _scene(rrt, bench_extra, variant, True)
    a = plain.render(cam)
    sa = plain.stats()
    b = solid.render(cam)
    sb = solid.stats()
    for x, y in zip(a, b):
        assert np.array_equal(np.ascontiguousarray(x).view(np.uint32), np.ascontiguousarray(y).view(np.uint32))
    for k in ("rays_primary", "rays_secondary", "rays_shadow", "bvh_nodes_visited", "bvh_tris_tested"):
        assert sa[k] == sb[k], k
    assert sa["rays_shadow"] > 0 and (variant == "phong" or sa["rays_secondary"] > 0)
    # the two scenes really differ in what the library was told: one record more, and its arithmetic priced per ray
    assert sb["flops_per_trace"] > sa["flops_per_trace"]


def _box_triangles(c, s):
    x0, y0, z0 = c
    x1, y1, z1 = (c[0] + s[0], c[1] + s[1], c[2] + s[2])
    v = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]], np.float32)
    f = [(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (3, 7, 6), (3, 6, 2), (0, 4, 7), (0, 7, 3), (1, 2, 6), (1, 6, 5)]
    return v[np.array(f)]


def _axis_rays():
    """Rays along the axes against axis-aligned boxes: for eight of a box's twelve triangles n . d is EXACTLY zero (the early
    exit of Triangle::intersect, triangle_object.rs:129-133), with origins inside the triangles' planes (U = 0: 0 / -0), beside them
    and outside the boxes; plus the same origins with slightly tilted directions."""
    rng = np.random.default_rng(7)
    axes = np.concatenate([np.eye(3, dtype=np.float32), -np.eye(3, dtype=np.float32)])
    o, d = [], []
    for face in (-1.0, 0.0, 0.5, 1.0, 2.0, 2.5):  # coordinates of box faces (boxes at [-1, 0], [0.5, 1] ... see the scene) and between
        for a in axes:
            for _ in range(24):
                p = rng.uniform(-1.4, 2.7, 3).astype(np.float32)
                k = int(rng.integers(0, 3))
                p[k] = face  # the origin lies IN the plane of the faces at this coordinate
                o.append(p - a * np.float32(4.0) * (np.float32(1.0) if rng.integers(0, 2) else np.float32(0.0)))
                d.append(a)
    o, d = np.array(o, np.float32), np.array(d, np.float32)
    tilt = d + rng.normal(0.0, 1e-3, d.shape).astype(np.float32)
    tilt /= np.linalg.norm(tilt, axis=1, keepdims=True)
    return np.concatenate([o, o]), np.concatenate([d, tilt.astype(np.float32)])


@pytest.mark.parametrize("mirror", [False, True])
def test_rays_parallel_to_triangle_planes(repo, mirror):
    """The triangle test's early exit for n . d == 0 exists in the fused single-level instance and is folded into the final
    predicate in all others (tri_test<EAGER>: the quotient of a parallel ray, 0 / -0 or x / 0, is computed and thrown away).
    Both must answer like the reference's exit: rays along the axes through a scene of axis-aligned box meshes, Phong (fused
    instance) and with a mirror part (child rays: the eager instance), against the oracle -- ids and depths bit for bit."""
    sys.path.insert(0, repo)
    sys.path.insert(0, os.path.join(repo, "tests"))
    import oracle_binding as ob
    import scene_kit as kit

    rrt = ob.rrt
    scenes = []
    for cls in (rrt.Scene, ob.OracleScene):
        s = cls()
        phong = kit.monochrome(s, (0.8, 0.3, 0.2), 0.1, 30.0)
        mat = kit.compound(s, [phong, kit.global_material(s, 0.6, 0.0, 1.0)]) if mirror else phong
        for c, size in (((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)), ((2.0, -1.0, 0.0), (0.5, 3.0, 2.5))):
            s.add_object(kit.mesh(s, _box_triangles(c, size), None, mat, False))
        s.add_object(kit.plane(s, (0.0, -2.0, 0.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), phong))
        kit.light(s, rrt._abi.LIGHT_POINT, position=(3.0, 6.0, -4.0), colour=(1.0, 1.0, 1.0))
        scenes.append(s)
    gpu, cpu = scenes
    gpu.pre_render(1)
    o, d = _axis_rays()
    rgb_g, depth_g, id_g = gpu.raytrace(o, d)
    rgb_c, depth_c, id_c = cpu.raytrace(o, d)
    assert np.array_equal(id_g, id_c)
    assert np.array_equal(np.ascontiguousarray(depth_g).view(np.uint32), np.ascontiguousarray(depth_c).view(np.uint32))
    assert np.abs(rrt.quantise(rgb_g).astype(int) - ob.quantise(rgb_c).astype(int)).max() <= 1
    st = gpu.stats()
    assert st["bvh_tris_tested"] > 0 and (st["rays_secondary"] > 0) == mirror
    hit_mesh = np.isin(id_g, [0, 1, 2]).sum()
    assert hit_mesh > len(o) // 8  # the rays do meet the boxes
