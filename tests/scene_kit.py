"""Small scene-building helpers shared by oracle tests and GPU parity tests.

Every helper takes a scene object that has the construction verbs of the C ABI
(add_material / object_define / object_transform / add_object / add_light): both
rrt.Scene (the product) and oracle_binding.OracleScene (the checker) do.
"""
import importlib

import numpy as np

rrt = importlib.import_module("rust-raytracer_amd")
abi = rrt._abi


def monochrome(scene, colour=(1.0, 0.0, 0.0), ambient=0.1, shininess=100.0):
    d = abi.MaterialDesc()
    d.kind = abi.MAT_MONOCHROME
    d.colour[:] = colour
    d.ambient, d.shininess = ambient, shininess
    d.diffuse_tex = d.normal_tex = -1
    return scene.add_material(d)


def global_material(scene, reflect, refract, ior):
    d = abi.MaterialDesc()
    d.kind = abi.MAT_GLOBAL
    d.reflect_weight, d.refract_weight, d.ior = reflect, refract, ior
    d.diffuse_tex = d.normal_tex = -1
    return scene.add_material(d)


def falsecolour(scene):
    d = abi.MaterialDesc()
    d.kind = abi.MAT_FALSECOLOUR
    d.diffuse_tex = d.normal_tex = -1
    return scene.add_material(d)


def texture_material(scene, diffuse, normal=-1, scale=1.0, ambient=0.1, shininess=100.0):
    d = abi.MaterialDesc()
    d.kind = abi.MAT_TEXTURE
    d.diffuse_tex, d.normal_tex = diffuse, normal
    d.scale, d.ambient, d.shininess = scale, ambient, shininess
    return scene.add_material(d)


def compound(scene, parts):
    d = abi.MaterialDesc()
    d.kind = abi.MAT_COMPOUND
    d.diffuse_tex = d.normal_tex = -1
    d.n_parts = len(parts)
    for i, p in enumerate(parts):
        d.parts[i] = p
    return scene.add_material(d)


def _obj(kind, material, values):
    d = abi.ObjectDesc()
    d.kind = kind
    d.material = material
    for i, v in enumerate(values):
        d.v[i] = v
    return d


def plane(scene, point, up, normal, material):
    return scene.object_define(_obj(abi.OBJ_PLANE, material, list(point) + list(up) + list(normal)))


def sphere(scene, centre, radius, material):
    return scene.object_define(_obj(abi.OBJ_SPHERE, material, list(centre) + [radius]))


def cuboid(scene, corner, size, material):
    return scene.object_define(_obj(abi.OBJ_CUBOID, material, list(corner) + list(size)))


def quadric(scene, coeffs, material):
    return scene.object_define(_obj(abi.OBJ_QUADRATIC, material, list(coeffs)))


def csg(scene, mode, left, right):
    d = abi.ObjectDesc()
    d.kind = abi.OBJ_CSG
    d.material = -1
    d.csg_mode, d.left, d.right = mode, left, right
    return scene.object_define(d)


def mesh(scene, positions, normals, material, smooth):
    import ctypes as C

    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3, 3)
    d = abi.ObjectDesc()
    d.kind = abi.OBJ_MESH
    d.material = material
    d.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
    if normals is not None:
        nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3, 3)
        d.normals = nrm.ctypes.data_as(C.POINTER(C.c_float))
    d.n_triangles = pos.shape[0]
    d.smooth = int(smooth)
    return scene.object_define(d)


def translation(t):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = t
    return m


def scaling(s):
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = m[1, 1] = m[2, 2] = s
    return m


def light(scene, kind, position=(0, 0, 0), direction=(0, 0, 0), colour=(1, 1, 1)):
    d = abi.LightDesc()
    d.kind = kind
    d.position[:] = position
    d.direction[:] = direction
    d.colour[:] = colour
    return scene.add_light(d)


def random_rays(n, seed, origin_scale=6.0, towards=(0.0, 1.0, 4.0), spread=4.0):
    """Rays from random origins aimed at random points near `towards` (plus some axis-parallel ones)."""
    rng = np.random.default_rng(seed)
    o = (rng.random((n, 3), dtype=np.float32) - 0.5) * 2 * origin_scale
    t = np.asarray(towards, np.float32) + (rng.random((n, 3), dtype=np.float32) - 0.5) * 2 * spread
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n // 16
    axes = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1)).astype(np.float32)
    d[:k] = axes
    return o.astype(np.float32), d.astype(np.float32)


def coherent_rays(n, seed, origin_scale=6.0, towards=(0.0, 1.0, 4.0), spread=4.0, cone=0.08):
    """Rays in bundles of 64 that share an origin (up to a small jitter) and aim into a narrow cone around a common
    target: what camera tiles and shadow rays of one surface patch look like.  Exercises the wave-level object cull."""
    rng = np.random.default_rng(seed)
    nb = (n + 63) // 64
    bo = (rng.random((nb, 1, 3), dtype=np.float32) - 0.5) * 2 * origin_scale
    bt = np.asarray(towards, np.float32) + (rng.random((nb, 1, 3), dtype=np.float32) - 0.5) * 2 * spread
    jitter = np.where(rng.random((nb, 1, 1)) < 0.5, 0.0, 0.05).astype(np.float32)
    o = bo + (rng.random((nb, 64, 3), dtype=np.float32) - 0.5) * jitter
    dist = np.linalg.norm(bt - bo, axis=2, keepdims=True)
    t = bt + (rng.random((nb, 64, 3), dtype=np.float32) - 0.5) * 2 * cone * dist
    d = t - o
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    return o.reshape(-1, 3)[:n].astype(np.float32), d.reshape(-1, 3)[:n].astype(np.float32)


def teapot_grid(obj_pos, obj_nrm, nx, ny, spacing=30.0):
    """Deterministic big mesh: an nx x ny grid of copies of a base mesh (BASELINE config 4 generator)."""
    pos = np.asarray(obj_pos, np.float32).reshape(-1, 3, 3)
    outs, outn = [], []
    for j in range(ny):
        for i in range(nx):
            off = np.array([(i - (nx - 1) / 2) * spacing, 0.0, (j - (ny - 1) / 2) * spacing], np.float32)
            outs.append(pos + off)
            if obj_nrm is not None:
                outn.append(np.asarray(obj_nrm, np.float32).reshape(-1, 3, 3))
    return np.concatenate(outs), (np.concatenate(outn) if outn else None)


def random_scene(scene, seed, with_mesh=True):
    """A deterministic random scene out of every primitive, CSG trees up to depth 3, per-object transforms and an optional
    small triangle soup; the same call sequence builds the product scene and the oracle scene."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi, n=None: rng.uniform(lo, hi, n).astype(np.float32) if n else np.float32(rng.uniform(lo, hi))
    mats = [monochrome(scene, tuple(u(0.1, 1.0, 3)), 0.1, 50.0) for _ in range(3)]
    mats.append(compound(scene, [monochrome(scene, (0.5, 0.5, 0.8)), global_material(scene, 0.5, 0.0, 1.0)]))
    mats.append(compound(scene, [monochrome(scene, (0.2, 0.3, 0.2)), global_material(scene, 0.8, 0.8, 1.4)]))

    def primitive(allow_plane=True):
        kind = rng.integers(0, 5 if allow_plane else 4)
        m = mats[rng.integers(0, len(mats))]
        c = u(-2.0, 2.0, 3)
        if kind == 0:
            return sphere(scene, tuple(c), float(u(0.3, 1.2)), m)
        if kind == 1:
            return cuboid(scene, tuple(c - 0.5), tuple(u(0.3, 1.8, 3)), m)
        if kind == 2:  # ellipsoid / cylinder / paraboloid-like quadrics around the origin, moved by the transform below
            q = np.zeros(10, np.float32)
            q[0], q[4], q[7] = u(0.5, 4.0), u(0.0, 4.0) * rng.integers(0, 2), u(0.5, 4.0)
            q[9] = -u(0.3, 1.5)
            if rng.integers(0, 4) == 0:
                q[6] = u(-0.5, 0.5)
            h = quadric(scene, tuple(q), m)
            scene.object_transform(h, translation(tuple(c)))
            return h
        if kind == 3:
            return sphere(scene, tuple(c), float(u(0.2, 0.6)), m)
        n = u(-1.0, 1.0, 3)
        n /= np.linalg.norm(n)
        up = np.cross(n, u(-1.0, 1.0, 3)).astype(np.float32)
        up /= np.linalg.norm(up)
        return plane(scene, tuple(u(-2.5, -1.0, 3)), tuple(up), tuple(n), m)

    def tree(depth):
        if depth == 0 or rng.integers(0, 3) == 0:
            return primitive(allow_plane=rng.integers(0, 6) == 0)
        mode = int(rng.integers(0, 3))
        return csg(scene, mode, tree(depth - 1), tree(depth - 1))

    for _ in range(int(rng.integers(4, 9))):
        h = tree(int(rng.integers(0, 4)))
        t = rng.integers(0, 3)
        if t == 1:
            scene.object_transform(h, translation(tuple(u(-1.0, 1.0, 3))))
        elif t == 2:
            scene.object_transform(h, scaling(float(u(0.6, 1.6))))
        scene.add_object(h)
    if with_mesh and rng.integers(0, 2) == 0:
        n_tri = int(rng.integers(8, 60))
        base = u(-2.0, 2.0, (n_tri, 1, 3))
        tris = base + u(-0.6, 0.6, (n_tri, 3, 3))
        scene.add_object(mesh(scene, tris, None, mats[0], False))
    light(scene, abi.LIGHT_POINT, position=tuple(u(-4.0, 4.0, 3)), colour=(1.0, 1.0, 1.0))
    return scene


def csg_mesh_scene(scene, seed):
    """CSG trees with MESH leaves (csg_object.rs:25-36 takes any Box<dyn Object>; PolyMesh::intersect hands the merge at
    most one hit, polymesh_object.rs:216-242): a closed box mesh and a triangle soup combined with spheres, cuboids and each
    other in all three modes, next to plain objects."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi, n=None: rng.uniform(lo, hi, n).astype(np.float32) if n else np.float32(rng.uniform(lo, hi))
    mats = [monochrome(scene, tuple(u(0.1, 1.0, 3)), 0.1, 50.0) for _ in range(2)]
    mats.append(compound(scene, [monochrome(scene, (0.5, 0.5, 0.8)), global_material(scene, 0.5, 0.0, 1.0)]))

    def box_mesh(c, s):
        x0, y0, z0 = c
        x1, y1, z1 = c + s
        v = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]], np.float32)
        f = [(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (3, 7, 6), (3, 6, 2), (0, 4, 7), (0, 7, 3), (1, 2, 6), (1, 6, 5)]
        return mesh(scene, v[np.array(f)], None, mats[int(rng.integers(0, 3))], False)

    def soup():
        n_tri = int(rng.integers(6, 40))
        tris = u(-1.5, 1.5, (n_tri, 1, 3)) + u(-0.7, 0.7, (n_tri, 3, 3))
        return mesh(scene, tris, None, mats[int(rng.integers(0, 3))], False)

    def leaf():
        k = int(rng.integers(0, 4))
        c = u(-1.5, 1.5, 3)
        if k == 0:
            return box_mesh(c - 0.5, u(0.5, 1.6, 3))
        if k == 1:
            return soup()
        if k == 2:
            return sphere(scene, tuple(c), float(u(0.4, 1.2)), mats[int(rng.integers(0, 3))])
        return cuboid(scene, tuple(c - 0.5), tuple(u(0.4, 1.5, 3)), mats[int(rng.integers(0, 3))])

    def tree(depth):
        if depth == 0:
            return leaf()
        return csg(scene, int(rng.integers(0, 3)), tree(depth - 1), tree(depth - 1))

    for _ in range(int(rng.integers(3, 6))):
        h = tree(int(rng.integers(1, 3)))
        t = rng.integers(0, 3)
        if t == 1:
            scene.object_transform(h, translation(tuple(u(-1.0, 1.0, 3))))
        elif t == 2:
            scene.object_transform(h, scaling(float(u(0.7, 1.5))))
        scene.add_object(h)
    scene.add_object(sphere(scene, tuple(u(-2.0, 2.0, 3)), 0.5, mats[0]))
    light(scene, abi.LIGHT_POINT, position=tuple(u(-4.0, 4.0, 3)), colour=(1.0, 1.0, 1.0))
    return scene
