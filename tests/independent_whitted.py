"""A second, independent restatement of the reference's raytracing-mode shading chain, written straight from the Rust sources in
numpy float32 scalars (test infrastructure, like tests/independent_scene.py for the parser): camera rays, sphere and plane
intersection, Scene::trace / shadowtrace / raytrace, the Phong terms, GlobalMaterial's reflection, refraction and Fresnel mix, the
compound mean, point and directional lights, the quantiser.  It shares no code with oracle/ or the product; the test compares it
with the oracle pixel by pixel, so that a misreading of the reference would have to be made twice, in two languages, to go unseen.

Every operation is a float32 operation in the reference's order (numpy float32 scalars round after every step, Python
evaluates left to right as Rust does); powf is taken through double and rounded once, as DESIGN.md section 2 explains for the oracle.
"""
import math

import numpy as np

f = np.float32
F32_MAX = f(3.4028234663852886e38)
INF = f(np.inf)


# ---- core/vector.rs -------------------------------------------------------------------------------------------------
def dot(a, b):  # vector.rs:78-80
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def length(a):  # vector.rs:70-76
    return np.sqrt(dot(a, a))


def normalised(a):  # vector.rs:61-68
    l = length(a)
    return (a[0] / l, a[1] / l, a[2] / l)


def neg(a):
    return (-a[0], -a[1], -a[2])


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def scale(a, s):  # Vector * f32
    return (a[0] * s, a[1] * s, a[2] * s)


def cross(a, b):  # vector.rs:106-112
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def reflection(n, i):  # vector.rs:82-90
    d = dot(n, i) * f(2.0)
    return (i[0] - d * n[0], i[1] - d * n[1], i[2] - d * n[2])


def vec(x, y, z):
    return (f(x), f(y), f(z))


BLACK = vec(0, 0, 0)


# ---- cameras/full_camera.rs --------------------------------------------------------------------------------------------
class Camera:
    def __init__(self, width, height, fov, position, lookat, up):  # full_camera.rs:23-58
        lookat = (lookat[0], -lookat[1], lookat[2])
        up = (up[0], -up[1], up[2])
        self.lookat, self.up = normalised(lookat), normalised(up)
        self.right = normalised(cross(self.lookat, self.up))
        self.width, self.height, self.fov, self.position = width, height, f(fov), position

    def ray(self, x, y):  # full_camera.rs:61-81
        fx = (f(x) + f(0.5)) / f(self.width)
        fy = (f(y) + f(0.5)) / f(self.height)
        v = (fx - f(0.5), fy - f(0.5), self.fov)
        r, u, l = self.right, self.up, self.lookat
        # the rotation matrix has rows right / up / lookat and is transposed before it is applied (vector.rs:121-131)
        d = (r[0] * v[0] + u[0] * v[1] + l[0] * v[2], r[1] * v[0] + u[1] * v[1] + l[1] * v[2], r[2] * v[0] + u[2] * v[1] + l[2] * v[2])
        return self.position, normalised(d)


# ---- objects: hits are (distance, entering, position, normal) ----------------------------------------------------------
class Sphere:
    def __init__(self, centre, radius, material):
        self.centre, self.radius, self.material = centre, f(radius), material

    def intersect(self, o, d):  # sphere_object.rs:35-91
        ro = sub(o, self.centre)
        a = dot(d, d)
        b = f(2.0) * dot(d, ro)
        c = dot(ro, ro) - self.radius * self.radius
        disc = b * b - f(4.0) * a * c
        if disc < 0:
            return []
        ds = np.sqrt(disc)
        hits = []
        for t, entering in (((-b - ds) / f(2.0), True), ((-b + ds) / f(2.0), False)):
            p = add(o, scale(d, t))
            n = normalised(sub(p, self.centre))
            if dot(n, d) > 0:
                n = neg(n)
            hits.append((t, entering, p, n))
        return hits


class Plane:
    def __init__(self, point, up, normal, material):  # plane_object.rs:27-41
        self.normal, self.material = normal, material
        self.d = -dot(normal, point)

    def intersect(self, o, d):  # plane_object.rs:54-113 (no textures here: tex coords and normal maps left out)
        U = dot(self.normal, o) + self.d
        V = dot(self.normal, d)
        zero = vec(0, 0, 0)
        if V == 0:
            return [(-INF, True, zero, zero), (INF, False, zero, zero)] if U >= 0 else []
        t = U / -V
        if V > 0:
            return [(-INF, True, zero, zero), (t, False, add(o, scale(d, t)), neg(self.normal))]
        return [(t, True, add(o, scale(d, t)), self.normal), (INF, False, zero, zero)]


# ---- materials -----------------------------------------------------------------------------------------------------------
class Monochrome:  # phong_material.rs:12-94
    def __init__(self, colour, ambient, shininess):
        self.colour, self.ambient, self.shininess = colour, f(ambient), f(shininess)

    def compute_once(self, scene, ray, hit, depth):
        return scale(self.colour, self.ambient)

    def compute_per_light(self, viewer, hit, ldir):
        n = hit[3]
        diffuse = scale(self.colour, -dot(n, ldir))
        refl = normalised(reflection(n, ldir))
        s = max(dot(viewer, refl), f(0.0))
        s = f(math.pow(float(s), float(self.shininess)))
        return add(diffuse, scale(vec(1, 1, 1), s))


class Global:  # global_material.rs:12-108
    def __init__(self, reflect, refract, ior):
        self.wr, self.wt, self.ior = f(reflect), f(refract), f(ior)

    def refraction(self, hit, incoming):  # :28-67
        if self.wt == 0:
            return None
        N, I = hit[3], neg(incoming)
        cos_i = dot(N, I)
        if hit[1]:
            N = neg(N)
            n1, n2 = f(1.0), self.ior
        else:
            cos_i = -cos_i
            n1, n2 = self.ior, f(1.0)
        eta = n1 / n2
        with np.errstate(invalid="ignore"):
            cos_t = np.sqrt(f(1.0) - (eta * eta) * (f(1.0) - cos_i * cos_i))
        if np.isnan(cos_t):
            return None
        T = sub(scale(I, eta), scale(N, cos_t - eta * cos_i))
        T = neg(normalised(T))
        r_par = (n2 * cos_i - n1 * cos_t) / (n2 * cos_i + n1 * cos_t)
        r_per = (n1 * cos_i - n2 * cos_t) / (n1 * cos_i + n2 * cos_t)
        kr = (r_par * r_par + r_per * r_per) / f(2.0)
        return add(hit[2], scale(T, f(0.0001))), T, kr

    def compute_once(self, scene, ray, hit, depth):  # :71-108
        if depth >= 5:
            return BLACK
        o, d = ray
        refl = refr = None
        if self.wr > 0:
            rd = normalised(reflection(hit[3], d))
            ro = add(hit[2], scale(rd, f(0.0001)))
            refl = scale(scene.raytrace((ro, rd), depth + 1)[0], self.wr)
        kr = f(0.0)
        if self.wt > 0:
            res = self.refraction(hit, d)
            if res is not None:
                kr = res[2]
                refr = scale(scene.raytrace((res[0], res[1]), depth + 1)[0], self.wt)
        if refl is not None and refr is not None:
            return add(scale(refl, kr), scale(refr, f(1.0) - kr))
        if refl is not None:
            return refl
        if refr is not None:
            return refr
        return BLACK

    def compute_per_light(self, viewer, hit, ldir):  # material.rs default
        return BLACK


class Compound:  # compound_material.rs:79-102
    def __init__(self, parts):
        self.parts = parts

    def compute_once(self, scene, ray, hit, depth):
        acc = BLACK
        for m in self.parts:
            acc = add(acc, m.compute_once(scene, ray, hit, depth))
        n = f(len(self.parts))
        return (acc[0] / n, acc[1] / n, acc[2] / n)

    def compute_per_light(self, viewer, hit, ldir):
        acc = BLACK
        for m in self.parts:
            acc = add(acc, m.compute_per_light(viewer, hit, ldir))
        n = f(len(self.parts))
        return (acc[0] / n, acc[1] / n, acc[2] / n)


# ---- lights --------------------------------------------------------------------------------------------------------------
class PointLight:  # point_light.rs:31-40
    def __init__(self, position, intensity):
        self.position, self.intensity = position, intensity

    def direction(self, surface):
        return normalised(sub(surface, self.position))


class DirectionalLight:  # directional_light.rs:19-27: the direction as given, NOT normalised
    def __init__(self, direction, intensity):
        self.dir, self.intensity = direction, intensity

    def direction(self, surface):
        return self.dir


# ---- environments/scene.rs ---------------------------------------------------------------------------------------------------
class Scene:
    def __init__(self):
        self.objects, self.lights = [], []
        self.rays = {"secondary": 0, "shadow": 0}

    @staticmethod
    def first_hit(hits):  # scene.rs:21-40
        best, best_d = None, F32_MAX
        for h in hits:
            if h[0] < 0 or not h[1]:
                continue
            if h[0] < best_d:
                best_d, best = h[0], h
        return best

    def trace(self, o, d):  # scene.rs:42-61
        best, best_d, which = None, F32_MAX, -1
        for k, obj in enumerate(self.objects):
            h = self.first_hit(obj.intersect(o, d))
            if h is None:
                continue
            if h[0] < best_d:
                best_d, best, which = h[0], h, k
        return best, which

    def shadowtrace(self, o, d, limit):  # scene.rs:65-79
        self.rays["shadow"] += 1
        for obj in self.objects:
            h = self.first_hit(obj.intersect(o, d))
            if h is None:
                continue
            if h[0] > f(0.0000001) and h[0] < limit:
                return True
        return False

    def raytrace(self, ray, depth):  # scene.rs:83-128
        if depth > 0:
            self.rays["secondary"] += 1
        o, d = ray
        hit, which = self.trace(o, d)
        if hit is None:
            return BLACK, f(0.0), -1
        material = self.objects[which].material
        colour = material.compute_once(self, ray, hit, depth)
        for light in self.lights:
            viewer = neg(normalised(hit[2]))
            ldir = light.direction(hit[2])
            if dot(ldir, hit[3]) > 0:
                continue
            sd = neg(ldir)
            so = add(hit[2], scale(sd, f(0.0001)))
            if self.shadowtrace(so, sd, length(ldir)):
                continue
            term = material.compute_per_light(viewer, hit, ldir)
            colour = add(colour, (term[0] * light.intensity[0], term[1] * light.intensity[1], term[2] * light.intensity[2]))
        return colour, hit[0], which


def quantise(v):  # framebuffer.rs:96-130: `(c * 255.0) as u8`, Rust's saturating cast (NaN -> 0)
    x = f(v) * f(255.0)
    if np.isnan(x) or x <= 0:
        return 0
    return 255 if x >= 255 else int(x)


def render(scene, cam):
    rgb = np.zeros((cam.height, cam.width, 3), np.float32)
    depth = np.zeros((cam.height, cam.width), np.float32)
    ids = np.zeros((cam.height, cam.width), np.uint32)
    for y in range(cam.height):
        for x in range(cam.width):
            c, t, which = scene.raytrace(cam.ray(x, y), 0)
            rgb[y, x] = c
            depth[y, x] = t
            ids[y, x] = np.uint32(which & 0xFFFFFFFF)
    return rgb, depth, ids
