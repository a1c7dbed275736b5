"""An independent restatement of the reference's scene loader (src/scene_file.rs) in Python, for tests only.

The oracle is normally fed by the PRODUCT's C++ parser, OBJ and PPM readers (through the builder vtable), so a bug in
those would move both sides of every parity test together.  This module reads the same files with none of that code:
its own paragraph grammar (scene_file.rs:124-285), its own object / material / light / camera builders (:287-601,
compound_material.rs:39-76, quadratic_object.rs:34-38), its own OBJ reader (polymesh_object.rs:34-194) and P6 reader
(texture.rs:41-103), and hands the result to a scene through the construction verbs of the C ABI (add_material,
object_define, object_transform, add_object, add_light, add_texture).  tests/test_host_side.py loads every scene file both
ways into the oracle and compares what rays see.
"""
import os

import numpy as np

import scene_kit as kit

abi = kit.abi
f32 = np.float32


# ---- grammar (scene_file.rs:124-253) ----------------------------------------------------------------------------------
class Paragraph:
    def __init__(self, kind, cls, attrs):
        self.kind, self.cls, self.attrs = kind, cls, attrs

    def take(self, key, default=None, required=False):
        if key in self.attrs:
            return self.attrs.pop(key)
        if required:
            raise ValueError("Missing required attribute: " + key)
        return default


def _indent(s):
    return len(s) - len(s.lstrip())


def _parse_paragraph(lines):
    head = lines[0].split()
    if len(head) != 2:
        raise ValueError("bad paragraph header: " + lines[0])
    base = _indent(lines[0])
    attrs = {}
    for i in range(1, len(lines)):
        line = lines[i]
        if not line or _indent(line) <= base:
            break
        words = line.split()
        key, words = words[0], words[1:]
        if key.startswith("#"):
            continue
        if len(words) == 3:
            value = ("vector", tuple(f32(w) for w in words))
        elif len(words) == 1:
            if i + 1 < len(lines) and _indent(lines[i + 1]) > _indent(line):
                value = ("paragraph", _parse_paragraph(lines[i:]))
            else:
                try:
                    value = ("float", f32(words[0]))
                except ValueError:
                    value = ("word", words[0])
        else:
            raise ValueError("Invalid word count in attribute value: " + line)
        attrs[key] = value          # HashMap insert: the last one wins
    return Paragraph(head[0], head[1], attrs)


def parse_paragraphs(text):
    out, cur = [], []
    for line in text.splitlines():
        if not line.strip() or line.startswith("#"):
            if cur:
                out.append(_parse_paragraph(cur))
                cur = []
            continue
        if not cur and _indent(line) == 0:
            cur = [line]
        elif cur and _indent(line) > 0:
            cur.append(line)
        else:
            raise ValueError("Started next paragraph without empty newline")
    if cur:
        out.append(_parse_paragraph(cur))
    return out


# ---- attribute conversions (scene_file.rs:603-660) ---------------------------------------------------------------------
def as_float(v):
    assert v[0] == "float", v
    return v[1]


def as_vector(v):
    if v[0] == "vector":
        return v[1]
    if v[0] == "float":
        return (v[1], v[1], v[1])
    raise ValueError("Invalid attribute value for vector")


def as_colour(v):
    if v[0] == "word":
        return {"White": (f32(1), f32(1), f32(1)), "Black": (f32(0), f32(0), f32(0))}[v[1]]
    return as_vector(v)


def as_word(v):
    assert v[0] == "word", v
    return v[1]


# ---- files ----------------------------------------------------------------------------------------------------------------
def read_p6(path):
    """texture.rs:41-103: P6, one or two bytes per sample (big endian); returns width, height, maxval, raw sample bytes."""
    data = open(path, "rb").read()
    pos, tokens = 0, []
    while len(tokens) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while data[pos:pos + 1] != b"\n":
                pos += 1
            continue
        start = pos
        while not data[pos:pos + 1].isspace():
            pos += 1
        tokens.append(data[start:pos])
    pos += 1  # the single whitespace byte behind maxval
    assert tokens[0] == b"P6"
    w, h, maxval = int(tokens[1]), int(tokens[2]), int(tokens[3])
    n = w * h * 3 * (1 if maxval <= 255 else 2)
    return w, h, maxval, data[pos:pos + n]


def read_obj(path):
    """PolyMesh::from_obj_file / parse_face (polymesh_object.rs:34-194): corner positions and normals per triangle."""
    verts, normals, vnorm, pos, nrm = [], [], {}, [], []
    for line in open(path):
        w = line.split()
        if not w or w[0] == "#":
            continue
        if w[0] == "v":
            verts.append([f32(x) for x in w[1:4]])
        elif w[0] == "vn":
            normals.append([f32(x) for x in w[1:4]])
        elif w[0] == "f":
            idx = []
            for c in w[1:]:
                p = c.split("/")
                idx.append((int(p[0]) - 1, int(p[2]) - 1 if len(p) > 2 and p[2] else None))
            for t in [(0, 1, 2)] + ([(0, 2, 3)] if len(idx) == 4 else []):
                for c in t:
                    vnorm[idx[c][0]] = idx[c][1]   # the vertex keeps the normal index of the last face that names it
                for c in t:
                    pos.append(verts[idx[c][0]])
                    nrm.append(normals[vnorm[idx[c][0]]] if vnorm[idx[c][0]] is not None else [0, 0, 0])
    return np.array(pos, np.float32).reshape(-1, 3, 3), np.array(nrm, np.float32).reshape(-1, 3, 3)


# ---- builders -------------------------------------------------------------------------------------------------------------
class Loader:
    def __init__(self, scene, asset_root):
        self.s, self.root = scene, asset_root

    def texture(self, name, which):
        path = os.path.join(self.root, "assets/textures", name, which + ".ppm")   # texture.rs:22-39: the .ppm sibling
        if not os.path.exists(path):
            return -1
        w, h, maxval, samples = read_p6(path)
        return self.s.add_texture(w, h, maxval, samples)

    def texture_material(self, name, scale, ambient, shininess):                  # Texture::import, texture.rs:127-152
        return kit.texture_material(self.s, self.texture(name, "diffuse"), self.texture(name, "normal"), scale, ambient, shininess)

    def material(self, v):                                                         # into_material, scene_file.rs:509-551
        assert v[0] == "paragraph", v
        p, s = v[1], self.s
        if p.cls == "Simple":                                                      # compound_material.rs:39-49
            colour, refl = as_colour(p.take("colour", required=True)), as_float(p.take("reflectiveness", required=True))
            shin = as_float(p.take("shininess", ("float", f32(100))))
            return kit.compound(s, [kit.monochrome(s, colour, f32(0.1), shin), kit.global_material(s, refl, f32(0), f32(1))])
        if p.cls == "Transparent":                                                 # :51-63
            colour, t = as_colour(p.take("colour", required=True)), as_float(p.take("transparency", required=True))
            ior, shin = as_float(p.take("ior", required=True)), as_float(p.take("shininess", ("float", f32(100))))
            opaque = f32(1) - t
            return kit.compound(s, [kit.monochrome(s, tuple(f32(c) * opaque for c in colour), f32(0.1), shin), kit.global_material(s, t, t, ior)])
        if p.cls == "Global":
            return kit.global_material(s, as_float(p.take("reflect", required=True)), as_float(p.take("refract", required=True)),
                                       as_float(p.take("ior", required=True)))
        if p.cls == "Monochrome":
            return kit.monochrome(s, as_colour(p.take("colour", required=True)), as_float(p.take("ambient", ("float", f32(0.1)))),
                                  as_float(p.take("shininess", required=True)))
        if p.cls == "Texture":
            return self.texture_material(as_word(p.take("name", required=True)), as_float(p.take("scale", required=True)),
                                         as_float(p.take("ambient", required=True)), as_float(p.take("shininess", required=True)))
        if p.cls == "TransparentTexture":                                          # :65-76
            name, scale = as_word(p.take("name", required=True)), as_float(p.take("scale", required=True))
            t = as_float(p.take("transparency", required=True))
            return kit.compound(s, [self.texture_material(name, scale, f32(0.1), f32(1000000.0)), kit.global_material(s, t, t, f32(1))])
        if p.cls == "FalseColour":
            return kit.falsecolour(s)
        raise ValueError("Invalid material class: " + p.cls)

    def cylinder(self, diameter, m):                                               # quadratic_object.rs:34-38
        radius = f32(diameter) / f32(2)
        a = f32(1) / (radius * radius)
        return kit.quadric(self.s, [a, 0, 0, 0, 0, 0, 0, a, 0, -1], m)

    def scale_translate(self, p):
        """identity with scale on the diagonal and the translation in row 3, then transposed (scene_file.rs:381-399, 440-453)"""
        m = np.eye(4, dtype=np.float32)
        sc, tr = p.take("scale"), p.take("translate")
        if tr is not None:
            m[:3, 3] = as_vector(tr)
        if sc is not None:
            m[0, 0] = m[1, 1] = m[2, 2] = as_float(sc)
        return m

    def object(self, p):                                                           # into_object, scene_file.rs:310-507
        s = self.s
        if p.cls == "Plane":
            return kit.plane(s, as_vector(p.take("point", required=True)), as_vector(p.take("up", required=True)),
                             as_vector(p.take("normal", required=True)), self.material(p.take("material", required=True)))
        if p.cls == "Sphere":
            return kit.sphere(s, as_vector(p.take("centre", required=True)), as_float(p.take("radius", required=True)),
                              self.material(p.take("material", required=True)))
        if p.cls == "Cuboid":
            return kit.cuboid(s, as_vector(p.take("corner", required=True)), as_vector(p.take("size", required=True)),
                              self.material(p.take("material", required=True)))
        if p.cls == "Quadratic":
            coeffs = [as_float(p.take(k, ("float", f32(0)))) for k in "abcdefghij"]
            h = kit.quadric(s, coeffs, self.material(p.take("material", required=True)))
            tr = p.take("translate")
            if tr is not None:
                s.object_transform(h, kit.translation(as_vector(tr)))
            return h
        if p.cls == "Model":
            pos, nrm = read_obj(os.path.join(self.root, "assets/models", as_word(p.take("obj", required=True))))
            m = self.material(p.take("material", required=True))
            smooth = as_float(p.take("smooth", ("float", f32(0)))) != 0
            h = kit.mesh(s, pos, nrm, m, smooth)
            if p.take("rotate_teapot") is not None:
                rot = np.eye(4, dtype=np.float32)
                rot[:3, :3] = [[1, 0, 0], [0, 0, 1], [0, -1, 0]]
                s.object_transform(h, rot)
            s.object_transform(h, self.scale_translate(p))                         # always applied, identity or not
            return h
        if p.cls == "MarioPipe":                                                   # :403-455
            m = self.material(p.take("material", required=True))
            main = kit.csg(s, abi.CSG_INTERSECTION, self.cylinder(1.0, m), kit.cuboid(s, (-0.5, 0.0, -0.5), (1.0, 3.0, 1.0), m))
            top = kit.csg(s, abi.CSG_INTERSECTION, self.cylinder(1.3, m), kit.cuboid(s, (-0.7, 2.7, -0.7), (1.3, 0.3, 1.3), m))
            pipe = kit.csg(s, abi.CSG_UNION, main, top)
            hole = kit.csg(s, abi.CSG_INTERSECTION, kit.sphere(s, (0.0, 3.0, 0.0), 0.5, kit.monochrome(s, (0, 0, 0), f32(0.1), f32(100000000.0))),
                           kit.cuboid(s, (-0.5, 3.0, -0.5), (1.0, 0.0001, 1.0), kit.monochrome(s, (0, 0, 0), f32(0.1), f32(100000000.0))))
            pipe = kit.csg(s, abi.CSG_UNION, pipe, hole)
            s.object_transform(pipe, self.scale_translate(p))                      # scale and translation in ONE matrix
            return pipe
        if p.cls == "MarioCoin":                                                   # :456-503
            m = self.material(p.take("material", required=True))
            coin = kit.csg(s, abi.CSG_INTERSECTION, kit.sphere(s, (0, 0, 0), 0.5, m), kit.cuboid(s, (-0.5, -0.5, -0.1), (1.0, 1.0, 0.2), m))
            shallow = kit.csg(s, abi.CSG_INTERSECTION, kit.sphere(s, (0, 0, 0), 0.4, m), kit.cuboid(s, (-0.4, -0.4, -0.11), (0.8, 0.8, 0.03), m))
            coin = kit.csg(s, abi.CSG_DIFFERENCE, coin, shallow)
            coin = kit.csg(s, abi.CSG_DIFFERENCE, coin, kit.cuboid(s, (-0.08, -0.28, -0.1), (0.16, 0.56, 0.05), m))
            sc, tr = p.take("scale"), p.take("translate")
            if sc is not None:                                                     # TWO matrices: scale first, then the translation
                s.object_transform(coin, kit.scaling(as_float(sc)))
            if tr is not None:
                s.object_transform(coin, kit.translation(as_vector(tr)))
            return coin
        raise ValueError("Invalid object class: " + p.cls)

    def light(self, p):                                                            # into_light, scene_file.rs:287-308
        colour = as_colour(p.take("colour", ("float", f32(1))))
        if p.cls == "Directional":
            return kit.light(self.s, abi.LIGHT_DIRECTIONAL, direction=as_vector(p.take("direction", required=True)), colour=colour)
        if p.cls == "Point":
            return kit.light(self.s, abi.LIGHT_POINT, position=as_vector(p.take("position", required=True)), colour=colour)
        if p.cls == "DirPoint":
            return kit.light(self.s, abi.LIGHT_DIRPOINT, position=as_vector(p.take("position", required=True)),
                             direction=as_vector(p.take("direction", required=True)), colour=colour)
        raise ValueError("Invalid light class: " + p.cls)


def camera_of(p, make_camera):
    """into_camera (scene_file.rs:553-586); p = None: the default camera (:107-119)"""
    take = (lambda k, d: p.take(k, d)) if p is not None else (lambda k, d: d)
    res = take("res", ("float", f32(1024)))
    width = as_float(take("width", res))
    height = as_float(take("height", ("float", width)))
    fov = f32(as_float(take("fov", ("float", f32(40)))) * f32(np.pi / 180.0))     # f32::to_radians
    position = as_vector(take("position", ("vector", (f32(0), f32(0), f32(0)))))
    lookat = as_vector(take("lookat", ("vector", (f32(0), f32(0), f32(1)))))
    up = as_vector(take("up", ("vector", (f32(0), lookat[2], -lookat[1]))))
    return make_camera(int(width), int(height), float(fov), position, lookat, up)


def load(text, scene_factory, make_camera, asset_root):
    """SceneFile::from_contents (scene_file.rs:59-122).  scene_factory(mode) -> a scene with the construction verbs."""
    paragraphs = parse_paragraphs(text)
    mode = abi.MODE_SCENE
    for p in paragraphs:
        if p.kind == "scene":
            mode = {"Scene": abi.MODE_SCENE, "PhotonScene": abi.MODE_PHOTON}[p.cls]
    scene = scene_factory(mode)
    L = Loader(scene, asset_root)
    cam = None
    for p in paragraphs:
        if p.kind == "light":
            L.light(p)
        elif p.kind == "object":
            scene.add_object(L.object(p))
        elif p.kind == "camera":
            cam = camera_of(p, make_camera)
        elif p.kind != "scene":
            raise ValueError("Invalid paragraph kind: " + p.kind)
    if cam is None:
        cam = camera_of(None, make_camera)
    return scene, cam, mode
