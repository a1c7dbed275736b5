"""Host logic of librrt without a GPU: the C-ABI surface, the scene-file grammar, OBJ / PPM readers,
camera construction.  Anything that computes on the device must fail loudly here (no CPU fallback).
"""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import oracle_binding as ob
import scene_kit as kit

rrt = ob.rrt
abi = rrt._abi


def _gpu():
    import torch

    return torch.cuda.is_available()


def test_library_exports_every_declared_symbol(repo):
    header = open(os.path.join(repo, "include/rrt.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = sorted(set(re.findall(r"\b(rrt_[a-z0-9_]+)\s*\(", header)))
    assert len(names) > 40
    lib = C.CDLL(rrt.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert missing == []
    lib.rrt_version.restype = C.c_char_p
    assert b"gfx950" in lib.rrt_version()


def test_product_does_not_link_the_oracle(repo):
    import subprocess

    out = subprocess.run(["ldd", rrt.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in out
    src = "".join(open(os.path.join(repo, "rust-raytracer_amd", f)).read() for f in ("__init__.py", "distributed.py", "_abi.py"))
    assert "oracle" not in src.replace("the CPU oracle in tests", "").replace("CPU oracle binds it", "")


def test_params_defaults_are_the_reference_constants():
    p = rrt.default_params()
    assert (p.photons_per_light, p.caustic_photons_per_light, p.max_depth) == (5_000_000, 10_000, 5)
    assert p.gather_half_width == np.float32(0.1)


def test_compute_calls_fail_loudly_without_gpu(repo):
    if _gpu():
        pytest.skip("GPU present")
    s, cam = rrt.Scene.load(os.path.join(repo, "assets/scenes/quadratics.txt"))
    with pytest.raises(rrt.RrtError) as e:
        s.pre_render(1)
    assert e.value.code == abi.ERR_DEVICE and "no CPU fallback" in e.value.message
    with pytest.raises(rrt.RrtError) as e:
        s.render(cam)
    assert e.value.code == abi.ERR_STATE


# ---- scene files -----------------------------------------------------------------------------------------
SCENES = ["quadratics", "reflection_refraction", "textures", "scene", "pm_diffuse", "pm_caustics", "scene2", "c2_quadric_csg_fresnel"]
EXPECT_OBJECTS = {"quadratics": 3, "reflection_refraction": 3, "textures": 6, "scene": 7, "pm_diffuse": 5, "pm_caustics": 5, "scene2": 12,
                  "c2_quadric_csg_fresnel": 8}


@pytest.mark.parametrize("name", SCENES)
def test_reference_scene_files_parse(repo, name):
    path = os.path.join(repo, "assets/scenes", name + ".txt")
    s, cam = rrt.Scene.load(path)
    assert s.object_count() == EXPECT_OBJECTS[name]
    o, _ = ob.OracleScene.load(path)
    assert o.object_count() == EXPECT_OBJECTS[name]
    # scene2.txt as committed selects PhotonScene (line 1 is a comment); all others Scene
    assert s.mode == (abi.MODE_PHOTON if name == "scene2" else abi.MODE_SCENE)
    assert o.mode == s.mode
    s2, _ = rrt.Scene.load(path, mode_override=abi.MODE_SCENE)
    assert s2.mode == abi.MODE_SCENE


def test_camera_defaults_and_parity_with_oracle(repo):
    # scene_file.rs:553-586 defaults; FullCamera::new product vs oracle bit for bit
    s, cam = rrt.Scene.from_text("camera Camera\n    position 0.0 3.0 0.5\n    lookat 0.0 0.4 1.0\n")
    assert (cam.width, cam.height) == (1024, 1024)
    assert cam.fov == np.float32(40.0) * (np.float32(math.pi) / np.float32(180.0))
    ref = ob.make_camera(1024, 1024, cam.fov, (0.0, 3.0, 0.5), (0.0, 0.4, 1.0))
    for f in ("lookat", "up", "right", "position"):
        assert list(getattr(cam, f)) == list(getattr(ref, f)), f
    for (x, y) in [(0, 0), (511, 511), (1023, 17), (3, 1000)]:
        o1, d1 = rrt.camera_ray(cam, x, y)
        o2, d2 = ob.camera_ray(ref, x, y)
        assert np.array_equal(d1.view(np.uint32), d2.view(np.uint32)) and np.array_equal(o1, o2)
    # no camera paragraph: scene_file.rs:107-119
    _, cam = rrt.Scene.from_text("scene Scene\n")
    assert list(cam.position) == [0.0, 3.0, 0.0] and (cam.width, cam.height) == (1024, 1024)
    # res / width / height chain
    _, cam = rrt.Scene.from_text("camera Camera\n    res 64\n    height 32\n")
    assert (cam.width, cam.height) == (64, 32)


def parse_error(text):
    with pytest.raises(rrt.RrtError) as e:
        rrt.Scene.from_text(text)
    assert e.value.code == abi.ERR_PARSE
    return e.value.line, e.value.message


def test_parse_errors_carry_the_reference_line_numbers():
    # scene_file.rs:171, :82, :195-199, :220, :239-243, :270, :282, :305, :504, :548, :591, :94, :99, :631
    assert parse_error("scene Scene\nscene Scene\n")[0] == 2                      # no blank line between paragraphs
    assert parse_error("scene Scene\n\nscene PhotonScene\n")[0] == 3              # multiple scenes
    assert "Too many words" in parse_error("object Sphere extra\n")[1]
    assert parse_error("light Point\n    position 1 2 x\n") == (2, "line 2: Invalid float")
    assert "Invalid word count" in parse_error("light Point\n    position 1 2\n")[1]
    assert "Invalid paragraph kind" in parse_error("thing Foo\n")[1]
    assert "Invalid scene class" in parse_error("scene Nope\n")[1]
    assert "Invalid light class" in parse_error("light Spot\n    position 0 0 0\n")[1]
    assert "Invalid object class" in parse_error("object Torus\n    material Monochrome\n        colour 1 1 1\n        shininess 1\n")[1]
    assert "Invalid material class" in parse_error("object Sphere\n    centre 0 0 0\n    radius 1\n    material Shiny\n        colour 1 1 1\n")[1]
    line, msg = parse_error("\n\nobject Sphere\n    centre 0 0 0\n    material Monochrome\n        colour 1 1 1\n        shininess 1\n")
    assert line == 3 and "Missing required attribute: radius" in msg
    assert "Multiple cameras" in parse_error("camera Camera\n    res 8\n\ncamera Camera\n    res 8\n")[1]
    assert "Cannot add material" in parse_error("material Monochrome\n    colour 1 1 1\n    shininess 1\n")[1]
    assert "Unknown colour name" in parse_error("light Point\n    position 0 0 0\n    colour Green\n")[1]
    assert "Invalid attribute value for material" in parse_error("object Sphere\n    centre 0 0 0\n    radius 1\n    material Monochrome\n")[1]
    assert "Started next paragraph" in parse_error("    position 1 2 3\n")[1]
    with pytest.raises(rrt.RrtError) as e:
        rrt.Scene.from_text("camera Camera\n    lookat 0 0 1\n    up 0 0.5 1\n")
    assert "not perpendicular" in e.value.message


def test_grammar_quirks(repo):
    # unknown keys are ignored (quadratics.txt gives a Point light a `direction`); `#` at column 0 ends a paragraph;
    # indented `#` lines are comments; colour accepts Float / Vector / White|Black
    text = (
        "light Point\n    direction 0 1 0\n    position 1 2 3\n    colour 3.0\n"
        "# comment ends the paragraph\n"
        "light Directional\n    # indented comment\n    direction 0.4 -0.6 1.0\n    colour White\n\n"
        "object Sphere\n    centre 0 0 5\n    radius 1\n    material Simple\n        colour 0.5 0 0\n        reflectiveness 0.25\n"
    )
    s, _ = rrt.Scene.from_text(text)
    assert s.object_count() == 1 and rrt.lib.rrt_light_count(s._h) == 2
    # nested sub-paragraph lines also land in the parent's attribute map, last key wins (scene_file.rs:203-253):
    # the Texture's `scale 2.0` overrides the Model's own `scale 0.04`
    quirk = (
        "object Model\n    obj teapot-low.obj\n    scale 0.04\n    translate 0 0 5\n    smooth 1.0\n"
        "    material Texture\n        name mio_block\n        scale 2.0\n        ambient 0.1\n        shininess 10\n"
    )
    o, _ = ob.OracleScene.from_text(quirk)
    # the OBJ teapot is z-up, ~16 units tall: at scale 2 a ray along +x at z = 5 + 2*6 meets its body,
    # at scale 0.04 the whole teapot stays below z = 5.7
    ray = ([(-100, 0, 17)], [(1, 0, 0)])
    assert o.trace(*ray)["object_id"][0] == 0
    plain = quirk.replace("        scale 2.0\n", "        scale 0.04\n")
    o2, _ = ob.OracleScene.from_text(plain)
    assert o2.trace(*ray)["object_id"][0] == abi.MISS
    assert o2.trace([(-100, 0, 5.24)], [(1, 0, 0)])["object_id"][0] == 0


def test_mario_composites_match_scene_file_rs(repo):
    # MarioPipe (scene_file.rs:403-455): cylinder of diameter 1 -> a = h = 4 (Quadratic::cylinder), clipped to y in [0,3]
    o, _ = ob.OracleScene.from_text("object MarioPipe\n    material Monochrome\n        colour 0 1 0\n        shininess 100\n")
    r = o.trace([(-5, 1, 0.3)], [(1, 0, 0)])                          # x = -sqrt(0.25 - 0.09) = -0.4
    assert r["distance"][0] == pytest.approx(4.6, abs=1e-6) and r["prim_id"][0] == 0   # leaf 0 = main cylinder
    # on the axis the cylinder and its clipping box tie at t = 4.5: `L.d > R.d` is false, ADrop, the box hit is emitted
    r = o.trace([(-5, 1, 0)], [(1, 0, 0)])
    assert r["distance"][0] == 4.5 and r["prim_id"][0] == 1
    r = o.trace([(-5, 2.8, 0)], [(1, 0, 0)])
    assert r["distance"][0] == pytest.approx(5 - 0.65, abs=1e-6) and r["prim_id"][0] == 2   # leaf 2 = top cylinder (d=1.3)
    # scale + translate go into ONE matrix for the pipe, TWO for the coin (:440-453 vs :485-500)
    o, _ = ob.OracleScene.from_text("object MarioCoin\n    translate 0 0 5\n    scale 2.0\n    material Monochrome\n        colour 1 1 0\n        shininess 100\n")
    r = o.trace([(0.5, 0, 0)], [(0, 0, 1)])
    # coin = sphere r=0.5 (x) slab |z|<=0.1, scaled by 2 then moved to z=5: front face of the (shallow-cut) coin
    assert r["object_id"][0] == 0 and 4.7 < r["distance"][0] < 5.0


# ---- OBJ / PPM ---------------------------------------------------------------------------------------------
def python_obj_loader(path):
    """Independent restatement of PolyMesh::from_obj_file / parse_face (polymesh_object.rs:34-194)."""
    verts, normals, vnorm, pos, nrm = [], [], {}, [], []
    for line in open(path):
        w = line.split()
        if not w or w[0] == "#":
            continue
        if w[0] == "v":
            verts.append([np.float32(x) for x in w[1:4]])
        elif w[0] == "vn":
            normals.append([np.float32(x) for x in w[1:4]])
        elif w[0] == "f":
            idx = []
            for c in w[1:]:
                p = c.split("/")
                idx.append((int(p[0]) - 1, int(p[2]) - 1 if len(p) > 2 else None))
            tris = [(0, 1, 2)] + ([(0, 2, 3)] if len(idx) == 4 else [])
            for t in tris:
                for c in t:
                    vnorm[idx[c][0]] = idx[c][1]
                for c in t:
                    pos.append(verts[idx[c][0]])
                    nrm.append(normals[vnorm[idx[c][0]]])
    return np.array(pos, np.float32).reshape(-1, 3, 3), np.array(nrm, np.float32).reshape(-1, 3, 3)


@pytest.mark.parametrize("name,ntri", [("teapot-low.obj", 240), ("teapot.obj", 6320)])
def test_obj_loader(repo, name, ntri, tmp_path):
    path = os.path.join(repo, "assets/models", name)
    pos, nrm = python_obj_loader(path)
    assert pos.shape[0] == ntri
    # the product's loader feeding the oracle must give the same triangles as the Python restatement feeding it
    a = ob.OracleScene()
    cam_a = rrt.parse_scene_text("object Model\n    obj %s\n    smooth 1.0\n    material FalseColour\n        _ _\n" % name, repo, a.builder())
    b = ob.OracleScene()
    h = kit.mesh(b, pos, nrm, kit.falsecolour(b), True)
    b.object_transform(h, np.eye(4, dtype=np.float32))  # scene_file.rs:399 always applies the (here identity) matrix: normals renormalise
    b.add_object(h)
    o, d = kit.random_rays(400, 11, origin_scale=40.0, towards=(0, 0, 5), spread=8.0)
    ra, rb = a.trace(o, d), b.trace(o, d)
    assert (ra["object_id"] != abi.MISS).sum() > 20
    for k in ra:
        assert np.array_equal(ra[k].view(np.uint32), rb[k].view(np.uint32)), k


def test_obj_loader_rejects_what_the_reference_panics_on(repo, tmp_path):
    p = tmp_path / "assets" / "models"
    p.mkdir(parents=True)
    (p / "nonormals.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    text = "object Model\n    obj nonormals.obj\n    smooth 1.0\n    material FalseColour\n        _ _\n"
    with pytest.raises(rrt.RrtError) as e:
        rrt.Scene.from_text(text, asset_root=str(tmp_path))
    assert e.value.code == abi.ERR_INVALID
    s, _ = rrt.Scene.from_text(text.replace("smooth 1.0", "smooth 0.0"), asset_root=str(tmp_path))  # flat shading needs no vn
    assert s.object_count() == 1
    with pytest.raises(rrt.RrtError) as e:
        rrt.Scene.from_text(text.replace("nonormals", "missing"), asset_root=str(tmp_path))
    assert e.value.code == abi.ERR_IO


def test_ppm_reader_and_writers(tmp_path):
    # texture.rs:41-103 (P6, 8 and 16 bit) and framebuffer.rs:96-130 (P6 / P5 writers)
    s = rrt.Scene()
    p8 = tmp_path / "a.ppm"
    p8.write_bytes(b"P6\n2 1\n255\n" + bytes([1, 2, 3, 4, 5, 6]))
    assert rrt.check(rrt.lib.rrt_add_texture_ppm(s._h, str(p8).encode())) == 0
    p16 = tmp_path / "b.ppm"
    p16.write_bytes(b"P6 1 1 65535\n" + bytes([0, 1, 2, 3, 4, 5]))
    assert rrt.check(rrt.lib.rrt_add_texture_ppm(s._h, str(p16).encode())) == 1
    bad = tmp_path / "c.ppm"
    bad.write_bytes(b"P5\n1 1\n255\n\0")
    assert rrt.lib.rrt_add_texture_ppm(s._h, str(bad).encode()) == abi.ERR_IO
    short = tmp_path / "d.ppm"
    short.write_bytes(b"P6\n2 2\n255\n\0\0\0")
    assert rrt.lib.rrt_add_texture_ppm(s._h, str(short).encode()) == abi.ERR_IO
    rgb = np.array([[[1.0, 0.999, -0.1], [2.0, np.nan, 0.5]]], np.float32)
    out = tmp_path / "o.ppm"
    rrt.write_rgb_ppm(str(out), rgb)
    assert out.read_bytes() == b"P6\n2 1\n255\n" + bytes([255, 254, 0, 255, 0, 127])
    outd = tmp_path / "o.pgm"
    rrt.write_depth_pgm(str(outd), np.array([[0.0, 0.5, 7.0]], np.float32))
    assert outd.read_bytes() == b"P5\n3 1\n255\n" + bytes([0, 127, 255])
    assert np.array_equal(rrt.quantise(rgb), ob.quantise(rgb))


def test_api_argument_checks():
    s = rrt.Scene()
    with pytest.raises(rrt.RrtError):
        kit.sphere(s, (0, 0, 0), 1.0, 5)  # unknown material
    m = kit.monochrome(s)
    g = kit.global_material(s, 1, 1, 1.5)
    with pytest.raises(rrt.RrtError) as e:
        kit.compound(s, [g, m])  # Global must be last
    assert e.value.code == abi.ERR_LIMIT
    c = kit.compound(s, [m, g])
    with pytest.raises(rrt.RrtError):
        kit.compound(s, [c, g])  # no nested compounds
    a = kit.sphere(s, (0, 0, 0), 1.0, m)
    with pytest.raises(rrt.RrtError):
        kit.csg(s, abi.CSG_UNION, a, a)
    b = kit.sphere(s, (0, 0, 1), 1.0, m)
    u = kit.csg(s, abi.CSG_UNION, a, b)
    with pytest.raises(rrt.RrtError):
        s.add_object(a)  # moved into the CSG node
    assert s.add_object(u) == 0
    tri = np.zeros((1, 3, 3), np.float32)
    with pytest.raises(rrt.RrtError):
        kit.mesh(s, tri, None, m, True)  # smooth without normals: smoothen_hit would panic
    q = kit.quadric(s, (1, 0, 0, 0, 1, 0, 0, 1, 0, -1), m)
    with pytest.raises(rrt.RrtError) as e:
        s.object_transform(q, np.zeros((4, 4), np.float32))
    assert "not invertible" in e.value.message


def test_device_lbvh_functions_pass_the_cpu_preflight(repo, tmp_path):
    """tools/cpu_emu/lbvh_check.cpp runs the bodies of the device BVH build kernels (bvh_lbvh.h) serially under
    AddressSanitizer and checks every tree exhaustively (structure, containment, depth)."""
    import subprocess

    exe = tmp_path / "lbvh_check"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++20", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I" + os.path.join(repo, "tools/cpu_emu"), "-I" + os.path.join(repo, "rust-raytracer_amd/csrc"), "-o", str(exe),
                           os.path.join(repo, "tools/cpu_emu/lbvh_check.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_device_pow_rounds_like_the_oracle_pow(repo, tmp_path):
    """rrt::pow_f32 (csrc/dev_pow.h, the specular power on the device) against the f64 std::pow rounded once to f32 — what
    the oracle computes — on 3 M (base, shininess) pairs and IEEE pow's special cases, bit for bit."""
    import subprocess

    exe = tmp_path / "pow_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(repo, "tools/cpu_emu"),
                           "-I" + os.path.join(repo, "rust-raytracer_amd/csrc"), "-o", str(exe), os.path.join(repo, "tools/cpu_emu/pow_check.cpp")])
    out = subprocess.run([str(exe), "3000000"], capture_output=True, text=True)
    assert out.returncode == 0 and "differ-in-last-bit 0 " in out.stdout and "special-cases-wrong 0" in out.stdout, out.stdout[-2000:]


def test_rust_bindings_cover_the_header(repo):
    """integration/rrt-sys/src/lib.rs (shipped uncompiled: no Rust toolchain in the image) against include/rrt.h: every
    function of the header is declared, no other, and every struct has the header's fields in the header's order."""
    hdr = open(os.path.join(repo, "include/rrt.h")).read()
    rs = open(os.path.join(repo, "integration/rrt-sys/src/lib.rs")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    c_funcs = re.findall(r"\b(rrt_[a-z0-9_]+)\s*\(", hdr_nc)
    rs_funcs = re.findall(r"pub fn (rrt_[a-z0-9_]+)\s*\(", rs)
    assert sorted(set(c_funcs)) == sorted(set(rs_funcs)), set(c_funcs) ^ set(rs_funcs)
    assert len(rs_funcs) == len(set(rs_funcs))
    for name in ("rrt_params", "rrt_material_desc", "rrt_object_desc", "rrt_light_desc", "rrt_camera", "rrt_stats", "rrt_builder"):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr_nc, re.S).group(1)
        c_fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fp = re.match(r".*\(\*(\w+)\)\(", decl, re.S)                  # function pointer member
            if fp:
                c_fields.append(fp.group(1))
                continue
            for part in decl.split(","):                                       # `uint32_t width, height`
                c_fields.append(re.search(r"(\w+)\s*(\[[^\]]*\])?\s*$", part.strip()).group(1))
        rbody = re.search(r"pub struct %s \{(.*?)\n\}" % name, rs, re.S).group(1)
        r_fields = re.findall(r"pub (\w+):", rbody)
        assert c_fields == r_fields, (name, c_fields, r_fields)
    # constants that size arrays
    assert "RRT_MAX_COMPOUND_PARTS: usize = 4" in rs and "#define RRT_MAX_COMPOUND_PARTS 4" in hdr
    assert "RRT_NOALLOC_MAX_RAYS: usize = 65536" in rs and "RRT_GATHER_K_MAX: u32 = 4096" in rs


def test_device_trace_code_in_the_wave_emulator(repo):
    """tools/cpu_emu: the device's trace code (dev_trace.h: record loop, CSG hit lists, the four-wide BVH walk with both LDS
    stack layouts) compiled for the CPU and run in 64-lane lockstep waves under ASan / UBSan with an ownership check on every
    LDS access, against the oracle: generated scenes and a 25 k-triangle mesh (deep tree, threaded SAH build, collapse)."""
    import subprocess
    import sys

    subprocess.check_call(["bash", os.path.join(repo, "tools/cpu_emu/build.sh")])
    out = subprocess.run([sys.executable, os.path.join(repo, "tools/cpu_emu/check_random_scenes.py"), "0", "6"], capture_output=True, text=True,
                         env=dict(os.environ, EMU_TEAPOTS="1"))
    assert out.returncode == 0 and "ALL OK" in out.stdout and "teapots: mesh hits" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    # a 101 120-triangle mesh: its tree is large enough for the threaded four-wide collapse (bvh_build.cpp: top breadth first,
    # subtrees on host threads, spliced) and the sliced triangle records (runtime.cpp)
    out = subprocess.run([sys.executable, os.path.join(repo, "tools/cpu_emu/check_random_scenes.py"), "0", "0"], capture_output=True, text=True,
                         env=dict(os.environ, EMU_TEAPOTS="1", EMU_TEAPOT_GRID="4", EMU_RAYS="1024"))
    assert out.returncode == 0 and "ALL OK" in out.stdout and "teapots: mesh hits" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    # CSG trees with MESH leaves (csg_object.rs:25-36): the BVH walk runs while hit lists are live (stack behind the lists),
    # and a mesh that meets no triangle can still hand the merge a plane's Hit::infinity (seed 2 has such meshes)
    out = subprocess.run([sys.executable, os.path.join(repo, "tools/cpu_emu/check_random_scenes.py"), "0", "4"], capture_output=True, text=True,
                         env=dict(os.environ, EMU_SCENE="csgmesh"))
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("name", SCENES)
def test_scene_files_through_an_independent_reader(repo, name):
    """Common-mode check of the scene plumbing.  The oracle is normally fed by the product's C++ parser, OBJ and PPM readers;
    here every scene file is ALSO read by tests/independent_scene.py (a Python restatement of scene_file.rs with its own
    grammar, builders, OBJ and P6 readers) into a second oracle scene.  Both must look the same to rays, bit for bit: hits
    (object, leaf / triangle, distance, position, normal, uv) on camera and random rays, shadow rays, and — materials,
    textures and lights — the raytraced colours."""
    import independent_scene as ind

    path = os.path.join(repo, "assets/scenes", name + ".txt")
    a, cam_a = ob.OracleScene.load(path, mode_override=abi.MODE_SCENE)
    b, cam_b, mode = ind.load(open(path).read(), lambda m: ob.OracleScene(abi.MODE_SCENE), ob.make_camera, repo)
    assert mode == (abi.MODE_PHOTON if name == "scene2" else abi.MODE_SCENE)
    assert bytes(cam_a) == bytes(cam_b)
    assert a.object_count() == b.object_count() == EXPECT_OBJECTS[name]
    rays = [ob.camera_ray(cam_a, x, y) for y in range(8, cam_a.height, cam_a.height // 20) for x in range(8, cam_a.width, cam_a.width // 20)]
    o = np.array([r[0] for r in rays], np.float32)
    d = np.array([r[1] for r in rays], np.float32)
    o2, d2 = kit.random_rays(300, 3, origin_scale=6.0, towards=(0.0, 1.0, 6.0), spread=5.0)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    ra, rb = a.trace(o, d), b.trace(o, d)
    assert (ra["object_id"] != abi.MISS).mean() > 0.15
    for k in ra:
        assert np.array_equal(ra[k].view(np.uint32), rb[k].view(np.uint32)), k
    lim = np.full(o.shape[0], 3.0, np.float32)
    assert np.array_equal(a.shadowtrace(o, d, lim), b.shadowtrace(o, d, lim))
    ca, cb = a.raytrace(o, d), b.raytrace(o, d)
    for x, y in zip(ca, cb):
        assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32))
