"""The workloads of bench.py and its extra single-GPU legs.

mesh    configs[3] (bench.py's headline): a 1 M-triangle mesh (13 x 13 grid of assets/models/teapot.obj = 1,068,080
        triangles, generated here deterministically — the reference ships no mesh of that size) at 3840 x 2160,
        Monochrome material, one point light: BVH traversal for primary and shadow rays.  Variants "mirror" and "glass"
        give the teapots reflective / refractive compound materials, so that incoherent secondary rays walk the BVH.
csg     configs[1]: the quadric + CSG + Fresnel scene at 1920 x 1080 (round 1's headline), one and three frames in flight.
photon  configs[2]: assets/scenes/scene2.txt in photon mode, 1 M photons + 10 k caustic, seed 1, 1024^2:
        emission/bounce pass, index build, gather render.
"""
import ctypes as C
import math
import os
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def load_obj_triangles(path):
    """v / vn / f (3 or 4 corners, fan) -> corner positions and normals, as polymesh_object.rs:57-194."""
    verts, normals, pos, nrm = [], [], [], []
    for line in open(path):
        w = line.split()
        if not w:
            continue
        if w[0] == "v":
            verts.append([float(x) for x in w[1:4]])
        elif w[0] == "vn":
            normals.append([float(x) for x in w[1:4]])
        elif w[0] == "f":
            idx = [(int(c.split("/")[0]) - 1, int(c.split("/")[2]) - 1) for c in w[1:]]
            for t in [(0, 1, 2)] + ([(0, 2, 3)] if len(idx) == 4 else []):
                for c in t:
                    pos.append(verts[idx[c][0]])
                    nrm.append(normals[idx[c][1]])
    return np.array(pos, np.float32).reshape(-1, 3, 3), np.array(nrm, np.float32).reshape(-1, 3, 3)


def teapot_grid(nx=13, nz=13, spacing=34.0):
    pos, nrm = load_obj_triangles(os.path.join(REPO, "assets/models/teapot.obj"))
    # the OBJ is z-up: rotate to y-up like scene_file.rs:372-381 (x, z, -y)
    rot = lambda a: np.stack([a[..., 0], a[..., 2], -a[..., 1]], axis=-1)
    pos, nrm = rot(pos), rot(nrm)
    P, N = [], []
    for j in range(nz):
        for i in range(nx):
            off = np.array([(i - (nx - 1) / 2) * spacing, 0.0, (j - (nz - 1) / 2) * spacing], np.float32)
            P.append(pos + off)
            N.append(nrm)
    return np.ascontiguousarray(np.concatenate(P), np.float32), np.ascontiguousarray(np.concatenate(N), np.float32)


def build_mesh_scene(rrt, smooth=True, bvh_builder=3, variant="phong", scene=None):
    """variant: "phong" (BASELINE configs[3]: Monochrome teapots, primary + shadow rays), "nolight" (the same without the
    light: primary rays only), "mirror" (teapots `Simple` = Compound[Monochrome, Global(reflect 0.6)], floor mirror too:
    reflection rays bounce between the teapots and the floor, compound_material.rs:39-49) and "glass" (`Transparent`,
    ior 1.5: reflection + refraction rays, compound_material.rs:51-63): secondary rays that walk the BVH incoherently."""
    abi = rrt._abi
    pos, nrm = teapot_grid()
    # `scene`: any object with the construction verbs of rrt.Scene (the CPU oracle's scene for bench.py's cpu_baseline leg)
    s = scene if scene is not None else rrt.Scene(params=rrt.default_params(bvh_builder=bvh_builder))

    def mono(colour, ambient, shininess):
        m = abi.MaterialDesc()
        m.kind = abi.MAT_MONOCHROME
        m.colour[:] = colour
        m.ambient, m.shininess = ambient, shininess
        m.diffuse_tex = m.normal_tex = -1
        return s.add_material(m)

    def compound(colour, reflect, refract, ior):
        g = abi.MaterialDesc()
        g.kind = abi.MAT_GLOBAL
        g.reflect_weight, g.refract_weight, g.ior = reflect, refract, ior
        g.diffuse_tex = g.normal_tex = -1
        c = abi.MaterialDesc()
        c.kind = abi.MAT_COMPOUND
        c.n_parts = 2
        c.parts[0], c.parts[1] = mono(colour, 0.1, 100.0), s.add_material(g)
        c.diffuse_tex = c.normal_tex = -1
        return s.add_material(c)

    if variant == "mirror":
        mat, floor_mat = compound((0.8, 0.6, 0.2), 0.6, 0.0, 1.0), compound((0.3, 0.5, 0.3), 0.4, 0.0, 1.0)
    elif variant == "glass":
        t = 0.7
        mat, floor_mat = compound((0.8 * (1 - t), 0.6 * (1 - t), 0.2 * (1 - t)), t, t, 1.5), mono((0.3, 0.5, 0.3), 0.1, 100.0)
    else:
        mat, floor_mat = mono((0.8, 0.6, 0.2), 0.1, 50.0), mono((0.3, 0.5, 0.3), 0.1, 100.0)
    d = abi.ObjectDesc()
    d.kind = abi.OBJ_MESH
    d.material = mat
    d.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
    d.normals = nrm.ctypes.data_as(C.POINTER(C.c_float))
    d.n_triangles = pos.shape[0]
    d.smooth = 1 if smooth else 0
    s.add_object(s.object_define(d))
    floor = abi.ObjectDesc()
    floor.kind = abi.OBJ_PLANE
    floor.material = floor_mat
    floor.v[:9] = (0, 0, 0, 0, 0, 1, 0, 1, 0)
    s.add_object(s.object_define(floor))
    if variant != "nolight":
        l = abi.LightDesc()
        l.kind = abi.LIGHT_POINT
        l.position[:] = (0.0, 400.0, -300.0)
        l.colour[:] = (1.2, 1.2, 1.2)
        s.add_light(l)
    cam = rrt.make_camera(3840, 2160, math.radians(40.0), (0.0, 260.0, -420.0), (0.0, 0.55, 1.0))  # file-space lookat: +y looks down (full_camera.rs:31-34)
    return s, cam, pos.shape[0]


def run_mesh(rrt, torch, steps=5, bvh_builder=3):
    t0 = time.perf_counter()
    scene, cam, ntri = build_mesh_scene(rrt, bvh_builder=bvh_builder)
    t_build_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    scene.pre_render(1)
    t_pre = time.perf_counter() - t0
    W, H = cam.width, cam.height
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    depth = torch.empty((H, W), dtype=torch.float32, device="cuda")
    ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), stream)
    torch.cuda.synchronize()
    rays = kernel_ms = flops = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), stream)
        st = scene.stats()
        rays += st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
        kernel_ms += st["ms_trace_kernels"]
        flops += st["flops_algorithmic"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    achieved = flops / (kernel_ms * 1e-3) / 1e12
    ms_3 = frames_in_flight_ms(torch, scene, cam, 9)
    return {
        "workload": "BASELINE configs[3]: %d-triangle mesh (13x13 teapot.obj grid) + floor, 3840x2160, 1 point light" % ntri,
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "ms_per_frame": dt / steps * 1e3,
        "ms_per_frame_3_frames_in_flight": ms_3,
        "value_3_frames_in_flight": (rays / steps) / (ms_3 * 1e-3) / 1e6,
        "rays_per_frame": rays // steps,
        "hit_fraction": float((ids >= 0).float().mean().item()),
        "bvh_nodes_per_ray": st["bvh_nodes_visited"] / (st["rays_primary"] + st["rays_shadow"]),
        "bvh_tris_per_ray": st["bvh_tris_tested"] / (st["rays_primary"] + st["rays_shadow"]),
        "bvh_builder": ("host binned SAH", "device LBVH", "device binned SAH", "auto: binned SAH, on the device for a mesh of this size")[bvh_builder],
        "ms_bvh_build": st["ms_bvh_build"],
        "host_seconds": {"mesh_generation": t_build_host, "pre_render_incl_bvh_build": t_pre},
        "roofline": {"kernel": "k_trace_shade", "bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_TFLOPS, "kernel_ms_per_frame": kernel_ms / steps},
    }


def frames_in_flight_ms(torch, scene, cam, frames, in_flight=3):
    """ms per frame with `in_flight` render calls on ONE scene handle at a time (a host thread and a HIP stream each; the
    library gives concurrent calls their own queues).  The same frame every time: throughput of the render pass."""
    import threading

    W, H = cam.width, cam.height
    dev = torch.cuda.current_device()
    bufs = [(torch.empty((H, W, 3), dtype=torch.float32, device="cuda"), torch.empty((H, W), dtype=torch.float32, device="cuda"),
             torch.empty((H, W), dtype=torch.int32, device="cuda"), torch.cuda.Stream()) for _ in range(in_flight)]
    errors = []

    def work(j, n):
        try:
            torch.cuda.set_device(dev)
            rgb, depth, ids, st = bufs[j]
            for _ in range(n):
                scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), st.cuda_stream)
        except Exception as e:
            errors.append(e)

    def go(n):
        th = [threading.Thread(target=work, args=(j, n)) for j in range(in_flight)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            raise errors[0]

    go(1)  # every pipeline sizes its queues once
    torch.cuda.synchronize()
    per = max(1, frames // in_flight)
    t0 = time.perf_counter()
    go(per)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (per * in_flight) * 1e3


def run_photon(rrt, torch, photons=1_000_000, steps=3):
    params = rrt.default_params(photons_per_light=photons, caustic_photons_per_light=10_000)
    scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/scene2.txt"), params=params)
    scene.pre_render(1)  # warm-up (allocations, code objects)
    W, H = cam.width, cam.height
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    depth = torch.empty((H, W), dtype=torch.float32, device="cuda")
    ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), stream)
    torch.cuda.synchronize()
    t_shoot = t_build = t_render = 0.0
    gathered = cand = queries = stored = 0
    kernel_ms = 0.0
    for i in range(steps):
        t0 = time.perf_counter()
        scene.pre_render(1 + i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), stream)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        st = scene.stats()
        t_shoot += st["ms_photon_shoot"] * 1e-3
        t_build += st["ms_photon_build"] * 1e-3
        t_render += t2 - t1
        kernel_ms += st["ms_trace_kernels"]
        gathered += st["photons_gathered"]
        cand += st["gather_candidates"]
        queries += st["gather_queries"]
        stored = st["photons_stored_regular"] + st["photons_stored_caustic"]
    ms_3 = frames_in_flight_ms(torch, scene, cam, 9)
    emitted = (photons + 10_000) * steps
    # algorithmic bytes of the gather: 36 B per photon inside a query cube + 8 B per touched cell
    gather_bytes = 36.0 * gathered + 8.0 * 8 * queries
    gbs = gather_bytes / (kernel_ms * 1e-3) / 1e9
    return {
        "workload": "BASELINE configs[2]: scene2 photon mode, %d + 10000 photons, seed 1.., 1024x1024" % photons,
        "mphotons_per_s_emitted_trace_pass": emitted / t_shoot / 1e6,
        "mphotons_per_s_gathered_render_pass": gathered / t_render / 1e6,
        "mphotons_per_s_gathered_render_pass_3_frames_in_flight": (gathered / steps) / (ms_3 * 1e-3) / 1e6,
        "ms": {"photon_shoot": t_shoot / steps * 1e3, "index_build": t_build / steps * 1e3, "render": t_render / steps * 1e3,
               "render_kernels": kernel_ms / steps, "render_3_frames_in_flight": ms_3},
        "records_stored": stored,
        "gathered_per_frame": gathered // steps,
        "candidates_read_per_frame": cand // steps,
        "roofline": photon_roofline(gbs, kernel_ms / steps),
    }


def photon_roofline(gbs, kernel_ms_per_frame):
    """The gather is priced in algorithmic bytes against the HBM roof, but the counters say what it really waits for:
    0.1 GB of HBM traffic per launch, waves waiting two thirds of their cycles on dependent L2 loads.  `traffic` is the HBM figure of
    the committed passes (profiles/rNN/c3_photon_pmc_traffic.json), withheld when the kernels no longer take the time they took there."""
    import json
    out = {"kernel": "k_trace_shade (photon mode: trace + gather)", "bound": "l2_latency", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "frac": gbs / PEAK_HBM_GBS, "achieved_is": "algorithmic bytes (36 B per gathered record + 8 B per cell run) over kernel time, against the HBM roof",
           "traffic": None, "kernel_ms_per_frame": kernel_ms_per_frame}
    for rnd in ("r03", "r02"):
        f = os.path.join(REPO, "profiles", rnd, "c3_photon_pmc_traffic.json")
        if os.path.exists(f):
            d = json.load(open(f))
            then = d.get("kernel_ms_per_frame")
            out["counters_from"] = {"file": "profiles/%s/c3_photon_pmc_traffic.json" % rnd, "git": d.get("git"), "kernel_ms_per_frame_then": then}
            if then and abs(kernel_ms_per_frame - then) / then <= 0.05:
                out["traffic"] = d.get("hbm_bytes_per_launch")
                out["traffic_unit"] = "HBM bytes per k_trace_shade launch (two launches per frame)"
                out["wait_fraction_of_wave_cycles"] = d.get("wait_fraction_of_wave_cycles")
            else:
                out["counters_from"]["stale"] = "kernel time differs by more than 5 % (or is unknown): traffic withheld"
            break
    return out


def render_loop(torch, scene, cam, steps, warmup=2):
    """`steps` frames one after the other on the current stream: (ms per frame, stats of the last frame, kernel ms, flops, rays)."""
    W, H = cam.width, cam.height
    rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    depth = torch.empty((H, W), dtype=torch.float32, device="cuda")
    ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), stream)
    torch.cuda.synchronize()
    rays = flops = 0
    kernel_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        scene.render_device(cam, 0, H, rgb.data_ptr(), depth.data_ptr(), ids.data_ptr(), stream)
        st = scene.stats()
        rays += st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
        kernel_ms += st["ms_trace_kernels"]
        flops += st["flops_algorithmic"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / steps * 1e3, st, kernel_ms / steps, flops // steps, rays // steps


def run_mesh_variant(rrt, torch, variant, steps=5):
    """configs[3] with secondary rays: the whole recursion (up to 5 bounces) over the 1 M-triangle BVH."""
    scene, cam, ntri = build_mesh_scene(rrt, variant=variant)
    scene.pre_render(1)
    ms, st, kernel_ms, flops, rays = render_loop(torch, scene, cam, steps)
    traced = st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
    return {
        "workload": "configs[3] with %s teapots (reflection%s rays through the BVH), %d triangles, 3840x2160" % (variant, " + refraction" if variant == "glass" else "", ntri),
        "value": rays / ms / 1e3, "unit": "Mrays/s", "ms_per_frame": ms, "kernel_ms_per_frame": kernel_ms,
        "rays_per_frame": rays, "rays_secondary": st["rays_secondary"], "rays_shadow": st["rays_shadow"],
        "bvh_box_tests_per_ray": st["bvh_nodes_visited"] / traced, "bvh_tris_per_ray": st["bvh_tris_tested"] / traced,
    }


def run_csg(rrt, torch, steps=60):
    """configs[1]: round 1's headline, one frame at a time and three in flight."""
    scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/c2_quadric_csg_fresnel.txt"))
    scene.pre_render(1)
    ms, st, kernel_ms, flops, rays = render_loop(torch, scene, cam, steps, warmup=3)
    ms_3 = frames_in_flight_ms(torch, scene, cam, steps)
    achieved = flops / (kernel_ms * 1e-3) / 1e12
    return {
        "workload": "BASELINE configs[1]: quadric + CSG + Fresnel scene, 1920x1080",
        "value": rays / ms / 1e3, "unit": "Mrays/s", "ms_per_frame": ms,
        "value_3_frames_in_flight": rays / ms_3 / 1e3, "ms_per_frame_3_frames_in_flight": ms_3,
        "rays_per_frame": rays, "rays_secondary": st["rays_secondary"], "rays_shadow": st["rays_shadow"],
        "roofline": {"kernel": "k_trace_shade (six recursion levels of a frame)", "bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS, "kernel_ms_per_frame": kernel_ms, "algorithmic_flops_per_frame": flops},
    }


def run(rrt, torch):
    out = {}
    legs = (("csg_1080p", run_csg), ("photon_1m", run_photon), ("mesh_1m_mirror", lambda r, t: run_mesh_variant(r, t, "mirror")),
            ("mesh_1m_glass", lambda r, t: run_mesh_variant(r, t, "glass")))
    for name, fn in legs:
        try:
            out[name] = fn(rrt, torch)
        except Exception as e:
            out[name] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    import importlib
    import json
    import sys

    import torch

    sys.path.insert(0, REPO)
    rrt = importlib.import_module("rust-raytracer_amd")
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    res = {}
    if which in ("all", "mesh"):
        res["mesh_1m"] = run_mesh(rrt, torch)
    if which in ("all", "mesh_lbvh"):
        res["mesh_1m_device_lbvh"] = run_mesh(rrt, torch, bvh_builder=1)
    if which in ("all", "mesh_hsah"):
        res["mesh_1m_host_sah"] = run_mesh(rrt, torch, bvh_builder=0)
    if which in ("all", "photon"):
        res["photon_1m"] = run_photon(rrt, torch)
    if which in ("all", "csg"):
        res["csg_1080p"] = run_csg(rrt, torch)
    if which in ("all", "mirror", "glass"):
        for v in ("mirror", "glass"):
            if which in ("all", v):
                res["mesh_1m_" + v] = run_mesh_variant(rrt, torch, v)
    print(json.dumps(res, indent=1))
