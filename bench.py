#!/usr/bin/env python3
"""bench.py — throughput of the render hot path on MI355X.

Workload (every N): BASELINE.json configs[3], the configuration north_star's target is quoted on — a 1 M-triangle mesh
(13 x 13 grid of assets/models/teapot.obj, 1,068,080 triangles, BVH in HBM) on a floor, one point light, 3840 x 2160.
One "step" = ONE frame: primary rays through the BVH, shadow rays, shading, frame written to HBM on rank 0.  Metric:
Mrays/s = (primary + secondary + shadow rays traced) / wall time, frames rendered one after the other.

N ranks render ONE frame together (north_star: "image tiles shard over GPUs"): its rows are dealt round-robin in groups
of 8 (rrt_render_interleaved_device: contiguous bands would leave the ranks with the sky idle), every rank renders its
share, quantises it on the device to what the reference finally writes (framebuffer.rs:96-130: u8 rgb + u8 depth, 4 bytes per
pixel: 33 MB per 4K frame instead of 166 MB of f32 rgb + depth + id) and ONE RCCL gather per frame brings the shares to rank 0,
which puts the rows back in image order with one kernel (rrt_assemble_frame_device).  Total work is fixed as N grows: "scaling": "strong".  The gather of frame f overlaps the rendering of
frame f + 1 (two buffers); `collective_ms` is that gather timed on its own (after the timed region, five repeats),
`collective_gb_per_s` the bytes rank 0 receives over that time, `collective_exposed_ms` what of it the timed region had to
wait for per step.  At N = 1 there is no collective and the frame is
rendered straight into the frame buffers: the N = 1 value IS the single-GPU number.

The same JSON line carries
  `roofline`      k_trace_shade (the traversal + shading kernel, > 95 % of the GPU time of a frame).  It is neither HBM-
                  nor MFMA-bound: BVH traversal and the reference's triangle test are divergent per-lane FP32 work whose
                  parity-relevant part may not be contracted into FMAs, so the roof is the vector FP32 rate
                  ("valu_fp32").  `achieved` = algorithmic flops (27 per ray/box and 61 per ray/triangle
                  test the traversal performed + the analytic primitives of every trace) / HIP-event time of the launches
                  on their stream; `peak` 157.3 TFLOP/s counts an FMA as two and packed issue, `frac_no_fma` prices the
                  same work against the 39.3 TFLOP/s that plain one-flop-per-lane instructions can reach, `frac_executed`
                  the lane-operations the vector ALUs really issued (PMC, address arithmetic and selects included) against
                  that same roof; `traffic` = HBM bytes per launch from the committed PMC passes (profiles/r02/).
  `cpu_baseline`  the CPU oracle (a port of the reference's brute-force loop over every triangle; the Rust binary cannot
                  be built here) on a bounded sample of the same frame's camera rays, min(16, cpus) threads.
  `extra`         N = 1: the other single-GPU configurations (configs[1] CSG scene, configs[2] photon-mapped scene2,
                  configs[3] with reflective / refractive teapots).  N > 1: `photon_pass`, north_star's second split —
                  photon emission by index range, the RCCL all-gather of the photon maps (bytes, GB/s against the
                  153 GB/s of one xGMI link), per-rank index build, interleaved render + gather (configs[4] scaled:
                  12.5 M photons per rank).
"""
import argparse
import importlib
import json
import math
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_TFLOPS = 157.3   # MI355X vector FP32 (packed FMA), MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP32_NO_FMA = 39.325  # the same lanes issuing one plain (unpacked, unfused) flop per cycle
PEAK_HBM_GBS = 8000.0
XGMI_LINK_GBS = 153.0


def cpu_baseline_mesh(rrt, cam, rays_wanted=2304):
    """The oracle's brute force (every ray against every triangle, polymesh_object.rs:216-242) on camera rays spread evenly
    over the same frame, traced from min(16, cpus) host threads."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import bench_extra
    import oracle_binding as ob

    so = ob.OracleScene()
    bench_extra.build_mesh_scene(rrt, scene=so)
    so.pre_render(1, threads=1)
    step = max(1, int(math.sqrt(cam.width * cam.height / rays_wanted)))
    xs, ys = np.arange(step // 2, cam.width, step), np.arange(step // 2, cam.height, step)
    rays = [rrt.camera_ray(cam, int(x), int(y)) for y in ys for x in xs]
    o = np.ascontiguousarray([r[0] for r in rays], np.float32)
    d = np.ascontiguousarray([r[1] for r in rays], np.float32)
    cores = min(16, os.cpu_count() or 1)
    before = so.stats()
    t0 = time.perf_counter()
    parts = np.array_split(np.arange(len(o)), cores)
    th = [threading.Thread(target=lambda idx=idx: so.raytrace(o[idx], d[idx])) for idx in parts if len(idx)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    after = so.stats()
    traced = sum(after[k] - before[k] for k in ("rays_primary", "rays_secondary", "rays_shadow"))
    return {
        "value": traced / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": "%d camera rays of the same 3840x2160 frame (every %dth pixel in x and y) + their shadow rays = %d rays, brute force over "
                  "1068080 triangles on %d threads, %.1f s" % (len(o), step, traced, cores, dt),
    }


def measured_counters(name="c4_mesh"):
    """The newest committed PMC summary of a workload (profiles/rNN/<name>_pmc_traffic.json, tools/collect_profiles.py): HBM
    bytes and VALU lane-operations per launch, together with the kernel time per launch and the git commit they were measured at,
    so that a bench line can say where its counters come from and drop them when its own kernel no longer takes that long."""
    for rnd in ("r03", "r02"):
        rel = "profiles/%s/%s_pmc_traffic.json" % (rnd, name)
        if os.path.exists(os.path.join(REPO, rel)):
            with open(os.path.join(REPO, rel)) as fh:
                d = json.load(fh)
            d["_path"] = rel
            return d
    return None


def counters_for(per_launch_ms, name="c4_mesh"):
    """(traffic, lane_ops, counters_from) for a roofline object: the counters are reported only if the kernel they were collected
    on ran within 5 % of the time measured now (otherwise the line would mix new timings with old counters)."""
    d = measured_counters(name)
    if d is None:
        return None, None, None
    then = d.get("kernel_ms_per_launch")
    src = {"file": d["_path"], "git": d.get("git"), "kernel_ms_per_launch_then": then}
    if not then or per_launch_ms <= 0 or abs(per_launch_ms - then) / then > 0.05:
        src["stale"] = "kernel time now %.4f ms: more than 5 %% from the run the counters come from (or unknown); traffic and frac_executed withheld" % per_launch_ms
        return None, None, src
    return d.get("hbm_bytes_per_launch"), d.get("valu_lane_ops_executed_per_launch"), src


def photon_pass_leg(rrt, torch, D, dist, rank, world, device, stage_host, photons_per_rank, width, height, interleave):
    """north_star's second split, timed phase by phase: photon emission by index range, the all-gather of the photon maps,
    the index build every rank does for itself, the interleaved render and its gather."""
    abi = rrt._abi
    photons, caustic = photons_per_rank * world, max(1000, photons_per_rank // 500) * world
    params = rrt.default_params(photons_per_light=photons, caustic_photons_per_light=caustic)
    scene, cam = rrt.Scene.load(os.path.join(REPO, "assets/scenes/scene2.txt"), params=params, mode_override=abi.MODE_PHOTON)
    cam.width, cam.height = width, height
    ex = D.StagedPhotonExchange(scene, "cuda") if stage_host else D.DevicePhotonExchange(scene, "cuda")
    info = None
    for seed in (1, 2):  # the first pass sizes buffers and warms RCCL up; the second is reported
        info = D.photon_pass(scene, ex, 1, photons, caustic, seed, rank, world, dist, "cpu" if stage_host else device)
    fg = D.FrameGather(rrt, cam, rank, world, dist, device, interleave=interleave, stage_host=stage_host, wire="u8" if world > 1 else "f32")
    stream = torch.cuda.current_stream().cuda_stream
    times = {}
    for it in range(2):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        fg.render(scene, cam, 0, stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fg.start_gather(0)
        fg.finish(0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        times = {"render_share": (t1 - t0) * 1e3, "gather_frame": (t2 - t1) * 1e3}
    st = scene.stats()
    n_records = scene.photon_count(0) + scene.photon_count(1)
    out = {"ms": dict(info["ms"], **times), "bytes": info["bytes"], "records": n_records, "gathered_this_rank": st["photons_gathered"]}
    return out, photons + caustic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--interleave", type=int, default=8, help="rows per group dealt round-robin to the ranks (0 = contiguous bands, camera.rs:14-31)")
    ap.add_argument("--variant", default="phong", help="mesh scene variant: phong (configs[3]), mirror, glass")
    ap.add_argument("--photons-per-rank", type=int, default=12_500_000, help="N > 1 extra leg: photons emitted per rank (configs[4]: 100 M over 8)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: librrt has no CPU fallback")
    # RRT_BENCH_REHEARSAL=1: run the N > 1 path on a box with fewer GPUs than ranks (ranks share devices; gloo with host
    # staging instead of RCCL, which refuses two ranks on one device).  Checks the sharding / collective plumbing only.
    rehearsal = os.environ.get("RRT_BENCH_REHEARSAL") == "1"
    device_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    reduce_device = "cpu" if rehearsal else device

    rrt = importlib.import_module("rust-raytracer_amd")
    D = importlib.import_module("rust-raytracer_amd.distributed")
    import bench_extra

    scene, cam, ntri = bench_extra.build_mesh_scene(rrt, variant=args.variant)
    t0 = time.perf_counter()
    scene.pre_render(seed=1)
    t_pre = time.perf_counter() - t0
    # N > 1: a rank sends its share as the reference finally writes it (u8 rgb + u8 depth, 4 bytes per pixel); N = 1 renders f32
    fg = D.FrameGather(rrt, cam, rank, world, dist, device, interleave=args.interleave, stage_host=rehearsal, wire="u8" if world > 1 else "f32")
    stream = torch.cuda.current_stream().cuda_stream

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    acc = {"rays": 0, "kernel_ms": 0.0, "flops": 0, "wait_ms": 0.0, "st": None}

    def run_steps(n):
        """n frames, one after the other; the gather of a frame is in flight while the next one renders."""
        for i in range(n):
            k = i % 2
            if i >= 2:
                t = time.perf_counter()
                fg.finish(k)  # buffer k is about to be rendered into again: its gather (frame i - 2) must be done
                acc["wait_ms"] += (time.perf_counter() - t) * 1e3
            fg.render(scene, cam, k, stream)
            st = scene.stats()
            acc["rays"] += st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
            acc["kernel_ms"] += st["ms_trace_kernels"]
            acc["flops"] += st["flops_algorithmic"]
            acc["st"] = st
            fg.start_gather(k)
        t = time.perf_counter()
        for i in range(max(0, n - 2), n):
            fg.finish(i % 2)
        acc["wait_ms"] += (time.perf_counter() - t) * 1e3

    run_steps(2)  # set-up: queues sized, both buffers used once
    run_steps(args.warmup)
    for k in ("rays", "flops"):
        acc[k] = 0
    acc["kernel_ms"] = acc["wait_ms"] = 0.0
    repeated_before = scene.stats()["frames_repeated"]
    sync()
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    repeated = scene.stats()["frames_repeated"] - repeated_before
    # the frame gather by itself (in the timed region it hides behind the next frame's kernels; only what is left of it
    # shows up there as waiting time)
    gather_alone_ms = 0.0
    if world > 1:
        fg.gather_only(0)
        sync()
        t = time.perf_counter()
        for _ in range(5):
            fg.gather_only(0)
        sync()
        gather_alone_ms = (time.perf_counter() - t) / 5 * 1e3
    rays, kernel_ms, flops, wait_ms, st = acc["rays"], acc["kernel_ms"], acc["flops"], acc["wait_ms"], acc["st"]
    if dist is not None:
        t = torch.tensor([dt, kernel_ms], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kernel_ms_max = float(t[0].item()), float(t[1].item())
        r = torch.tensor([rays, flops], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays, total_flops = float(r[0].item()), float(r[1].item())
    else:
        total_rays, total_flops, kernel_ms_max = float(rays), float(flops), kernel_ms

    extra = {}
    if world > 1 and not args.no_extra:
        try:
            leg, emitted = photon_pass_leg(rrt, torch, D, dist, rank, world, device, rehearsal, args.photons_per_rank, cam.width, cam.height, args.interleave)
            ms, by = leg["ms"], leg["bytes"]
            ag = ms.get("allgather_maps", 0.0)
            leg.update({
                "workload": "configs[4] scaled: scene2 photon mode, %d photons per rank (%d emitted in all), %dx%d, %d ranks" % (args.photons_per_rank, emitted, cam.width, cam.height, world),
                "mphotons_per_s_emitted": emitted / (ms["shoot"] * 1e-3) / 1e6 if ms.get("shoot") else None,
                "collective_ms": ag, "collective_bytes": by.get("allgather_maps", 0),
                "collective_gb_per_s": by.get("allgather_maps", 0) / (ag * 1e-3) / 1e9 if ag else None,
                "xgmi_link_gb_per_s": XGMI_LINK_GBS,
            })
            extra["photon_pass"] = leg
        except Exception as e:  # extra legs must never hide the headline number
            extra["photon_pass"] = {"error": repr(e)}

    if rank == 0:
        # this rank's launches (its share of every frame): flops of the share / HIP-event time of the share's launches
        achieved = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        per_launch_ms = kernel_ms / max(1, args.steps)
        traffic, lane_ops, counters_from = counters_for(per_launch_ms) if world == 1 else (None, None, None)
        gather_bytes = fg.bytes_per_frame()
        out = {
            "metric": "Mrays/s (primary+secondary+shadow rays traced)",
            "value": total_rays / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            # the gather of one frame to rank 0 timed on its own, and what of it the timed region had to wait for per step
            "collective_ms": gather_alone_ms if world > 1 else 0.0,
            "collective_gb_per_s": (gather_bytes / (gather_alone_ms * 1e-3) / 1e9) if world > 1 and gather_alone_ms > 0 else None,
            "collective_exposed_ms": wait_ms / args.steps if world > 1 else 0.0,
            "collective_bytes": gather_bytes if world > 1 else 0,
            "config": {
                "workload": "BASELINE configs[3]: %d-triangle mesh (13x13 grid of assets/models/teapot.obj) + floor, 1 point light, 3840x2160, "
                            "%s; one frame per step, frames one at a time" % (ntri, args.variant),
                "rays_per_frame": int(total_rays // max(1, args.steps)),
                "parallelism": ("one GPU" if world == 1 else "rows of each frame dealt to %d ranks in groups of %d, one gather per frame to rank 0 (%d bytes of u8 rgb + depth), "
                                "overlapped with the next frame" % (world, args.interleave, gather_bytes)) + (" (REHEARSAL: ranks share devices, gloo + host staging)" if rehearsal else ""),
                "frames_repeated": repeated,
                "bvh_box_tests_per_ray": st["bvh_nodes_visited"] / max(1, st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]),
                "bvh_tris_per_ray": st["bvh_tris_tested"] / max(1, st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]),
                "pre_render_s": t_pre, "ms_bvh_build": st["ms_bvh_build"],
            },
            "roofline": {
                "kernel": "k_trace_shade<0, 6, 64, true> (traversal + shading; one launch per frame; the instance for scenes of planes and meshes only)",
                "bound": "valu_fp32",
                "achieved": achieved,
                "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS,
                "peak_no_fma": PEAK_FP32_NO_FMA,
                "frac_no_fma": achieved / PEAK_FP32_NO_FMA,
                # what the vector ALUs really issued (SQ_THREAD_CYCLES_VALU: active lanes summed over every VALU instruction,
                # address arithmetic, compares and selects included) against the same one-op-per-lane roof
                "valu_lane_ops_executed_per_launch": lane_ops,
                "frac_executed": lane_ops / (per_launch_ms * 1e-3) / 1e12 / PEAK_FP32_NO_FMA if lane_ops and per_launch_ms > 0 else None,
                "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE + WRITE_SIZE, corrected as the MI355X guide prescribes), from the committed passes named in counters_from",
                "counters_from": counters_from,
                "hbm_gb_per_s": traffic / (per_launch_ms * 1e-3) / 1e9 if traffic and per_launch_ms > 0 else None,
                "hbm_frac": traffic / (per_launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if traffic and per_launch_ms > 0 else None,
                "kernel_ms_per_launch": per_launch_ms,
                "algorithmic_flops_per_launch": int(flops // max(1, args.steps)),
                "pass": "the timed region itself: frames run one at a time, so HIP-event durations add up to GPU time",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_mesh(rrt, cam)
        if world == 1 and not args.no_extra:
            try:
                extra.update(bench_extra.run(rrt, torch))
                ms3 = bench_extra.frames_in_flight_ms(torch, scene, cam, 30)
                out["config"]["value_3_frames_in_flight"] = (total_rays / args.steps) / (ms3 * 1e-3) / 1e6
                # the same scene flattened once more, warm (pre_render_s / ms_bvh_build above include the first use of the
                # builder's kernels in this process)
                scene2, _, _ = bench_extra.build_mesh_scene(rrt, variant=args.variant)
                t0 = time.perf_counter()
                scene2.pre_render(seed=1)
                out["config"]["pre_render_warm_s"] = time.perf_counter() - t0
                out["config"]["ms_bvh_build_warm"] = scene2.stats()["ms_bvh_build"]
                del scene2
            except Exception as e:
                extra["error"] = repr(e)
        if extra:
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
