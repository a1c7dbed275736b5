#!/usr/bin/env python3
"""bench.py — throughput of the render hot path on MI355X.

Workload at N=1 (BASELINE.json configs[1]): the quadric + CSG + Fresnel scene
`assets/scenes/c2_quadric_csg_fresnel.txt` at 1920x1080.  One "step" = one full frame through
the wavefront pipeline (primary rays, up to 5 levels of reflection/refraction rays, shadow rays),
scene resident in HBM, outputs written to HBM.  Metric: Mrays/s = (primary + secondary + shadow
rays traced) / wall time.  With N ranks every rank renders its own frame of a camera orbit per step
(frames are independent objects: no data-path collective), so scaling is "weak".

Frames in flight.  The deep recursion levels of a frame hold few rays: their launches cost one wave's latency each and
leave most of the GPU idle, and every frame ends with a host round trip for its counters.  Independent frames fill
those gaps, so a rank drives `--frames-in-flight` (default 3: with torch's own stream that makes the 4 hardware queues a
process has on this pool) frame pipelines -- a host thread and a HIP stream each, all rendering on ONE scene handle
(concurrent render calls on a handle are part of the C ABI, as `raytrace(&self)` is in the reference), exactly what a
host application rendering a camera path would do with this library -- and the K timed steps are dealt round-robin to
them.  `config.value_one_frame_in_flight` is the same measurement with one pipeline.

Extra legs (same JSON line):
  `cpu_baseline`  the CPU oracle (a port of the reference's brute-force algorithm; the Rust binary cannot be
                  built here: no Rust toolchain) rendering whole frames of the same workload for >= 12 s on
                  min(16, cpus) threads, row bands like camera.rs:14-42.
  `roofline`      k_trace_shade, the kernel that holds ~88 % of the GPU time.  It is neither HBM- nor MFMA-bound:
                  ray/primitive intersection is divergent scalar-per-lane FP32 work with IEEE division and
                  square roots that may not be contracted into FMAs (bit-exact parity), so the ceiling that
                  applies is the vector FP32 rate ("valu_fp32", 157.3 TFLOP/s, which counts an FMA as two).
                  achieved = algorithmic flops (per-primitive flops of the reference's brute
                  force loop x rays traced, + 27 per ray/box and 61 per ray/triangle test) / HIP-event time of
                  the launches on their stream (rrt_stats.ms_trace_kernels), taken in a pass of its own that
                  renders the same frames one at a time (overlapping launches share the GPU, so their durations
                  would not add up to GPU time).  `traffic` = HBM bytes per launch
                  from the committed PMC passes (profiles/r01_final/pmc_traffic.json, FETCH_SIZE doubled as
                  the MI355X guide prescribes for 16-B-per-lane reads); `hbm_frac` prices the same launches
                  against 8 TB/s to show how far from the HBM roof the kernel is.
  `extra`         the other two single-GPU configs of BASELINE.json (1M-triangle mesh at 4K, 1M photons),
                  short runs (bench_extra.py).
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def orbit_camera(rrt, base, rank, step):
    """Frame `step` of rank `rank`: the C2 camera moved along x (keeps lookat/up perpendicular)."""
    import ctypes as C

    cam = type(base)()
    C.memmove(C.byref(cam), C.byref(base), C.sizeof(base))
    if rank or step:
        cam.position[0] = base.position[0] + 0.15 * rank + 0.01 * (step % 8)
    return cam


def cpu_baseline(scene_path, cam, min_seconds=12.0):
    """Oracle (CPU port of the reference's brute-force algorithm, threaded by row bands like camera.rs:14-42) on
    whole frames of the same workload until at least `min_seconds` of wall time have passed."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_binding as ob

    so, _ = ob.OracleScene.load(scene_path)
    cores = min(16, os.cpu_count() or 1)
    frames = 0
    t0 = time.perf_counter()
    while True:
        so.render(cam, threads=cores)
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or frames >= 200:
            break
    st = so.stats()
    rays = st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
    return {
        "value": rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d whole 1920x1080 frames of the same scene on %d threads (%d rays, %.1f s)" % (frames, cores, rays, dt),
    }


def measured_traffic(scene_path, width, height):
    """HBM bytes per k_trace_shade launch from the committed PMC passes; only valid for the workload they were taken on."""
    f = os.path.join(REPO, "profiles/r01_final/pmc_traffic.json")
    if not (os.path.exists(f) and scene_path.endswith("c2_quadric_csg_fresnel.txt") and (width, height) == (1920, 1080)):
        return None
    with open(f) as fh:
        return json.load(fh).get("hbm_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="independent frame pipelines per rank (scene handle + host thread + HIP stream each); the steps "
                         "are dealt round-robin to them.  1 = every frame finishes before the next one starts")
    ap.add_argument("--streams", type=int, default=0, help="rrt_params.render_streams (0 = library default)")
    ap.add_argument("--static-camera", action="store_true", help="render the same frame every step")
    ap.add_argument("--scene", default=os.path.join(REPO, "assets/scenes/c2_quadric_csg_fresnel.txt"))
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: librrt has no CPU fallback")
    # RRT_BENCH_REHEARSAL=1: run the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, gloo instead
    # of RCCL, which refuses two ranks on one device).  For checking the launch / sharding / reduction plumbing only.
    rehearsal = os.environ.get("RRT_BENCH_REHEARSAL") == "1"
    device_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
    reduce_device = "cpu" if rehearsal else "cuda"

    rrt = importlib.import_module("rust-raytracer_amd")
    F = max(1, args.frames_in_flight)

    shared_scene, shared_cam = rrt.Scene.load(args.scene, params=rrt.default_params(render_streams=args.streams))
    shared_scene.pre_render(seed=1)

    class Worker:
        """One frame pipeline: a host thread with its own output buffers and HIP stream, rendering on the SHARED scene
        handle (concurrent render calls on one handle each get their own queues inside the library; the flattened scene
        is in HBM once).  rrt_get_stats returns the calling thread's own last frame."""

        def __init__(self, k):
            self.k = k
            self.scene, self.cam0 = shared_scene, shared_cam
            W, H = self.cam0.width, self.cam0.height
            self.rgb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
            self.depth = torch.empty((H, W), dtype=torch.float32, device="cuda")
            self.ids = torch.empty((H, W), dtype=torch.int32, device="cuda")
            self.stream = torch.cuda.Stream() if F > 1 else torch.cuda.current_stream()
            self.reset()

        def reset(self):
            self.rays = self.flops = self.repeated = 0
            self.kernel_ms = 0.0
            self.frames = 0
            self.error = None

        def step(self, i):
            cam = self.cam0 if args.static_camera else orbit_camera(rrt, self.cam0, rank, i)
            self.scene.render_device(cam, 0, cam.height, self.rgb.data_ptr(), self.depth.data_ptr(), self.ids.data_ptr(), self.stream.cuda_stream)
            st = self.scene.stats()
            self.rays += st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
            self.kernel_ms += st["ms_trace_kernels"]
            self.flops += st["flops_algorithmic"]
            self.repeated = st["frames_repeated"]
            self.frames += 1

        def run(self, first, count, stride):  # frames first, first + stride, ... < count
            try:
                torch.cuda.set_device(device_index)  # HIP's current device is per thread; new threads start on device 0
                for i in range(first, count, stride):
                    self.step(i)
            except Exception as e:  # surfaces in the main thread
                self.error = e

    workers = [Worker(k) for k in range(F)]
    cam0 = workers[0].cam0
    W, H = cam0.width, cam0.height

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The pipelines' host threads live for the whole run and wait at a barrier between phases: starting threads inside
    # the timed region would cost about a frame's time each.
    import threading

    phase = {"n": 0, "stop": False}
    gate_in, gate_out = threading.Barrier(F + 1), threading.Barrier(F + 1)

    def pipeline_thread(j):
        torch.cuda.set_device(device_index)
        while True:
            gate_in.wait()
            if phase["stop"]:
                return
            workers[j].run(j, phase["n"], F)
            gate_out.wait()

    threads = [threading.Thread(target=pipeline_thread, args=(j,), daemon=True) for j in range(F)] if F > 1 else []
    for t in threads:
        t.start()

    def run_steps(n, active):
        """n steps (frames 0 .. n-1), dealt round-robin to the pipelines (all of them, or pipeline 0 alone)."""
        if len(active) == 1:
            active[0].run(0, n, 1)
        else:
            phase["n"] = n
            gate_in.wait()
            gate_out.wait()
        for w in active:
            if w.error is not None:
                raise w.error

    # set-up, not warm-up: every pipeline renders one frame on its OWN thread (queue sizes, and whatever HIP sets up the
    # first time a thread launches)
    run_steps(F, workers)
    run_steps(args.warmup, workers)
    for w in workers:
        w.reset()
    sync()
    t0 = time.perf_counter()
    run_steps(args.steps, workers)
    sync()
    dt = time.perf_counter() - t0
    rays = sum(w.rays for w in workers)
    repeated = sum(w.repeated for w in workers)
    st_frames = sum(w.frames for w in workers)
    assert st_frames == args.steps, (st_frames, args.steps)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        r = torch.tensor([rays], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays = float(r.item())
    else:
        total_rays = float(rays)

    # Roofline pass: the same frames ONE AT A TIME on pipeline 0.  Per-launch HIP-event durations only add up to GPU time
    # when launches of different frames do not overlap; with several frames in flight each launch shares the GPU with
    # the neighbours' launches and its duration says nothing about the kernel.
    w0 = workers[0]
    w0.reset()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    w0.run(0, min(args.steps, 40), 1)
    torch.cuda.synchronize()
    dt_one = time.perf_counter() - t1
    if w0.error is not None:
        raise w0.error
    kernel_ms, flops, one_frames, one_rays = w0.kernel_ms, w0.flops, w0.frames, w0.rays
    if threads:
        phase["stop"] = True
        gate_in.wait()
        for t in threads:
            t.join()

    if rank == 0:
        achieved = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        traffic = measured_traffic(args.scene, W, H)
        launches = 6  # recursion levels 0..5 of a frame
        hbm_gbs = traffic * launches / (kernel_ms / max(1, one_frames) * 1e-3) / 1e9 if traffic and kernel_ms > 0 else None
        out = {
            "metric": "Mrays/s (primary+secondary+shadow rays traced)",
            "value": total_rays / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: quadric + CSG + Fresnel scene, 1920x1080, one frame per rank per step",
                "frames_in_flight": F,
                "value_one_frame_in_flight": one_rays / dt_one / 1e6,
                "scene": os.path.relpath(args.scene, REPO),
                "rays_per_frame": rays // max(1, args.steps),
                "parallelism": "frames sharded over %d rank(s), no collective" % world + (" (REHEARSAL: ranks share devices)" if rehearsal else ""),
                "frames_repeated": repeated,
            },
            "roofline": {
                "kernel": "k_trace_shade (all recursion levels of a frame)",
                "bound": "valu_fp32",
                "achieved": achieved,
                "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS,
                "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_final/pmc_traffic.json)",
                "hbm_gb_per_s": hbm_gbs,
                "hbm_frac": hbm_gbs / PEAK_HBM_GBS if hbm_gbs else None,
                "launches_per_frame": launches,
                "kernel_ms_per_frame": kernel_ms / max(1, one_frames),
                "algorithmic_flops_per_frame": flops // max(1, one_frames),
                "pass": "%d of the same frames, one at a time on one pipeline, right after the timed region" % one_frames,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.scene, cam0)
        if world == 1 and not args.no_extra:
            try:
                import bench_extra

                out["extra"] = bench_extra.run(rrt, torch)
            except Exception as e:  # extra legs must never hide the headline number
                out["extra"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
