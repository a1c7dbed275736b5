"""One process per GPU: how the two shardable passes split over ranks.

* render: the frame's rows are dealt to the ranks — contiguous bands, the reference's own thread split
  (cameras/camera.rs:14-31), or 8-row groups round-robin (`rrt_render_interleaved_device`) so that no rank is left
  with the sky — and gathered to rank 0 on the device (one RCCL gather of one packed buffer per frame);
* photon pass: the photon index range of every (light, pass) is split like lights/light.rs:28-40 (n / world each,
  remainder on the last rank).  Streams are keyed by photon index, so the union of the ranks' records equals the
  single-GPU record set.  Two exchanges: the pass-1 caustic photons of each light before pass 2 (small), and both
  photon maps before the index build (the big one).

Collectives.  Every rank's share has another length.  The lengths go round first (one 8-byte all-gather); then each
array is exported straight into a buffer padded to the longest share and ONE `all_gather_into_tensor` per array fills
one preallocated [world][cap] buffer — every rank's segment lands at its own offset, nothing is concatenated or
copied on the way.  The library then takes the segments where they lie (`rrt_photon_import_device` for the first,
`rrt_photon_append_device` for the rest).  Backend "nccl" is RCCL over xGMI on the GPU box; the same code runs over
gloo with CPU tensors in the tests (host arrays standing in for the device).

This module is plumbing: it moves opaque [n][4] f32 / [n] u64 arrays between an `exchange` adapter and
torch.distributed.
"""
import time

import torch


def shard_range(n, rank, world):
    """lights/light.rs:28-40 / cameras/camera.rs:16-31: n // world each, remainder on the last."""
    per = n // world
    first = per * rank
    count = per + (n % world if rank == world - 1 else 0)
    return first, count


def row_band(height, rank, world):
    first, count = shard_range(height, rank, world)
    return first, first + count


def exchange_counts(n, dist, device):
    """Every rank's share length (one small all-gather; the only host synchronisation of an exchange)."""
    mine = torch.tensor([n], dtype=torch.int64, device=device)
    out = torch.empty(dist.get_world_size(), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine)
    return [int(c) for c in out.tolist()]


def all_gather_segments(padded, dist):
    """padded: this rank's array in a buffer of `cap` rows (the same cap on every rank).  Returns [world][cap] + shape[1:],
    rank r's rows at [r]: one collective into one preallocated buffer."""
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(padded.shape), dtype=padded.dtype, device=padded.device)
    dist.all_gather_into_tensor(out.view((world * padded.shape[0],) + tuple(padded.shape[1:])), padded)
    return out


class Timer:
    """Wall-clock phases of an exchange; synchronises the device around a phase when there is one."""

    def __init__(self, device):
        self.cuda = str(device).startswith("cuda")
        self.ms = {}
        self.bytes = {}

    def _sync(self):
        if self.cuda:
            torch.cuda.synchronize()

    def phase(self, name):
        timer = self

        class _P:
            def __enter__(self):
                timer._sync()
                self.t0 = time.perf_counter()

            def __exit__(self, *exc):
                timer._sync()
                timer.ms[name] = timer.ms.get(name, 0.0) + (time.perf_counter() - self.t0) * 1e3

        return _P()


class DevicePhotonExchange:
    """Exchange adapter for librrt: torch device tensors filled / read through device pointers."""

    def __init__(self, scene, device):
        self.scene = scene
        self.device = device

    def _alloc(self, n, cols, dtype):
        shape = (max(n, 1), cols) if cols else (max(n, 1),)
        return torch.empty(shape, dtype=dtype, device=self.device)

    def seed_count(self, light):
        return self.scene.caustic_count(light)

    def seeds(self, light, cap=None):
        """This rank's pass-1 caustic photons of `light` in buffers of `cap` rows (default: exactly as many as there are)."""
        n = self.scene.caustic_count(light)
        cap = n if cap is None else cap
        pos, inc, key = self._alloc(cap, 4, torch.float32), self._alloc(cap, 4, torch.float32), self._alloc(cap, 0, torch.int64)
        self.scene.caustic_export_device(light, pos.data_ptr(), inc.data_ptr(), key.data_ptr())
        return (pos, inc, key) if cap != n else (pos[:n], inc[:n], key[:n])

    def set_seeds(self, light, pos, inc, key):
        pos, inc, key = pos.contiguous(), inc.contiguous(), key.contiguous()
        self.scene.caustic_import_device(light, pos.data_ptr(), inc.data_ptr(), key.data_ptr(), pos.shape[0])

    def record_count(self, which):
        return self.scene.photon_count(which)

    def records(self, which, cap=None):
        n = self.scene.photon_count(which)
        cap = n if cap is None else cap
        pos, inc, rgb = (self._alloc(cap, 4, torch.float32) for _ in range(3))
        meta = self._alloc(cap, 0, torch.int64)
        self.scene.photon_export_device(which, pos.data_ptr(), inc.data_ptr(), rgb.data_ptr(), meta.data_ptr())
        return (pos, inc, rgb, meta) if cap != n else (pos[:n], inc[:n], rgb[:n], meta[:n])

    def set_records(self, which, pos, inc, rgb, meta):
        pos, inc, rgb, meta = pos.contiguous(), inc.contiguous(), rgb.contiguous(), meta.contiguous()
        self.scene.photon_import_device(which, pos.data_ptr(), inc.data_ptr(), rgb.data_ptr(), meta.data_ptr(), pos.shape[0])

    def set_record_segments(self, which, segments, counts):
        """segments: the four [world][cap] arrays of all_gather_segments; counts: valid rows per rank.  The segments are
        taken where they lie, one device-to-device copy each."""
        pos, inc, rgb, meta = segments
        self.scene.photon_reserve(which, sum(counts))
        first = True
        for r, c in enumerate(counts):
            args = (which, pos[r].data_ptr(), inc[r].data_ptr(), rgb[r].data_ptr(), meta[r].data_ptr(), c)
            if first:
                self.scene.photon_import_device(*args)  # replaces this rank's own records
                first = False
            elif c:
                self.scene.photon_append_device(*args)


class StagedPhotonExchange(DevicePhotonExchange):
    """Rehearsals with several ranks on ONE GPU: RCCL refuses that, gloo wants host tensors, so every array is staged
    through host memory on its way to and from the collective.  Everything else is the real path."""

    def seeds(self, light, cap=None):
        return tuple(t.cpu() for t in super().seeds(light, cap))

    def records(self, which, cap=None):
        return tuple(t.cpu() for t in super().records(which, cap))

    def set_seeds(self, light, pos, inc, key):
        super().set_seeds(light, pos.to(self.device), inc.to(self.device), key.to(self.device))

    def set_records(self, which, pos, inc, rgb, meta):
        super().set_records(which, *(t.to(self.device) for t in (pos, inc, rgb, meta)))

    def set_record_segments(self, which, segments, counts):
        super().set_record_segments(which, tuple(t.to(self.device) for t in segments), counts)


def _set_record_segments(exchange, which, segments, counts):
    if hasattr(exchange, "set_record_segments"):
        exchange.set_record_segments(which, segments, counts)
    else:  # adapters without a segment import (host-array adapters in the tests): trim and join
        exchange.set_records(which, *(torch.cat([a[r, :c] for r, c in enumerate(counts)], dim=0) for a in segments))


def photon_pass(scene, exchange, n_lights, photons_per_light, caustic_photons_per_light, seed, rank=0, world=1, dist=None, device="cpu"):
    """PhotonScene::build_photon_maps (photon_scene.rs:42-83) split over `world` ranks.  Returns the phases' wall-clock
    milliseconds and the bytes each collective delivered to this rank."""
    T = Timer(device)
    with T.phase("begin"):
        scene.photon_begin(seed)
    for light in range(n_lights):
        first, count = shard_range(photons_per_light, rank, world)
        with T.phase("shoot"):
            scene.photon_shoot(light, 0, first, count)
        if world > 1:
            with T.phase("allgather_seeds"):
                counts = exchange_counts(exchange.seed_count(light), dist, device)
                cap = max(max(counts), 1)
                segs = [all_gather_segments(a, dist) for a in exchange.seeds(light, cap)]
                # small (thousands of records): trimmed and joined on the device
                exchange.set_seeds(light, *(torch.cat([a[r, :c] for r, c in enumerate(counts)], dim=0) for a in segs))
                T.bytes["allgather_seeds"] = T.bytes.get("allgather_seeds", 0) + world * cap * 40
        first, count = shard_range(caustic_photons_per_light, rank, world)
        with T.phase("shoot"):
            scene.photon_shoot(light, 1, first, count)
    if world > 1:
        for which in (0, 1):
            with T.phase("allgather_maps"):
                counts = exchange_counts(exchange.record_count(which), dist, device)
                cap = max(max(counts), 1)
                segs = [all_gather_segments(a, dist) for a in exchange.records(which, cap)]
                T.bytes["allgather_maps"] = T.bytes.get("allgather_maps", 0) + world * cap * 56
            with T.phase("import"):
                _set_record_segments(exchange, which, segs, counts)
                del segs
    with T.phase("build"):
        scene.photon_build()
    return {"ms": T.ms, "bytes": T.bytes}


class FrameGather:
    """Row shares of one frame, rendered by every rank into ONE packed buffer and gathered to rank 0, which puts the rows back
    in image order.

    wire = "u8" (what bench.py uses): a rank sends what the reference finally writes, FrameBuffer::write_rgb_file /
    write_depth_file's `(c * 255.0) as u8` (framebuffer.rs:96-130) — 4 bytes r g b d per pixel, quantised and packed on the
    device (rrt_quantise_pack_device), 33 MB for a 4K frame; rank 0 turns the gathered shares into the P6 / P5 payloads with
    ONE kernel that knows the split (rrt_assemble_frame_device), no index arrays, no per-plane copies.
    wire = "f32" (tests: bit-exact frames with object ids): rgb | depth | id planes of f32, 20 bytes per pixel.

    interleave > 0: rank r renders the 8 x interleave row groups r, r + world, ... (rrt_render_interleaved_device);
    interleave == 0: contiguous bands (camera.rs:14-31)."""

    def __init__(self, rrt, cam, rank, world, dist, device, interleave=8, buffers=2, stage_host=False, wire="f32"):
        self.rrt, self.cam, self.rank, self.world, self.dist, self.device = rrt, cam, rank, world, dist, device
        self.interleave = interleave
        assert wire in ("f32", "u8")
        self.wire = wire
        # stage_host: the collective runs on host copies (gloo rehearsals with several ranks on one GPU)
        self.stage_host = bool(stage_host) and world > 1
        self.on_gpu = str(device).startswith("cuda")
        cdev = "cpu" if self.stage_host else device
        H, W = cam.height, cam.width
        if interleave:
            rows = [rrt.interleaved_rows(H, interleave, r, world) for r in range(world)]
        else:
            import numpy as np

            rows = [np.arange(*row_band(H, r, world), dtype=np.int64) for r in range(world)]
        self.rows = rows
        self.n_rows = [len(r) for r in rows]
        self.rows_max = max(self.n_rows)
        self.plane = self.rows_max * W  # elements per plane
        # per buffer: [5][rows_max][W] 4-byte words: rgb as 3 planes' worth of interleaved floats, then depth, then ids
        self.local = [torch.empty(5 * self.plane, dtype=torch.float32, device=device) for _ in range(buffers)]
        words = 5 * self.plane if wire == "f32" else self.plane  # 4-byte words a rank sends per frame
        wdtype = torch.float32 if wire == "f32" else torch.uint8
        wlen = words if wire == "f32" else 4 * words
        self.wire_bytes = 4 * words
        # u8: the share packed r g b d per pixel (the f32 wire sends `local` itself)
        self.packed = [torch.empty(wlen, dtype=torch.uint8, device=device) if wire == "u8" else None for _ in range(buffers)]
        self.stage = [torch.empty(wlen, dtype=wdtype, device="cpu") if self.stage_host else None for _ in range(buffers)]
        self.parts = [torch.empty((world, wlen), dtype=wdtype, device=cdev) if rank == 0 and world > 1 else None for _ in range(buffers)]
        self.pending = [None] * buffers
        if rank == 0:
            if wire == "f32" or world == 1:
                H_rank = torch.empty(H, dtype=torch.int64)   # which rank rendered image row y, and which of its rows it is
                H_row = torch.empty(H, dtype=torch.int64)
                for r, rr in enumerate(rows):
                    H_rank[torch.as_tensor(rr)] = r
                    H_row[torch.as_tensor(rr)] = torch.arange(len(rr))
                self.row_rank, self.row_local = H_rank.to(cdev), H_row.to(cdev)
                self.frame_rgb = torch.empty((H, W, 3), dtype=torch.float32, device=cdev)
                self.frame_depth = torch.empty((H, W), dtype=torch.float32, device=cdev)
                self.frame_ids = torch.empty((H, W), dtype=torch.int32, device=cdev)
            if wire == "u8":  # the frame as FrameBuffer::write_rgb_file / write_depth_file would write it
                self.frame_rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device=device)
                self.frame_depth8 = torch.empty((H, W), dtype=torch.uint8, device=device)
                self.parts_dev = [torch.empty((world, wlen), dtype=torch.uint8, device=device) if self.stage_host else None for _ in range(buffers)]

    def pointers(self, k):
        if self.world == 1 and self.wire == "f32":  # one rank renders straight into the frame
            return self.frame_rgb.data_ptr(), self.frame_depth.data_ptr(), self.frame_ids.data_ptr()
        base = self.local[k].data_ptr()
        return base, base + 3 * self.plane * 4, base + 4 * self.plane * 4

    def render(self, scene, cam, k, stream=None):
        """This rank's share of the frame into buffer k (returns when the share is rendered; the u8 wire's packing kernel
        is queued behind it on the same stream)."""
        rgb, depth, ids = self.pointers(k)
        if self.n_rows[self.rank] == 0:
            return
        if self.interleave:
            scene.render_interleaved_device(cam, self.interleave, self.rank, self.world, rgb, depth, ids, stream)
        else:
            y0, y1 = row_band(cam.height, self.rank, self.world)
            scene.render_device(cam, y0, y1, rgb, depth, ids, stream)
        if self.wire == "u8":
            self.rrt.quantise_pack_device(rgb, depth, self.n_rows[self.rank] * cam.width, self.packed[k].data_ptr(), stream)

    def _send_buffer(self, k):
        return self.local[k] if self.wire == "f32" else self.packed[k]

    def start_gather(self, k):
        """Collective of buffer k (asynchronous: the next frame renders into the other buffer meanwhile)."""
        if self.world > 1:
            outs = list(self.parts[k].unbind(0)) if self.rank == 0 else None
            src = self._send_buffer(k)
            if self.stage_host:
                self.stage[k].copy_(src)
                src = self.stage[k]
            self.pending[k] = self.dist.gather(src, outs, dst=0, async_op=True)

    def gather_only(self, k):
        """The collective of buffer k by itself, complete on return (for timing it; no assembly)."""
        self.start_gather(k)
        if self.pending[k] is not None:
            self.pending[k].wait()
            self.pending[k] = None
        if self.on_gpu:
            torch.cuda.synchronize()

    def finish(self, k):
        """Waits for buffer k's collective; rank 0 then assembles the frame (rows back in image order)."""
        if self.pending[k] is not None:
            self.pending[k].wait()
            self.pending[k] = None
        if self.rank != 0:
            return None
        W, H = self.cam.width, self.cam.height
        if self.wire == "u8":
            parts = self.parts[k] if self.world > 1 else self.packed[k].view(1, -1)
            if self.stage_host:
                self.parts_dev[k].copy_(parts)
                parts = self.parts_dev[k]
            if self.on_gpu:  # one kernel; the split tells it where every row lies
                self.rrt.assemble_frame_device(parts.data_ptr(), self.world, self.rows_max, W, H, self.interleave, self.frame_rgb8.data_ptr(),
                                               self.frame_depth8.data_ptr(), torch.cuda.current_stream().cuda_stream)
            else:  # host tensors standing in for the device (the gloo tests): the same assembly with torch indexing
                px = parts.view(self.world, self.rows_max, W, 4)
                idx_rank, idx_row = self._row_index()
                rows = px[idx_rank, idx_row]
                self.frame_rgb8.copy_(rows[..., :3])
                self.frame_depth8.copy_(rows[..., 3])
            return self.frame_rgb8, self.frame_depth8, None
        if self.world > 1:
            src = self.parts[k]  # [world][5 * plane]: strided views of the three planes, gathered row by row in one pass each
            rgb = src[:, : 3 * self.plane].unflatten(1, (self.rows_max, W * 3))
            depth = src[:, 3 * self.plane: 4 * self.plane].unflatten(1, (self.rows_max, W))
            ids = src[:, 4 * self.plane:].unflatten(1, (self.rows_max, W))
            self.frame_rgb.view(-1, W * 3).copy_(rgb[self.row_rank, self.row_local])
            self.frame_depth.copy_(depth[self.row_rank, self.row_local])
            self.frame_ids.view(torch.float32).copy_(ids[self.row_rank, self.row_local])
        return self.frame_rgb, self.frame_depth, self.frame_ids

    def _row_index(self):
        H = self.cam.height
        idx_rank, idx_row = torch.empty(H, dtype=torch.int64), torch.empty(H, dtype=torch.int64)
        for r, rr in enumerate(self.rows):
            idx_rank[torch.as_tensor(rr)] = r
            idx_row[torch.as_tensor(rr)] = torch.arange(len(rr))
        return idx_rank, idx_row

    def bytes_per_frame(self):
        """Bytes rank 0 receives per frame from the other ranks."""
        return (self.world - 1) * self.wire_bytes
