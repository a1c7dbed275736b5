"""The N>1 host path on CPU: gloo with 2, 4 and 8 ranks, the oracle standing in for the device.

Exercises rust-raytracer_amd/distributed.py (index-range sharding, the two variable-length
all-gathers into preallocated segment buffers, the row shares of a frame — contiguous bands and
interleaved 8-row groups — and their gather to rank 0) with CPU tensors; on the GPU box the same code
runs over RCCL with DevicePhotonExchange and device buffers.
"""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PHOTONS, N_CAUSTIC, SEED = 2001, 203, 9
# per world size: photons, caustic photons (remainders on the last rank; with 8 ranks some shoot no caustic photon in pass 1)
CASES = {2: (2001, 203), 4: (1203, 101), 8: (171, 37)}


class OracleExchange:
    """Exchange adapter with the oracle's host arrays (tests only)."""

    def __init__(self, scene):
        self.scene = scene

    @staticmethod
    def _t(a):
        return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a)

    @staticmethod
    def _pad(t, cap):
        if cap is None or cap == t.shape[0]:
            return t
        out = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype)
        out[: t.shape[0]] = t
        return out

    def seed_count(self, light):
        return self.scene.caustic_count(light)

    def record_count(self, which):
        return self.scene.photon_count(which)

    def seeds(self, light, cap=None):
        return tuple(self._pad(self._t(a), cap) for a in self.scene.caustic_export(light))

    def set_seeds(self, light, pos, inc, key):
        self.scene.caustic_import(light, pos.numpy(), inc.numpy(), key.numpy().view(np.uint64))

    def records(self, which, cap=None):
        return tuple(self._pad(self._t(a), cap) for a in self.scene.photon_export(which))

    def set_records(self, which, pos, inc, rgb, meta):
        self.scene.photon_import(which, pos.numpy(), inc.numpy(), rgb.numpy(), meta.numpy().view(np.uint64))


def _scene(world=2):
    for p in (REPO, os.path.join(REPO, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_binding as ob

    n_photons, n_caustic = CASES[world]
    params = ob.rrt.default_params(photons_per_light=n_photons, caustic_photons_per_light=n_caustic)
    s, cam = ob.OracleScene.load(os.path.join(REPO, "assets/scenes/pm_caustics.txt"), params=params, mode_override=1)
    cam.width, cam.height = 48, 37
    return ob, s, cam


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ob, s, cam = _scene(world)
        n_photons, n_caustic = CASES[world]
        D = importlib.import_module("rust-raytracer_amd.distributed")
        ex = OracleExchange(s)
        seeds_seen = []
        count0 = ex.seed_count
        ex.seed_count = lambda light: (seeds_seen.append(count0(light)), seeds_seen[-1])[1]  # this rank's pass-1 caustic photons
        info = D.photon_pass(s, ex, 1, n_photons, n_caustic, SEED, rank, world, dist, "cpu")
        assert info["bytes"]["allgather_maps"] > 0 and "allgather_seeds" in info["ms"]
        all_seeds = D.exchange_counts(seeds_seen[0], dist, "cpu")
        frames = {}
        for interleave in (0, 8):   # contiguous bands (camera.rs:14-31), then 8-row groups dealt round-robin
            fg = D.FrameGather(ob.rrt, cam, rank, world, dist, "cpu", interleave=interleave)
            fg8 = D.FrameGather(ob.rrt, cam, rank, world, dist, "cpu", interleave=interleave, wire="u8")
            W, plane = cam.width, fg.plane
            rows = fg.rows[rank]
            # the oracle renders [y0, y1) bands: one call per run of consecutive rows of this rank's share
            runs = np.split(rows, np.where(np.diff(rows) != 1)[0] + 1) if len(rows) else []
            parts = [s.render(cam, int(r[0]), int(r[-1]) + 1, threads=1) for r in runs]
            n = len(rows) * W
            if n:
                rgb, depth, ids = (np.concatenate([p[k] for p in parts]) for k in range(3))
                fg.local[0][: 3 * n] = torch.from_numpy(rgb.reshape(-1))
                fg.local[0][3 * plane: 3 * plane + n] = torch.from_numpy(depth.reshape(-1))
                fg.local[0][4 * plane: 4 * plane + n] = torch.from_numpy(ids.reshape(-1).view(np.float32))
                # the u8 wire: what rrt_quantise_pack_device leaves in the send buffer (r g b d per pixel)
                rgbd = np.concatenate([ob.quantise(rgb).reshape(-1, 3), ob.quantise(depth).reshape(-1, 1)], axis=1)
                fg8.packed[0][: 4 * n] = torch.from_numpy(np.ascontiguousarray(rgbd).reshape(-1))
            fg.start_gather(0)
            out = fg.finish(0)
            fg8.start_gather(0)
            out8 = fg8.finish(0)
            if rank == 0:
                frames["rgb%d" % interleave], frames["depth%d" % interleave] = out[0].numpy().copy(), out[1].numpy().copy()
                frames["ids%d" % interleave] = out[2].numpy().copy()
                frames["rows%d" % interleave] = np.array(fg.n_rows)
                frames["rgb8_%d" % interleave], frames["depth8_%d" % interleave] = out8[0].numpy().copy(), out8[1].numpy().copy()
                frames["wire%d" % interleave] = np.array([fg.bytes_per_frame(), fg8.bytes_per_frame()])
        if rank == 0:
            np.savez(os.path.join(out_dir, "multi.npz"), n0=s.photon_count(0), n1=s.photon_count(1), key0=s.photon_read(0)["key"],
                     pos1=s.photon_read(1)["position"], seeds=np.array(all_seeds), **frames)
    finally:
        dist.destroy_process_group()


def test_shard_range_matches_light_rs():
    D = importlib.import_module("rust-raytracer_amd.distributed")
    # lights/light.rs:28-40: n / threads each, remainder on the last
    assert [D.shard_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 2), (4, 2), (6, 4)]
    assert [D.shard_range(5_000_000, r, 8)[1] for r in range(8)] == [625000] * 8
    assert D.shard_range(3, 0, 1) == (0, 3)
    assert [D.row_band(1080, r, 8) for r in (0, 7)] == [(0, 135), (945, 1080)]
    assert sum(D.shard_range(7, r, 3)[1] for r in range(3)) == 7


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_equal_one_rank(tmp_path, world):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    multi = np.load(os.path.join(str(tmp_path), "multi.npz"))
    ob, s, cam = _scene(world)
    n_photons, n_caustic = CASES[world]
    s.pre_render(SEED, threads=2)
    rgb, depth, ids = s.render(cam, threads=2)
    if world == 2:
        assert list(multi["rows0"]) == [18, 19]  # 37 rows in bands: 18 + (18 + remainder 1)
        assert list(multi["rows8"]) == [21, 16]  # 8-row groups 0, 2, 4 (the last one 5 rows) and 1, 3
    if world == 4:
        assert list(multi["rows0"]) == [9, 9, 9, 10] and list(multi["rows8"]) == [13, 8, 8, 8]
    if world == 8:
        assert list(multi["rows0"]) == [4] * 7 + [9] and list(multi["rows8"]) == [8, 8, 8, 8, 5, 0, 0, 0]  # ranks without a row
        assert n_photons % world and n_caustic % world                    # remainders on the last rank
        assert multi["seeds"].min() == 0 and multi["seeds"].sum() > 0     # a rank that has no pass-1 caustic photon to offer
    assert int(multi["n0"]) == s.photon_count(0) and int(multi["n1"]) == s.photon_count(1) and int(multi["n0"]) > n_photons // 2
    assert np.array_equal(multi["key0"], s.photon_read(0)["key"])
    assert np.array_equal(multi["pos1"].view(np.uint32), s.photon_read(1)["position"].view(np.uint32))
    for k in ("0", "8"):
        assert np.array_equal(multi["depth" + k].view(np.uint32), depth.view(np.uint32))
        assert np.array_equal(multi["ids" + k].view(np.uint32), ids)
        # same photon set, same query order inside the oracle (it sorts by distance): bit-identical frames
        assert np.array_equal(multi["rgb" + k].view(np.uint32), rgb.view(np.uint32))
        # the u8 wire: the frame as FrameBuffer::write_rgb_file / write_depth_file write it, a fifth of the bytes
        assert np.array_equal(multi["rgb8_" + k], ob.quantise(rgb)) and np.array_equal(multi["depth8_" + k], ob.quantise(depth))
        assert multi["wire" + k][1] * 5 == multi["wire" + k][0]
