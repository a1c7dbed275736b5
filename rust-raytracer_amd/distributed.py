"""One process per GPU: how the two shardable passes split over ranks.

* render: contiguous row bands, the reference's own thread split (cameras/camera.rs:14-31): no
  data-path collective, rank 0 may gather the bands afterwards;
* photon pass: the photon index range of every (light, pass) is split like
  lights/light.rs:28-40 (n / world each, remainder on the last rank).  Streams are keyed by photon
  index, so the union of the ranks' records equals the single-GPU record set.  Two exchanges:
  the pass-1 caustic photons of each light before pass 2 (small), and both photon maps before the
  index build (the big one) — all-gathers over RCCL/xGMI via torch.distributed ("nccl" backend),
  or over gloo with CPU tensors in tests.

This module is plumbing: it moves opaque [n][4] f32 / [n] u64 arrays between an `exchange`
adapter (device pointers for librrt, host arrays for the CPU oracle in tests) and torch.distributed.
"""
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


def all_gather_variable(tensors, dist, device):
    """All-gather a tuple of per-rank arrays that share dim 0 (different length on every rank).

    Exchange of the lengths, then one padded all_gather per array (ring collectives want equal
    sizes; padding to the maximum costs at most the imbalance between ranks), then trim + concat.
    """
    world = dist.get_world_size()
    n = torch.tensor([tensors[0].shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    out = []
    for t in tensors:
        pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=device)
        pad[: t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out.append(torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0))
    return out, counts


class DevicePhotonExchange:
    """Exchange adapter for librrt: torch CUDA tensors filled / read through device pointers."""

    def __init__(self, scene, device):
        self.scene = scene
        self.device = device

    def _alloc(self, n, cols, dtype):
        shape = (max(n, 1), cols) if cols else (max(n, 1),)
        return torch.empty(shape, dtype=dtype, device=self.device)

    def seeds(self, light):
        n = self.scene.caustic_count(light)
        pos, inc, key = self._alloc(n, 4, torch.float32), self._alloc(n, 4, torch.float32), self._alloc(n, 0, torch.int64)
        self.scene.caustic_export_device(light, pos.data_ptr(), inc.data_ptr(), key.data_ptr())
        torch.cuda.synchronize()
        return pos[:n], inc[:n], key[:n]

    def set_seeds(self, light, pos, inc, key):
        pos, inc, key = pos.contiguous(), inc.contiguous(), key.contiguous()
        self.scene.caustic_import_device(light, pos.data_ptr(), inc.data_ptr(), key.data_ptr(), pos.shape[0])

    def records(self, which):
        n = self.scene.photon_count(which)
        pos, inc, rgb = (self._alloc(n, 4, torch.float32) for _ in range(3))
        meta = self._alloc(n, 0, torch.int64)
        self.scene.photon_export_device(which, pos.data_ptr(), inc.data_ptr(), rgb.data_ptr(), meta.data_ptr())
        torch.cuda.synchronize()
        return pos[:n], inc[:n], rgb[:n], meta[:n]

    def set_records(self, which, pos, inc, rgb, meta):
        pos, inc, rgb, meta = pos.contiguous(), inc.contiguous(), rgb.contiguous(), meta.contiguous()
        self.scene.photon_import_device(which, pos.data_ptr(), inc.data_ptr(), rgb.data_ptr(), meta.data_ptr(), pos.shape[0])


def photon_pass(scene, exchange, n_lights, photons_per_light, caustic_photons_per_light, seed, rank=0, world=1, dist=None, device="cpu"):
    """PhotonScene::build_photon_maps (photon_scene.rs:42-83) split over `world` ranks."""
    scene.photon_begin(seed)
    for light in range(n_lights):
        first, count = shard_range(photons_per_light, rank, world)
        scene.photon_shoot(light, 0, first, count)
        if world > 1:
            (pos, inc, key), _ = all_gather_variable(exchange.seeds(light), dist, device)
            exchange.set_seeds(light, pos, inc, key)
        first, count = shard_range(caustic_photons_per_light, rank, world)
        scene.photon_shoot(light, 1, first, count)
    if world > 1:
        for which in (0, 1):
            (pos, inc, rgb, meta), _ = all_gather_variable(exchange.records(which), dist, device)
            exchange.set_records(which, pos, inc, rgb, meta)
    scene.photon_build()
