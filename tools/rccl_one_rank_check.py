"""The collectives of the N > 1 paths on the real backend ("nccl" = RCCL) with ONE rank on the box's one GPU.

The multi-rank rehearsals share a GPU and therefore go over gloo with host staging; this run checks what they cannot: that every
collective call of rust-raytracer_amd/distributed.py and bench.py is accepted by RCCL for the dtypes, shapes and views used (u8
frame shares gathered into rows of one buffer, f32 / i64 record arrays all-gathered into [world][cap] buffers, f64 reductions),
with device tensors and the asynchronous gather on the process group's own stream.  A world of one moves no data between GPUs.
Run under the environment torch.distributed.run would set (tests/test_gpu_zzz_rccl_one_rank.py does)."""
import importlib
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
D = importlib.import_module("rust-raytracer_amd.distributed")


def main():
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=device)
    world, rank = dist.get_world_size(), dist.get_rank()
    assert (world, rank) == (1, 0)
    # photon exchange: counts, then each record array as one fixed-stride all-gather (distributed.py: photon_pass)
    counts = D.exchange_counts(12345, dist, device)
    assert counts == [12345], counts
    cap = 4096 * 4
    g = torch.Generator(device="cpu").manual_seed(1)
    pos = torch.rand((cap, 4), generator=g).to(device)
    meta = torch.randint(0, 2 ** 62, (cap,), generator=g, dtype=torch.int64).to(device)
    seg_pos, seg_meta = D.all_gather_segments(pos, dist), D.all_gather_segments(meta, dist)
    assert seg_pos.shape == (1, cap, 4) and seg_meta.shape == (1, cap)
    assert torch.equal(seg_pos[0], pos) and torch.equal(seg_meta[0], meta)
    # frame gather: a rank's u8 share into row r of rank 0's buffer, asynchronously (FrameGather.start_gather / finish)
    wlen = 4 * 3840 * 8 * 34
    share = torch.randint(0, 256, (wlen,), generator=g, dtype=torch.uint8).to(device)
    parts = torch.zeros((world, wlen), dtype=torch.uint8, device=device)
    work = dist.gather(share, list(parts.unbind(0)), dst=0, async_op=True)
    work.wait()
    torch.cuda.synchronize()
    assert torch.equal(parts[0], share)
    # the f32 wire of the tests
    share32 = torch.rand((5 * 1024,), generator=g).to(device)
    parts32 = torch.zeros((world, 5 * 1024), dtype=torch.float32, device=device)
    dist.gather(share32, list(parts32.unbind(0)), dst=0, async_op=True).wait()
    torch.cuda.synchronize()
    assert torch.equal(parts32[0], share32)
    # bench.py's reductions and barrier
    t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    r = torch.tensor([3.0, 4.0], dtype=torch.float64, device=device)
    dist.all_reduce(r, op=dist.ReduceOp.SUM)
    dist.barrier()
    torch.cuda.synchronize()
    assert t.tolist() == [1.5, 2.5] and r.tolist() == [3.0, 4.0]
    dist.destroy_process_group()
    print("rccl one rank: every collective of the N > 1 paths accepted (backend %s)" % "nccl")


if __name__ == "__main__":
    main()
