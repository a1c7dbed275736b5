"""Two ranks SHARING one GPU rehearse the multi-GPU photon path end to end on the device (the real job uses one GPU per
rank and RCCL; two ranks on one device must use gloo, so the arrays are staged through host memory here):
photon index ranges per rank -> all-gather of the pass-1 caustic photons -> pass 2 -> all-gather of both maps -> every
rank builds the index -> every rank renders its share of the frame's rows (bands, then interleaved groups) -> one gather to rank 0, which compares with a single-rank
run, bit for bit.
   torchrun --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/two_rank_photon_rehearsal.py"""
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
rrt = importlib.import_module("rust-raytracer_amd")
D = importlib.import_module("rust-raytracer_amd.distributed")
abi = rrt._abi

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
PHOTONS, CAUSTIC, SEED = 200_003, 8_001, 17   # (remainders on the last rank)
params = rrt.default_params(photons_per_light=PHOTONS, caustic_photons_per_light=CAUSTIC)
path = os.path.join(REPO, "assets/scenes/pm_caustics.txt")
scene, cam = rrt.Scene.load(path, params=params, mode_override=abi.MODE_PHOTON)


info = D.photon_pass(scene, D.StagedPhotonExchange(scene, "cuda"), 1, PHOTONS, CAUSTIC, SEED, rank=rank, world=world, dist=dist, device="cpu")
cam.width, cam.height = 320, 236   # 29.5 groups of 8 rows: ragged shares
frames, frames8 = {}, {}
for interleave in (0, 8):         # contiguous bands, then 8-row groups dealt round-robin (rrt_render_interleaved_device)
    fg = D.FrameGather(rrt, cam, rank, world, dist, "cuda", interleave=interleave, stage_host=True)
    fg.render(scene, cam, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fg.start_gather(0)
    out = fg.finish(0)
    # the u8 wire (what bench.py sends): quantised and packed on the device, put together by one kernel on rank 0
    fg8 = D.FrameGather(rrt, cam, rank, world, dist, "cuda", interleave=interleave, stage_host=True, wire="u8")
    fg8.render(scene, cam, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fg8.start_gather(0)
    out8 = fg8.finish(0)
    torch.cuda.synchronize()
    if rank == 0:
        frames[interleave] = tuple(t.numpy().copy() for t in out)
        frames8[interleave] = (out8[0].cpu().numpy().copy(), out8[1].cpu().numpy().copy())
        assert fg8.bytes_per_frame() * 5 == fg.bytes_per_frame()
if rank == 0:
    one, _ = rrt.Scene.load(path, params=params, mode_override=abi.MODE_PHOTON)
    one.pre_render(SEED)
    want = one.render(cam)
    for interleave, got in frames.items():
        for k, name in enumerate(("rgb", "depth", "id")):
            assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), (interleave, name)
        assert np.array_equal(frames8[interleave][0], rrt.quantise(want[0])), (interleave, "u8 rgb")
        assert np.array_equal(frames8[interleave][1], rrt.quantise(want[1])), (interleave, "u8 depth")
    assert info["bytes"]["allgather_maps"] > 0
    for which in (0, 1):
        a, b = one.photon_read(which), scene.photon_read(which)
        assert len(a["key"]) > 0
        for f in a:
            assert np.array_equal(a[f].view(np.uint8), b[f].view(np.uint8)), (which, f)
    print("%d ranks == one rank: %d + %d photon records, %dx%d frame in %d shares (f32 and u8 wire), bit for bit" %
          (world, len(one.photon_read(0)["key"]), len(one.photon_read(1)["key"]), cam.width, cam.height, world))
dist.barrier()
dist.destroy_process_group()
