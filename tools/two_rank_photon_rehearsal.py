"""Two ranks SHARING one GPU rehearse the multi-GPU photon path end to end on the device (the real job uses one GPU per
rank and RCCL; two ranks on one device must use gloo, so the arrays are staged through host memory here):
photon index ranges per rank -> all-gather of the pass-1 caustic photons -> pass 2 -> all-gather of both maps -> every
rank builds the index -> every rank renders its row band -> rank 0 compares with a single-rank run, bit for bit.
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
PHOTONS, CAUSTIC, SEED = 200_000, 8_000, 17
params = rrt.default_params(photons_per_light=PHOTONS, caustic_photons_per_light=CAUSTIC)
path = os.path.join(REPO, "assets/scenes/pm_caustics.txt")
scene, cam = rrt.Scene.load(path, params=params, mode_override=abi.MODE_PHOTON)


class Staged(D.DevicePhotonExchange):
    """gloo hands back host tensors: put them on the device before librrt reads them"""

    def set_seeds(self, light, pos, inc, key):
        super().set_seeds(light, pos.cuda(), inc.cuda(), key.cuda())

    def set_records(self, which, pos, inc, rgb, meta):
        super().set_records(which, pos.cuda(), inc.cuda(), rgb.cuda(), meta.cuda())


D.photon_pass(scene, Staged(scene, "cuda"), 1, PHOTONS, CAUSTIC, SEED, rank=rank, world=world, dist=dist, device="cpu")
cam.width, cam.height = 320, 240
y0, y1 = D.row_band(cam.height, rank, world)
rgb, depth, ids = scene.render(cam, y0, y1)
parts = [None] * world
dist.gather_object((rgb, depth, ids), parts if rank == 0 else None, dst=0)
if rank == 0:
    one, _ = rrt.Scene.load(path, params=params, mode_override=abi.MODE_PHOTON)
    one.pre_render(SEED)
    want = one.render(cam)
    for k, name in enumerate(("rgb", "depth", "id")):
        got = np.concatenate([p[k] for p in parts])
        assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32)), name
    for which in (0, 1):
        a, b = one.photon_read(which), scene.photon_read(which)
        assert len(a["key"]) > 0
        for f in a:
            assert np.array_equal(a[f].view(np.uint8), b[f].view(np.uint8)), (which, f)
    print("two ranks == one rank: %d + %d photon records, %dx%d frame in %d bands, bit for bit" %
          (len(one.photon_read(0)["key"]), len(one.photon_read(1)["key"]), cam.width, cam.height, world))
dist.barrier()
dist.destroy_process_group()
