"""The multi-GPU photon path end to end ON THE DEVICE with two ranks (tools/two_rank_photon_rehearsal.py): the ranks share
the one GPU of the test box, so the exchange goes over gloo with host staging instead of RCCL; everything else is the
real path: index ranges per rank, caustic-seed all-gather, map all-gather, per-rank index build, row bands.  The union
must equal the single-rank run bit for bit (records and frame)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_equal_one_rank_on_device(repo, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "tools/two_rank_photon_rehearsal.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=repo)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "%d ranks == one rank" % world in out.stdout and "bit for bit" in out.stdout


def test_bench_with_two_ranks_measures_both_splits(repo):
    """bench.py --gpus 2 in rehearsal mode (both ranks on the box's one GPU, gloo + host staging instead of RCCL): the row
    shares of the 4K mesh frame with their gather to rank 0, and the photon pass with its all-gather; the JSON line must
    carry the collective figures and the ray total of a whole frame."""
    import json

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RRT_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--photons-per-rank", "200000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=repo, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["steps"] == 4
    assert j["collective_ms"] >= 0.0 and j["collective_gb_per_s"] is not None
    assert 0 < j["collective_bytes"] <= 45_000_000   # u8 rgb + depth on the wire: half a 4K frame at 4 bytes per pixel
    assert 15_000_000 < j["config"]["rays_per_frame"] < 16_500_000          # both shares add up to the whole frame
    p = j["extra"]["photon_pass"]
    assert "error" not in p, p
    assert p["collective_bytes"] > 0 and p["collective_ms"] > 0 and p["records"] > 400000
    for k in ("shoot", "allgather_seeds", "allgather_maps", "import", "build", "render_share", "gather_frame"):
        assert k in p["ms"], k
