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


def test_two_ranks_equal_one_rank_on_device(repo):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "tools/two_rank_photon_rehearsal.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=repo)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "two ranks == one rank" in out.stdout and "bit for bit" in out.stdout
