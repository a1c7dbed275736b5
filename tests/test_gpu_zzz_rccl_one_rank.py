"""The collective calls of the multi-GPU paths on the real backend (RCCL) with one rank on the test box's one GPU
(tools/rccl_one_rank_check.py says what that covers and what it cannot)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_collectives_are_accepted_by_rccl(repo):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, os.path.join(repo, "tools/rccl_one_rank_check.py")], capture_output=True, text=True, timeout=240, cwd=repo,
                         env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "every collective of the N > 1 paths accepted" in out.stdout
