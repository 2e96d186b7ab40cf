"""The sharded path's collectives over the REAL RCCL backend ("nccl") with a one-rank world - a single GPU cannot host
two RCCL ranks, but one rank exercises what gloo cannot: device-tensor all-gathers between kernels enqueued through the
C ABI (stream ordering), the accept-count exchange in both its forms (Python callback; the library's own ncclAllReduce on a
communicator made by TorchDistComm.rccl_direct, with the step size adapted in the next step's prologue), the reference fit's
moments summed by the library, and the WHOLE sampler through the sharded code path (Comm.force_sharded) against the
single-rank run: same schedule, same log Z, same particles.  Runs tools/nccl_world1.py in a process of its own."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_collectives_over_rccl_one_rank_world():
    for attempt in range(4):  # (a port picked as free can be taken before the rendezvous listens on it: repeat on a fresh one)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_world1.py")], env=env, capture_output=True, text=True,
                           timeout=600)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "nccl world-1 checks ok" in r.stdout
