"""The multi-GPU exchange on the REAL backend (torch.distributed "nccl" = RCCL), as far as one GPU allows: a single-rank process
group whose only rank sends its tile slab to itself with the same batched point-to-point calls TileExchange issues, then stitches.
What this pins: RCCL initialises with device_id, barrier / all_reduce work, and the stream ordering the bench relies on
(tile kernel on the current stream -> RCCL picks the slab up -> wait() -> stitch kernel on the current stream) gives the frame
an unsharded render gives.  The N > 1 wiring itself is covered by the gloo tests (tests/test_dist_gloo.py) and bench.py --rehearse."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_loopback_exchange():
    import subprocess, sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_loopback_check.py")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "LOOPBACK OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
