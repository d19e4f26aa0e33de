"""RCCL under `pytest -m gpu`: tod_amd/sharded.py::ShardedMatcher over GpuOps -- real all_gather_into_tensor / all_to_all_single
on torch.distributed's "nccl" backend (RCCL), the comm stream, the event edges of the overlapped choreography -- in a child
process that initialises the process group before anything else touches the GPU (one rank: all a one-GPU box can hold; the
world-2 / world-3 choreography runs over gloo in tests/test_sharded_cpu.py with the same class). Every step's merged matches
== todhip_match_device on the whole DB, 6 consecutive steps (every double buffer is reused), overlapped and serial, both
exchanges; the stress variant delays every producer so that a missing cross-stream edge shows as a wrong result."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
CHILD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sharded_gpu_child.py")


@pytest.mark.parametrize("mode,port", [("plain", 29641), ("stress", 29642)])
def test_sharded_matcher_on_rccl_one_rank(mode, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, CHILD] + ([mode] if mode == "stress" else []), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ok: 24 steps checked" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_the_stress_variant_notices_missing_edges():
    """The same run with every cross-stream wait removed (and the producers still late): the overlapped form must come out wrong,
    or the stress test above proves nothing."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29643", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, CHILD, "broken"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "AssertionError" in p.stderr, p.stdout[-1000:] + p.stderr[-2000:]
