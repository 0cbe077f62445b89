"""GPU suite: the train-sharded matcher paths through the RCCL backend.  One GPU allows a
one-rank group only; communicator set-up, the all-gather on device tensors and its ordering
against the library's kernels are those of an N-rank job (N ranks run on CPU over gloo in
tests/test_distributed_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_sharded_paths_over_rccl_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29633")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py")], env=env,
                         capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["equal_to_unsharded_calls"] is True and line["world_size"] == 1
