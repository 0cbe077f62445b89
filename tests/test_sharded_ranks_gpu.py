"""GPU suite: the train-sharded matcher at world_size 2 on the product's kernels.

Two fresh processes share GPU 0 in a gloo group (one GPU is what a test box has; RCCL needs
one GPU per rank) and each runs distributed.knn_train_sharded and
distributed.match_crosscheck_train_sharded on its own contiguous shard of the train rows:
per-shard search with global indices, the all-gather, the merge.  Both ranks must get the
oracle's unsharded answer, on random and tie-stress data and on all three kernel settings
(tests/sharded_rank_worker.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_sharded_two_ranks_share_the_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                   WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "sharded_rank_worker.py")], env=env,
            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=110))
    finally:
        for p in procs:          # exactly the children started above, nothing by pattern
            if p.poll() is None:
                p.kill()
    for rank, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank}: {out[-2000:]}{err[-2000:]}"
        line = json.loads(out.strip().splitlines()[-1])
        assert line["ok"] is True and line["rank"] == rank and line["world_size"] == world
        assert len(line["cases"]) == 12
        assert all(c["knn"] and c["crosscheck"] for c in line["cases"].values())
        # the cross-check kept something wherever true correspondences exist
        assert line["cases"]["by_size/random"]["matches"] > 1000
