"""GPU suite: examples/multi_gpu_frames.cpp -- a C++ host dealing frames to several contexts
from its own threads (no Python, no torch.distributed).  The test box has one GPU, so the
three workers share device 0, each through its own context; every frame's k-NN rows and
ratio verdicts are replayed on the CPU oracle."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.mark.gpu
def test_frames_dealt_to_three_contexts_from_cpp_threads(tmp_path):
    import oracle_bind as orc
    exe = os.path.join(ROOT, "examples", "multi_gpu_frames")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    dump = tmp_path / "dump.bin"
    n_frames, rows, db_rows = 25, 700, 20000          # 25 frames over 3 workers: blocks of 8, 8, 9
    out = subprocess.run([exe, "--devices", "0,0,0", "--frames", str(n_frames), "--rows", str(rows),
                          "--db", str(db_rows), str(dump)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    blocks = re.findall(r"worker (\d) device 0 frames (\d+)\.\.(\d+)", out.stdout)
    assert [(int(a), int(b)) for _, a, b in blocks] == [(0, 8), (8, 16), (16, 25)]
    raw = open(dump, "rb").read()
    arrays, pos = [], 0
    for dt in (np.uint8, np.uint8, np.int32, np.int32, np.uint8):
        n = int(np.frombuffer(raw, np.int64, 1, pos)[0])
        pos += 8
        arrays.append(np.frombuffer(raw, dt, n, pos).copy())
        pos += n * np.dtype(dt).itemsize
    assert pos == len(raw)
    db, frames, idx, dist, keep = arrays
    db, frames = db.reshape(db_rows, 32), frames.reshape(n_frames, rows, 32)
    idx, dist = idx.reshape(n_frames, rows, 2), dist.reshape(n_frames, rows, 2)
    keep = keep.reshape(n_frames, rows)
    kept_total = 0
    for f in range(n_frames):
        ri, rd = orc.knn(frames[f], db, 2, threads=8)
        assert np.array_equal(idx[f], ri) and np.array_equal(dist[f], rd), f
        kept = [m[0] for m in orc.ratio_filter(ri, rd, 0.75)]
        assert np.nonzero(keep[f])[0].tolist() == kept, f
        assert (rd[::97, 0] == 0).all()               # the planted copies of database rows
        kept_total += len(kept)
    assert kept_total >= n_frames * 8
    assert int(re.search(r"kept (\d+)", out.stdout).group(1)) == kept_total
