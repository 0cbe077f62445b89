"""One rank of a train-sharded job whose ranks SHARE one GPU (gloo group; started once per
rank by tests/test_sharded_ranks_gpu.py, never collected by pytest itself).

Every rank runs poseestimation_amd.distributed.knn_train_sharded and
match_crosscheck_train_sharded themselves -- the product's per-shard kernels, the
all-gather (host-staged under gloo) and the product's merge / unpack kernels -- and
compares what it gets with the oracle's UNSHARDED answer over the whole train set.
Prints one JSON line; exit status 0 iff every case is equal on this rank.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

import datagen
import oracle_bind as orc
from poseestimation_amd import _lib
from poseestimation_amd.distributed import (block_bounds, knn_train_sharded,
                                            match_crosscheck_train_sharded)


def cases():
    """(name, queries, train rows): random rows with true correspondences, a tie-stress
    codebook (the lowest-index rule decides across the shard boundary), a train set of one
    row (rank 0's shard is empty) and a larger set whose shards take the matrix cores."""
    q = datagen.random_descriptors(3000, 201)
    t = np.concatenate([datagen.perturbed_copy(q, 202)[:2500], datagen.random_descriptors(2501, 203)])
    yield "random", q, t
    yield "tie_stress", datagen.codebook_descriptors(2000, 204, words=12), \
        datagen.codebook_descriptors(3333, 205, words=12)
    yield "one_train_row", datagen.random_descriptors(300, 206), datagen.random_descriptors(1, 207)
    q = datagen.random_descriptors(9000, 208)
    t = np.concatenate([datagen.random_descriptors(11000, 209), datagen.perturbed_copy(q, 210)[:4001]])
    yield "large", q, t


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib.init(0)
    lib = _lib.load()
    report, ok = {}, True
    try:
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for path_name, path in (("by_size", 0), ("matrix_cores", 2), ("popcount", 1)):
                _lib.check(lib.ph_knn_set_path(path))
                for name, q, t in cases():
                    nq, nt = q.shape[0], t.shape[0]
                    lo, hi = block_bounds(nt, world, rank)
                    d_q = torch.from_numpy(q).to(dev)
                    d_t = torch.from_numpy(np.ascontiguousarray(t[lo:hi])).to(dev)
                    idx, dst, keep = knn_train_sharded(d_q, d_t, lo, 2, ratio=0.75)
                    cidx, cdst = match_crosscheck_train_sharded(d_q, d_t, lo)
                    stream.synchronize()
                    idx, dst, keep = idx.cpu().numpy(), dst.cpu().numpy(), keep.cpu().numpy()
                    cidx, cdst = cidx.cpu().numpy(), cdst.cpu().numpy()
                    ridx, rdist = orc.knn(q, t, 2, threads=4)
                    kept = [m[0] for m in orc.ratio_filter(ridx, rdist, 0.75)]
                    knn_ok = bool(np.array_equal(idx, ridx) and np.array_equal(dst, rdist)
                                  and np.nonzero(keep)[0].tolist() == kept)
                    got = [(int(i), int(cidx[i]), 0, float(cdst[i])) for i in range(nq) if cidx[i] >= 0]
                    want = orc.match(q, t, cross_check=True)
                    cc_ok = bool(got == want and (cdst[cidx < 0] == 2147483647).all())
                    report[f"{path_name}/{name}"] = {"knn": knn_ok, "crosscheck": cc_ok,
                                                     "matches": len(got), "shard": [lo, hi]}
                    ok = ok and knn_ok and cc_ok
        _lib.check(lib.ph_knn_set_path(0))
        dist.barrier()
    finally:
        dist.destroy_process_group()
    print(json.dumps({"rank": rank, "world_size": world, "backend": "gloo", "ok": ok,
                      "cases": report}), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
