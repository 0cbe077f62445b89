"""The train-sharded matcher paths through the RCCL backend ("nccl") in a one-rank group: the
collectives are trivial, but communicator set-up, the all-gather on device tensors and its
ordering against the library's kernels on torch's current stream are the ones an N-rank job
uses.  Prints one JSON line; exit status 0 iff the results equal the unsharded calls."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

from poseestimation_amd import _lib
from poseestimation_amd.distributed import knn_train_sharded, match_crosscheck_train_sharded

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29631")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
_lib.init(0)
lib = _lib.load()
g = torch.Generator(device=dev)
g.manual_seed(5)
nq, nt = 20000, 30000
d_q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev, generator=g)
d_t = torch.randint(0, 256, (nt, 32), dtype=torch.uint8, device=dev, generator=g)
d_t[:5000] = d_q[:5000]          # exact copies: the cross-check has matches to keep
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    dist.barrier()
    el = torch.tensor([1.25], dtype=torch.float64, device=dev)   # the benchmark's MAX over ranks
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    assert float(el.item()) == 1.25
    idx, dst, keep = knn_train_sharded(d_q, d_t, 0, 2, ratio=0.75)
    cidx, cdst = match_crosscheck_train_sharded(d_q, d_t, 0)
    stream.synchronize()
t_init = time.perf_counter() - t0
ri = torch.empty((nq, 2), dtype=torch.int32, device=dev)
rd = torch.empty_like(ri)
rk = torch.empty(nq, dtype=torch.uint8, device=dev)
xi = torch.empty(nq, dtype=torch.int32, device=dev)
xd = torch.empty_like(xi)
torch.cuda.synchronize()
_lib.check(lib.ph_knn_hamming256_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 2, 0, ri.data_ptr(), rd.data_ptr(),
                                     rk.data_ptr(), 0.75, None))
_lib.check(lib.ph_match_crosscheck_hamming256_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, xi.data_ptr(),
                                                  xd.data_ptr(), None))
_lib.check(lib.ph_stream_synchronize(None))
ok = bool(torch.equal(idx, ri) and torch.equal(dst, rd) and torch.equal(keep, rk)
          and torch.equal(cidx, xi) and torch.equal(cdst, xd) and int((xi >= 0).sum()) >= 5000)
dist.destroy_process_group()
print(json.dumps({"backend": "nccl (RCCL)", "world_size": 1, "nq": nq, "nt": nt, "seconds_incl_init": t_init,
                  "equal_to_unsharded_calls": ok}))
sys.exit(0 if ok else 1)
