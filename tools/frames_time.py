"""Times F independent nq x nt frames as one batched launch against F single-frame calls."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from poseestimation_amd import _lib
_lib.init(0); lib = _lib.load()
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
F = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
d_q = torch.randint(0, 256, (F * nq, 32), dtype=torch.uint8, device=dev, generator=g)
d_t = torch.randint(0, 256, (F * nt, 32), dtype=torch.uint8, device=dev, generator=g)
idx = torch.empty((F * nq, 2), dtype=torch.int32, device=dev); dist = torch.empty_like(idx)
keep = torch.empty(F * nq, dtype=torch.uint8, device=dev)
idx2 = torch.empty_like(idx); dist2 = torch.empty_like(idx); keep2 = torch.empty_like(keep)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); sp = C.c_void_p(st.cuda_stream)
qo = (np.arange(F + 1) * nq).astype(np.int32); to = (np.arange(F + 1) * nt).astype(np.int32)
i32 = C.POINTER(C.c_int32)
def batched():
    _lib.check(lib.ph_knn_hamming256_frames_dev(d_q.data_ptr(), qo.ctypes.data_as(i32), d_t.data_ptr(), to.ctypes.data_as(i32), F, 2,
                                                idx.data_ptr(), dist.data_ptr(), keep.data_ptr(), 0.75, sp))
def single():
    for f in range(F):
        _lib.check(lib.ph_knn_hamming256_dev(d_q[f * nq:].data_ptr(), nq, d_t[f * nt:].data_ptr(), nt, 2, 0,
                                             idx2[f * nq:].data_ptr(), dist2[f * nq:].data_ptr(), keep2[f * nq:].data_ptr(), 0.75, sp))
for name, fn in (("single", single), ("batched", batched)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 50
    for _ in range(R): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print({"mode": name, "nq": nq, "nt": nt, "frames": F, "us_per_frame": dt / F * 1e6, "evals_per_s": F * nq * nt / dt}, flush=True)
print({"equal": bool(torch.equal(idx, idx2) and torch.equal(dist, dist2) and torch.equal(keep, keep2))})
