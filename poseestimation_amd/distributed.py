"""One-process-per-GPU partitioning of the matcher (torch.distributed; backend
"nccl" is RCCL on ROCm).

Three modes, none of which exists in the reference (it is single-process):

frame batch   independent frames against one replicated train set: frames are
              dealt to ranks in contiguous blocks and nothing is exchanged.
train shards  one large train set split into contiguous row ranges, one per
              rank; every rank searches all queries in its range (emitting
              GLOBAL train indices), the per-shard k-best lists are all-gathered
              (the one exchange step) and every rank merges them.  Shards are
              gathered in rank order = ascending index order, and the merge key is
              (distance, global index), so OpenCV's lowest-index tie-break holds
              across shards.
cross-check   the same sharding for the cross-check matcher: every rank finds each
              query's nearest row within its shard and whether the query is that row's
              nearest query (the swapped search is local to the shard that owns the row);
              the candidates are all-gathered and folded the same way, and the winner's
              flag decides (match_crosscheck_train_sharded).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


def block_bounds(n, parts, part):
    """Contiguous block `part` of `n` items dealt to `parts` owners: [lo, hi)."""
    if parts < 1 or not 0 <= part < parts:
        raise ValueError("bad partition")
    lo = n * part // parts
    hi = n * (part + 1) // parts
    return lo, hi


def frames_of_rank(n_frames, world, rank):
    """Frame indices owned by `rank` (contiguous block)."""
    lo, hi = block_bounds(n_frames, world, rank)
    return range(lo, hi)


def allgather_candidates(cand_idx, cand_dist, group=None):
    """All-gather of per-shard k-best lists.

    cand_idx, cand_dist: int32 [nq, k] on this rank (global train indices).
    Returns ([world, nq, k] idx, [world, nq, k] dist), shards in rank order, on the
    device of the inputs.  Under RCCL ("nccl") GPU tensors are exchanged in place, in the
    order of torch's current stream.  Any other backend (gloo: ranks that share a GPU, or
    CPU tensors) gets host tensors: GPU candidates are staged through the host, the copy
    out being ordered behind the kernels on torch's current stream.
    """
    world = dist.get_world_size(group)
    if cand_idx.dtype != torch.int32 or cand_dist.dtype != torch.int32:
        raise ValueError("candidate lists must be int32")
    if cand_idx.shape != cand_dist.shape or cand_idx.dim() != 2:
        raise ValueError("candidate lists must both be [nq, k]")
    # one message per rank: idx and dist travel together ([2, nq, k])
    mine = torch.stack([cand_idx, cand_dist]).contiguous()
    nq, k = cand_idx.shape
    dev = mine.device
    staged = dev.type != "cpu" and dist.get_backend(group) != "nccl"
    if staged:
        mine = mine.cpu()  # waits for the current stream's work on `mine`
    gathered = torch.empty((world * 2, nq, k), dtype=torch.int32, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine, group=group)  # concatenates along dim 0
    if staged:
        gathered = gathered.to(dev)
    gathered = gathered.view(world, 2, nq, k)
    return gathered[:, 0].contiguous(), gathered[:, 1].contiguous()


def _gather_shards(idx, dst, group):
    """The exchange step of a train-sharded search.  A process that has no process group
    is a job of one rank: its shard is the whole train set and nothing is exchanged."""
    if dist.is_available() and dist.is_initialized():
        return allgather_candidates(idx, dst, group)
    return idx.unsqueeze(0), dst.unsqueeze(0)


def _stream_ptr():
    s = torch.cuda.current_stream().cuda_stream
    return C.c_void_p(s) if s else None


def knn_train_sharded(d_query, d_train_shard, shard_base, k, ratio=None, group=None,
                      events=None):
    """k-NN of all queries against a train set sharded over the ranks of `group`.

    d_query: uint8 [nq, 32] on this rank's GPU (the same queries on every rank);
    d_train_shard: uint8 [nt_local, 32], rows [shard_base, shard_base + nt_local)
    of the global train set.  Returns (idx [nq, k], dist [nq, k], keep [nq] or
    None) as GPU tensors, identical on every rank.  events: optionally four
    torch.cuda.Event objects, recorded on the current stream before the search, behind
    it, behind the all-gather and behind the merge (a benchmark's phase clock).
    """
    lib = _lib.load()
    nq, nt = d_query.shape[0], d_train_shard.shape[0]
    dev = d_query.device
    idx = torch.empty((nq, k), dtype=torch.int32, device=dev)
    dst = torch.empty((nq, k), dtype=torch.int32, device=dev)
    # kernels run on torch's current stream so the collective that follows is ordered
    # after them without a host sync
    sp = _stream_ptr()
    if sp is None:
        raise RuntimeError("run under `with torch.cuda.stream(torch.cuda.Stream())`: torch's "
                           "default stream has no handle to pass through the C ABI")
    if events:
        events[0].record()
    _lib.check(lib.ph_knn_hamming256_dev(d_query.data_ptr(), nq,
                                         d_train_shard.data_ptr() if nt else None, nt, k,
                                         int(shard_base), idx.data_ptr(), dst.data_ptr(), None,
                                         0.0, sp))
    if events:
        events[1].record()
    all_idx, all_dst = _gather_shards(idx, dst, group)
    if events:
        events[2].record()
    world = all_idx.shape[0]
    out_idx = torch.empty_like(idx)
    out_dst = torch.empty_like(dst)
    keep = torch.empty(nq, dtype=torch.uint8, device=dev) if ratio is not None else None
    _lib.check(lib.ph_knn_merge_dev(all_idx.data_ptr(), all_dst.data_ptr(), world, nq, k,
                                    out_idx.data_ptr(), out_dst.data_ptr(),
                                    keep.data_ptr() if keep is not None else None,
                                    float(ratio) if ratio is not None else 0.0, sp))
    if events:
        events[3].record()
    return out_idx, out_dst, keep


def match_crosscheck_train_sharded(d_query, d_train_shard, shard_base, group=None):
    """Cross-check match (BFMatcher(NORM_HAMMING, crossCheck=True).match) of all queries
    against a train set sharded over the ranks of `group`.

    OpenCV's cross-check keeps a pair (q, t) iff t is q's nearest train row and q is t's
    nearest query.  With the queries replicated, a train row's nearest query is known to
    the shard that owns the row, and a query's nearest row over the whole train set is the
    smallest (distance, global index) among its nearest rows per shard.  So every rank
    emits, per query, its shard's nearest row with a flag "the query is this row's nearest
    query" packed below the GLOBAL index, the lists are all-gathered (the one exchange
    step) and folded by (distance, index) -- the lowest-index rule holds across shards and
    the flag rides along --, and the winner is kept where its flag is set.

    d_query: uint8 [nq, 32] (the same on every rank); d_train_shard: uint8 [nt_local, 32],
    rows [shard_base, shard_base + nt_local) of the global train set.  Returns (idx [nq],
    dist [nq]) GPU tensors, identical on every rank: the matched global train row or -1.
    """
    lib = _lib.load()
    nq, nt = d_query.shape[0], d_train_shard.shape[0]
    dev = d_query.device
    idx = torch.empty((nq, 1), dtype=torch.int32, device=dev)
    dst = torch.empty((nq, 1), dtype=torch.int32, device=dev)
    sp = _stream_ptr()
    if sp is None:
        raise RuntimeError("run under `with torch.cuda.stream(torch.cuda.Stream())`: torch's "
                           "default stream has no handle to pass through the C ABI")
    _lib.check(lib.ph_match_crosscheck_hamming256_shard_dev(
        d_query.data_ptr(), nq, d_train_shard.data_ptr() if nt else None, nt, int(shard_base),
        idx.data_ptr(), dst.data_ptr(), sp))
    all_idx, all_dst = _gather_shards(idx, dst, group)
    out_idx = torch.empty_like(idx)
    out_dst = torch.empty_like(dst)
    _lib.check(lib.ph_knn_merge_dev(all_idx.data_ptr(), all_dst.data_ptr(), all_idx.shape[0], nq,
                                    1, out_idx.data_ptr(), out_dst.data_ptr(), None, 0.0, sp))
    _lib.check(lib.ph_match_crosscheck_unpack_dev(out_idx.data_ptr(), out_dst.data_ptr(), nq, sp))
    return out_idx.view(nq), out_dst.view(nq)


def knn_frame_block_resident(db, d_query, k, ratio=None):
    """This rank's block of a frame-sharded job against a train set resident in HBM (a
    model database uploaded once with ph_db_upload, replicated on every rank): the frames
    of the block are independent, all match the same database, and their query rows lie
    one after the other in d_query, so the whole block is ONE search and nothing is
    exchanged between ranks.

    db: handle from ph_db_upload (int or ctypes.c_void_p); d_query: uint8 [rows, 32] GPU
    tensor, the rows of frames_of_rank(...) concatenated.  Returns (idx [rows, k],
    dist [rows, k], keep [rows] or None); row r belongs to frame r // rows_per_frame of
    the block, train indices are those of the database.
    """
    lib = _lib.load()
    sp = _stream_ptr()
    if sp is None:
        raise RuntimeError("run under `with torch.cuda.stream(torch.cuda.Stream())`: torch's "
                           "default stream has no handle to pass through the C ABI")
    rows, dev = d_query.shape[0], d_query.device
    idx = torch.empty((rows, k), dtype=torch.int32, device=dev)
    dst = torch.empty((rows, k), dtype=torch.int32, device=dev)
    keep = torch.empty(rows, dtype=torch.uint8, device=dev) if ratio is not None else None
    _lib.check(lib.ph_db_knn_hamming256_dev(
        db, d_query.data_ptr() if rows else None, rows, k, idx.data_ptr(), dst.data_ptr(),
        keep.data_ptr() if keep is not None else None,
        float(ratio) if ratio is not None else 0.0, sp))
    return idx, dst, keep


def knn_frames(d_query, q_offsets, d_train, t_offsets, k, ratio=None):
    """k-NN of a batch of independent frames in one launch (this rank's block of a
    frame-sharded job): frame f matches query rows [q_offsets[f], q_offsets[f+1]) of
    d_query against train rows [t_offsets[f], t_offsets[f+1]) of d_train.

    d_query / d_train: uint8 [rows, 32] GPU tensors (may be the same tensor);
    q_offsets / t_offsets: host sequences of n_frames + 1 ascending ints.
    Returns (idx [rows, k], dist [rows, k], keep [rows] or None) indexed by query
    row of d_query; train indices are relative to the frame's first train row.
    Rows outside [q_offsets[0], q_offsets[-1]) are left as (-1, INT_MAX, 0).
    """
    import numpy as np
    lib = _lib.load()
    qo = np.ascontiguousarray(q_offsets, dtype=np.int32)
    to = np.ascontiguousarray(t_offsets, dtype=np.int32)
    if qo.ndim != 1 or qo.shape != to.shape or qo.size < 1:
        raise ValueError("q_offsets and t_offsets must be 1-D and of equal length >= 1")
    if qo.size > 1 and (int(qo[-1]) > d_query.shape[0] or int(to[-1]) > d_train.shape[0]):
        raise ValueError("offsets exceed the descriptor buffers")
    sp = _stream_ptr()
    if sp is None:
        raise RuntimeError("run under `with torch.cuda.stream(torch.cuda.Stream())`: torch's "
                           "default stream has no handle to pass through the C ABI")
    rows, dev = d_query.shape[0], d_query.device
    idx = torch.full((rows, k), -1, dtype=torch.int32, device=dev)
    dst = torch.full((rows, k), 2147483647, dtype=torch.int32, device=dev)
    keep = torch.zeros(rows, dtype=torch.uint8, device=dev) if ratio is not None else None
    i32p = C.POINTER(C.c_int32)
    _lib.check(lib.ph_knn_hamming256_frames_dev(
        d_query.data_ptr(), qo.ctypes.data_as(i32p), d_train.data_ptr(), to.ctypes.data_as(i32p),
        qo.size - 1, k, idx.data_ptr(), dst.data_ptr(),
        keep.data_ptr() if keep is not None else None,
        float(ratio) if ratio is not None else 0.0, sp))
    return idx, dst, keep


def match_crosscheck_frames(d_query, q_offsets, d_train, t_offsets):
    """Cross-check match (BFMatcher(NORM_HAMMING, crossCheck=True).match) of a batch of
    frames in one launch; frames are laid out as for knn_frames.  Returns (idx [rows],
    dist [rows]) indexed by query row: the matched frame-local train row or -1."""
    import numpy as np
    lib = _lib.load()
    qo = np.ascontiguousarray(q_offsets, dtype=np.int32)
    to = np.ascontiguousarray(t_offsets, dtype=np.int32)
    if qo.ndim != 1 or qo.shape != to.shape or qo.size < 1:
        raise ValueError("q_offsets and t_offsets must be 1-D and of equal length >= 1")
    if qo.size > 1 and (int(qo[-1]) > d_query.shape[0] or int(to[-1]) > d_train.shape[0]):
        raise ValueError("offsets exceed the descriptor buffers")
    sp = _stream_ptr()
    if sp is None:
        raise RuntimeError("run under `with torch.cuda.stream(torch.cuda.Stream())`: torch's "
                           "default stream has no handle to pass through the C ABI")
    rows, dev = d_query.shape[0], d_query.device
    idx = torch.full((rows,), -1, dtype=torch.int32, device=dev)
    dst = torch.full((rows,), 2147483647, dtype=torch.int32, device=dev)
    i32p = C.POINTER(C.c_int32)
    _lib.check(lib.ph_match_crosscheck_hamming256_frames_dev(
        d_query.data_ptr(), qo.ctypes.data_as(i32p), d_train.data_ptr(), to.ctypes.data_as(i32p),
        qo.size - 1, idx.data_ptr(), dst.data_ptr(), sp))
    return idx, dst
