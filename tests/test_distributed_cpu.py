"""CPU suite: the multi-rank plumbing under gloo, world_size 2 (no GPU).

The per-shard searches are produced by the oracle here (the product's kernels
need a GPU); what runs from the product is the partition arithmetic and the
all-gather of the candidate lists, and the gathered layout is checked by merging
with the (distance, global index) rule and comparing with the unsharded answer.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import datagen
import oracle_bind as orc
from poseestimation_amd.distributed import allgather_candidates, block_bounds, frames_of_rank


def test_block_bounds_cover_everything_once():
    for n in (0, 1, 7, 8, 4096, 10_000_001):
        for parts in (1, 2, 3, 8):
            edges = [block_bounds(n, parts, p) for p in range(parts)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(parts - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    assert list(frames_of_rank(10, 4, 1)) == [2, 3, 4]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _merge_top2(all_idx, all_dist, k):
    """Oracle-side merge: smallest (distance, global index) keys over the shards."""
    world, nq, _ = all_idx.shape
    out_i = np.full((nq, k), -1, np.int32)
    out_d = np.full((nq, k), 2147483647, np.int32)
    for q in range(nq):
        cands = [(int(all_dist[s, q, r]), int(all_idx[s, q, r]))
                 for s in range(world) for r in range(k) if all_idx[s, q, r] >= 0]
        cands.sort()
        for r, (d, i) in enumerate(cands[:k]):
            out_i[q, r], out_d[q, r] = i, d
    return out_i, out_d


def _worker(rank, world, port, nq, nt, k, tie_stress, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gen = datagen.codebook_descriptors if tie_stress else datagen.random_descriptors
        q = gen(nq, 5)
        t = gen(nt, 6)
        lo, hi = block_bounds(nt, world, rank)
        idx, dst = orc.knn(q, t[lo:hi], k)
        idx = np.where(idx >= 0, idx + lo, idx).astype(np.int32)   # global indices
        ai, ad = allgather_candidates(torch.from_numpy(idx), torch.from_numpy(dst))
        assert ai.shape == (world, nq, k)
        # shard s of the gathered tensor must be rank s's list
        assert np.array_equal(ai[rank].numpy(), idx) and np.array_equal(ad[rank].numpy(), dst)
        mi, md = _merge_top2(ai.numpy(), ad.numpy(), k)
        ri, rd = orc.knn(q, t, k)
        ret[rank] = bool(np.array_equal(mi, ri) and np.array_equal(md, rd))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tie_stress", [False, True])
def test_train_sharded_allgather_world2(tie_stress):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 150, 401, 2, tie_stress, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _crosscheck_worker(rank, world, port, nq, nt, tie_stress, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if tie_stress:
            q, t = datagen.codebook_descriptors(nq, 7, words=8), datagen.codebook_descriptors(nt, 8, words=8)
        else:
            q = datagen.random_descriptors(nq, 7)
            t = np.concatenate([datagen.perturbed_copy(q, 9)[:nt // 2],
                                datagen.random_descriptors(nt - nt // 2, 10)])
        lo, hi = block_bounds(nt, world, rank)
        # the shard's candidates, as ph_match_crosscheck_hamming256_shard_dev emits them: every
        # query's nearest row within the shard (global index) with the flag "the query is
        # that row's nearest query" packed below the index
        si, sd = orc.knn(q, t[lo:hi], 1)
        ti, _ = orc.knn(t[lo:hi], q, 1)
        flag = (ti[si[:, 0], 0] == np.arange(nq)).astype(np.int32)
        idx = (2 * (si + lo) + flag[:, None]).astype(np.int32)
        dst = sd.astype(np.int32)
        ai, ad = allgather_candidates(torch.from_numpy(idx), torch.from_numpy(dst))
        mi, md = _merge_top2(ai.numpy(), ad.numpy(), 1)
        # ph_match_crosscheck_unpack_dev
        got = [(qi, int(mi[qi, 0]) >> 1, 0, float(md[qi, 0])) for qi in range(nq)
               if mi[qi, 0] >= 0 and mi[qi, 0] & 1]
        ret[rank] = got == orc.match(q, t, cross_check=True) and len(got) > 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tie_stress", [False, True])
def test_crosscheck_train_sharded_world2(tie_stress):
    """Cross-check under train sharding: per-shard candidates (global index + mutual flag)
    all-gathered, folded by (distance, index) and unpacked equal the unsharded cross-check
    match."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_crosscheck_worker, args=(world, port, 180, 333, tie_stress, ret), nprocs=world,
             join=True)
    assert dict(ret) == {0: True, 1: True}
