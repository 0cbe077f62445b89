#!/usr/bin/env python3
"""Headline benchmark: Hamming evals/s of the k=2 + ratio-test matcher on MI355X.

Workload at every N (weak scaling): per rank, a step is one pass over a batch of
`--frames-per-step` independent frames; each frame is ONE 10k x 10k
`knnMatch(k=2)` + Lowe ratio test problem (BASELINE.json configs[1]) with its own
query and train descriptors, inputs resident in HBM; the batch goes down as one
`ph_knn_hamming256_frames_dev` launch.  Frames are independent, so ranks share
nothing and no collective is on the data path.  The same frames issued as one
`ph_knn_hamming256_dev` call each are timed beside it (`per_call`), and so is the same
step on the popcount kernel (`popcount_kernel`) when the matrix cores ran it.

`python bench.py --gpus N` started bare (no WORLD_SIZE in the environment) launches the N
ranks itself, as child processes of a parent that never touches the GPU; under
torch.distributed.run it is one of the ranks.  Beside the frame-sharded headline every
run also times BASELINE configs[3] as the frame-sharded job deals it (512 frames x 2000
descriptors per rank against a resident 50k-row database: `frame_sharded_database`) and
the train-sharded form of the matcher (BASELINE configs[4] scaled to the
rank count: queries replicated, train rows in contiguous shards, one all-gather of the
per-shard candidates, merge) and reports it as `train_sharded`.

`value` comes from EXACTLY --steps timed steps after the warm-up (at least --warmup steps
and 50 ms, past the clock ramp).  When those steps are less than --min-seconds of work, a
second, longer region (the same count on every rank) is timed as well and reported as
`sustained`.

Prints one JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--nt", type=int, default=10000)
    ap.add_argument("--frames-per-step", type=int, default=64)
    ap.add_argument("--ratio", type=float, default=0.75)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--ransac-hyps", type=int, default=4096)
    ap.add_argument("--ransac-points", type=int, default=2000)
    ap.add_argument("--ransac-problems", type=int, default=64,
                    help="independent RANSAC problems (frames) per step per GPU, one batched call; "
                         "the default equals --frames-per-step: a step is 64 frames either way")
    ap.add_argument("--ransac-steps", type=int, default=10)
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, the default) or gloo (rehearsing N ranks on fewer GPUs)")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="length of the sustained region timed beside the exact --steps region")
    ap.add_argument("--min-warmup-seconds", type=float, default=0.05)
    ap.add_argument("--shard-nq", type=int, default=1_000_000,
                    help="train-sharded leg: queries (replicated on every rank)")
    ap.add_argument("--shard-nt", type=int, default=1_250_000,
                    help="train-sharded leg: train rows per rank (BASELINE configs[4]: 10M / 8)")
    ap.add_argument("--shard-steps", type=int, default=3)
    ap.add_argument("--db-frames", type=int, default=512,
                    help="frame-sharded leg: frames per rank (BASELINE configs[3]: 4096 / 8)")
    ap.add_argument("--db-rows-per-frame", type=int, default=2000)
    ap.add_argument("--db-rows", type=int, default=50_000,
                    help="frame-sharded leg: rows of the model database resident on every rank")
    ap.add_argument("--db-steps", type=int, default=5)
    return ap.parse_args()


def visible_gpu_count():
    """GPUs this process could open, counted WITHOUT the HIP runtime (the parent of the ranks
    must stay GPU-free): compute nodes with SIMDs in the KFD topology, capped by the
    *_VISIBLE_DEVICES lists."""
    import glob
    n = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(prop) as f:
                for line in f:
                    if line.startswith("simd_count ") and int(line.split()[1]) > 0:
                        n += 1
        except (OSError, ValueError):
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def launch_ranks(args):
    """`bench.py --gpus N` started bare: start the N ranks as children (one process per GPU,
    torch.distributed.run on 127.0.0.1) from this process, which has made no GPU call and
    makes none (it neither imports torch nor loads the HIP runtime), and pass their output
    and exit code on."""
    have = visible_gpu_count()
    if args.dist_backend == "nccl" and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def all_reduce_max(value, torch, dist, dev, backend):
    """MAX over ranks of a host float (device tensor under RCCL, host tensor under gloo)."""
    t = torch.tensor([value], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_region(step, steps, warmup, min_seconds, min_warmup_seconds, torch, dist, dev, world,
                 backend):
    """Warm up (at least `warmup` steps and `min_warmup_seconds`), then time EXACTLY `steps`
    steps between barrier + synchronize on both sides, MAX over ranks: that region is the
    headline.  When it is shorter than `min_seconds`, a second, longer region of
    proportionally more steps (every rank derives the same count from the all-reduced time)
    is timed as well and reported beside it as the sustained rate.
    Returns (seconds, steps) of the exact region and (seconds, steps) of the longest one."""
    def barrier():
        if world > 1:
            dist.barrier()

    def run(n):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return all_reduce_max(el, torch, dist, dev, backend) if world > 1 else el

    done, w0 = 0, time.perf_counter()
    while done < warmup or time.perf_counter() - w0 < min_warmup_seconds:
        step()
        done += 1
        if done % 16 == 0:
            torch.cuda.synchronize()   # bound the queue while the clock is being read
    exact = (run(steps), steps)
    el, n = exact
    for _ in range(3):
        if el >= min_seconds:
            break
        n = int(math.ceil(n * min_seconds * 1.15 / max(el, 1e-6)))
        el = run(n)
    return exact, (el, n)


def ransac_roofline(args, torch, lib, _lib, step, problems, hyps, n_points):
    """Roofline of the RANSAC half: the solve kernel (fp64 VALU) and, beside it, the scoring
    kernel, timed with HIP events on the launch stream inside the batched step."""
    a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
    _lib.check(lib.ph_set_profiling(1))
    first, last, score = [], [], []
    for r in range(14):
        step()
        step()
        _lib.check(lib.ph_ransac_last_timing(C.byref(a), C.byref(b), C.byref(c)))
        if r >= 2:
            first.append(a.value)
            last.append(b.value)
            score.append(c.value)
    _lib.check(lib.ph_set_profiling(0))
    torch.cuda.synchronize()
    fma, mix = C.c_double(0), C.c_double(0)
    _lib.check(lib.ph_valu_fp64_peak(2000, C.byref(fma), C.byref(mix)))
    counts = json.load(open(os.path.join(ROOT, "profiles", "ransac_flop_count.json")))
    solve_flop = counts["solve_fp64_per_hyp"]["flop"]
    score_flop = counts["score_fp64_per_hyp"]["flop"] / counts["n_points"] * n_points \
        + 5.0 * n_points
    total = problems * hyps
    n_cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    round_hyps = min(total, n_cus * 128)        # one GPU-full of solves per launch
    n_rounds = math.ceil(total / round_hyps)
    last_hyps = total - (n_rounds - 1) * round_hyps
    # the solve launch timed with nothing else on the GPU (the first round of a call; the
    # later rounds share the chip with the previous round's scoring, and the event interval
    # around them also holds the time their workgroups wait for CU slots, which a kernel
    # trace does not count: that figure is reported beside it)
    solve_ms = float(np.mean(first))
    hyps_per_launch = total / n_rounds
    achieved = hyps_per_launch * solve_flop / (solve_ms * 1e-3)
    peak = 2.0 * fma.value
    score_ms = float(np.mean(score))
    traffic = None   # HBM bytes per solve launch from the PMC passes of the same shape
    try:
        pmc_path = os.path.join(ROOT, "profiles", "r03", "pmc_ransac_16x4096x2000.json")
        if not os.path.exists(pmc_path):
            pmc_path = os.path.join(ROOT, "profiles", "r02", "pmc_ransac_16x4096x2000.json")
        pmc = json.load(open(pmc_path))
        if (hyps, n_points) == (4096, 2000) and problems * hyps >= n_cus * 128 == 256 * 128:  # full rounds
            traffic = pmc["derived"]["void posehip::ransac_solve_kernel<32>"]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    # One wave per SIMD is all the solve's registers and LDS allow, and a lone wave cannot
    # issue fp64 instructions as fast as the pipe takes them from eight: the ratio of the two
    # measured issue intervals (independent mul+add chains, tools/fp64_rate) scales the
    # no-fma rate down to what a single wave per SIMD can reach at all.
    lone, full = 7.16, 4.54
    try:
        rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r02",
                                                         "fp64_issue_rate_by_waves.jsonl"))]
        pick = {r["waves_per_simd"]: r["cycles_per_instr_per_simd"] for r in rows
                if r["op"] == "mul+add" and r["chains"] == 4}
        lone, full = pick[1], pick[8]
    except Exception:
        pass
    single_wave_ceiling = mix.value * full / lone
    return {
        "bound": "fp64 valu",
        "kernel": "ransac_solve_kernel<32>",
        "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
        "frac": achieved / peak,
        "frac_of_mul_add_rate": achieved / mix.value,
        "frac_of_single_wave_ceiling": achieved / single_wave_ceiling,
        "single_wave_ceiling_TFLOPs": single_wave_ceiling / 1e12,
        "single_wave_ceiling_note": f"no-fma rate x {full:.2f} / {lone:.2f}: clocks per fp64 "
                                    f"instruction of a SIMD with eight waves against one wave "
                                    f"(profiles/r02/fp64_issue_rate_by_waves.jsonl); the kernel "
                                    f"runs one wave per SIMD (424 registers, LDS for 128 "
                                    f"hypotheses per CU), and about half of its instructions "
                                    f"are algorithmic operations (the rest expand correctly "
                                    f"rounded quotients and roots, select and address)",
        "traffic": traffic,
        "kernel_ms": solve_ms,
        "kernel_ms_with_scoring_underneath": float(np.mean(last)) if n_rounds > 1 else None,
        "hypotheses_per_launch": hyps_per_launch,
        "launches_per_step": n_rounds,
        "work_per_launch": f"{hyps_per_launch:.0f} five-point EPnP solves x {solve_flop:.0f} fp64 "
                           f"operations each (exact count of the sequential solve on this "
                           f"workload's samples, profiles/ransac_flop_count.json; no fma: "
                           f"the solve rounds every operation separately)",
        "peak_source": "measured live: v_fma_f64, 8 waves/SIMD on every CU, 2 FLOP per lane-op",
        "mul_add_rate_TFLOPs": mix.value / 1e12,
        "mul_add_rate_note": "the ceiling of code without fma (v_mul_f64 + v_add_f64, one FLOP "
                             "each), measured the same way",
        "score_kernel": {
            "kernel": "ransac_score_kernel", "kernel_ms": score_ms,
            "hypotheses": last_hyps,
            "flop_per_hypothesis_nominal": score_flop,
            "nominal_TFLOPs": last_hyps * score_flop / (score_ms * 1e-3) / 1e12,
            "nominal_frac_of_fp64_peak": last_hyps * score_flop / (score_ms * 1e-3) / peak,
            "nominal_note": "operations of the exact (sequential-statement) evaluation per "
                            "verdict; the kernel settles most verdicts in packed fp32 with a "
                            "rigorous error bound and runs the exact form only where that cannot "
                            "decide, so this is work delivered, not instructions executed",
            "bytes_per_hypothesis": 20 * n_points,
            "what": f"{n_points} reprojections per hypothesis: 25 fp64 + 5 fp32 operations and "
                    f"20 bytes of correspondence each"},
    }


def bench_ransac(args, torch, dist, lib, _lib, dev, rank, world, sptr):
    """Second half of the metric: RANSAC-EPnP hypotheses/s (BASELINE configs[2]:
    4096 hypotheses over 2000 correspondences, 30 % outliers), weak-scaled like the
    matcher: every rank solves and scores its own problems."""
    import datagen
    H, n, P = args.ransac_hyps, args.ransac_points, args.ransac_problems
    K = np.ascontiguousarray(datagen.CAMERA_K).reshape(9)
    Kp = K.ctypes.data_as(C.POINTER(C.c_double))
    objs, imgs, subs = [], [], []
    for p in range(P):
        obj, img, R, t, inl = datagen.pnp_scene(n=n, outlier_frac=0.3, seed=3 + p + 100 * rank)
        sub = np.empty((H, 5), np.int32)
        _lib.check(lib.ph_ransac_subsets(0xFFFFFFFFFFFFFFFF, n, H,
                                         sub.ctypes.data_as(C.POINTER(C.c_int32))))
        objs.append(obj)
        imgs.append(img)
        subs.append(sub)
    # the P problems (frames) of a step go down in ONE batched launch
    d_obj = torch.from_numpy(np.concatenate(objs)).to(dev)
    d_img = torch.from_numpy(np.concatenate(imgs)).to(dev)
    d_sub = torch.from_numpy(np.concatenate(subs)).to(dev)
    d_off = torch.arange(0, (P + 1) * n, n, dtype=torch.int32, device=dev)
    d_m = torch.empty((P * H, 6), dtype=torch.float64, device=dev)
    d_c = torch.empty(P * H, dtype=torch.int32, device=dev)

    def step():
        _lib.check(lib.ph_ransac_epnp_hypotheses_batched_dev(
            d_obj.data_ptr(), d_img.data_ptr(), d_off.data_ptr(), P, Kp, d_sub.data_ptr(), H, 8.0,
            d_m.data_ptr(), d_c.data_ptr(), sptr))

    # latency of a single problem (one call), for reference
    def one_problem():
        _lib.check(lib.ph_ransac_epnp_hypotheses_dev(d_obj.data_ptr(), d_img.data_ptr(), n, Kp,
                                                     d_sub.data_ptr(), H, 8.0, d_m.data_ptr(),
                                                     d_c.data_ptr(), sptr))
    one_problem()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        one_problem()
    torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t0) / 5 * 1e3

    _, (el, r_steps) = timed_region(step, args.ransac_steps, 2, args.min_seconds / 2,
                               args.min_warmup_seconds, torch, dist, dev, world, args.dist_backend)
    res = {
        "value": float(H) * P * world * r_steps / el,
        "unit": "hypotheses/s",
        "steps": r_steps,
        "ms_per_step": el / r_steps * 1e3,
        "single_problem_ms": single_ms,
        "single_problem_hyps_per_s": H / (single_ms * 1e-3),
        "config": {"workload": f"{H} EPnP hypotheses (5-point samples, fp64) each scored on {n} "
                               f"2D-3D correspondences, 30% outliers (BASELINE configs[2]); "
                               f"{P} problems per step per GPU in one batched launch",
                   "n_hyp": H, "n_points": n, "problems_per_step": P},
        "dtype": "f64 solve, f32 reprojection error",
    }
    if rank == 0:
        res["roofline"] = ransac_roofline(args, torch, lib, _lib, step, P, H, n)
    if rank == 0:
        # whole solvePnPRansac calls (sampling, adaptive budget, inlier list and refit) from host
        # buffers: a batch of frames of the reference's size in ONE call, and one frame per call
        import poseestimation_amd as pa
        fr = [datagen.pnp_scene(n=500, outlier_frac=0.3, seed=500 + f) for f in range(256)]
        fo, fi = [f[0] for f in fr], [f[1] for f in fr]
        pa.solvePnPRansacBatch(fo, fi, datagen.CAMERA_K, None, iterationsCount=100)
        c0 = time.perf_counter()
        for _ in range(5):
            pa.solvePnPRansacBatch(fo, fi, datagen.CAMERA_K, None, iterationsCount=100)
        tb = (time.perf_counter() - c0) / 5
        for f in range(8):
            pa.solvePnPRansac(fo[f], fi[f], datagen.CAMERA_K, None, iterationsCount=100)
        t1 = float("inf")
        for _ in range(3):   # best of three runs of 64 calls, one frame per call
            c0 = time.perf_counter()
            for f in range(64):
                pa.solvePnPRansac(fo[f], fi[f], datagen.CAMERA_K, None, iterationsCount=100)
            t1 = min(t1, (time.perf_counter() - c0) / 64)
        res["solvepnp_ransac_host_calls"] = {
            "workload": "256 frames x 500 correspondences, 30% outliers, iterationsCount 100 "
                        "(BASELINE configs[0] size), host buffers, PCIe and host replay included",
            "batched_poses_per_s": 256 / tb, "batched_call_ms": tb * 1e3,
            "single_call_ms": t1 * 1e3,
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU beside it: at N = 1 only
        import oracle_bind as orc
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        threads = max(1, min(16, cores))
        # a bounded sample of the same step: whole problems, one after the other, for about
        # half of --cpu-seconds; every one of them is compared with the GPU's step
        step()
        torch.cuda.synchronize()
        g_c, g_m = d_c.cpu().numpy(), d_m.cpu().numpy()
        done, spent, same, worst = 0, 0.0, True, 0.0
        while done < P and (done == 0 or spent < args.cpu_seconds / 2):
            c0 = time.perf_counter()
            m, c = orc.pnp_hypotheses(objs[done], imgs[done], datagen.CAMERA_K, subs[done],
                                      threads=threads)
            spent += time.perf_counter() - c0
            same = same and bool(np.array_equal(g_c[done * H:(done + 1) * H], c))
            fin = np.isfinite(m)
            if fin.any():
                worst = max(worst, float(np.abs(g_m[done * H:(done + 1) * H][fin] - m[fin]).max()))
            done += 1
        res["cpu_baseline"] = {
            "value": done * H / spent, "unit": "hypotheses/s", "cores": threads, "kind": "port",
            "sample": f"the first {done} of the step's {P} problems, {H} hypotheses x {n} "
                      f"correspondences each (oracle/pnp_ransac.c, {threads} pthreads)",
            "gpu_counts_match_cpu": same,
            "gpu_models_max_abs_diff": worst,
        }
    return res


def bench_train_sharded(args, torch, dist, lib, _lib, dev, rank, world, sptr):
    """The train-sharded form of the matcher (BASELINE configs[4] scaled to the rank count):
    every rank holds all queries and one contiguous shard of the train rows, searches its
    shard with global indices, the per-shard k=2 candidates are all-gathered (RCCL) and
    every rank merges them.  Weak scaling: the train set grows with the ranks."""
    nq, ntl = args.shard_nq, args.shard_nt
    if nq <= 0 or ntl <= 0:
        return None

    def rows(n, seed):   # device-side generator: any rank can rebuild any shard
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        return torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)

    from poseestimation_amd.distributed import knn_train_sharded
    d_q = rows(nq, 4999)
    d_t = rows(ntl, 5000 + rank)
    base = rank * ntl
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    last = {}

    def step(timed=False):
        # the product's own sequence: per-shard search with global indices, all-gather of
        # the candidates in rank order (host-staged under gloo), merge + ratio test
        last["out"] = knn_train_sharded(d_q, d_t, base, 2, ratio=args.ratio,
                                        events=ev if timed else None)

    _, (el, steps) = timed_region(step, args.shard_steps, 1, 0.0, 0.0, torch, dist, dev, world,
                             args.dist_backend)
    step(timed=True)
    torch.cuda.synchronize()
    res = {
        "value": float(nq) * ntl * world * steps / el, "unit": "Hamming evals/s",
        "steps": steps, "ms_per_step": el / steps * 1e3, "scaling": "weak",
        "config": {"workload": f"{nq} queries (replicated) x {ntl * world} train rows in {world} "
                               f"contiguous shard(s) of {ntl}, kNN k=2 + ratio {args.ratio} "
                               f"(BASELINE configs[4] scaled to the rank count)",
                   "nq": nq, "nt_per_rank": ntl, "nt_total": ntl * world,
                   "parallelism": f"train shards x{world}, one all-gather of [2,nq,2] int32 per "
                                  f"rank + merge" if world > 1 else "one shard (no exchange)"},
        "search_ms": ev[0].elapsed_time(ev[1]),
        "allgather_ms": ev[1].elapsed_time(ev[2]) if world > 1 else None,
        "allgather_bytes_sent_per_rank": 2 * nq * 2 * 4 if world > 1 else 0,
        "allgather_bytes_received_per_rank": 2 * nq * 2 * 4 * (world - 1),
        "merge_ms": ev[2].elapsed_time(ev[3]),
    }
    out_i, out_d, keep = last["out"]
    ratio = C.c_float(args.ratio)
    if rank == 0:
        # spot check: sampled queries against the whole train set in one unsharded call
        pick = torch.randperm(nq, device=dev, generator=torch.Generator(device=dev).manual_seed(1))[:256]
        full = torch.cat([d_t] + [rows(ntl, 5000 + r) for r in range(1, world)])
        sq = d_q[pick].contiguous()
        ci = torch.empty((256, 2), dtype=torch.int32, device=dev)
        cd = torch.empty((256, 2), dtype=torch.int32, device=dev)
        ck = torch.empty(256, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        _lib.check(lib.ph_knn_hamming256_dev(sq.data_ptr(), 256, full.data_ptr(), full.shape[0], 2,
                                             0, ci.data_ptr(), cd.data_ptr(), ck.data_ptr(), ratio,
                                             sptr))
        torch.cuda.synchronize()
        res["sampled_rows_equal_unsharded_call"] = bool(
            torch.equal(out_i[pick], ci) and torch.equal(out_d[pick], cd)
            and torch.equal(keep[pick], ck))
        del full
    return res


def bench_reference_frame(args, torch, lib, _lib, dev, sptr):
    """BASELINE configs[0], the reference's own per-frame workload: 500 ORB descriptors against
    500 with BFMatcher(NORM_HAMMING, crossCheck=True).match (matcher.py:18,21) -- one frame per
    host call through the C ABI (PCIe, launch and synchronisation included), and 256 such
    frames resident in HBM in one batched launch.  Rank 0 only: latency, not a scaling figure."""
    import datagen
    from poseestimation_amd.distributed import match_crosscheck_frames
    n, F = 500, 256
    q = [datagen.random_descriptors(n, 9000 + f) for f in range(8)]
    t = [datagen.perturbed_copy(x, 9100 + f) for f, x in enumerate(q)]
    u8 = C.POINTER(C.c_uint8)
    out = (_lib.DMatchC * n)()
    n_out = C.c_int(0)
    fn = lib.ph_match_crosscheck_hamming256

    def one(f):
        st = fn(q[f].ctypes.data_as(u8), n, 32, t[f].ctypes.data_as(u8), n, 32, out, C.byref(n_out))
        if st != 0:
            _lib.check(st)

    for f in range(8):
        one(f)
    best = float("inf")
    for _ in range(3):
        c0 = time.perf_counter()
        for r in range(200):
            one(r % 8)
        best = min(best, (time.perf_counter() - c0) / 200)
    kept = n_out.value
    d_q = torch.from_numpy(np.concatenate([q[f % 8] for f in range(F)])).to(dev)
    d_t = torch.from_numpy(np.concatenate([t[f % 8] for f in range(F)])).to(dev)
    off = np.arange(F + 1, dtype=np.int32) * n
    match_crosscheck_frames(d_q, off, d_t, off)
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(20):
        idx, _d = match_crosscheck_frames(d_q, off, d_t, off)
    torch.cuda.synchronize()
    batched = (time.perf_counter() - c0) / 20
    return {
        "workload": f"{n} x {n} ORB descriptors, BFMatcher(NORM_HAMMING, crossCheck=True).match "
                    f"(BASELINE configs[0]; the pose half is ransac_epnp.solvepnp_ransac_host_calls)",
        "single_call_us": best * 1e6, "matches_kept_last_call": kept,
        "batched_us_per_frame": batched / F * 1e6, "batched_frames": F,
        "batched_matches_kept_per_frame": int((idx >= 0).sum().item()) / F,
        "what": "single: host buffers through the C ABI, one frame per call; batched: 256 frames "
                "resident in HBM, one ph_match_crosscheck_hamming256_frames_dev launch",
    }


def bench_frame_sharded(args, torch, dist, lib, _lib, dev, rank, world, sptr):
    """The frame-sharded form of BASELINE configs[3]: 4096 frames x 2000 descriptors against a
    50k-row model database, 512 frames per rank at 8 ranks.  The database is uploaded once and
    stays resident on every rank; a rank's block of frames is ONE search
    (distributed.knn_frame_block_resident); nothing is exchanged.  Weak scaling: every rank
    has its own 512 frames."""
    F, R, T = args.db_frames, args.db_rows_per_frame, args.db_rows
    if F <= 0 or R <= 0 or T <= 0:
        return None
    from poseestimation_amd.distributed import knn_frame_block_resident
    g = torch.Generator(device=dev)
    g.manual_seed(7000)
    d_db = torch.randint(0, 256, (T, 32), dtype=torch.uint8, device=dev, generator=g)
    g.manual_seed(7001 + rank)
    d_q = torch.randint(0, 256, (F * R, 32), dtype=torch.uint8, device=dev, generator=g)
    db_host = d_db.cpu().numpy()
    h = C.c_void_p()
    _lib.check(lib.ph_db_upload(db_host.ctypes.data_as(C.POINTER(C.c_uint8)), T, 32, C.byref(h)))
    last = {}

    def step():
        last["out"] = knn_frame_block_resident(h, d_q, 2, ratio=args.ratio)

    try:
        _, (el, steps) = timed_region(step, args.db_steps, 1, 0.0, 0.0, torch, dist, dev, world,
                                      args.dist_backend)
        res = {
            "value": float(F) * R * T * world * steps / el, "unit": "Hamming evals/s",
            "steps": steps, "ms_per_step": el / steps * 1e3, "scaling": "weak",
            "config": {"workload": f"{F * world} frames x {R} descriptors against a resident "
                                   f"{T}-row model database, kNN k=2 + ratio {args.ratio} "
                                   f"(BASELINE configs[3]: 4096 frames over 8 GPUs), {F} frames per "
                                   f"rank in one search",
                       "frames_per_rank": F, "rows_per_frame": R, "db_rows": T,
                       "parallelism": f"frames x{world} (no collective)"},
            "us_per_frame": el / steps / F * 1e6,
        }
        if rank == 0:
            # spot check: sampled query rows through the plain device entry point on the raw rows
            idx, dst, keep = last["out"]
            pick = torch.randperm(F * R, device=dev,
                                  generator=torch.Generator(device=dev).manual_seed(2))[:256]
            sq = d_q[pick].contiguous()
            ci = torch.empty((256, 2), dtype=torch.int32, device=dev)
            cd = torch.empty((256, 2), dtype=torch.int32, device=dev)
            ck = torch.empty(256, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            _lib.check(lib.ph_knn_hamming256_dev(sq.data_ptr(), 256, d_db.data_ptr(), T, 2, 0,
                                                 ci.data_ptr(), cd.data_ptr(), ck.data_ptr(),
                                                 C.c_float(args.ratio), sptr))
            torch.cuda.synchronize()
            res["sampled_rows_equal_plain_call"] = bool(
                torch.equal(idx[pick], ci) and torch.equal(dst[pick], cd)
                and torch.equal(keep[pick], ck))
    finally:
        _lib.check(lib.ph_db_free(h))
    return res


def main():
    args = parse()
    # RCCL shares buffers between the ranks' processes through dmabuf handles on this platform
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # one rank per GPU; under --dist-backend gloo several ranks may share a GPU (rehearsal)
    dev_index = local_rank if args.dist_backend == "nccl" else local_rank % torch.cuda.device_count()
    if args.dist_backend == "nccl" and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, but this node shows "
                         f"{torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from poseestimation_amd import _lib
    _lib.init(dev_index)
    lib = _lib.load()

    nq, nt, F = args.nq, args.nt, args.frames_per_step
    # synthetic descriptors: iid uniform bytes (config "C2"); every rank and frame has
    # its own query and train rows
    t_host = np.random.default_rng(2000 + rank).integers(0, 256, size=(F, nt, 32), dtype=np.uint8)
    q_host = np.random.default_rng(1000 + rank).integers(0, 256, size=(F, nq, 32), dtype=np.uint8)
    d_t = torch.from_numpy(t_host).to(dev)
    d_q = torch.from_numpy(q_host).to(dev)
    d_idx = torch.empty((F, nq, 2), dtype=torch.int32, device=dev)
    d_dist = torch.empty((F, nq, 2), dtype=torch.int32, device=dev)
    d_keep = torch.empty((F, nq), dtype=torch.uint8, device=dev)
    # a stream of our own: the handle of torch's default stream is 0, which the
    # C ABI reads as "use the library's stream"
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sptr = C.c_void_p(stream.cuda_stream)
    qp = [d_q[f].data_ptr() for f in range(F)]
    tp = [d_t[f].data_ptr() for f in range(F)]
    ip = [d_idx[f].data_ptr() for f in range(F)]
    dp = [d_dist[f].data_ptr() for f in range(F)]
    kp = [d_keep[f].data_ptr() for f in range(F)]
    ratio = C.c_float(args.ratio)
    i32p = C.POINTER(C.c_int32)
    q_off = (np.arange(F + 1, dtype=np.int64) * nq).astype(np.int32)
    t_off = (np.arange(F + 1, dtype=np.int64) * nt).astype(np.int32)
    q_off_p, t_off_p = q_off.ctypes.data_as(i32p), t_off.ctypes.data_as(i32p)
    fn_frames = lib.ph_knn_hamming256_frames_dev
    fn = lib.ph_knn_hamming256_dev
    args_frames = (d_q.data_ptr(), q_off_p, d_t.data_ptr(), t_off_p, F, 2, d_idx.data_ptr(),
                   d_dist.data_ptr(), d_keep.data_ptr(), ratio, sptr)

    def step():
        st = fn_frames(*args_frames)
        if st != 0:
            _lib.check(st)

    def step_per_call():
        for f in range(F):
            st = fn(qp[f], nq, tp[f], nt, 2, 0, ip[f], dp[f], kp[f], ratio, sptr)
            if st != 0:
                _lib.check(st)

    def barrier():
        if world > 1:
            dist.barrier()

    (elapsed, steps_run), (long_s, long_steps) = timed_region(step, args.steps, args.warmup, args.min_seconds,
                                      args.min_warmup_seconds, torch, dist, dev, world,
                                      args.dist_backend)

    evals_per_step_all = float(nq) * nt * F * world
    value = evals_per_step_all * steps_run / elapsed

    sharded = bench_train_sharded(args, torch, dist, lib, _lib, dev, rank, world, sptr)
    framed = bench_frame_sharded(args, torch, dist, lib, _lib, dev, rank, world, sptr)
    ref_frame = bench_reference_frame(args, torch, lib, _lib, dev, sptr) if rank == 0 else None
    ransac = bench_ransac(args, torch, dist, lib, _lib, dev, rank, world, sptr)

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (distance kernel), measured live with
        # HIP events on the launch stream, outside the timed region above
        _lib.check(lib.ph_set_profiling(1))
        ks, ms, ov = [], [], []
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        for r in range(40):
            # a run of back-to-back calls keeps the GPU in the steady state of the timed
            # region; the events of the last call of the run are read
            for _ in range(3):
                step()
            _lib.check(lib.ph_knn_last_timing(C.byref(a), C.byref(b)))
            _lib.check(lib.ph_knn_last_timing_overhead(C.byref(c)))
            if r >= 5:
                ks.append(a.value)
                ms.append(b.value)
                ov.append(c.value)
        _lib.check(lib.ph_set_profiling(0))
        # an event-to-event interval costs a fixed amount even when empty; it is measured
        # right behind every call and taken off both kernel figures
        event_ms = float(np.mean(ov))
        kernel_ms = float(np.mean(ks)) - event_ms
        merge_ms = float(np.mean(ms)) - event_ms
        mfma_path = lib.ph_knn_last_path() == 1
        prepass_ms = 0.0
        if mfma_path:
            _lib.check(lib.ph_set_profiling(1))
            pp = []
            for r in range(12):
                for _ in range(3):
                    step()
                _lib.check(lib.ph_knn_last_prepass_timing(C.byref(a)))
                if r >= 2:
                    pp.append(a.value)
            _lib.check(lib.ph_set_profiling(0))
            prepass_ms = float(np.mean(pp)) - event_ms
        torch.cuda.synchronize()  # the peak kernel runs on the library's stream: nothing beside it
        peak_ops = C.c_double(0)
        _lib.check(lib.ph_valu_popcount_peak(4000, C.byref(peak_ops)))
        peak_evals = peak_ops.value / 16.0
        achieved = float(nq) * nt * F / (kernel_ms * 1e-3)
        kernel_name = "knn_mfma_frames_kernel" if mfma_path else "knn_partial_frames_kernel"
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and (nq, nt) == (10000, 10000):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(kernel_name + "_bytes_per_launch_by_frames", {}).get(str(F))
                if traffic is None and F == 16:
                    traffic = tj.get(kernel_name + "_bytes_per_launch")
            except Exception:
                traffic = None

        def timed(fn_step, n):
            for _ in range(3):
                fn_step()
            torch.cuda.synchronize()
            p0 = time.perf_counter()
            for _ in range(n):
                fn_step()
            torch.cuda.synchronize()
            return (time.perf_counter() - p0) / n

        # the same frames as one call each (outside the timed region above)
        pc_steps = max(5, min(steps_run, 30))
        per_call_s = timed(step_per_call, pc_steps)
        per_call = {"value": float(nq) * nt * F / per_call_s, "unit": "Hamming evals/s",
                    "us_per_frame": per_call_s / F * 1e6, "steps": pc_steps,
                    "what": f"the same {F} frames, one ph_knn_hamming256_dev call per frame, rank 0"}
        # and the same batched step on the other distance kernel (xor + popcount on the VALU)
        popcount = None
        if mfma_path:
            _lib.check(lib.ph_knn_set_path(1))
            pop_s = timed(step, pc_steps)
            # its distance kernel against the integer-popcount peak, measured like the main one
            _lib.check(lib.ph_set_profiling(1))
            pk = []
            for r in range(12):
                for _ in range(2):
                    step()
                _lib.check(lib.ph_knn_last_timing(C.byref(a), C.byref(b)))
                if r >= 2:
                    pk.append(a.value)
            _lib.check(lib.ph_set_profiling(0))
            _lib.check(lib.ph_knn_set_path(0))
            pop_kernel_ms = float(np.mean(pk)) - event_ms
            pop_achieved = float(nq) * nt * F / (pop_kernel_ms * 1e-3)
            popcount = {"value": float(nq) * nt * F / pop_s, "unit": "Hamming evals/s",
                        "ms_per_step": pop_s * 1e3,
                        "roofline": {"bound": "valu", "kernel": "knn_partial_frames_kernel",
                                     "achieved": pop_achieved / 1e12, "peak": peak_evals / 1e12,
                                     "unit": "Teval/s", "frac": pop_achieved / peak_evals,
                                     "kernel_ms": pop_kernel_ms,
                                     "peak_source": "measured live: v_xor_b32+v_bcnt_u32_b32 mix, "
                                                    "8 waves/SIMD, /16"},
                        "what": "the same step with ph_knn_set_path(1): the popcount kernel "
                                "(v_xor_b32 + v_bcnt_u32_b32, 16 VALU lane-ops per evaluation), "
                                "whose ceiling is the integer VALU peak"}
        if mfma_path:
            flops = 512.0 * nq * nt * F  # 256 multiply-adds per pair
            peak_tflops = 10000.0        # dense FP4 MFMA peak of MI355X
            # and what this GPU sustains on the kernel's MFMA alone, at the clock it holds
            bare = C.c_double(0)
            _lib.check(lib.ph_mfma_fp4_peak(3000, C.byref(bare)))
            roofline = {
                "bound": "mfma",
                "kernel": kernel_name,
                "achieved": flops / (kernel_ms * 1e-3) / 1e12,
                "peak": peak_tflops,
                "unit": "TFLOP/s",
                "frac": flops / (kernel_ms * 1e-3) / 1e12 / peak_tflops,
                "traffic": traffic,
                "work_per_launch": f"{F} frames x {nq}x{nt} pairs x 256 FP4 multiply-adds "
                                   f"(v_mfma_scale_f32_32x32x64_f8f6f4, exact in fp32)",
                "evals_per_s": achieved,
                "kernel_ms": kernel_ms,
                "prepass_ms": prepass_ms,
                "merge_kernel_ms": merge_ms,
                "event_interval_overhead_ms": event_ms,
                "peak_source": "dense FP4 MFMA peak (~10 PFLOP/s at 2.4 GHz)",
                "peak_measured": bare.value / 1e12,
                "frac_of_measured_peak": flops / (kernel_ms * 1e-3) / bare.value,
                "peak_measured_source": "measured live: v_mfma_scale_f32_32x32x64_f8f6f4 alone on "
                                        "random FP4 operands, 4 independent chains, 2 waves/SIMD "
                                        "on every CU (the clock the chip holds under MFMA load is "
                                        "below the 2.4 GHz of the spec figure)",
                "hbm_algorithmic_GBps": 128.0 * (nq + nt) * F / (kernel_ms * 1e-3) / 1e9,
            }
        else:
            roofline = {
                "bound": "valu",
                "bound_note": "integer VALU issue (xor + popcount): the path is neither HBM- nor "
                              "MFMA-bound; hbm_algorithmic_GBps and traffic show how far it is from HBM",
                "kernel": kernel_name,
                "achieved": achieved / 1e12,
                "peak": peak_evals / 1e12,
                "unit": "Teval/s",
                "frac": achieved / peak_evals,
                "traffic": traffic,
                "work_per_launch": f"{F} frames x {nq}x{nt} Hamming-256 evals (16 VALU lane-ops each)",
                "kernel_ms": kernel_ms,
                "merge_kernel_ms": merge_ms,
                "event_interval_overhead_ms": event_ms,
                "peak_source": "measured live: v_xor_b32+v_bcnt_u32_b32 mix, 8 waves/SIMD, /16",
                "peak_spec": 256 * 64 * 2.4e9 / 16 / 1e12,
                "hbm_algorithmic_GBps": 32.0 * (nq + nt) * F / (kernel_ms * 1e-3) / 1e9,
            }

        cpu = None
        if world == 1 and not args.no_cpu_baseline:   # at N = 1 only; the other ranks would wait
            import oracle_bind as orc
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            threads = max(1, min(16, cores))
            q0 = q_host[0]
            reps, spent = 0, 0.0
            ridx = rdist = None
            while spent < args.cpu_seconds and reps < 5000:
                c0 = time.perf_counter()
                ridx, rdist = orc.knn(q0, t_host[0], 2, threads=threads)
                kept = orc.ratio_filter(ridx, rdist, args.ratio)
                spent += time.perf_counter() - c0
                reps += 1
            cpu = {
                "value": float(nq) * nt * reps / spent,
                "unit": "Hamming evals/s",
                "cores": threads,
                "kind": "port",
                "sample": f"{reps} x one {nq}x{nt} k=2 + ratio frame (oracle/bf_match.c, "
                          f"{threads} pthreads)",
            }
            # the same frame on the GPU (from the batched launch) must agree bit for bit
            step()
            torch.cuda.synchronize()
            g_idx = d_idx[0].cpu().numpy()
            g_dist = d_dist[0].cpu().numpy()
            g_keep = d_keep[0].cpu().numpy()
            kept_q = np.array([m[0] for m in kept], dtype=np.int64)
            ok = (np.array_equal(g_idx, ridx) and np.array_equal(g_dist, rdist)
                  and np.array_equal(np.nonzero(g_keep)[0], kept_q))
            cpu["gpu_matches_cpu"] = bool(ok)

        out = {
            "metric": "Hamming evals/sec (value; kNN k=2 + ratio test, 256-bit ORB descriptors) "
                      "+ RANSAC-EPnP hypotheses/sec (ransac_epnp.value)",
            "value": value,
            "unit": "Hamming evals/s",
            "n_gpus": world,
            "steps": steps_run,
            "warmup": args.warmup,
            "timed_region_s": elapsed,
            "ms_per_step": elapsed / steps_run * 1e3,
            "sustained": {"value": evals_per_step_all * long_steps / long_s, "steps": long_steps,
                          "timed_region_s": long_s, "ms_per_step": long_s / long_steps * 1e3,
                          "what": f"the same step timed the same way over at least "
                                  f"{args.min_seconds} s (the {steps_run} steps asked for are "
                                  f"{elapsed * 1e3:.1f} ms of work)"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp4 e2m1 x fp4 -> f32 (exact integers)" if mfma_path else "u32 (xor + popcount)",
            "data": "synthetic",
            "config": {
                "workload": f"{nq}x{nt} ORB-descriptor kNN k=2 + ratio {args.ratio} per frame "
                            f"(BASELINE configs[1]), {F} independent frames per step per GPU in "
                            f"one batched launch",
                "nq": nq, "nt": nt, "k": 2, "frames_per_step": F,
                "parallelism": f"frames x{world} (no collective)",
            },
            "us_per_frame": elapsed / steps_run / F * 1e6,
            "reference_frame_crosscheck": ref_frame,
            "frame_sharded_database": framed,
            "train_sharded": sharded,
            "per_call": per_call,
            "popcount_kernel": popcount,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "ransac_epnp": ransac,
        }
        print(json.dumps(out), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
