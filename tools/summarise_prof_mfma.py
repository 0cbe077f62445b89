"""Turns the raw output of tools/run_prof_mfma.sh (gpurun_out/prof_r01m_F<F>/) into the
summaries kept under profiles/r01/ and the traffic entry bench.py reads.
    python tools/summarise_prof_mfma.py 64 [--primary] [--tag r03m --round r03]
--primary: also copy the kernel-trace summary and the bench lines (the shape bench.py runs).
--tag / --round: the raw directory's tag (tools/run_prof_mfma.sh's second argument) and the
directory under profiles/ the summaries go to (r01m / r01 by default)."""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

F = int(sys.argv[1])
primary = "--primary" in sys.argv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "r01m"
ROUND = sys.argv[sys.argv.index("--round") + 1] if "--round" in sys.argv else "r01"
O = os.path.join(ROOT, "gpurun_out", f"prof_{TAG}_F{F}")
P = os.path.join(ROOT, "profiles")
os.makedirs(os.path.join(P, ROUND), exist_ok=True)


def newest(pat):
    f = glob.glob(os.path.join(O, pat))
    f.sort(key=os.path.getmtime)
    return f[-1]


res = {}
for d in ("fetch", "write", "sq", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(newest(f"{d}/vm/*_counter_collection.csv"))):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "knn_mfma" in k or "expand" in k or "merge" in k:
            res.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
            res[k]["launches_averaged"] = len(next(iter(v.values())))
m = res[[x for x in res if "knn_mfma_frames" in x][0]]
bench = json.load(open(os.path.join(O, "bench_line.json")))
n_splits = 1 if F >= 64 else {16: 4, 32: 2}.get(F, 0)
rows = F * 10000
alg = rows * 32 + rows * 128 + rows * 2 * 4 * (n_splits if n_splits > 1 else 2)
alg_note = (f"{rows} raw query rows (32 B) + FP4 image of {rows} train rows (128 B) read once + "
            + (f"{n_splits} splits x 2 keys x {rows} queries x 4 B written once" if n_splits > 1 else
               f"k-NN rows (idx, dist: 16 B per query) written once by the distance kernel itself"))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for r in csv.DictReader(open(newest("kt/vm/*_kernel_trace.csv")))
        if "knn_mfma_frames" in r["Kernel_Name"]]
under = json.loads(open(os.path.join(O, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
summary = {
    "command": f"tools/run_prof_mfma.sh {F}: tools/knn_time 10000 10000 {F} (ph_knn_hamming256_frames_dev, "
               f"{F} frames of 10k x 10k, k=2), one rocprofv3 --pmc pass per counter group",
    "per_launch_averages": res,
    "derived_knn_mfma_frames_kernel": {
        "mfma_instructions": m["SQ_INSTS_MFMA"],
        "mfma_busy_cycles_per_instruction": m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_INSTS_MFMA"],
        "valu_instructions_per_mfma": m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"],
        "kernel_cycles(GRBM_GUI_ACTIVE/8)": m["GRBM_GUI_ACTIVE"] / 8,
        "mfma_pipe_utilisation": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024),
        "wave_time_split": {
            "issuing": m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"],
            "issue_stalled(pipe busy / operand not ready)": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
            "parked(s_waitcnt / barrier)": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]},
        "hbm_bytes_per_launch": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024,
        "hbm_note": "FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE doubled (gfx950 reports half of wide reads)",
        "algorithmic_bytes_per_launch": alg,
        "algorithmic_note": alg_note,
    },
    "kernel_trace_of_bench_py": {
        "launches": len(durs), "median_us": statistics.median(durs), "average_us": sum(durs) / len(durs),
        "min_us": min(durs), "max_us": max(durs),
        "hip_event_kernel_ms_same_run": under["roofline"]["kernel_ms"]},
    "bench_line_unprofiled": {k: bench[k] for k in ("value", "ms_per_step")} | {
        "roofline_frac": bench["roofline"]["frac"], "kernel_ms": bench["roofline"]["kernel_ms"],
        "prepass_ms": bench["roofline"]["prepass_ms"], "merge_kernel_ms": bench["roofline"]["merge_kernel_ms"]},
}
json.dump(summary, open(os.path.join(P, ROUND, f"pmc_knn_mfma_frames_{F}x10k_x_10k.json"), "w"), indent=1)
t = json.load(open(os.path.join(P, "hbm_traffic.json")))
t.setdefault("knn_mfma_frames_kernel_bytes_per_launch_by_frames", {})[str(F)] = int(
    summary["derived_knn_mfma_frames_kernel"]["hbm_bytes_per_launch"])
if F == 16:
    t["knn_mfma_frames_kernel_bytes_per_launch"] = t["knn_mfma_frames_kernel_bytes_per_launch_by_frames"]["16"]
if F == 64:
    t["mfma_frames_64_how"] = (f"tools/run_prof_mfma.sh 64 {TAG} (profiles/{ROUND}/pmc_knn_mfma_frames_64x10k_x_10k.json): "
                               f"64 frames per launch, one split, the distance kernel writes the k-NN rows itself; "
                               f"algorithmic bytes = 20.48 MB of raw query rows + 81.92 MB of train image read once + "
                               f"10.24 MB of k-NN rows (idx, dist: 16 B per query) written once = 112.6 MB")
    t["mfma_frames_64_algorithmic_compulsory_bytes"] = alg
json.dump(t, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
if primary:
    shutil.copy(newest("kt/vm/*_kernel_stats.csv"), os.path.join(P, ROUND, "bench_kernel_stats_mfma.csv"))
    shutil.copy(os.path.join(O, "bench_under_rocprof.json"), os.path.join(P, ROUND, "bench_under_rocprof_mfma.json"))
    shutil.copy(os.path.join(O, "bench_line.json"), os.path.join(P, ROUND, "bench_line_mfma.json"))
d = summary["derived_knn_mfma_frames_kernel"]
print(json.dumps({k: d[k] for k in ("mfma_pipe_utilisation", "valu_instructions_per_mfma", "wave_time_split",
                                    "hbm_bytes_per_launch", "algorithmic_bytes_per_launch")}, indent=1))
print(summary["kernel_trace_of_bench_py"], summary["bench_line_unprofiled"])
