"""Turns the raw output of tools/run_prof_round.sh (gpurun_out/prof_<tag>) into the files kept
under profiles/<round>/: the bench line, the rocprofv3 kernel-trace summary of the same command
with the bench line of that profiled run, and the PMC counters (mean per launch) of the RANSAC
kernels.   python tools/summarise_prof_round.py <tag> <round>   e.g.  r02a r02"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + sys.argv[1])
DST = os.path.join(ROOT, "profiles", sys.argv[2])
os.makedirs(DST, exist_ok=True)


def newest(pattern):
    return max(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)


def last_json_line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


json.dump(last_json_line(os.path.join(SRC, "bench_line.json")), open(os.path.join(DST, "bench_line.json"), "w"))
json.dump(last_json_line(os.path.join(SRC, "bench_under_rocprof.json")),
          open(os.path.join(DST, "bench_under_rocprof.json"), "w"))
shutil.copy(newest("kt/*/*_kernel_stats.csv"), os.path.join(DST, "bench_kernel_stats.csv"))
shutil.copy(newest("kt_pnp/*/*_kernel_stats.csv"), os.path.join(DST, "ransac_kernel_stats.csv"))
kernels = collections.defaultdict(lambda: collections.defaultdict(list))
for group in ("sq1", "sq2", "fetch", "write"):
    with open(newest(group + "/*/*_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            if "ransac" in name:
                kernels[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {
    "command": "rocprofv3 --pmc <group> --output-format csv -- tools/pnp_time_bin 16 4096 2000  "
               "(tools/run_prof_round.sh; groups: 8 SQ | 6 SQ | FETCH_SIZE | WRITE_SIZE); mean per kernel "
               "launch; one call = 65536 hypotheses in two rounds of 32768 (2 launches of "
               "ransac_solve_kernel<32>, 2 of ransac_score_kernel; the first round's scoring runs under "
               "the second round's solves)",
    "kernels": {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in kernels.items()},
}
for k, cs in out["kernels"].items():
    d = {}
    if "SQ_WAVE_CYCLES" in cs and "SQ_WAIT_INST_ANY" in cs:
        d["wait_share_of_wave_time"] = cs["SQ_WAIT_INST_ANY"] / cs["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in cs and "SQ_WAVES" in cs:
        d["valu_instructions_per_wave"] = cs["SQ_INSTS_VALU"] / cs["SQ_WAVES"]
    if "SQ_ACTIVE_INST_VALU" in cs and "SQ_WAVE_CYCLES" in cs:
        d["valu_active_share_of_wave_time"] = cs["SQ_ACTIVE_INST_VALU"] / cs["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in cs:
        d["hbm_bytes_per_launch"] = cs["FETCH_SIZE"] * 1024 * 2 + cs.get("WRITE_SIZE", 0) * 1024
    out.setdefault("derived", {})[k] = d
json.dump(out, open(os.path.join(DST, "pmc_ransac_16x4096x2000.json"), "w"), indent=1)
for k, cs in out["kernels"].items():
    print(k, {c: round(v) for c, v in cs.items()}, out["derived"][k])
