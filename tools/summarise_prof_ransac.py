"""Turns the raw output of tools/run_prof_ransac.sh (gpurun_out/prof_r01h) into
profiles/r01/ransac_kernel_stats.csv and profiles/r01/pmc_ransac_16x4096x2000.json
(mean counter value per kernel launch).  python tools/summarise_prof_ransac.py"""
import csv, glob, json, os, shutil, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r01h")
DST = os.path.join(ROOT, "profiles", "r01")


def newest(pattern):
    files = glob.glob(os.path.join(SRC, pattern))
    return max(files, key=os.path.getmtime)


shutil.copy(newest("kt/*/*_kernel_stats.csv"), os.path.join(DST, "ransac_kernel_stats.csv"))
kernels = collections.defaultdict(lambda: collections.defaultdict(list))
for group in ("sq1", "sq2", "fetch", "write"):
    with open(newest(group + "/*/*_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            if "ransac" in name:
                kernels[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {
    "command": "rocprofv3 --pmc <group> --output-format csv -- tools/pnp_time_bin 16 4096 2000  "
               "(tools/run_prof_ransac.sh; groups: 8 SQ | 6 SQ | FETCH_SIZE | WRITE_SIZE); mean per kernel launch; "
               "one call = 65536 hypotheses in two rounds of 32768 (2 launches of ransac_solve_kernel<32>, "
               "2 of ransac_score_kernel)",
    "kernels": {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in kernels.items()},
}
with open(os.path.join(DST, "pmc_ransac_16x4096x2000.json"), "w") as f:
    json.dump(out, f, indent=1)
for k, cs in out["kernels"].items():
    print(k, {c: round(v) for c, v in cs.items()})
