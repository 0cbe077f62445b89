"""The exact fp64 operation count of one RANSAC-EPnP hypothesis (bench.py's roofline uses it).

tests/flop_count.cpp compiles the CPU oracle with `double` replaced by a counting type and
runs it over the benchmark's problem (datagen.pnp_scene(n=2000, outlier_frac=0.3, seed=3),
cv::RNG's first 4096 samples).  The GPU solve performs the same operations in the same
order, so the oracle's count is the algorithmic work of a hypothesis.  The committed figures
(profiles/ransac_flop_count.json) must equal a fresh count; regenerate them with
`python tests/test_flop_count.py --write` after a change of the oracle's solve.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
COMMITTED = os.path.join(ROOT, "profiles", "ransac_flop_count.json")


def count(n=2000, n_hyp=4096, seed=3):
    import datagen
    import oracle_bind as orc
    obj, img, R, t, inl = datagen.pnp_scene(n=n, outlier_frac=0.3, seed=seed)
    sub = orc.ransac_subsets(n, n_hyp)
    with tempfile.TemporaryDirectory() as tmp:
        exe, scene = os.path.join(tmp, "flop_count"), os.path.join(tmp, "scene.bin")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-ffp-contract=off",
                               os.path.join(HERE, "flop_count.cpp"), "-o", exe, "-lpthread"])
        with open(scene, "wb") as f:
            f.write(obj.tobytes())
            f.write(img.tobytes())
            f.write(np.ascontiguousarray(datagen.CAMERA_K, np.float64).tobytes())
            f.write(np.ascontiguousarray(sub, np.int32).tobytes())
        out = subprocess.check_output([exe, scene, str(n), str(n_hyp)], text=True)
    return json.loads(out)


def test_committed_flop_count_is_current():
    got = count()
    want = json.load(open(COMMITTED))
    assert got["solve_fp64_per_hyp"] == want["solve_fp64_per_hyp"]
    assert got["score_fp64_per_hyp"] == want["score_fp64_per_hyp"]
    s = got["solve_fp64_per_hyp"]
    assert abs(s["add"] + s["mul"] + s["div"] + s["sqrt"] + s["transcendental"] - s["flop"]) < 1e-2
    # scoring: per correspondence 11 add + 13 mul + 1 div in fp64 (R X + t, 1/z, K), and
    # Rodrigues once per hypothesis; the 5 fp32 operations of the squared error come on top
    sc = got["score_fp64_per_hyp"]
    assert sc["add"] == 11 * 2000 + 21 and sc["mul"] == 13 * 2000 + 43 and sc["div"] == 2001


if __name__ == "__main__":
    if "--write" in sys.argv:
        res = count()
        res["what"] = ("fp64 operations per hypothesis, mean over cv::RNG's first 4096 samples of "
                       "datagen.pnp_scene(n=2000, outlier_frac=0.3, seed=3) (bench.py's problem 0); "
                       "add/sub/mul/div/sqrt and acos/sin/cos one FLOP each, compares zero; "
                       "scoring adds 5 fp32 operations per correspondence")
        json.dump(res, open(COMMITTED, "w"), indent=1)
        print(json.dumps(res))
    else:
        print(json.dumps(count()))
