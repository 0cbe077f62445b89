"""The C++ shim (include/posehip.hpp): it must compile without OpenCV (CPU suite)
and the host example built on it must run on the GPU (GPU suite)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_compiles_without_opencv(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "posehip.hpp"\n'
                   'int main() { posehip::DMatch m; std::vector<posehip::DMatch> v; (void)m; '
                   'return (int)v.size(); }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only",
                           "-I" + os.path.join(ROOT, "include"), str(src)])
    # and as plain C for the C ABI header
    csrc = tmp_path / "t.c"
    csrc.write_text('#include "posehip.h"\nint main(void) { return sizeof(ph_dmatch) == 16 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only",
                           "-I" + os.path.join(ROOT, "include"), str(csrc)])


@pytest.mark.gpu
def test_cpp_host_example_runs():
    exe = os.path.join(ROOT, "examples", "cereal_box_matcher")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"matches (\d+) good (\d+) crosscheck (\d+) ratio (\d+)", out.stdout)
    assert m, out.stdout
    n_match, n_good, n_cc, n_ratio = map(int, m.groups())
    assert n_match == 480 and 0 < n_good <= 480 and 0 < n_cc <= 480 and 0 < n_ratio <= 480
    assert "pnp ok 1" in out.stdout
