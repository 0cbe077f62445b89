"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (host code only):
tests/oracle_selftest.c calls every entry point with exactly-sized heap buffers."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_clean_under_asan_ubsan(tmp_path):
    exe = tmp_path / "selftest"
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-march=x86-64-v2", "-mpopcnt",
                           "-ffp-contract=off", "-pthread", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined",
                           os.path.join(ROOT, "tests", "oracle_selftest.c"),
                           os.path.join(ROOT, "oracle", "bf_match.c"),
                           os.path.join(ROOT, "oracle", "pnp_ransac.c"), "-lm", "-o", str(exe)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)   # the sanitizer runtime must come first in this child only
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "oracle selftest ok" in out.stdout
