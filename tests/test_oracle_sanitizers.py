"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the CPU build; GPU sanitizers are not
available on the pool).  The oracle is what every GPU parity test trusts, so its own memory safety is checked here on a small workload."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_runs_clean_under_asan_ubsan():
    build = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "oracle_sanitize")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
                           os.path.join(ROOT, "tests", "cpp", "oracle_sanitize.cpp"), os.path.join(ROOT, "oracle", "hs_oracle.cpp"),
                           os.path.join(ROOT, "oracle", "hs_oracle_match.cpp"), "-pthread", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "ORACLE SANITIZE OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
