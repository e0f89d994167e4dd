"""The PRODUCT's host code under AddressSanitizer + UndefinedBehaviorSanitizer, CPU only (tests/cpp/host_sanitize/): every translation unit of
hyslam_amd/csrc compiled host-only (`hipcc --cuda-host-only`: no device code) and linked against a stand-in for the HIP runtime on host memory
(hip_stub.cpp: device calls stubbed at the hipMalloc seam, the logic runs for real).  The driver sweeps random geometries through hs_orb_create /
hs_orb_reserve / the host-pointer entry points (planners, table builders, workspace sizing, staging, the ingest tickets) and feeds the vocabulary
loaders truncated and corrupted files.  Any sanitizer report aborts the driver.  (GPU AddressSanitizer is not available on this pool; the oracle has
its own sanitizer test, tests/test_oracle_sanitizers.py.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "cpp", "host_sanitize")
EXE = os.path.join(DIR, "_build", "host_sanitize")

# tests/cpp/host_sanitize/ is listed in .gpurunignore: the GPU pool refuses snapshots that hold a hipcc recipe with -fsanitize=address (GPU ASan is not
# available there), and this CPU-only build never runs on a GPU box
pytestmark = pytest.mark.skipif((shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc")) or not os.path.exists(os.path.join(DIR, "Makefile")),
                                reason="needs hipcc (host-only compile) and tests/cpp/host_sanitize/ (not shipped to GPU boxes)")


def build():
    subprocess.check_call(["make", "-s", "-j4", "-C", DIR], timeout=1500)


def test_product_host_code_is_sanitizer_clean():
    build()
    for seed, geo, voc in ((1, 150, 500), (2, 150, 500), (7, 100, 300)):
        r = subprocess.run([EXE, str(seed), str(geo), str(voc)], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "HOST SANITIZE OK" in r.stdout, (r.stdout + r.stderr)[-4000:]
        assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_small_batch_pyramid_plan_is_one_launch_at_1080p():
    """the host-side plan of BASELINE's geometry: levels 1-7 of a 1920x1080 frame in ONE k_resize_chain launch for small batches (3 in the standard plan)"""
    build()
    r = subprocess.run([EXE, "plan", "1920", "1080"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "standard plan 3, small-batch plan 1 (longest chain 7 levels" in r.stdout, r.stdout
