"""The header-only C++ adaptor (hyslam_amd/host/HipORBExtractor.h: FeatureExtractor / Stereomatcher call surface over the
C ABI) compiled against host/cv_compat.h and run like ImageProcessing::ProcessStereoImage uses it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_adaptor")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "cpp", "test_adaptor.cpp"), "-o", EXE,
                           "-L" + os.path.join(ROOT, "hyslam_amd"), "-lhyslam_amd", "-L" + os.path.join(ROOT, "oracle", "_build"), "-lhs_oracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "hyslam_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle", "_build")])


def run():
    from hyslam_amd.synth import synth_stereo_pair
    L, R = synth_stereo_pair(7, 640, 480)
    return subprocess.run([EXE, "640", "480"], input=L.tobytes() + R.tobytes(), capture_output=True, timeout=300)


def test_adaptor_compiles_and_fails_loudly_without_gpu():
    build()
    r = run()
    assert r.returncode == 0, r.stdout + r.stderr
    assert b"NO DEVICE" in r.stdout or b"ADAPTOR OK" in r.stdout


@pytest.mark.gpu
def test_adaptor_bit_exact_on_gpu(gpu):
    build()
    r = run()
    assert r.returncode == 0 and b"ADAPTOR OK" in r.stdout, r.stdout + r.stderr
