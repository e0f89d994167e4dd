"""tools/typecheck_adaptors.py: g++ -std=c++14 -fsyntax-only -DHYSLAM_AMD_WITH_HYSLAM over every header of hyslam_amd/host/ (and host/replace/FeatureMatcher.cc)
against the reference's REAL headers, with declaration-only stand-ins for the third-party headers those include (tests/cpp/thirdparty_stubs/).  The
adaptor BODIES are type-checked against Frame.h, KeyFrame.h, MapPoint.h, FeatureViews.h, FeatureFactory.h, FeatureMatcher.h, LandMarkMatches.h — which
tools/check_signatures.py (text comparison of declarations) cannot do.  Build container only — skipped where /root/reference is absent (the GPU box)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "typecheck_adaptors.py")
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "features")), reason="the reference's headers are not on this machine")


def run(*extra):
    return subprocess.run([sys.executable, TOOL, REF] + list(extra), capture_output=True, text=True, timeout=300)


def test_adaptors_compile_against_the_real_hyslam_headers():
    r = run()
    assert r.returncode == 0, r.stdout + r.stderr
    assert "typecheck unpatched: ok" in r.stdout and "typecheck patched: ok" in r.stdout


@pytest.mark.parametrize("header,old,new,expect", [
    # a hySLAM accessor that does not exist (the real FeatureViews has getViews() on Frame / KeyFrame, keypt(), descriptor(), uR() ...)
    ("HipFeatureMatcher.h", "pKF->getViews()", "pKF->getViewz()", "getViewz"),
    # a wrong argument type for a real member: Frame::associateLandMark(int, MapPoint*, bool)
    ("HipFeatureMatcher.h", "int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3) override {",
     "int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const double th = 3) override {", "override"),
    # the extractor's call operator no longer matches FeatureExtractor's pure virtual: the class stays abstract
    ("HipORBExtractor.h", "void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints,",
     "void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::Point2f>& _keypoints,", "error"),
])
def test_the_typecheck_fires_on_a_drifted_adaptor(tmp_path, header, old, new, expect):
    host = tmp_path / "hyslam_amd" / "host"                         # (the adaptors include "../../include/hyslam_amd.h")
    shutil.copytree(os.path.join(ROOT, "hyslam_amd", "host"), host)
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    src = (host / header).read_text()
    assert old in src, old
    (host / header).write_text(src.replace(old, new))
    r = run("--host-dir", str(host))
    assert r.returncode != 0 and "FAILED" in r.stdout and expect in r.stdout, r.stdout[-2000:]
