"""tools/check_signatures.py: the adaptors' signatures (hyslam_amd/host/cv_compat.h + HipStereomatcher) against the reference's real headers.
Build container only — skipped where /root/reference is absent (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_signatures.py")
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "features")), reason="the reference's headers are not on this machine")


def run(env=None):
    return subprocess.run([sys.executable, TOOL, REF], capture_output=True, text=True, timeout=120, env=env)


def test_adaptor_signatures_agree_with_the_reference_headers():
    r = run()
    assert r.returncode == 0, r.stdout + r.stderr
    first = r.stdout.splitlines()[0]
    assert "agree" in first and int(first.split()[1]) >= 100, first          # not vacuous: > 100 member functions were compared
    assert "MISMATCH" not in r.stdout


@pytest.mark.parametrize("old,new,expect", [
    # a parameter type of an overridden search entry point
    ("virtual int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3)",
     "virtual int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const int th = 3)", "FeatureMatcher::int SearchByProjection"),
    # const-ness of an accessor the adaptors call
    ("int numViews() const { return N; }", "int numViews() { return N; }", "FeatureViews::int numViews"),
    # a return type
    ("cv::KeyPoint keypt(int i) const { return mvKeys[i]; }", "cv::Point2f keypt(int i) const { return cv::Point2f(); }", "FeatureViews::cv::Point2f keypt"),
    # arity of the extractor's call operator
    ("virtual void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,\n                            std::vector<FeatureDescriptor>& descriptors) = 0;",
     "virtual void operator()(cv::InputArray image, std::vector<cv::KeyPoint>& keypoints, std::vector<FeatureDescriptor>& descriptors) = 0;", "operator()"),
    # a `virtual` that is NOT part of the documented patch
    ("FeatureMatcherSettings getFeatureMatcherSettings() const { return matcher_settings; }", "virtual FeatureMatcherSettings getFeatureMatcherSettings() const { return matcher_settings; }", "virtual"),
    # a member function dropped from a class the adaptors derive from
    ("    virtual float GetScaleFactor() = 0;\n", "", "GetScaleFactor"),
])
def test_the_check_fires_on_a_drifted_declaration(tmp_path, old, new, expect):
    src = open(os.path.join(ROOT, "hyslam_amd", "host", "cv_compat.h")).read()
    assert old in src
    mutated = tmp_path / "cv_compat.h"
    mutated.write_text(src.replace(old, new))
    r = run(dict(os.environ, HS_CHECK_COMPAT=str(mutated)))
    assert r.returncode != 0 and "MISMATCH" in r.stdout and expect in r.stdout, r.stdout
