#!/usr/bin/env python3
"""Puts a COMPILER on the adaptors' `HYSLAM_AMD_WITH_HYSLAM` branches (build container only: needs /root/reference).

    g++ -std=c++14 -fsyntax-only -DHYSLAM_AMD_WITH_HYSLAM -I<every directory of /root/reference/src> -Itests/cpp/thirdparty_stubs ...

over a translation unit that includes every header of hyslam_amd/host/ — so every `Frame::` / `KeyFrame::` / `MapPoint::` / `FeatureViews::` /
`FeatureFactory::` / `LandMarkMatches::` member the adaptor BODIES touch is type-checked against hySLAM's REAL headers (C++14, as the reference's
CMakeLists.txt:16 sets), and `HipORBFactory::LoadSettings` against cv::FileStorage's declared interface.  The third-party headers hySLAM includes
(OpenCV 3.4, Eigen, DBoW2: absent from this image) are DECLARATION-ONLY stand-ins under tests/cpp/thirdparty_stubs/ — test infrastructure: nothing
has a body, nothing is linked, shipped or used for arithmetic.  Two integrations:

  unpatched  -DHYSLAM_AMD_UNPATCHED_MATCHER + host/replace/FeatureMatcher.cc against the reference's headers as they are
  patched    the two-line `virtual` patch of INTEGRATION.md §3(a) applied to COPIES of FeatureMatcher.h / FeatureFactory.h in a temporary directory
             (never written into this repository), then HipFeatureMatcher's `override`s and HipORBFactory::getFeatureMatcher() must bind

usage: typecheck_adaptors.py [/root/reference] [--host-dir DIR]     exit 0 = both modes compile; prints the compiler's errors otherwise."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = ["HipORBFactory.h", "HipFeatureMatcher.h", "HipAssociationReplay.h", "HipORBExtractor.h"]
SEARCHES = ("SearchByProjection", "SearchByBoW", "SearchByBoW2", "SearchForTriangulation", "SearchForInitialization", "Fuse", "SearchBySim3")


def patched_headers(ref, out_dir):
    """INTEGRATION.md §3(a) applied to copies: `virtual` in front of the twelve search entry points + a virtual destructor; `virtual getFeatureMatcher()`."""
    src = open(os.path.join(ref, "src", "features", "FeatureMatcher.h")).read()
    lines, n = [], 0
    for line in src.split("\n"):
        m = re.match(r"^(\s*)int\s+(\w+)\(", line)
        if m and m.group(2) in SEARCHES:
            line = m.group(1) + "virtual " + line[len(m.group(1)):]
            n += 1
        lines.append(line)
    if n != 12:
        raise SystemExit("typecheck_adaptors: expected 12 search entry points in FeatureMatcher.h, found %d" % n)
    s = "\n".join(lines)
    anchor = "FeatureMatcher(FeatureMatcherSettings settings);"
    if anchor not in s:
        raise SystemExit("typecheck_adaptors: FeatureMatcher.h has no `%s`" % anchor)
    open(os.path.join(out_dir, "FeatureMatcher.h"), "w").write(s.replace(anchor, anchor + "\n    virtual ~FeatureMatcher() {}", 1))
    f = open(os.path.join(ref, "src", "features", "FeatureFactory.h")).read()
    decl = "std::unique_ptr<FeatureMatcher> getFeatureMatcher();"
    if decl not in f:
        raise SystemExit("typecheck_adaptors: FeatureFactory.h has no `%s`" % decl)
    open(os.path.join(out_dir, "FeatureFactory.h"), "w").write(f.replace(decl, "virtual " + decl, 1))


def compile_mode(ref, host_dir, mode, tmp):
    inc = []
    for d, _, _ in os.walk(os.path.join(ref, "src")):
        inc.append("-I" + d)
    tu = os.path.join(tmp, "tu_%s.cpp" % mode)
    body = "".join('#include "%s"\n' % h for h in HEADERS)
    flags = ["-DHYSLAM_AMD_WITH_HYSLAM"]
    first = []
    if mode == "unpatched":
        body += '#include "replace/FeatureMatcher.cc"\n'
        flags.append("-DHYSLAM_AMD_UNPATCHED_MATCHER")
    else:
        pd = os.path.join(tmp, "patched")
        os.makedirs(pd, exist_ok=True)
        patched_headers(ref, pd)
        first = ["-I" + pd]
    # touch what a caller would: the factory as hySLAM's System.cc:77-85 would make it, an extractor through the FeatureFactory interface, a matcher
    body += ("int main() {\n"
             "    std::unique_ptr<HYSLAM::FeatureFactory> f = std::make_unique<HYSLAM::HipORBFactory>(std::string(\"cfg.yaml\"));\n"
             "    std::shared_ptr<HYSLAM::FeatureExtractor> e = f->getExtractor(std::string(\"SLAM\"));\n"
             "    std::unique_ptr<HYSLAM::FeatureMatcher> m = f->getFeatureMatcher();\n"
             "    std::vector<cv::KeyPoint> k; std::vector<HYSLAM::FeatureDescriptor> d; cv::Mat img;\n"
             "    (*e)(img, cv::Mat(), k, d);\n"
             "    HYSLAM::FeatureViews v(k, k, d, d, f->getFeatureExtractorSettings()); HYSLAM::Camera cam;\n"
             "    HYSLAM::HipStereomatcher sm(v, cam, f->getFeatureMatcherSettings()); sm.computeStereoMatches(); sm.getData(v);\n"
             "    return (int)k.size() + (m ? 1 : 0);\n}\n")
    open(tu, "w").write(body)
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wno-unused", "-Wno-reorder", "-Wno-sign-compare"] + flags + first + inc + \
          ["-I" + os.path.join(ROOT, "tests", "cpp", "thirdparty_stubs"), "-I" + host_dir, "-I" + os.path.join(ROOT, "include"), tu]
    r = subprocess.run(cmd, capture_output=True, text=True)
    errs = [l for l in r.stderr.splitlines() if "error" in l]
    return r.returncode, errs, r.stderr


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ref = args[0] if args else "/root/reference"
    host_dir = os.path.join(ROOT, "hyslam_amd", "host")
    if "--host-dir" in sys.argv:
        host_dir = sys.argv[sys.argv.index("--host-dir") + 1]
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for mode in ("unpatched", "patched"):
            rc, errs, full = compile_mode(ref, host_dir, mode, tmp)
            if rc == 0:
                print("typecheck %s: ok (hyslam_amd/host/*.h%s against the real hySLAM headers, C++14)" % (mode, " + replace/FeatureMatcher.cc" if mode == "unpatched" else ", `virtual` patch applied to copies"))
            else:
                bad += 1
                print("typecheck %s: FAILED\n%s" % (mode, "\n".join(errs[:40]) or full[-3000:]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
