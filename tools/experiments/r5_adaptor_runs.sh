#!/bin/bash
# round 5: the call-site bench (tests/cpp/bench_adaptor) several times — helper threads following the caller's L3 domain (default) / not (HYSLAM_AMD_PIN_HELPERS=0) /
# the whole process confined with taskset: the run-to-run spread of TrackLocalMap's gather is thread placement on the two-socket host
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
lscpu | grep -E "NUMA node[0-9]|Socket|Model name" | head -4
python3 - <<PY
import sys, json, subprocess, os
sys.path.insert(0, "tests")
import test_adaptor as t
t.build("bench_adaptor.cpp", t.EXE_B)
from hyslam_amd.synth import synth_stereo_pair
L, R = synth_stereo_pair(2, 1920, 1080)
fl, fr = os.path.join(t.BUILD, "bench_L.raw"), os.path.join(t.BUILD, "bench_R.raw")
L.tofile(fl); R.tofile(fr)
for name, pre, env in (("helpers follow the caller", [], {}), ("HYSLAM_AMD_PIN_HELPERS=0", [], {"HYSLAM_AMD_PIN_HELPERS": "0"}), ("taskset -c 0-7", ["taskset", "-c", "0-7"], {}), ("helpers follow the caller", [], {})):
    for k in range(4):
        e = dict(os.environ); e.update(env)
        r = subprocess.run(pre + [t.EXE_B, "1920", "1080", fl, fr, "30", "50000"], capture_output=True, timeout=600, env=e)
        try:
            j = json.loads(r.stdout.decode()); tl = j["TrackLocalMap_SearchByProjection_ms"]
            print(name, "| TrackLocalMap", tl["total"], "gather", tl["gather"], "| ProcessStereoImage", j["ProcessStereoImage_ms"]["total"])
        except Exception as ex:
            print(name, "failed", r.stderr.decode()[-200:])
PY
