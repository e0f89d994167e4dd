#!/bin/bash
# round 5: the extractor adaptor's descriptor scatter with 0 / 1 / 2 helper threads (HYSLAM_AMD_SCATTER_THREADS; helpers follow their caller's L3 domain), free placement and the process confined to one L3 domain
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 - <<PY
import sys, json, subprocess, os
sys.path.insert(0, "tests")
import test_adaptor as t
t.build("bench_adaptor.cpp", t.EXE_B)
from hyslam_amd.synth import synth_stereo_pair
L, R = synth_stereo_pair(2, 1920, 1080)
fl, fr = os.path.join(t.BUILD, "bench_L.raw"), os.path.join(t.BUILD, "bench_R.raw")
L.tofile(fl); R.tofile(fr)
for pre in ([], ["taskset", "-c", "0-7,128-135"]):
  for n in ("default", "0", "1", "2"):
    for k in range(3):
        e = dict(os.environ)
        if n != "default": e["HYSLAM_AMD_SCATTER_THREADS"] = n
        r = subprocess.run(pre + [t.EXE_B, "1920", "1080", fl, fr, "30", "50000"], capture_output=True, timeout=600, env=e)
        j = json.loads(r.stdout.decode()); p = j["ProcessStereoImage_ms"]
        print(" ".join(pre) or "free", "scatter helpers", n, "| ProcessStereoImage", p["total"], "extract_LR", p["extract_LR_threads"], "scatter", p["extract_scatter"], "| frontend", j["HipStereoFrontend_ms"]["process_total"], "| TrackLocalMap", j["TrackLocalMap_SearchByProjection_ms"]["total"])
PY
