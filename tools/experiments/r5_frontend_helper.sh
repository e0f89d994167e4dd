cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_adaptor.py -m gpu -q -x 2>&1 | tail -1
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
  for n in ("default", "0"):
    for k in range(3):
        e = dict(os.environ)
        if n != "default": e["HYSLAM_AMD_SCATTER_THREADS"] = n
        r = subprocess.run(pre + [t.EXE_B, "1920", "1080", fl, fr, "30", "5000"], capture_output=True, timeout=600, env=e)
        j = json.loads(r.stdout.decode()); f = j["HipStereoFrontend_ms"]
        print(" ".join(pre) or "free", "helpers", n, "| frontend", f, "| ProcessStereoImage", j["ProcessStereoImage_ms"]["total"])
PY
