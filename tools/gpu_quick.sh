#!/bin/bash
# quick GPU iteration: extraction parity tests, bench (no CPU leg), FAST / quadtree phase profiles when the instrumented builds are present
OUT=gpurun_out/${1:-q}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
timeout 300 python3 bench.py --cpu-seconds 0 > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python3 bench.py --cpu-seconds 0 --handles 1 > $OUT/bench_h1.json 2>/dev/null
timeout 300 python3 bench.py --cpu-seconds 0 --pairs 1 > $OUT/bench_p1.json 2>/dev/null
[ -f hyslam_amd/libhyslam_amd_fprof.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_fprof.so timeout 300 python3 tools/fast_phase_profile.py > $OUT/fast_phase.txt 2>&1
[ -f hyslam_amd/libhyslam_amd_qprof.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so timeout 300 python3 tools/quadtree_phase_profile.py > $OUT/qt_phase.txt 2>&1
python3 - <<PY
import json
for f in ("bench","bench_h1","bench_p1"):
    try:
        d=json.loads([l for l in open("$OUT/%s.json"%f).read().splitlines() if l.startswith("{")][-1]); print(f, d["value"], d["stage_ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
cat $OUT/fast_phase.txt 2>/dev/null
[ -n "$2" ] && timeout 120 tools/micro/valu_peak > $OUT/valu_peak.txt 2>&1 && cat $OUT/valu_peak.txt
tail -3 $OUT/bench.err
for v in nostore x1 x2; do
  [ -f hyslam_amd/libhyslam_amd_$v.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_$v.so timeout 300 python3 bench.py --cpu-seconds 0 --handles 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v (1 handle):', d['value'], d['stage_ms_per_step'])"
done
HS_PYRAMID_NO_FUSE=1 timeout 300 python3 bench.py --cpu-seconds 0 --handles 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused pyramid (1 handle):', d['value'], d['stage_ms_per_step'])"
HS_PYRAMID_NO_FUSE=1 timeout 300 python3 bench.py --cpu-seconds 0 --pairs 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused pyramid (pairs 1):', d['value'], d['stage_ms_per_step'])"
