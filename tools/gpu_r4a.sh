#!/bin/bash
# round 4, first GPU pass: batch-1 anatomy of FAST (phase stamps per wave), quadtree phase profile, baseline bench lines
OUT=gpurun_out/${1:-r4a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_fwav.so timeout 300 python3 tools/fast_b1_timeline.py 2 > $OUT/fast_b1_lc6.txt 2>&1
HS_FAST_COLS=32 HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_fwav.so timeout 300 python3 tools/fast_b1_timeline.py 2 > $OUT/fast_b1_lc5.txt 2>&1
HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_fwav.so timeout 300 python3 tools/fast_b1_timeline.py 32 > $OUT/fast_b16_lc6.txt 2>&1
HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so timeout 300 python3 tools/quadtree_phase_profile.py > $OUT/qt_phase.txt 2>&1
timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --pairs 1 > $OUT/bench_p1.json 2> $OUT/bench_p1.err
timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/fast_b1_lc6.txt $OUT/fast_b1_lc5.txt
python3 - <<PY
import json
for f in ("bench","bench_p1"):
    try:
        d=json.loads([l for l in open("$OUT/%s.json"%f).read().splitlines() if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d["inner_repeats"], d["stage_ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
