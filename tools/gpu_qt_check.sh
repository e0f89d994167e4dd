#!/bin/bash
# quadtree iteration: its parity tests, the fuzz slice, bench at 16 pairs / 1 pair / C4, the phase profile when the instrumented build is there
OUT=gpurun_out/${1:-qt}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
for a in "" "--pairs 1" "--config c4 --steps 60"; do
  timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 $a 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', d['value'], d.get('stage_ms_per_step', d.get('stage_ms')))"
done
[ -f hyslam_amd/libhyslam_amd_qprof.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so timeout 300 python3 tools/quadtree_phase_profile.py > $OUT/qt_phase.txt 2>&1 && cat $OUT/qt_phase.txt
[ -f hyslam_amd/libhyslam_amd_qprof.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so timeout 300 python3 tools/quadtree_phase_profile.py 4000 3000 3000 1.4 > $OUT/qt_phase_4k.txt 2>&1 && cat $OUT/qt_phase_4k.txt
