#!/bin/bash
OUT=gpurun_out/${1:-r4p}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so timeout 300 python3 tools/quadtree_phase_profile.py > $OUT/qt_phase.txt 2>&1; cat $OUT/qt_phase.txt
for b in 1 16; do
    timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --pairs $b --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $b:', d['value'], round(d['ms_per_step']/d['inner_repeats']*1000/$b,1), 'us/pair', d['parity_checksum_ok'], d['stage_ms_per_step'])"
done 2>&1 | tee $OUT/sweep.txt
