#!/bin/bash
# round 4: keys for the largest levels only; faster pre-mode set-up and cell sweep in the quadtree
OUT=gpurun_out/${1:-r4g}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so timeout 300 python3 tools/quadtree_phase_profile.py > $OUT/qt_phase.txt 2>&1; cat $OUT/qt_phase.txt
for b in 1 16; do
  for k in 0 1 2 3 8; do
    if [ $k = 0 ]; then export HS_FAST_KEYS=0; else export HS_FAST_KEYS=1 HS_FAST_KEYS_LEVELS=$k; fi
    timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --pairs $b --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $b key levels $k:', d['value'], d['ms_per_step'], d['inner_repeats'], d['parity_checksum_ok'], d['stage_ms_per_step'])"
  done
done 2>&1 | tee $OUT/sweep.txt
