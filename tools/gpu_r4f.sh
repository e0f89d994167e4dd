#!/bin/bash
# round 4: geometric keys computed by the FAST kernel (histogram + best candidate per key in global memory), quadtree without the gather
OUT=gpurun_out/${1:-r4f}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_ingest.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so timeout 300 python3 tools/quadtree_phase_profile.py > $OUT/qt_phase.txt 2>&1; cat $OUT/qt_phase.txt
for b in 1 4 16; do
  for f in 1 0; do
    HS_FAST_KEYS=$f timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --pairs $b --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $b keys $f:', d['value'], d['ms_per_step'], d['inner_repeats'], d['parity_checksum_ok'], d['stage_ms_per_step'])"
  done
done 2>&1 | tee $OUT/sweep.txt
