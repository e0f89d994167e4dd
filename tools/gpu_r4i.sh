#!/bin/bash
# round 4: deep chain with 16 waves per workgroup at one pair per call; ticket tests; padded records
OUT=gpurun_out/${1:-r4i}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ingest.py tests/test_comm.py tests/test_gpu_matchers.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
for v in "HS_PYRAMID_DEEP_MAX=0" "HS_PYRAMID_NW16=0" "HS_PYRAMID_NW16=1"; do
    env $v timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --pairs 1 --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs 1 $v:', d['value'], round(d['ms_per_step']/d['inner_repeats']*1000,1), 'us/pair', d['parity_checksum_ok'], d['stage_ms_per_step'])"
done 2>&1 | tee $OUT/sweep.txt
