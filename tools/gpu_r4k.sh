#!/bin/bash
# round 4: k_resize_chain with software-pipelined per-stage fetches
OUT=gpurun_out/${1:-r4k}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
for v in "HS_PYRAMID_DEEP_MAX=0" "HS_PYRAMID_DEEP_MAX=2"; do
  for b in 1 16; do
    env $v timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --pairs $b --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $b $v:', d['value'], round(d['ms_per_step']/d['inner_repeats']*1000/$b,1), 'us/pair', d['parity_checksum_ok'], d['stage_ms_per_step'])"
  done
done 2>&1 | tee $OUT/sweep.txt
bash tools/kernel_timeline.sh --pairs 1 --min-timed-ms 0 --pcie-seconds 0 --call-site 0 > $OUT/kt1.txt 2>&1; cat $OUT/kt1.txt
