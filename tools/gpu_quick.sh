#!/bin/bash
# quick GPU check of a kernel change: the parity suites, then the bench line's stage times at several batch sizes (HS_* environment passes through)
OUT=gpurun_out/${1:-quick}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for p in ${PAIRS:-1 4 16 64}; do
  timeout -k 10 200 python3 bench.py --cpu-seconds 0 --call-site 0 --pcie-seconds 0 --min-timed-ms 1500 --pairs $p 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $p:', d['value'], d['stage_ms_per_step'], 'parity', d['parity_checksum_ok'])" | tee -a $OUT/bench.txt
done
