#!/bin/bash
# round 4: fused stereo (strips in the describe launch, median by the pair's last matcher workgroup) + narrow FAST items: parity, then bench
OUT=gpurun_out/${1:-r4c}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_ingest.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for b in 1 2 4 16; do
  for f in 1 0; do
    HS_STEREO_FUSE=$f timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --pairs $b --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $b fuse $f:', d['value'], d['ms_per_step'], d['inner_repeats'], d['stage_ms_per_step'])"
  done
done 2>&1 | tee $OUT/sweep.txt
bash tools/kernel_timeline.sh --pairs 1 > $OUT/kt1.txt 2>&1; cat $OUT/kt1.txt
