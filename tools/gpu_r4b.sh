#!/bin/bash
# round 4: narrow items + folded static schedule for small launches — parity first, then the batch sweep with both widths forced
OUT=gpurun_out/${1:-r4b}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_fwav.so timeout 300 python3 tools/fast_b1_timeline.py 2 > $OUT/fast_b1_auto.txt 2>&1
cat $OUT/fast_b1_auto.txt
for b in 1 2 3 4 8 16; do
  for c in auto 32 64; do
    if [ $c = auto ]; then unset HS_FAST_COLS; else export HS_FAST_COLS=$c; fi
    timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --pairs $b --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $b cols $c:', d['value'], d['ms_per_step'], d['inner_repeats'], d['stage_ms_per_step']['fast_cells'], d['stage_ms_per_step']['quadtree'])"
  done
done 2>&1 | tee $OUT/sweep.txt
unset HS_FAST_COLS
HS_FAST_NO_FOLD=1 timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --pairs 1 --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs 1 nofold:', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" | tee -a $OUT/sweep.txt
