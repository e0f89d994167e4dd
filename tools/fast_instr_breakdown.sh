#!/bin/bash
# Instruction budget of k_fast_rows by phase.  Here:  bash tools/fast_instr_breakdown.sh build   (libraries cut short after phase n, FR_STOP=n)
# On the GPU box:  gpurun -- 'bash tools/fast_instr_breakdown.sh run'   -> SQ counters of k_fast_rows per variant; differences = cost of a phase.
if [ "$1" = build ]; then
  for n in 1 2 3 4; do make -s -C hyslam_amd/csrc BUILD=_build_stop$n OUT=../libhyslam_amd_stop$n.so EXTRA=-DFR_STOP=$n || exit 1; done
  exit 0
fi
OUT=gpurun_out/fast_budget
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in stop1 stop2 stop3 stop4 full; do
  if [ $v = full ]; then unset HYSLAM_AMD_LIB; else export HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_$v.so; fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/$v -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --handles 1 > /dev/null 2>&1
  timeout 300 python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0 --handles 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v fast_cells ms/step (32 frames):', d['stage_ms_per_step']['fast_cells'])"
done
python3 - <<PY
import csv, glob, collections
for v in ("stop1","stop2","stop3","stop4","full"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fast_rows" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: round(sum(x)/len(x)/1e6, 2) for k, x in sorted(acc.items())}, "M per launch (32 frames)")
PY
