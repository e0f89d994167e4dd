#!/bin/bash
# Instruction budget of k_fast_rows by phase.  Here:  bash tools/fast_instr_breakdown.sh build   (libraries cut short after phase n, FR_STOP=n)
# On the GPU box:  gpurun -- 'bash tools/fast_instr_breakdown.sh run'   -> SQ counters of k_fast_rows per variant; differences = cost of a phase.
#   stop1 = tile staging + scan A masks, stop2 = + list expansion, stop3 / stop4 = + corner pass (segment test = score network), full = + score tile, NMS, emit
# Round 5 (--classes): SQ_ACTIVE_INST_VALU beside SQ_INSTS_VALU.  tools/micro/valu_peak shows two rates on gfx950 — plain integer ALU ops issue at up to twice
# the rate of everything else (min/max, mul24, perm, dot, packed, DPP, three-operand ops) — and the same run under rocprofv3 (`calib`) shows what
# SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU is for a pure stream of each class: a phase's ratio then places its dynamic mix between the two, and the
# mix-weighted ceiling follows from the class rates at the kernel's occupancy (tools/fast_class_ceiling.py prints the table for profiles/README.md).
if [ "$1" = build ]; then
  for n in 1 2 3 4; do make -s -C hyslam_amd/csrc BUILD=_build_stop$n OUT=../libhyslam_amd_stop$n.so EXTRA=-DFR_STOP=$n || exit 1; done
  exit 0
fi
OUT=gpurun_out/fast_budget
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BF="--cpu-seconds 0 --pcie-seconds 0 --call-site 0 --handles 1"
if [ "$1" = calib ]; then      # the class calibration: every op of the micro-benchmark as its own kernel
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/calib -- tools/micro/valu_peak > $OUT/calib_valu_peak.txt 2>&1
  python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_issue<(\d+)>", r["Kernel_Name"])
        if m: acc[(int(m.group(1)), int(r["Grid_Size"]) if "Grid_Size" in r else 0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("op grid  INSTS_VALU  ACTIVE_INST_VALU/INSTS  BUSY_CU/INSTS  (averages over the launches)")
for (op, grid), d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m.get("SQ_INSTS_VALU"): print(op, grid, "%.3g" % m["SQ_INSTS_VALU"], "%.3f" % (m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_INSTS_VALU"]), "%.3f" % (m.get("SQ_BUSY_CU_CYCLES", 0) / m["SQ_INSTS_VALU"]))
PY
  exit 0
fi
for v in stop1 stop2 stop3 stop4 full; do
  if [ $v = full ]; then unset HYSLAM_AMD_LIB; else export HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_$v.so; fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/$v -- python3 bench.py --steps 3 --warmup 1 --min-timed-ms 0 $BF > /dev/null 2>&1
  timeout 300 python3 bench.py --steps 50 --warmup 5 --min-timed-ms 1000 $BF 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v fast_cells ms/step (32 frames):', d['stage_ms_per_step']['fast_cells'])"
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
