#!/bin/bash
# Where k_fast_rows' cycles go: SQ "active" cycle counters per instruction class (two --pmc passes).  GPU box:  bash tools/sq_fast_detail.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sqd; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --pcie-seconds 0 "$@" > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_INT32 SQ_INSTS --output-format csv -d $OUT/b -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --pcie-seconds 0 "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]
        if k.startswith("k_"): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, " ".join("%s=%.2fM" % (c.replace("SQ_", ""), v / 1e6) for c, v in sorted(m.items())))
PY
