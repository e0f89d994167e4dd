#!/bin/bash
# SQ counters per kernel for one short bench run (GPU box):  gpurun -- 'bash tools/sq_kernel_counters.sh [bench args]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sqk; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $OUT/a -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
        if k.startswith("k_"): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print("%-28s VALU %6.2fM SALU %6.2fM LDS %5.2fM | per CU: busy %7.0f cyc, VALU/busy %.2f, LDS active %.2f (conflicts %.2f), launches %d" % (
        k, m["SQ_INSTS_VALU"] / 1e6, m["SQ_INSTS_SALU"] / 1e6, m["SQ_INSTS_LDS"] / 1e6, m["SQ_BUSY_CU_CYCLES"] / 256,
        m["SQ_INSTS_VALU"] / m["SQ_BUSY_CU_CYCLES"], m["SQ_LDS_IDX_ACTIVE"] / m["SQ_BUSY_CU_CYCLES"], m["SQ_LDS_BANK_CONFLICT"] / m["SQ_BUSY_CU_CYCLES"], len(d["SQ_INSTS_VALU"])))
PY
