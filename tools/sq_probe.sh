#!/bin/bash
# Runs ON THE GPU BOX: SQ instruction / utilisation counters per kernel for the single-handle bench (two --pmc passes, no trace domains).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sqx; mkdir -p gpurun_out/sqx
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/sqx/a -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --handles 1 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sqx/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")[:40]
        if k.startswith("k_"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel launches VALU SALU LDS waves busy_cycles/SE VALU_util wait_frac")
for k,dd in agg.items():
    m={c: sum(v)/len(v) for c,v in dd.items()}
    busy=m["SQ_BUSY_CYCLES"]/32
    print("%-22s %3d %6.1fM %6.1fM %6.1fM %7d %9.0f   %4.0f%%   %4.0f%%"%(k,len(dd["SQ_WAVES"]),m["SQ_INSTS_VALU"]/1e6,m["SQ_INSTS_SALU"]/1e6,m["SQ_INSTS_LDS"]/1e6,m["SQ_WAVES"],busy,100*m["SQ_INSTS_VALU"]/256/busy,100*m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"]))
PY
