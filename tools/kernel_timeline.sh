#!/bin/bash
# Kernel timeline of the last steps of a bench run without stage events (GPU box):  gpurun -- 'bash tools/kernel_timeline.sh --pairs 1'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kt; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --profile-steps 0 "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0].replace("void ","")
        if n.startswith("k_"): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:24]))
rows.sort()
first=[i for i,r in enumerate(rows) if r[2].startswith("k_resize")]          # a step starts with the first pyramid launch (k_resize_two_levels, or k_resize_chain alone in the small-batch plan)
starts=[i for i in first if i==0 or not rows[i-1][2].startswith("k_resize")]
a,b=starts[-3],starts[-2]
t0=rows[a][0]; prev=None; busy=0
for s,e,n in rows[a:b]:
    print("%-26s start %7.1f us  dur %6.1f us  gap %5.1f" % (n,(s-t0)/1e3,(e-s)/1e3,0 if prev is None else (s-prev)/1e3)); prev=e; busy+=e-s
print("step period %.1f us, kernels busy %.1f us" % ((rows[b][0]-t0)/1e3, busy/1e3))
PY
