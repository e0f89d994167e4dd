#!/bin/bash
# per-launch durations and grids of the pyramid kernels, packed lanes 0 / 1 / 2
set -e
mkdir -p gpurun_out/r4u && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pk in 0 1 2; do
  export HS_PYRAMID_PACKED=$pk
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4u/pk$pk -o t -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --call-site 0 --pcie-seconds 0 --min-timed-ms 0 --profile-steps 0 > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/r4u/pk$pk/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "k_resize" in n:
        key=(n.split("(")[0][:34], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"], r.get("LDS_Block_Size",""))
        d[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
print("== packed $pk")
for k,v in d.items():
    v=sorted(v); print("  %-36s grid %3s x %3s x %3s lds %6s  n %3d  median %.1f us" % (k[0],k[1],k[2],k[3],k[4],len(v),v[len(v)//2]))
PY
done
