#!/bin/bash
# diagnostic: pyramid launches without their source fetch (garbage results) vs the real ones
set -e
mkdir -p gpurun_out/pyr_noload && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in real noload; do
  if [ $v = noload ]; then export HYSLAM_AMD_LIB=$R/hyslam_amd/libhyslam_amd_pnoload.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pyr_noload/$v -o t -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --call-site 0 --pcie-seconds 0 --min-timed-ms 0 --profile-steps 0 > $R/gpurun_out/pyr_noload/$v.out 2>&1 || true
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pyr_noload/$v/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "k_resize" in n:
        key=(n.split("(")[0][:34], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"])
        d[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
print("== $v")
for k,v in d.items():
    v=sorted(v); print("  %-36s grid %3s x %3s x %3s  n %3d  median %.1f us" % (k[0],k[1],k[2],k[3],len(v),v[len(v)//2]))
PY
done
