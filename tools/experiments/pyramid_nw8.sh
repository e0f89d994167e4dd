#!/bin/bash
# 8 waves per workgroup for the pyramid kernels at 16 pairs: HS_PYRAMID_NW8 = workgroups-per-CU threshold (3 = default)
set -e
mkdir -p gpurun_out/pyr_nw8 && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nw in 3 10 20 50; do
  export HS_PYRAMID_NW8=$nw
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pyr_nw8/nw$nw -o t -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --call-site 0 > $R/gpurun_out/pyr_nw8/nw$nw.json 2> $R/gpurun_out/pyr_nw8/nw$nw.err
  echo "== NW8=$nw"; python3 - <<PY
import csv,glob,json
print(json.loads(open("$R/gpurun_out/pyr_nw8/nw$nw.json").read().strip().splitlines()[-1])["value"])
f=glob.glob("$R/gpurun_out/pyr_nw8/nw$nw/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("k_"): print("  %-28s %4s %9.1f" % (r["Name"][:28], r["Calls"], float(r["AverageNs"])/1000))
PY
done
