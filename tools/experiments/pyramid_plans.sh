#!/bin/bash
# pyramid launch plans at 16 pairs (HS_PYRAMID_PLAN: chain lengths from level 1)
set -e
mkdir -p gpurun_out/pyr_plans
for plan in default 2,2,3 3,2,2 2,3,2 3,4 4,3 2,5 2,2,2,1 3,3,1 7; do
  if [ $plan = default ]; then unset HS_PYRAMID_PLAN; else export HS_PYRAMID_PLAN=$plan; fi
  python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --call-site 0 --pcie-seconds 0 > gpurun_out/pyr_plans/p_$plan.json 2> gpurun_out/pyr_plans/p_$plan.err || { echo "plan $plan FAILED"; tail -3 gpurun_out/pyr_plans/p_$plan.err; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/pyr_plans/p_$plan.json").read().strip().splitlines()[-1])
print("plan %-8s value %8.1f parity %s pyramid %.4f ms" % ("$plan", d["value"], d.get("parity_checksum_ok"), d["stage_ms_per_step"]["pyramid"]))
PY
done
