#!/bin/bash
# packed-lane chain kernel: parity variants, then pyramid time at 16 pairs for packed 0 / 1 / 2
set -e
mkdir -p gpurun_out/r4t
export PYTHONPATH=$PWD
for v in "HS_PYRAMID_PACKED=1" "HS_PYRAMID_PACKED=2" "HS_PYRAMID_PACKED=1 HS_PYRAMID_PLAN=3,4" "HS_PYRAMID_PACKED=1 HS_PYRAMID_PLAN=1,2,3,1" "HS_PYRAMID_PACKED=0 HS_PYRAMID_PLAN=1,2,3,1"; do
  env $v HS_PYRAMID_DEEP_MAX=0 timeout -k 10 300 python tests/_fast_variant_check.py > gpurun_out/r4t/v.log 2>&1 || { echo "variant $v FAILED"; tail -15 gpurun_out/r4t/v.log; exit 1; }
  echo "variant $v: $(tail -1 gpurun_out/r4t/v.log)"
done
for pk in 0 1 2; do
  export HS_PYRAMID_PACKED=$pk
  python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --call-site 0 --pcie-seconds 0 > gpurun_out/r4t/pk$pk.json 2> gpurun_out/r4t/pk$pk.err || { echo "packed $pk FAILED"; tail -3 gpurun_out/r4t/pk$pk.err; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4t/pk$pk.json").read().strip().splitlines()[-1])
print("packed $pk value %8.1f parity %s pyramid %.4f ms" % (d["value"], d.get("parity_checksum_ok"), d["stage_ms_per_step"]["pyramid"]))
PY
done
