#!/bin/bash
# lanes inside one handle at the default workload: is the overlap of two half-batches worth anything with this round's kernels?
set -e
mkdir -p gpurun_out
for l in 1 2; do
  for p in 16 32; do
    python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --call-site 0 --pairs $p --lanes $l > gpurun_out/lanes_l${l}_p${p}.json 2> gpurun_out/lanes_l${l}_p${p}.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/lanes_l${l}_p${p}.json").read().strip().splitlines()[-1])
print("lanes $l pairs $p:", d["value"], d["ms_per_step"], d.get("parity_checksum_ok"))
PY
  done
done
