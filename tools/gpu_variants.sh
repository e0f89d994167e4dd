#!/bin/bash
# bench.py under a list of environment variants (GPU box):  bash tools/gpu_variants.sh TAG "VAR=1" "VAR2=x --pairs 1" ...
# each argument = environment assignments followed by optional bench.py flags; prints value + per-stage ms
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for spec in "base" "$@"; do
  envs=""; flags=""
  if [ "$spec" != base ]; then for tok in $spec; do case $tok in *=*) envs="$envs $tok";; *) flags="$flags $tok";; esac; done; fi
  env $envs timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 $flags 2>/dev/null > $OUT/v$i.json
  python3 - "$spec" $OUT/v$i.json <<PY
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1]); print("%-40s %9.1f %s" % (sys.argv[1], d["value"], d["stage_ms_per_step"]))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
  i=$((i+1))
done
