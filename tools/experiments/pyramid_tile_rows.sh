#!/bin/bash
# tile height of the fused pyramid kernels at 16 pairs: FZ_ROWS = 8 / 16 / 24 (variant libraries)
mkdir -p gpurun_out/pyr_rows
for v in "" _fz8 _fz24; do
  export HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd$v.so
  python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --call-site 0 --pcie-seconds 0 > gpurun_out/pyr_rows/b$v.json 2> gpurun_out/pyr_rows/b$v.err || { echo "variant '$v' FAILED"; tail -3 gpurun_out/pyr_rows/b$v.err; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/pyr_rows/b$v.json").read().strip().splitlines()[-1])
print("variant '%s' value %8.1f parity %s pyramid %.4f ms" % ("$v", d["value"], d.get("parity_checksum_ok"), d["stage_ms_per_step"]["pyramid"]))
PY
done
