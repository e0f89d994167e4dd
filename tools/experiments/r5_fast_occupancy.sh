#!/bin/bash
# round 5: what occupancy buys k_fast_rows.  The wide items (256 px x 34 rows, 14 KB of LDS) run 11 single-wave workgroups per CU (2.75 waves per SIMD), the
# narrow ones (128 px, <= 10 KB) 16 (4 per SIMD).  Same kernel, same frames: narrow items at 16 / 12 / 11 / 8 workgroups per CU separate the occupancy
# effect from the geometry effect (narrow items scan 5 % more halo columns and are twice as many); wide items at 11 / 8 / 6 give the other half of the curve.
OUT=gpurun_out/r5o; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { env "$@" timeout -k 10 200 python3 bench.py --cpu-seconds 0 --call-site 0 --pcie-seconds 0 --min-timed-ms 1500 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*:', d['value'], 'fast_cells', d['stage_ms_per_step']['fast_cells'], 'parity', d['parity_checksum_ok'])"; }
for cfg in "HS_FAST_COLS=64" "HS_FAST_COLS=64 HS_FAST_WG_PER_CU=8" "HS_FAST_COLS=64 HS_FAST_WG_PER_CU=6" "HS_FAST_COLS=32" "HS_FAST_COLS=32 HS_FAST_WG_PER_CU=12" "HS_FAST_COLS=32 HS_FAST_WG_PER_CU=11" "HS_FAST_COLS=32 HS_FAST_WG_PER_CU=8"; do
  one $cfg | tee -a $OUT/occupancy.txt
done
