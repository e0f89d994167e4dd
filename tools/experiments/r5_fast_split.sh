#!/bin/bash
# round 5: the two-pass wide kernel (k_fast_rows<6, 24, K, 2>: half-height tile, 10 240 bytes of LDS, 128 VGPRs) against the one-pass one (HS_FAST_SPLIT=0) by workgroups per CU
OUT=gpurun_out/r5u; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { env "$@" timeout -k 10 200 python3 bench.py --cpu-seconds 0 --call-site 0 --pcie-seconds 0 --min-timed-ms 1500 --pairs ${PAIRS:-64} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*:', d['value'], 'fast_cells', d['stage_ms_per_step']['fast_cells'], 'parity', d['parity_checksum_ok'])"; }
for cfg in "HS_FAST_SPLIT=1" "HS_FAST_SPLIT=1 HS_FAST_WG_PER_CU=12" "HS_FAST_SPLIT=1 HS_FAST_WG_PER_CU=8" "HS_FAST_SPLIT=0" "HS_FAST_SPLIT=0 HS_FAST_WG_PER_CU=8"; do
  one $cfg | tee -a $OUT/split.txt
done
