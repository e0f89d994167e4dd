#!/bin/bash
# round 5: what a shorter pixel list costs k_fast_rows at EQUAL occupancy (HS_FAST_PCAP only moves the list end; the LDS grant stays 14 080 bytes = 11 per CU):
# the price side of trading list capacity for a twelfth workgroup per CU
OUT=gpurun_out/r5p; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { env "$@" timeout -k 10 200 python3 bench.py --cpu-seconds 0 --call-site 0 --pcie-seconds 0 --min-timed-ms 1500 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*:', d['value'], 'fast_cells', d['stage_ms_per_step']['fast_cells'], 'parity', d['parity_checksum_ok'])"; }
for cfg in "HS_X=0" "HS_FAST_PCAP=800" "HS_FAST_PCAP=624" "HS_FAST_PCAP=512" "HS_FAST_PCAP=384"; do
  one $cfg | tee -a $OUT/pcap.txt
done
