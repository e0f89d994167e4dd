#!/bin/bash
# round 6: a quadtree instance of 512 list entries + 1024 points in LDS (76.7 KB: two workgroups per CU) for large batches (HS_QT_MID=1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--cpu-seconds 0 --pcie-seconds 0 --call-site 0 --copy-gib 0 --latency-calls 0 --min-timed-ms 1500 --steps 20"
for v in "HS_QT_MID=0" "HS_QT_MID=1" "HS_QT_SMALL=1"; do
  for p in 32 64; do
    env $v timeout -k 10 200 python3 bench.py $Q --pairs $p 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v pairs $p: quadtree %.4f ms, %.0f pairs/s, parity %s' % (d['stage_ms_per_step']['quadtree'], d['value'], d['parity_checksum_ok']))"
  done
done
