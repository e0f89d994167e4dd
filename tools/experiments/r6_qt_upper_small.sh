#!/bin/bash
# round 6: the upper pyramid levels of a large-batch call on the two-per-CU quadtree instance, in a launch of their own (HS_QT_UPPER_SMALL = first such level)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--cpu-seconds 0 --pcie-seconds 0 --call-site 0 --copy-gib 0 --latency-calls 0 --min-timed-ms 1500 --steps 20"
for k in 0 1 2 3 4 5 6; do
  HS_QT_UPPER_SMALL=$k timeout -k 10 200 python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HS_QT_UPPER_SMALL=$k: quadtree %.4f ms per 128 frames, %.0f pairs/s, parity %s' % (d['stage_ms_per_step']['quadtree'], d['value'], d['parity_checksum_ok']))"
done
