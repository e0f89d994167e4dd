#!/bin/bash
# round 5: the standard-batch pyramid plan (chain lengths from level 1; default "2,2,3": two-level kernel for levels 1-2 and 3-4, a three-level chain for 5-7) against
# the alternatives at 128 frames per launch (round 4 compared them at 32)
OUT=gpurun_out/r5y; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { env "$@" timeout -k 10 200 python3 bench.py --cpu-seconds 0 --call-site 0 --pcie-seconds 0 --min-timed-ms 1500 --pairs ${PAIRS:-64} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*:', d['value'], 'pyramid', d['stage_ms_per_step']['pyramid'], 'parity', d['parity_checksum_ok'])"; }
for plan in "HS_X=0" "HS_PYRAMID_PLAN=2,2,3" "HS_PYRAMID_PLAN=2,2,2,1" "HS_PYRAMID_PLAN=2,2,1,2" "HS_PYRAMID_PLAN=2,3,2" "HS_PYRAMID_PLAN=3,2,2" "HS_PYRAMID_PLAN=1,2,2,2" "HS_PYRAMID_PLAN=2,2,1,1,1" "HS_PYRAMID_PLAN=2,2,3 HS_PYRAMID_NW8=0"; do
  one $plan | tee -a $OUT/plans.txt
done
