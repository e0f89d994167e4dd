#!/bin/bash
# round 4: the whole GPU suite on the round's kernels, then handles / lanes at 16 pairs
OUT=gpurun_out/${1:-r4j}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1700 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for v in "--handles 1" "--handles 2" "--lanes 2" "--pairs 64" "--pairs 32"; do
    timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --steps 100 $v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v:', d['value'], d['parity_checksum_ok'], d['stage_ms_per_step'])"
done 2>&1 | tee $OUT/sweep.txt
