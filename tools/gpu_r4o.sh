#!/bin/bash
# round 4: adaptor changes (parallel scatter / gather) on the GPU + the bench's call_site block
OUT=gpurun_out/${1:-r4o}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 900 python3 -m pytest tests/test_adaptor.py tests/test_gpu_ingest.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --steps 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], json.dumps(d['call_site']))"
