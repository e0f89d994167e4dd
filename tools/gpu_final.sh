#!/bin/bash
# end-of-round check on the GPU box: the whole GPU suite, smoke(), the default bench line (what the driver runs)
OUT=gpurun_out/final; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.json
