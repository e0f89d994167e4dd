#!/bin/bash
# round 3, first GPU pass: the new adaptor / comm / ingest / C5 tests first, then the whole GPU suite, the adaptor call-site timings,
# bench.py and the host-fed rates.  Everything lands in gpurun_out/$1.
OUT=gpurun_out/${1:-r3a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_adaptor.py tests/test_comm.py tests/test_gpu_ingest.py -m gpu -q -x > $OUT/pytest_new.log 2>&1; echo "rc=$?" >> $OUT/pytest_new.log
tail -15 $OUT/pytest_new.log
timeout 1500 python3 -m pytest tests -m gpu -q -x --deselect tests/test_adaptor.py --deselect tests/test_comm.py --deselect tests/test_gpu_ingest.py > $OUT/pytest_all.log 2>&1; echo "rc=$?" >> $OUT/pytest_all.log
tail -8 $OUT/pytest_all.log
python3 - <<PY
from hyslam_amd.synth import synth_stereo_pair
L, R = synth_stereo_pair(2, 1920, 1080)
L.tofile("$OUT/L.raw"); R.tofile("$OUT/R.raw")
PY
timeout 600 tests/cpp/_build/bench_adaptor 1920 1080 $OUT/L.raw $OUT/R.raw 30 50000 > $OUT/adaptor_bench.json 2> $OUT/adaptor_bench.err
rm -f $OUT/L.raw $OUT/R.raw
cat $OUT/adaptor_bench.json
timeout 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-600; tail -3 $OUT/bench.err
timeout 300 python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 > $OUT/bench_20.json 2>/dev/null; tail -1 $OUT/bench_20.json | cut -c1-300
timeout 300 python3 tools/bench_pcie.py > $OUT/pcie.json 2> $OUT/pcie.err; cat $OUT/pcie.json; tail -3 $OUT/pcie.err
timeout 300 python3 bench.py --config c5 --steps 50 --warmup 5 > $OUT/c5.json 2> $OUT/c5.err; tail -1 $OUT/c5.json | cut -c1-400; tail -3 $OUT/c5.err
