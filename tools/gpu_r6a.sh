#!/bin/bash
# round 6, first GPU iteration: the new parity tests (bench launch shape, init extractors), the whole GPU suite, the default bench line with the new fields
OUT=gpurun_out/r6a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "init_extractors or quota_beyond" > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -3 $OUT/pytest_new.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "bench_launch_shape" > $OUT/pytest_shape.log 2>&1; echo "shape test rc=$?"; tail -3 $OUT/pytest_shape.log
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_all.log 2>&1; echo "all rc=$?"; tail -3 $OUT/pytest_all.log
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.err
timeout -k 10 300 python3 tools/fuzz_parity.py --stress --cases 120 --seed 901 > $OUT/fuzz901.log 2>&1; echo "fuzz rc=$?"; tail -8 $OUT/fuzz901.log
