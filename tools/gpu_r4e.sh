#!/bin/bash
# round 4: bench.py's new self-checks on the GPU (default line incl. parity checksum, call_site, 200 ms window), driver-style 20-step run, C3/C4/C5
OUT=gpurun_out/${1:-r4e}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 300 python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --pcie-seconds 0 --call-site 0 > $OUT/bench_20.json 2>/dev/null; python3 -c "import json; d=json.loads(open('$OUT/bench_20.json').read().strip().splitlines()[-1]); print('20 steps:', d['value'], d['ms_per_step'], d['inner_repeats'], d['timed_region_ms'], d['parity_checksum_ok'])"
timeout 300 python3 bench.py --config c5 --steps 50 --warmup 5 > $OUT/c5.json 2> $OUT/c5.err; tail -c 1500 $OUT/c5.json; tail -3 $OUT/c5.err
timeout 300 python3 bench.py --config c3 --steps 20 --warmup 3 2>/dev/null | tail -c 600
timeout 300 python3 bench.py --config c4 --steps 30 --warmup 3 2>/dev/null | tail -c 900
timeout 600 python3 -m pytest tests/test_adaptor.py tests/test_comm.py -m gpu -q -x 2>&1 | tail -3
