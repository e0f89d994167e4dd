#!/bin/bash
# first GPU pass of round 2: full -m gpu suite, the bench configs, the VALU issue micro-benchmark, baseline kernel stats
OUT=gpurun_out/r2a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 120 tools/micro/valu_peak > $OUT/valu_peak.txt 2>&1
timeout 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python3 bench.py --config c3 --steps 20 --warmup 3 > $OUT/c3.json 2>$OUT/c3.err
timeout 300 python3 bench.py --config c4 --steps 30 --warmup 3 > $OUT/c4.json 2>$OUT/c4.err
timeout 300 python3 bench.py --config c5 --steps 50 --warmup 5 > $OUT/c5.json 2>$OUT/c5.err
timeout 300 python3 bench.py --pairs 1 --steps 50 --warmup 5 --cpu-seconds 0 > $OUT/bench_p1.json 2>/dev/null
timeout 300 python3 bench.py --handles 1 --steps 30 --warmup 5 --cpu-seconds 0 > $OUT/bench_h1.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 > $OUT/bench_under_rocprof.json 2>/dev/null
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
timeout 300 python3 tools/bench_pcie.py > $OUT/pcie.json 2>/dev/null
head -c 600 $OUT/bench.json; echo; cat $OUT/c4.json | head -c 800; echo; tail -3 $OUT/c5.err; head -30 $OUT/valu_peak.txt
