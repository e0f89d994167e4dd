#!/bin/bash
# one GPU iteration (round 5): extraction parity + fuzz slices, the default bench line (no CPU / call-site / host-fed legs), SQ counters per kernel.
#   gpurun -- 'bash tools/gpu_iter.sh TAG [pytest files...]'
TAG=${1:-it}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TESTS=${@:-tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py}
timeout 900 python3 -m pytest $TESTS -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
grep -q "pytest rc=0" $OUT/pytest.log || exit 1
timeout 300 python3 bench.py --cpu-seconds 0 --call-site 0 --pcie-seconds 0 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
timeout 300 python3 bench.py --cpu-seconds 0 --call-site 0 --pcie-seconds 0 --pairs 1 > $OUT/bench_p1.json 2>/dev/null
bash tools/sq_kernel_counters.sh --call-site 0 --pcie-seconds 0 --min-timed-ms 0 > $OUT/sq.txt 2>&1; cat $OUT/sq.txt
python3 - <<PY
import json
for f in ("bench", "bench_p1"):
    try:
        d = json.loads([l for l in open("$OUT/%s.json" % f) if l.startswith("{")][-1]); print(f, d["value"], d["stage_ms_per_step"], "parity", d.get("parity_checksum_ok"))
    except Exception as e: print(f, "ERR", e)
PY
