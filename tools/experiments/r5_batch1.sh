#!/bin/bash
# round 5, GPU batch 1: new parity cases, the quadtree's two-per-CU instance, scatter-thread variants of the extractor adaptor, the box's CPU share,
# the extended issue-rate table, describe LDS pitches
OUT=gpurun_out/r5f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1000 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_ingest.py tests/test_adaptor.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for b in 16 32 64; do for q in 1 0; do
  HS_QT_SMALL=$q timeout 200 python3 bench.py --cpu-seconds 0 --call-site 0 --pcie-seconds 0 --min-timed-ms 1500 --pairs $b 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs $b HS_QT_SMALL=$q:', d['value'], d['stage_ms_per_step'], 'parity', d['parity_checksum_ok'])"
done; done
( echo "nproc $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "v1 quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1)"; grep Cpus_allowed_list /proc/self/status; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)" ) > $OUT/cpu_share.txt 2>&1; cat $OUT/cpu_share.txt
for t in 0 1 2; do HYSLAM_AMD_SCATTER_THREADS=$t python3 - <<PY
import sys, json
sys.path.insert(0, "tests")
import test_adaptor as t
r = t.run_bench(1920, 1080, 40, 50000)
open("$OUT/adaptor_scatter$t.json", "w").write(r.stdout.decode())
try:
    j = json.loads(r.stdout.decode()); print("scatter helpers $t", j["ProcessStereoImage_ms"], "| TrackLocalMap", j["TrackLocalMap_SearchByProjection_ms"]["total"])
except Exception as e: print(r.stdout.decode()[-500:], r.stderr.decode()[-300:])
PY
done
timeout 300 tools/micro/valu_peak > $OUT/valu_peak.txt 2>&1; tail -3 $OUT/valu_peak.txt
for v in "" hp50 hp54 bp48 bp56; do
  if [ -z "$v" ]; then unset HYSLAM_AMD_LIB; else export HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_$v.so; fi
  timeout 200 python3 bench.py --cpu-seconds 0 --call-site 0 --pcie-seconds 0 --min-timed-ms 1500 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('describe variant [$v]:', d['value'], d['stage_ms_per_step']['describe'], 'parity', d['parity_checksum_ok'])"
done
unset HYSLAM_AMD_LIB
timeout 400 python3 bench.py --min-timed-ms 2000 --call-site 0 --pcie-seconds 0 > $OUT/bench_cpu.json 2>/dev/null; python3 -c "
import json; d=json.loads([l for l in open('$OUT/bench_cpu.json') if l.startswith('{')][-1]); print(json.dumps(d['cpu_baseline']))"
