#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh TAG'): every measurement behind profiles/ for one tag.
# Counter passes are separate runs with --pmc only (no trace domains), as MI355X_MICROARCH.md prescribes.
TAG=${1:-x}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT      # a tag used before must not leave its files behind (summarize_profiles.py takes the first match)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "import json; from hyslam_amd._native import source_digests; json.dump(source_digests(), open('$OUT/source_digests.json', 'w'))"
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --latency-calls 0 --copy-gib 0 --min-timed-ms 0 > $OUT/bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_traffic.py run > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_traffic.py run > /dev/null 2>&1
python3 tools/pmc_traffic.py report $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic.csv
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --latency-calls 0 --copy-gib 0 --min-timed-ms 0 > /dev/null 2>&1
timeout 300 python3 tools/bench_matchers.py > $OUT/matchers.json 2>/dev/null
timeout 300 python3 tools/bench_pcie.py > $OUT/pcie.json 2>/dev/null
timeout 300 python3 bench.py --config c3 --steps 20 --warmup 3 --min-timed-ms 2000 > $OUT/c3.json 2>/dev/null
timeout 300 python3 bench.py --config c4 --steps 30 --warmup 3 --min-timed-ms 2000 > $OUT/c4.json 2>/dev/null
timeout 300 python3 bench.py --config c5 --steps 50 --warmup 5 --min-timed-ms 2000 > $OUT/c5.json 2>/dev/null
timeout 300 python3 bench.py --config c5 --c5-match bow --steps 50 --warmup 5 --min-timed-ms 2000 > $OUT/c5_bow.json 2>/dev/null
Q="--cpu-seconds 0 --pcie-seconds 0 --call-site 0 --copy-gib 0 --min-timed-ms 2000"      # the sweeps: 2 s of timed region each, no secondary legs
for b in 1 4 16 128; do timeout 300 python3 bench.py --pairs $b --steps 30 --warmup 3 $Q 2>/dev/null | tail -1 > $OUT/bench_pairs$b.json; done
for hn in 2 3; do timeout 300 python3 bench.py --handles $hn --steps 30 --warmup 3 $Q 2>/dev/null | tail -1 > $OUT/bench_handles$hn.json; done
timeout 300 python3 bench.py --density 3 --steps 100 $Q 2>/dev/null | tail -1 > $OUT/bench_density3.json
# what hySLAM's call sites would see through the C++ adaptors (tests/cpp/bench_adaptor.cpp)
python3 - <<PY
import subprocess, sys
sys.path.insert(0, "tests")
import test_adaptor as t
t.build("bench_adaptor.cpp", t.EXE_B)
r = t.run_bench(1920, 1080, 30, 50000)
open("$OUT/adaptor.json", "w").write(r.stdout.decode())
PY
# batch-1 kernel timeline (un-instrumented steps)
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt1 -- python3 bench.py --pairs 1 --steps 30 --warmup 5 --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --latency-calls 0 --copy-gib 0 --min-timed-ms 0 --profile-steps 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt16 -- python3 bench.py --pairs 16 --steps 30 --warmup 5 --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --latency-calls 0 --copy-gib 0 --min-timed-ms 0 --profile-steps 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt64 -- python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --latency-calls 0 --copy-gib 0 --min-timed-ms 0 --profile-steps 0 > /dev/null 2>&1
# the 16-pair launch shape (the bench default until round 4) under the profiler too: comparable with r03 / r04
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats16 -- python3 bench.py --pairs 16 --steps 20 --warmup 3 --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --latency-calls 0 --copy-gib 0 --min-timed-ms 0 > $OUT/bench16_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq16 -- python3 bench.py --pairs 16 --steps 4 --warmup 1 --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --latency-calls 0 --copy-gib 0 --min-timed-ms 0 > /dev/null 2>&1
[ -f hyslam_amd/libhyslam_amd_qprof.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so timeout 300 python3 tools/quadtree_phase_profile.py > $OUT/qt_phase_1080p.txt 2>&1
[ -f hyslam_amd/libhyslam_amd_qprof.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so timeout 300 python3 tools/quadtree_phase_profile.py 4000 3000 3000 1.4 > $OUT/qt_phase_4000x3000.txt 2>&1
[ -f hyslam_amd/libhyslam_amd_fwav.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_fwav.so timeout 300 python3 tools/fast_b1_timeline.py 2 > $OUT/fast_b1_timeline.txt 2>&1
[ -f hyslam_amd/libhyslam_amd_fwav.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_fwav.so timeout 300 python3 tools/fast_wave_timeline.py 32 > $OUT/fast_wave_timeline.txt 2>&1
[ -f hyslam_amd/libhyslam_amd_pprof.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_pprof.so timeout 300 python3 tools/pyramid_phase_profile.py 2 > $OUT/pyr_phase_b1.txt 2>&1
[ -f hyslam_amd/libhyslam_amd_pprof.so ] && HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_pprof.so timeout 300 python3 tools/pyramid_phase_profile.py 32 > $OUT/pyr_phase_b16.txt 2>&1
for b in 2 8 32 64; do timeout 300 python3 bench.py --pairs $b --steps 30 --warmup 3 $Q 2>/dev/null | tail -1 > $OUT/bench_pairs$b.json; done
timeout 120 tools/micro/valu_peak > $OUT/valu_issue_rates.txt 2>&1
timeout 300 python3 tools/bench_preprocess.py > $OUT/preprocess.json 2>/dev/null
[ -x tools/micro/lds_gather ] && timeout 120 tools/micro/lds_gather > $OUT/lds_gather.txt 2>&1
tail -1 $OUT/bench.json | cut -c1-400
