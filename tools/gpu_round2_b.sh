#!/bin/bash
# second GPU pass: full -m gpu suite (no -x), phase profiles (instrumented builds), counter passes (baseline of round 2)
OUT=gpurun_out/r2b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
( time timeout 300 python3 bench.py --cpu-seconds 5 ) > $OUT/bench.json 2> $OUT/bench.err
HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_fprof.so timeout 300 python3 tools/fast_phase_profile.py > $OUT/fast_phase.txt 2>&1
HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so timeout 300 python3 tools/quadtree_phase_profile.py > $OUT/qt_phase.txt 2>&1
timeout 300 python3 tools/bench_pcie.py > $OUT/pcie.json 2>$OUT/pcie.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_traffic.py run > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_traffic.py run > /dev/null 2>&1
python3 tools/pmc_traffic.py report $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic.csv 2>&1
cat $OUT/fast_phase.txt; cat $OUT/qt_phase.txt | tail -40; cat $OUT/traffic.csv; tail -3 $OUT/pcie.err; cat $OUT/pcie.json; tail -4 $OUT/bench.err
