#!/bin/bash
# round 4: flatter tiles for the deep chains of single-pair calls
OUT=gpurun_out/${1:-r4n}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
for r in 16 8 6 4; do
    HS_PYRAMID_DEEP_ROWS=$r timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --pairs 1 --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs 1 deep rows $r:', d['value'], round(d['ms_per_step']/d['inner_repeats']*1000,1), 'us/pair', d['parity_checksum_ok'], d['stage_ms_per_step'])"
    HS_PYRAMID_DEEP_ROWS=$r HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_pprof.so python3 tools/pyramid_phase_profile.py 2 | head -1
done 2>&1 | tee $OUT/sweep.txt
HS_PYRAMID_DEEP_MAX=4 HS_PYRAMID_DEEP_ROWS=8 timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --pairs 2 --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs 2 deep rows 8:', d['value'], d['stage_ms_per_step'])"
HS_PYRAMID_DEEP_MAX=0 timeout 300 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --pairs 2 --steps 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs 2 standard:', d['value'], d['stage_ms_per_step'])"
