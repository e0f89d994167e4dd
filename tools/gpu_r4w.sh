#!/bin/bash
# after the hoisted best-candidate loads (quadtree) and the batched source fetch (pyramid): parity of the extraction suites, batch-1 timeline, quadtree phases
set -e
mkdir -p gpurun_out/r4w
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > gpurun_out/r4w/pytest.log 2>&1 || { tail -30 gpurun_out/r4w/pytest.log; exit 1; }
tail -2 gpurun_out/r4w/pytest.log
HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_qprof.so python tools/quadtree_phase_profile.py > gpurun_out/r4w/qt_phase.txt 2>&1 && cat gpurun_out/r4w/qt_phase.txt | tail -17
bash tools/kernel_timeline.sh --pairs 1 --call-site 0 --pcie-seconds 0 --min-timed-ms 0 > gpurun_out/r4w/kt1.txt 2>&1 || true
tail -14 gpurun_out/r4w/kt1.txt
for p in 1 16; do python bench.py --pairs $p --steps 30 --warmup 5 --cpu-seconds 0 --call-site 0 --pcie-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pairs $p', d['value'], d['parity_checksum_ok'] if 'parity_checksum_ok' in d else '', d['stage_ms_per_step'])"; done
