#!/bin/bash
# the stress sweep under environment variants that force the paths large batches take (wide items, work queues, no keys, standard pyramid plan) and other extremes
run() { name=$1; shift; env "$@" timeout -k 10 500 python3 tools/fuzz_parity.py --stress --cases 250 --seed $SEED > gpurun_out/_sv.log 2>&1; echo "$name seed $SEED rc=$?: $(tail -1 gpurun_out/_sv.log); compared $(grep -c 'keypoints ok' gpurun_out/_sv.log)"; grep -n "MISMATCH\|Traceback" -A16 gpurun_out/_sv.log | head -50; }
SEED=3400; run "large-batch paths" HS_FAST_COLS=64 HS_FAST_NO_FOLD=1 HS_FAST_KEYS=0 HS_PYRAMID_DEEP_MAX=0
SEED=3401; run "large-batch paths" HS_FAST_COLS=64 HS_FAST_NO_FOLD=1 HS_FAST_KEYS=0 HS_PYRAMID_DEEP_MAX=0
SEED=3402; run "keys always, deep always" HS_FAST_KEYS_MAX_BATCH=100000 HS_PYRAMID_DEEP_MAX=100000
SEED=3403; run "point domain, split" HS_QT_POINT_DOMAIN=1 HS_EXTRACT_SPLIT=1
SEED=3404; run "narrow items + queues, chains for pairs" HS_FAST_COLS=32 HS_FAST_NO_FOLD=1 HS_PYRAMID_CHAIN=2 HS_PYRAMID_DEEP_MAX=0
SEED=3405; run "small lists, scan B" HS_FAST_TEST_SMALL_LISTS=1 HS_FAST_TEST_SCAN_B=1
true
