cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--cpu-seconds 0 --pcie-seconds 0 --call-site 0 --copy-gib 0 --min-timed-ms 1500 --pairs 1 --steps 30"
for v in "" "HS_EXTRACT_SPLIT=1" "HS_FAST_COLS=64" "HS_PYRAMID_DEEP_MAX=0"; do
  env $v timeout -k 10 120 python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['latency_one_pair'], d['stage_ms_per_step'])"
done
timeout -k 10 200 python3 bench.py --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --min-timed-ms 1500 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['measured_copy_peak'])"
