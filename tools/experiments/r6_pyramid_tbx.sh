#!/bin/bash
# round 6: how much of the two-level pyramid kernel's time is lane utilisation?  The level-B passes of a tile use TBX / 256 of the lanes (208 at scale 1.2: 81 %),
# the level-A passes (TBX * 1.2 + halo) / 256.  Narrower tiles lower BOTH utilisations by a known amount; the slope of time against utilisation says what full
# lanes would be worth.  GPU box: bash tools/experiments/r6_pyramid_tbx.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--cpu-seconds 0 --pcie-seconds 0 --call-site 0 --copy-gib 0 --latency-calls 0 --min-timed-ms 1500 --steps 20"
for t in 0 208 192 176 160 144 128; do
  HS_PYRAMID_TBX_MAX=$t timeout -k 10 200 python3 bench.py $Q 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HS_PYRAMID_TBX_MAX=$t: pyramid %.4f ms per 128 frames, %.0f pairs/s, parity %s' % (d['stage_ms_per_step']['pyramid'], d['value'], d['parity_checksum_ok']))"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_pyr_tbx -- python3 bench.py --steps 5 --warmup 2 --min-timed-ms 0 --cpu-seconds 0 --pcie-seconds 0 --call-site 0 --copy-gib 0 --latency-calls 0 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r6_pyr_tbx/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resize" in r["Name"]: print(r["Name"][:40], r["Calls"], r["AverageNs"])
PY
