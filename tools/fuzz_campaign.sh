#!/bin/bash
# A long randomised parity campaign (GPU box):  gpurun -- 'bash tools/fuzz_campaign.sh FIRST_SEED N_SEEDS [stress]'  -> gpurun_out/fuzz_campaign.txt
# `stress`: the extraction sweep draws big / saturated / banded frames, tiny and huge quotas and several frames per call (tools/fuzz_parity.py --stress, 300 cases per seed)
F=${1:-100}; N=${2:-16}; MODE=${3:-}
if [ "$MODE" = stress ]; then FZ="--stress --cases 300"; FM="--stress --cases 40"; else FZ="--cases 700"; FM="--cases 150"; fi
OUT=gpurun_out/fuzz_campaign.txt
: > $OUT
: > gpurun_out/_refusals.log
for ((s=F; s<F+N; s++)); do
  timeout 400 python3 tools/fuzz_parity.py $FZ --seed $s > gpurun_out/_fz.log 2>&1; rc=$?
  echo "extraction${MODE:+ ($MODE)} seed $s rc=$rc: $(tail -1 gpurun_out/_fz.log); compared $(grep -c 'keypoints ok' gpurun_out/_fz.log), refused $(grep -c 'refused' gpurun_out/_fz.log), stereo checks $(grep -c 'stereo shift' gpurun_out/_fz.log)" >> $OUT
  [ $rc -ne 0 ] && grep -n "MISMATCH\|Traceback\|UNEXPECTED" -A16 gpurun_out/_fz.log | head -40 >> $OUT
  grep '^refusals:' gpurun_out/_fz.log >> gpurun_out/_refusals.log
  timeout 400 python3 tools/fuzz_matchers.py $FM --seed $s > gpurun_out/_fm.log 2>&1; rc=$?
  echo "matchers   seed $s rc=$rc: $(tail -1 gpurun_out/_fm.log)" >> $OUT
  [ $rc -ne 0 ] && grep -n "MISMATCH\|Traceback" -A6 gpurun_out/_fm.log | head -20 >> $OUT
done
# refusals of the whole campaign by reason (every one is checked against an independent statement of the rules: an UNEXPECTED one fails its seed)
echo "refusals by reason, all seeds:" >> $OUT
sed 's/^refusals: *\([0-9]*\)  \(.*\)$/\1\t\2/' gpurun_out/_refusals.log | awk -F'\t' '{n[$2]+=$1} END {for (r in n) printf "  %6d  %s\n", n[r], r}' | sort -rn >> $OUT
rm -f gpurun_out/_fz.log gpurun_out/_fm.log gpurun_out/_refusals.log
grep -c "rc=0" $OUT; grep -v "rc=0" $OUT | head
