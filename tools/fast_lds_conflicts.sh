#!/bin/bash
# k_fast_rows: SQ_LDS_BANK_CONFLICT by phase (round 6).  Here: bash tools/fast_instr_breakdown.sh build (libraries cut short after phase n);
# GPU box: bash tools/fast_lds_conflicts.sh  -> gpurun_out/fast_lds/table.txt: counters of the launch cut short after phase n; differences = the phase's own.
#   stop1 = tile staging + scan A masks, stop2 = + list expansion, stop4 = + corner pass (ring gathers, score network, in-place compaction),
#   full = + score tile (zero, scatter) + NMS + emit
OUT=gpurun_out/fast_lds; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BF="--cpu-seconds 0 --pcie-seconds 0 --call-site 0 --copy-gib 0 --latency-calls 0 --handles 1 --pairs ${PAIRS:-64}"
for v in stop1 stop2 stop4 full; do
  if [ $v = full ]; then unset HYSLAM_AMD_LIB; else export HYSLAM_AMD_LIB=$PWD/hyslam_amd/libhyslam_amd_$v.so; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU --output-format csv -d $OUT/$v -- python3 bench.py --steps 3 --warmup 1 --min-timed-ms 0 $BF > /dev/null 2>&1 || echo "$v: rocprofv3 failed"
done
python3 - <<PY | tee $OUT/table.txt
import csv, glob, collections
rows = {}
for v in ("stop1", "stop2", "stop4", "full"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fast_rows" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows[v] = {k: sum(x) / len(x) / 1e6 for k, x in acc.items()}
names = ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL"]
print("k_fast_rows, M per launch (%s pairs per call), cumulative up to the phase:" % "${PAIRS:-64}")
print("%-8s" % "build", " ".join("%22s" % n.replace("SQ_", "") for n in names))
for v in ("stop1", "stop2", "stop4", "full"):
    print("%-8s" % v, " ".join("%22.2f" % rows[v].get(n, float("nan")) for n in names))
print("the phase's own (difference to the build before it):")
prev = None
for v, label in (("stop1", "staging + scan A"), ("stop2", "list expansion"), ("stop4", "corner pass"), ("full", "score tile + NMS + emit")):
    d = {n: rows[v].get(n, 0) - (rows[prev].get(n, 0) if prev else 0) for n in names}
    conf, act = d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_IDX_ACTIVE"]
    print("%-26s conflict %7.2f M of %7.2f M LDS-active (%4.1f %%), %6.2f M LDS instructions -> %.2f conflict cycles per LDS instruction" % (label, conf, act, 100 * conf / act if act else 0, d["SQ_INSTS_LDS"], conf / d["SQ_INSTS_LDS"] if d["SQ_INSTS_LDS"] else 0))
    prev = v
PY
