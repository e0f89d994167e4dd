#!/usr/bin/env python3
"""Turns gpurun_out/prof_TAG (made by tools/collect_profiles.sh on the GPU box) into the committed files under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

tag, rnd = sys.argv[1], sys.argv[2]          # e.g. final r01_e
src = "gpurun_out/prof_" + tag
os.makedirs("profiles", exist_ok=True)


def first(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return g[0] if g else None


f = first("stats/**/*kernel_stats.csv")
rows = list(csv.reader(open(f)))
with open("profiles/%s_kernel_stats.csv" % rnd, "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0\n")
    o.write("# bench.py default: 16 stereo pairs of 1920x1080 per step dealt over 2 handles -> every kernel launch covers 16 frames; MI355X; tag %s\n" % tag)
    w = csv.writer(o)
    for r in rows:
        r[0] = r[0].split("(")[0][:60]
        w.writerow(r)
open("profiles/%s_hbm_traffic_pmc.csv" % rnd, "w").write(open(os.path.join(src, "traffic.csv")).read())
tr = {}
for r in csv.reader(l for l in open(os.path.join(src, "traffic.csv")) if not l.startswith("#")):
    if r[0] == "kernel":
        continue
    tr[r[0].split("<")[0]] = {"read_MB": float(r[4]), "written_MB": float(r[5]), "frames_per_launch": int(r[6]), "launches": int(r[1])}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/pmc_traffic.py; FETCH_SIZE x2 (gfx950, calibrated on "
                     "a 1 GiB copy at 4 B/lane and 16 B/lane), WRITE_SIZE x1", "pairs_per_step": 16, "kernels": tr},
          open("profiles/%s_hbm_traffic.json" % rnd, "w"), indent=1)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(first("sq/**/*counter_collection.csv"))):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_"):
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("profiles/%s_sq_counters.csv" % rnd, "w") as o:
    o.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0\n")
    o.write("# per-launch averages; bench.py default (2 handles x 8 pairs): 16 frames of 1920x1080 per launch; MI355X; tag %s\n" % tag)
    names = sorted({c for d in agg.values() for c in d})
    o.write("kernel,launches," + ",".join(names) + "\n")
    for k, d in agg.items():
        o.write(k.replace(",", ";") + "," + str(len(next(iter(d.values())))) + "," + ",".join(str(round(sum(d[c]) / len(d[c]))) for c in names) + "\n")
summary = {}
for name in ("bench", "matchers", "pcie", "c3", "c5", "bench_pairs1", "bench_pairs4", "bench_pairs64", "bench_under_rocprof"):
    p = os.path.join(src, name + ".json")
    try:
        line = [l for l in open(p).read().splitlines() if l.startswith("{")][-1]
        summary[name] = json.loads(line)
    except Exception as e:
        summary[name] = None
json.dump(summary, open("profiles/%s_bench_lines.json" % rnd, "w"), indent=1)
b = summary["bench"]
print("value", b["value"], b["stage_ms_per_step"], b["roofline"])
