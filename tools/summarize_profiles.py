#!/usr/bin/env python3
"""Turns gpurun_out/prof_TAG (made by tools/collect_profiles.sh on the GPU box) into the committed files under profiles/.
usage: python tools/summarize_profiles.py TAG PREFIX          e.g.  a r02"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hyslam_amd._native import source_digests          # noqa: E402  (no library load: the digests are of the source files)

tag, rnd = sys.argv[1], sys.argv[2]
try:                                # written on the GPU box by tools/collect_profiles.sh: the sources the counters were collected from
    DIGESTS = json.load(open(os.path.join("gpurun_out/prof_" + tag, "source_digests.json")))
except OSError:
    DIGESTS = source_digests()


def stamp(kernel):
    for prefix, d in DIGESTS.items():
        if kernel.startswith(prefix):
            return d
    return None

src = "gpurun_out/prof_" + tag
os.makedirs("profiles", exist_ok=True)
import re as _re
_m = _re.search(r'^DEFAULT_PAIRS = (\d+)', open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')).read(), _re.M)
PAIRS, HANDLES = int(os.environ.get('HS_PROFILE_PAIRS', _m.group(1) if _m else 16)), 1          # bench.py's defaults: PAIRS pairs per step on one handle -> every launch sequence covers 2 * PAIRS frames


def first(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    return g[0] if g else None


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


f = first("stats/**/*kernel_stats.csv")
rows = list(csv.reader(open(f)))
with open("profiles/%s_kernel_stats.csv" % rnd, "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0\n")
    o.write("# bench.py default: %d stereo pairs of 1920x1080 per step on one handle -> every launch sequence covers %d frames; MI355X; tag %s\n" % (PAIRS, 2 * PAIRS, tag))
    w = csv.writer(o)
    for r in rows:
        r[0] = short(r[0])[:60]
        w.writerow(r)
# per-launch durations of the pyramid launches by grid (kernel trace)
kt = first("stats/**/*kernel_trace.csv")
if kt:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        k = short(r["Kernel_Name"])
        if k.startswith("k_resize"):
            d[(k, int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open("profiles/%s_pyramid_launches.csv" % rnd, "w") as o:
        o.write("# per-launch duration of the pyramid kernels by grid (workgroups x, y, frames), rocprofv3 --kernel-trace of the same run\nkernel,grid_x,grid_y,frames,launches,avg_us\n")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            o.write("%s,%d,%d,%d,%d,%.2f\n" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v) / 1e3))
open("profiles/%s_hbm_traffic_pmc.csv" % rnd, "w").write(open(os.path.join(src, "traffic.csv")).read())
tr = {}
for r in csv.reader(l for l in open(os.path.join(src, "traffic.csv")) if not l.startswith("#")):
    if r[0] == "kernel":
        continue
    # pmc_traffic.py runs 3 steps of one handle = 3 launch sequences
    tr[r[0].split("<")[0]] = {"read_MB": float(r[4]), "written_MB": float(r[5]), "frames_per_launch": int(r[6]), "launches": int(r[1]),
                              "launches_per_sequence": int(r[1]) / 3.0, "source_sha16": stamp(r[0])}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/pmc_traffic.py; FETCH_SIZE x2 (gfx950, calibrated on "
                     "a 1 GiB copy at 4 B/lane and 16 B/lane), WRITE_SIZE x1", "pairs_per_step": PAIRS, "kernels": tr},
          open("profiles/%s_hbm_traffic.json" % rnd, "w"), indent=1)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(first("sq/**/*counter_collection.csv"))):
    k = short(r["Kernel_Name"])
    if k.startswith("k_"):
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for d in agg.values() for c in d})
# launch sequences of the counter run = launches of the FAST kernel (one per sequence); bench.py runs two warm-ups since round 4, so the count is read, not assumed
n_seq = max([len(next(iter(d.values()))) for k, d in agg.items() if k.startswith("k_fast_rows")] or [5 * HANDLES])
sq = {}
with open("profiles/%s_sq_counters.csv" % rnd, "w") as o:
    o.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0\n")
    o.write("# per-launch averages; bench.py default (one handle, %d pairs): %d frames of 1920x1080 per launch; MI355X; tag %s\n" % (PAIRS, 2 * PAIRS, tag))
    o.write("kernel,launches," + ",".join(names) + "\n")
    for k, d in agg.items():
        n = len(next(iter(d.values())))
        o.write(k.replace(",", ";") + "," + str(n) + "," + ",".join(str(round(sum(d[c]) / len(d[c]))) for c in names) + "\n")
        sq[k.split("<")[0]] = dict({c: round(sum(d[c]) / len(d[c])) for c in names}, frames_per_launch=2 * PAIRS // HANDLES, launches=n, launches_per_sequence=n / float(n_seq),
                                  source_sha16=stamp(k))
json.dump({"source": "rocprofv3 --pmc SQ_* (one pass, no trace domains) of `python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0`; per-launch averages",
           "pairs_per_step": PAIRS, "kernels": sq}, open("profiles/%s_sq_counters.json" % rnd, "w"), indent=1)
summary = {}
for name in ("bench", "matchers", "pcie", "c3", "c4", "c5", "c5_bow", "bench_pairs1", "bench_pairs2", "bench_pairs4", "bench_pairs8", "bench_pairs16", "bench_pairs32", "bench_pairs64", "bench_pairs128", "bench_handles2", "bench_handles3", "bench_under_rocprof", "bench16_under_rocprof",
             "bench_density3", "adaptor", "preprocess"):
    p = os.path.join(src, name + ".json")
    try:
        text = open(p).read()
        try:
            summary[name] = json.loads(text)                     # a multi-line JSON object (the adaptor bench)
        except ValueError:
            summary[name] = json.loads([l for l in text.splitlines() if l.startswith("{")][-1])
    except Exception:
        summary[name] = None
json.dump(summary, open("profiles/%s_bench_lines.json" % rnd, "w"), indent=1)
try:
    open("profiles/%s_valu_issue_rates.txt" % rnd, "w").write(
        "# tools/micro/valu_peak: wave-instructions per cycle and CU (at the nominal 2.4 GHz) of the instruction classes the kernels are made of, by waves per SIMD\n" +
        "".join(l for l in open(os.path.join(src, "valu_issue_rates.txt")) if l.startswith("v_")))
except Exception:
    pass
# un-instrumented kernel timelines (one step each) and the quadtree's phase stamps
def timeline(d, out):
    kt = first(d + "/**/*kernel_trace.csv")
    if not kt:
        return
    rows = []
    for r in csv.DictReader(open(kt)):
        n = short(r["Kernel_Name"])
        if n.startswith("k_"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:28]))
    rows.sort()
    firsts = [i for i, r in enumerate(rows) if r[2].startswith("k_resize") and (i == 0 or not rows[i - 1][2].startswith("k_resize"))]
    if len(firsts) < 3:
        return
    a, b2 = firsts[-3], firsts[-2]
    t0, prev, busy = rows[a][0], None, 0
    for s_, e_, n in rows[a:b2]:
        out.write("%-30s start %7.1f us  dur %6.1f us  gap %5.1f\n" % (n, (s_ - t0) / 1e3, (e_ - s_) / 1e3, 0 if prev is None else (s_ - prev) / 1e3))
        prev = e_; busy += e_ - s_
    out.write("step period %.1f us, kernels busy %.1f us\n" % ((rows[b2][0] - t0) / 1e3, busy / 1e3))


with open("profiles/%s_kernel_timeline.txt" % rnd, "w") as o:
    o.write("# rocprofv3 --kernel-trace of bench.py without stage events (--profile-steps 0): one step of the timed loop, tag %s\n" % tag)
    o.write("## --pairs 1 (one stereo pair per call)\n"); timeline("kt1", o)
    o.write("## 16 stereo pairs per call (the bench default until round 4)\n"); timeline("kt16", o)
    o.write("## default (%d stereo pairs per call)\n" % PAIRS); timeline("kt64", o)
for name in ("qt_phase_1080p", "qt_phase_4000x3000"):
    try:
        body = open(os.path.join(src, name + ".txt")).read()      # read FIRST: a missing source (no HS_QT_PROFILE build on the box) must not truncate the committed file
        open("profiles/%s_%s.txt" % (rnd, name), "w").write(
            "# tools/quadtree_phase_profile.py: shader-clock stamps of the level-0 quadtree workgroup of image 0 (HS_QT_PROFILE build); tags: 20 set-up, 22 item scan, 23/25 record run\n"
            "# (search / fetch+key+histogram), 1 gather done, 30 pyramid, 31 closed form, 5 list built, 110 order, 11 child counts, 12 cut+children, 13 survivors, 14 relabel (point domain),\n"
            "# 3 geometric keys -> nodes, 40 best point per node, 41 emitted, 4 tile order\n" + body)
    except OSError:
        pass
for name, head in (("fast_b1_timeline", "# tools/fast_b1_timeline.py 2 (HS_FAST_WAVES build): one stereo pair per call — phases of every persistent workgroup's first work item, by pyramid level\n"),
                   ("fast_wave_timeline", "# tools/fast_wave_timeline.py 32 (HS_FAST_WAVES build): 32 frames per launch\n"),
                   ("pyr_phase_b1", "# tools/pyramid_phase_profile.py 2 (HS_PYR_PROFILE build): the middle workgroup of the single k_resize_chain launch of a single-pair call (levels 1-7)\n"),
                   ("pyr_phase_b16", "# tools/pyramid_phase_profile.py 32 (HS_PYR_PROFILE build): the middle workgroup of the three-level k_resize_chain launch (levels 5-7) at 16 pairs per call\n")):
    try:
        body = open(os.path.join(src, name + ".txt")).read()
        open("profiles/%s_%s.txt" % (rnd, name), "w").write(head + body)
    except OSError:
        pass
# the 16-pair launch shape (bench default until round 4) for comparison with r03 / r04: kernel stats and SQ counters of `bench.py --pairs 16`
f16 = first("stats16/**/*kernel_stats.csv")
if f16:
    with open("profiles/%s_kernel_stats_pairs16.csv" % rnd, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --pairs 16 --steps 20 --warmup 3 --cpu-seconds 0: 32 frames per launch (the default shape of rounds 2-4); tag %s\n" % tag)
        w = csv.writer(o)
        for r in csv.reader(open(f16)):
            r[0] = short(r[0])[:60]
            w.writerow(r)
s16 = first("sq16/**/*counter_collection.csv")
if s16:
    agg16 = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(s16)):
        k = short(r["Kernel_Name"])
        if k.startswith("k_"):
            agg16[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names16 = sorted({c for d in agg16.values() for c in d})
    with open("profiles/%s_sq_counters_pairs16.csv" % rnd, "w") as o:
        o.write("# rocprofv3 --pmc SQ_* -- python3 bench.py --pairs 16 --steps 4 --warmup 1: per-launch averages, 32 frames per launch (comparable with r03 / r04); tag %s\n" % tag)
        o.write("kernel,launches," + ",".join(names16) + "\n")
        for k, d in agg16.items():
            o.write(k.replace(",", ";") + "," + str(len(next(iter(d.values())))) + "," + ",".join(str(round(sum(d[c]) / len(d[c]))) for c in names16) + "\n")
try:
    open("profiles/%s_lds_gather.txt" % rnd, "w").write("# tools/micro/lds_gather: a FAST ring gather (16 ring pixels + centre of a pseudo-random pixel per lane) as byte reads and as wider LDS reads\n" + open(os.path.join(src, "lds_gather.txt")).read())
except OSError:
    pass
b = summary["bench"]
print("value", b["value"], b["stage_ms_per_step"], b["roofline"])
