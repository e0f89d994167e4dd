#!/usr/bin/env python3
"""bench.py — stereo frames/s of ORB extract + left<->right match at 1920x1080 / 2000 features (BASELINE.json C2).

A *step* is one pass of the hot path (pyramid -> FAST/NMS cells -> quadtree distribution -> blur+orientation+rBRIEF for
left and right frames, then the stereo matcher) over one batch of `--pairs` synthetic stereo pairs that are already
resident in HBM.  One process per GPU; frames shard across ranks with no data-path collective (weak scaling).

  python bench.py                      N = 1
  python bench.py --gpus N             spawns N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1) when it was not
                                       started by torch.distributed.run; under torchrun it uses the environment it is given and insists
                                       that WORLD_SIZE == --gpus
Prints ONE JSON line on rank 0:
  value         whole-job stereo pairs per second (all ranks)
  roofline      the dominant kernel of the step: algorithmic bytes per launch / its HIP-event-measured duration vs 8 TB/s, the PMC traffic
                and the VALU issue fraction of the same launch shape from the committed counter passes
  cpu_baseline  the oracle ("port" of the reference's CPU path) timed on this box, N = 1 only: 2 threads (the reference's structure) and all cores
Other BASELINE configurations: --config c3 | c4 | c5.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
CLOCK_GHZ = 2.4                # max shader clock (MI355X_MICROARCH.md); issue-rate fractions are quoted against it
W, H, NFEAT = 1920, 1080, 2000
PROFILE_TAG = "r06"            # profiles/<tag>_hbm_traffic.json, profiles/<tag>_sq_counters.json hold the committed counter passes
DEFAULT_PAIRS = 64             # stereo pairs per step; tools/pmc_traffic.py and tools/summarize_profiles.py read this constant (HS_PROFILE_PAIRS overrides)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (200 x 16 pairs = 0.11 s of GPU work at the default)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=DEFAULT_PAIRS,
                    help="stereo pairs per step (= per call of the stereo front end) per GPU.  Default 64 since round 5 (16 until round 4): a launch's duration is a fixed part "
                         "(launch floor, ramp-up, the tail of the persistent FAST waves) + a part per frame — FAST: 26 us + 4.1 us per frame — so larger calls amortise more; "
                         "profiles/README.md lists 1 / 2 / 4 / 8 / 16 / 32 / 64 pairs per call for every round")
    ap.add_argument("--distinct", type=int, default=4,
                    help="distinct synthetic pairs generated per rank; the batch holds --pairs separate copies (pair i = distinct pair i %% distinct), "
                         "so no two frames of a step share an address")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of EACH of the two CPU baseline samples (0 = skip)")
    ap.add_argument("--handles", type=int, default=1,
                    help="extractor handles per GPU; the pairs of a step are dealt over them and each runs on its own stream (the reference also "
                         "keeps separate extractor instances side by side, ImageProcessing.cpp:31-32).  Default 1: every kernel then has the GPU "
                         "to itself and its HIP-event duration is its own cost (the roofline block); 2 handles overlap the latency-bound stages of "
                         "one with the wide kernels of the other (+4 %% pairs/s) but each kernel's duration then includes the time it shares")
    ap.add_argument("--profile-steps", type=int, default=-1,
                    help="timed steps that record per-stage HIP events (the roofline's kernel durations); default: a tenth of --steps, at least 1.  "
                         "Every recorded event drains the stream between two stages, so the remaining steps run un-instrumented")
    ap.add_argument("--lanes", type=int, default=1, choices=[1, 2],
                    help="launch sequences INSIDE one handle (hs_orb_set_lanes): same effect for callers that own a single handle")
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c2",
                    help="c2 = stereo extract+match (the headline metric); c3 = batched 64 mono frames, extract only (per-kernel GB/s); "
                         "c4 = dual camera (stereo pair + 4000x3000 'Imaging' frame) + 50k-landmark projection search; "
                         "c5 = one mono stream per GPU + all-gather + cross-camera match")
    ap.add_argument("--c5-match", choices=["knn2", "bow"], default="knn2", help="c5: brute-force 2-NN or vocabulary-grouped (BoW) matching against every peer")
    ap.add_argument("--c5-comm", choices=["hs", "torch"], default="hs",
                    help="c5 exchange: hs = the C ABI's own RCCL all-gather (hs_comm_*, enqueued on the step's stream; the id travels over torch.distributed), "
                         "torch = torch.distributed.all_gather_into_tensor issued on the same stream")
    ap.add_argument("--density", type=int, choices=[1, 3], default=1,
                    help="scene density: 1 = the default synthetic scene (w*h/800 shapes: ~5 k FAST corners on level 0 of a 1080p frame), "
                         "3 = three times the shapes (>= 12 k level-0 corners): FAST's sparse phases scale with the corner load")
    ap.add_argument("--min-timed-ms", type=float, default=10000.0,
                    help="every timed step repeats its batch `inner_repeats` times so that the K timed steps cover at least this much GPU time "
                         "(0 = one pass per step); value counts every pass.  Default 10 s: an outside observer that samples GPU activity every few "
                         "seconds (the driver's 5-s sampler saw 0 %% in round 4: the timed region was 0.22 s of a 26-s run) then lands inside the region at least once")
    ap.add_argument("--cpu-worker", type=int, default=-1, help=argparse.SUPPRESS)      # internal: one worker PROCESS of the all-cores CPU baseline (no GPU, no torch)
    ap.add_argument("--cpu-frames", type=str, default="", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-workers", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--call-site", type=int, choices=[0, 1], default=1,
                    help="c2, N = 1: also run tests/cpp/_build/bench_adaptor (when __graft_entry__.build() has made it) AFTER the timed loop and report, as "
                         "`call_site`, what hySLAM's own call sites would see through the drop-in C++ classes (ProcessStereoImage ms per pair, split gather / C ABI / scatter)")
    ap.add_argument("--copy-gib", type=float, default=2.0,
                    help="c2: GiB of the device-to-device copy that measures this box's own HBM copy bandwidth (hipMemcpyAsync and a 16-byte-per-lane kernel), "
                         "printed as roofline.peak_measured_copy beside the vendor peak (0 = skip)")
    ap.add_argument("--latency-calls", type=int, default=400, help="c2, N = 1: calls of the one-pair-per-call latency leg after the timed loop (0 = skip)")
    ap.add_argument("--pcie-seconds", type=float, default=1.5, help="c2, N = 1: budget of the host-fed (PCIe-inclusive) secondary measurement (0 = skip)")
    ap.add_argument("--c4-split", type=int, choices=[0, 1], default=1,
                    help="c4: hs_orb_set_split(1) on both camera handles (level 0's FAST + quadtree on a second stream beside the pyramid); 0 = the library default")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: ranks rendezvous over gloo, exchange one tensor and rank 0 prints a JSON line (covers the --gpus spawn path on CPU)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ N > 1 without torchrun: spawn the ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """Parent of `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment): starts one child per GPU and relays rank 0's JSON
    line.  The parent never touches the GPU (no torch import, no HIP call), so starting children is safe on this pool."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll every child: a rank that dies at start-up would leave rank 0 blocked in the rendezvous until the process-group timeout
    out0, rc, deadline = b"", 0, time.time() + 3600
    import selectors
    sel = selectors.DefaultSelector()
    sel.register(procs[0].stdout, selectors.EVENT_READ)
    alive = set(range(args.gpus))
    eof0 = False
    while alive or not eof0:
        if not eof0:
            for key, _ in sel.select(timeout=0.2):
                chunk = os.read(key.fileobj.fileno(), 1 << 16)
                if chunk:
                    out0 += chunk
                else:
                    eof0 = True
                    sel.unregister(key.fileobj)
        else:
            time.sleep(0.05)
        for r in list(alive):
            code = procs[r].poll()
            if code is not None:
                alive.discard(r)
                rc = rc or (1 if code else 0)
        if (rc or time.time() > deadline) and alive:      # one rank failed (or nothing finishes): stop the others, never report success
            for r in alive:
                procs[r].kill()
            for r in alive:
                procs[r].wait()
            alive.clear()
            rc = 1
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    if rc:
        sys.stderr.write("bench.py: a rank failed (exit codes %s)\n" % [p.returncode for p in procs])
    return 1 if rc else 0


def dry_run(args, rank, world):
    if os.environ.get("HS_BENCH_TEST_FAIL_RANK") == str(rank):       # tests: a rank that dies before the rendezvous
        sys.exit(3)
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert int(t.item()) == world
    line = {"metric": "dry run (no GPU work)", "value": 0.0, "unit": "stereo_pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "dry_run": True}
    rc = 0
    if args.config == "c5":
        # the self-check of the cross-camera exchange on CPU tensors: seeded stand-in records (no extraction), the real all-gather + verify_exchange;
        # HS_BENCH_TEST_CORRUPT_RANK makes one rank flip a byte of ITS copy of the gathered buffer so that the check is seen to fire
        import numpy as np
        from hyslam_amd import distributed as D
        from hyslam_amd._native import KP_DTYPE
        cap = 64
        rng = np.random.default_rng(500 + rank)
        n = 40 + rank
        k = np.zeros(n, KP_DTYPE); k["x"] = rng.random(n); k["y"] = rng.random(n); k["octave"] = rng.integers(0, 8, n)
        rec = torch.from_numpy(D.pack_record(k, rng.integers(0, 256, (n, 32), dtype=np.uint8), cap))
        g = D.all_gather_records(rec) if world > 1 else rec.view(1, -1).clone()
        if os.environ.get("HS_BENCH_TEST_CORRUPT_RANK") == str(rank):
            g[(rank + 1) % world, 100] ^= 0xFF
        check = D.verify_exchange(g, n, rank, world)
        line["ranks_consistent"] = check["ranks_consistent"]
        line["exchange_check"] = check
        rc = 0 if check["ranks_consistent"] else 4
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


# ------------------------------------------------------------------------------------------------ accounting
def pyramid_pixels(ex, w, h):
    inv = ex.GetInverseScaleFactors()
    sizes = [(int(np.rint(np.float32(w) * s)), int(np.rint(np.float32(h) * s))) for s in inv]
    return [a * b for a, b in sizes]


def algorithmic_bytes(px, k):
    """compulsory HBM bytes per mono frame at the reference's stage granularity."""
    P, p0, p7 = sum(px), px[0], px[-1]
    per_stage = {
        "pyramid": (P - p7) + (P - p0),          # each level read once to make the next + levels 1.. written
        "fast_cells": P,                         # FAST reads every level once
        "quadtree": 0,                           # host stage in the reference; candidate lists only
        "describe": 2 * P + k * 1369 + k * 60,   # the reference's full-level blur (read + write) + 37x37 window per keypoint + record out
    }
    return per_stage, sum(per_stage.values())


KERNEL_OF_STAGE = {"fast_cells": "k_fast_rows", "pyramid": "k_resize", "describe": "k_describe", "quadtree": "k_quadtree"}


def committed_counters(stage, pairs_per_step, frames_per_launch):
    """Per-launch HBM traffic and VALU wave-instructions of `stage`'s kernel from the committed rocprofv3 counter passes of THIS workload
    (profiles/<tag>_hbm_traffic.json: --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 calibration;
    profiles/<tag>_sq_counters.json: --pmc SQ_INSTS_VALU ...).  Counters cannot be read from inside this process, so values are only
    reported when the profiled launch shape matches; otherwise null."""
    traffic = valu = None
    sq = {}
    src = []
    from hyslam_amd._native import source_digests
    want = source_digests().get(KERNEL_OF_STAGE[stage])

    def fresh(entries):      # a counter is replayed only if it was collected from the kernel sources this library was built from
        return all(e.get("source_sha16") == want for e in entries)
    try:       # a stage can be several kernels (the pyramid): per-launch averages x launches per launch sequence, summed
        t = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_hbm_traffic.json")))
        if t["pairs_per_step"] == pairs_per_step:
            m = [e for k, e in t["kernels"].items() if k.startswith(KERNEL_OF_STAGE[stage]) and e["frames_per_launch"] == frames_per_launch]
            if m and not fresh(m):
                src.append("STALE: profiles/%s_hbm_traffic.json was collected from other kernel sources" % PROFILE_TAG)
            elif m:
                traffic = int(sum((e["read_MB"] + e["written_MB"]) * 1e6 * e.get("launches_per_sequence", 1) for e in m))
                src.append("profiles/%s_hbm_traffic.json" % PROFILE_TAG)
    except Exception:
        pass
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_sq_counters.json")))
        if t["pairs_per_step"] == pairs_per_step:
            m = [e for k, e in t["kernels"].items() if k.startswith(KERNEL_OF_STAGE[stage]) and e["frames_per_launch"] == frames_per_launch]
            if m and not fresh(m):
                src.append("STALE: profiles/%s_sq_counters.json was collected from other kernel sources" % PROFILE_TAG)
            elif m:
                valu = float(sum(e["SQ_INSTS_VALU"] * e.get("launches_per_sequence", 1) for e in m))
                for c in ("SQ_BUSY_CU_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
                    if all(c in e for e in m):
                        sq[c] = float(sum(e[c] * e.get("launches_per_sequence", 1) for e in m))
                src.append("profiles/%s_sq_counters.json" % PROFILE_TAG)
    except Exception:
        pass
    return traffic, valu, src, sq


def roofline_block(stage, stage_ms, per_stage_bytes, frames_per_launch, launches, pairs_per_step, moved_bytes=None):
    """roofline of one stage (= `launches` kernel launches): algorithmic bytes / HIP-event time against the HBM peak (`bound` "hbm": the roof `frac`
    is priced against — the path is integer / byte work, no MFMA), the measured traffic, and what the counters say limits the kernel (`limiter`):
    VALU wave-instructions per busy CU cycle against the instruction-class ceilings of profiles/r05_valu_issue_rates.txt."""
    ms = stage_ms[stage]
    algo = per_stage_bytes[stage] * frames_per_launch
    achieved = algo / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic, valu, src, sq = committed_counters(stage, pairs_per_step, frames_per_launch)
    rl = {"bound": "hbm", "kernel": stage, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None if traffic is None else int(traffic / launches), "counter_source": src or None,
          "algorithmic_bytes_per_launch": int(algo / launches), "launch_ms": round(ms / launches, 5), "launches_per_step_and_handle": launches,
          "frames_per_launch": frames_per_launch}
    if moved_bytes is not None:       # describe: the kernel does not move the reference's full-level blur; GB/s on the bytes it requests
        rl["moved_bytes_per_launch"] = int(moved_bytes * frames_per_launch)
        rl["achieved_on_moved_bytes"] = round(moved_bytes * frames_per_launch / (ms * 1e-3) / 1e9, 2) if ms > 0 else 0.0
    if valu is not None and ms > 0:
        rl["valu_insts_per_launch"] = int(valu / launches)
        busy = sq.get("SQ_BUSY_CU_CYCLES")
        if busy:      # counters of the same profiled launch: VALU wave-instructions per busy CU cycle, no clock assumption (ceilings by instruction class
            # and occupancy: 0.90-0.96 for everything but plain 32-bit ALU operations, 1.44-1.77 for those: profiles/r05_valu_issue_rates.txt)
            rl["valu_issue_frac"] = round(valu / busy, 4)
            if "SQ_LDS_IDX_ACTIVE" in sq:
                rl["lds_active_frac"] = round(sq["SQ_LDS_IDX_ACTIVE"] / busy, 4)
                rl["lds_conflict_frac"] = round(sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / busy, 4)
        else:         # older counter files: against the nominal clock (the clock under load is lower: this understates the fraction)
            rl["valu_issue_frac"] = round(valu / (256.0 * ms * 1e-3 * CLOCK_GHZ * 1e9), 4)
        # what the counters say limits the kernel (`frac` stays the HBM fraction of the algorithmic bytes either way).  k_fast_rows: round 5 measured
        # t = a + b / workgroups per CU with b / n more than half of the launch (profiles/r05_late_experiments.txt).
        if stage == "fast_cells":
            rl["limiter"] = "half SIMD throughput, half per-wave issue at 3 waves per SIMD (12 single-wave workgroups per CU by LDS; t = a + b / n); VALU issue %.2f per busy CU cycle of a mix-weighted ceiling ~1.1" % rl["valu_issue_frac"]
        elif rl["valu_issue_frac"] >= 0.7:
            rl["limiter"] = "VALU issue (%.2f per busy CU cycle; ceiling 0.90-0.96 for this instruction class)" % rl["valu_issue_frac"]
    return rl


def fence_fn(torch, dist, world):
    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    return fence


def timed(step, args, fence, torch, dist, world, dev, begin=None, end=None, pause=None):
    """W untimed warm-up steps (twice: once with stage events, once without — the second sizes `inner_repeats`), then exactly K steps bracketed by
    barrier + synchronize on both sides; MAX over ranks.
    The per-stage HIP events (begin / pause / end) are recorded on the FIRST `profiled_steps(args)` of the K timed steps only: an event record
    drains the stream between two stages (28 us per step on MI355X: 5 % of a 16-pair step, 16 % of a single-pair step), so instrumenting every
    step would make the measured throughput a property of the instrumentation."""
    if begin:
        begin()
    for _ in range(max(args.warmup, 1)):        # first warm-up: instrumented (the event pool is created here), includes first-touch costs
        step()
    torch.cuda.synchronize()
    if end:
        end()
    nw = max(args.warmup, 3)                     # second warm-up: exactly what the timed steps run (no events) -> the duration of one pass
    tw = time.perf_counter()
    for _ in range(nw):
        step()
    torch.cuda.synchronize()
    t_pass = (time.perf_counter() - tw) / nw
    # the driver may ask for so few steps that the timed region is a few milliseconds (20 steps = 10 ms): repeat the batch inside a step so that
    # the K steps cover >= --min-timed-ms of GPU time.  `steps` stays the caller's unit; every rank uses the same factor.
    inner = 1
    if args.min_timed_ms > 0 and t_pass > 0:
        inner = max(1, int(np.ceil(1.15 * args.min_timed_ms * 1e-3 / (args.steps * t_pass))))      # 15 % margin: the few warm-up passes carry the final synchronisation, so t_pass overestimates a pass
    if world > 1:
        t = torch.tensor([inner], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        inner = int(t.item())
    args.inner_repeats = inner
    fence()
    n_prof = profiled_steps(args) if begin else 0
    if n_prof:
        begin()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == n_prof and n_prof and pause:
            pause()
        for _ in range(inner):
            step()
    fence()
    t1 = time.perf_counter()
    prof = end() if end else None
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    args.local_elapsed = t1 - t0                 # this rank's own time for the K steps (the line reports min / max over ranks beside the aggregate)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    return float(elapsed.item()), prof


def profiled_steps(args):
    """timed steps that record stage events: --profile-steps, default a tenth of the timed steps (at least one)"""
    p = args.profile_steps if args.profile_steps >= 0 else max(1, args.steps // 10)
    return min(p, args.steps)


def base_line(metric, value, unit, args, world, elapsed):
    return {"metric": metric, "value": round(value, 2), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "inner_repeats": getattr(args, "inner_repeats", 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic"}


# ------------------------------------------------------------------------------------------------ C2 (headline)
def run_c2(args, rank, world, local_rank, dev, torch, dist, HS, N):
    from hyslam_amd.synth import synth_stereo_pair
    from hyslam_amd.distributed import shard_range
    B = args.pairs
    nd = max(1, min(args.distinct, B))
    n_shapes = args.density * max(40, (W * H) // 800)
    pairs = [synth_stereo_pair(1000 + 97 * rank + i, W, H, n_shapes) for i in range(nd)]
    left = torch.from_numpy(np.stack([pairs[i % nd][0] for i in range(B)])).to(dev)
    right = torch.from_numpy(np.stack([pairs[i % nd][1] for i in range(B)])).to(dev)

    nh = max(1, min(args.handles, B))
    exs = [HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NFEAT), device=local_rank) for _ in range(nh)]
    ex = exs[0]
    cap = ex.max_keypoints()
    sp = HS.stereo_params(HS.Camera(fx=1050.0, mbf=1050.0 * 0.12, mnMaxY=float(H)))
    kp_bytes = N.KP_DTYPE.itemsize
    kL = torch.empty(B * cap * kp_bytes, dtype=torch.uint8, device=dev)
    kR = torch.empty_like(kL)
    dL = torch.empty(B * cap * 32, dtype=torch.uint8, device=dev)
    dR = torch.empty_like(dL)
    nL = torch.zeros(B, dtype=torch.int32, device=dev)
    nR = torch.zeros(B, dtype=torch.int32, device=dev)
    uR = torch.empty(B * cap, dtype=torch.float32, device=dev)
    depth = torch.empty_like(uR)
    parts = [shard_range(B, i, nh) for i in range(nh)]          # contiguous blocks of pairs per handle
    lanes = args.lanes if min(b - a for a, b in parts) >= 2 else 1
    for e, (a, b) in zip(exs, parts):
        e.set_lanes(lanes)
        e.reserve(W, H, 2 * (b - a))

    def step():
        # every handle enqueues on its own stream (stream argument 0); nothing synchronises between handles or between steps
        for e, (a, b) in zip(exs, parts):
            e.stereo_frontend_batch_device(left.data_ptr() + a * W * H, right.data_ptr() + a * W * H, b - a, W, H, W, W * H,
                                           kL.data_ptr() + a * cap * kp_bytes, dL.data_ptr() + a * cap * 32, nL.data_ptr() + 4 * a,
                                           kR.data_ptr() + a * cap * kp_bytes, dR.data_ptr() + a * cap * 32, nR.data_ptr() + 4 * a,
                                           cap, sp, uR.data_ptr() + 4 * a * cap, depth.data_ptr() + 4 * a * cap, 0)

    def profile_begin():
        for e in exs:
            e.profile_begin()

    def profile_pause():
        for e in exs:
            e.profile_pause()

    def profile_end():
        tot = {}
        for e in exs:
            for k, (ms, c) in e.profile_end().items():
                m0, c0 = tot.get(k, (0.0, 0))
                tot[k] = (m0 + ms, c0 + c)
        return tot

    elapsed, prof = timed(step, args, fence_fn(torch, dist, world), torch, dist, world, dev, profile_begin, profile_end, profile_pause)
    value = world * B * args.steps * args.inner_repeats / elapsed
    n_left = nL.cpu().numpy()
    n_match = int((depth.view(B, cap)[0] > 0).sum().item())
    # what the timed loop left in EVERY pair's output buffers, hashed like tools/make_golden.py hashed the oracle's outputs for the same seeded pairs
    # (tests/golden/bench_c2_seed1000.json: data only); every rank checks its own pairs, the line carries the conjunction
    parity = parity_checksum(rank, args, N, cap, kL, dL, nL, kR, dR, nR, uR, depth)
    per_rank = [B * args.steps * args.inner_repeats / args.local_elapsed]
    if world > 1:
        t = torch.tensor([1 if parity["ok"] else (0 if parity["ok"] is False else 2), int(per_rank[0]), parity["pairs_checked"]], dtype=torch.int64, device=dev)
        allt = torch.empty((world, 3), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allt.view(-1), t)
        flags, per_rank = allt[:, 0].tolist(), [float(v) for v in allt[:, 1].tolist()]
        parity["pairs_checked_per_rank"] = allt[:, 2].tolist()
        parity["ok"] = None if all(f == 2 for f in flags) else all(f != 0 for f in flags)
        parity["ranks_checked"] = sum(f != 2 for f in flags)
        parity["ranks_failed"] = [r for r, f in enumerate(flags) if f == 0]
    if rank != 0:
        rccl_probe_guarded(None, ex, rank, world, dev)
        return
    copy_peak = measured_copy_peak(ex, torch, dev) if args.copy_gib > 0 else None
    lat1 = one_pair_latency(HS, N, torch, dev, local_rank, pairs[0], sp) if (world == 1 and args.latency_calls > 0) else None
    px = pyramid_pixels(ex, W, H)
    per_stage, per_frame = algorithmic_bytes(px, NFEAT)
    frames_per_launch = 2 * max(b - a for a, b in parts) // lanes     # every launch sequence covers its own share of the pairs
    stage_ms = {s: (ms / max(c, 1)) for s, (ms, c) in prof.items()}   # per stage invocation (one launch sequence)
    # dominant kernel among those with an HBM-byte model, by single-launch duration (the quadtree is an LDS kernel: a host stage in the
    # reference with no compulsory HBM bytes in SURVEY.md's accounting; its time is still listed in stage_ms_per_step)
    launches = {"pyramid": ex.pyramid_launches(), "fast_cells": 1, "describe": 1}
    dom = max(launches, key=lambda s: stage_ms[s] / launches[s])
    moved = {"describe": NFEAT * (43 * 48 + 60)}                     # what k_describe requests: 43 rows x 48 B per keypoint + the record
    rls = {s: roofline_block(s, stage_ms, per_stage, frames_per_launch, launches[s], B, moved.get(s)) for s in launches}
    out = base_line("stereo frames/sec ORB extract+match, 1920x1080 @2000 feat", value, "stereo_pairs/s", args, world, elapsed)
    out["config"] = {"workload": "C2: 1920x1080 stereo pair, 2000 features/frame, 8 levels @1.2, extract L+R + stereo match",
                     "pairs_per_step_per_gpu": B * args.inner_repeats, "pairs_per_pass": B, "handles": nh, "lanes_per_handle": lanes, "scene_density": args.density,
                     "distinct_pairs": "%d distinct synthetic pairs per rank; the %d pairs of a step are separate copies in HBM (pair i = distinct pair i %% %d): "
                                       "no address reuse inside a step, every step re-reads the same %d MB of frames" % (nd, B, nd, 2 * B * W * H // 1000000),
                     "sharding": "pairs sharded over ranks, no collective",
                     "keypoints_left_frame0": int(n_left[0]), "stereo_matches_frame0": n_match}
    out["parity_checksum_ok"] = parity["ok"]
    out["pairs_checked"] = parity["pairs_checked"]          # of rank 0's `pairs_per_pass` (N > 1: parity_checksum.pairs_checked_per_rank)
    out["parity_checksum"] = {k: v for k, v in parity.items() if k != "ok"}
    out["per_rank_pairs_per_s"] = {"min": round(min(per_rank), 1), "max": round(max(per_rank), 1)}
    out["timed_region_ms"] = round(elapsed * 1e3, 1)
    if copy_peak is not None:         # both peaks beside each other: the vendor's 8 TB/s and this box's own device-to-device copy
        out["measured_copy_peak"] = copy_peak
        for rl in rls.values():
            rl["peak_measured_copy"] = copy_peak.get("GBps")
            rl["frac_of_measured"] = round(rl["achieved"] / copy_peak["GBps"], 5) if copy_peak.get("GBps") else None
            if "achieved_on_moved_bytes" in rl and copy_peak.get("GBps"):      # describe: its algorithmic bytes include the reference's full-level blur, which it never moves
                rl["frac_on_moved_bytes"] = round(rl["achieved_on_moved_bytes"] / HBM_PEAK_GBS, 5)
                rl["frac_of_measured_on_moved_bytes"] = round(rl["achieved_on_moved_bytes"] / copy_peak["GBps"], 5)
    out["roofline"] = rls[dom]
    out["roofline_other_kernels"] = {s: rls[s] for s in rls if s != dom}
    if lat1 is not None:
        out["latency_one_pair_ms"] = lat1.get("back_to_back_ms")
        out["latency_one_pair"] = lat1
    out["stage_ms_per_step"] = {s: round(v, 5) for s, v in stage_ms.items()}
    out["profiled_steps"] = profiled_steps(args)      # the first of the timed steps; the others record no events
    pair_bytes = 2 * per_frame
    out["end_to_end"] = {"algorithmic_bytes_per_pair": int(pair_bytes), "achieved_GBps": round(value / world * pair_bytes / 1e9, 2),
                         "frac_of_hbm_peak": round(value / world * pair_bytes / 1e9 / HBM_PEAK_GBS, 5),
                         "note": "algorithmic bytes include the reference's full-level blur (2P per frame) that k_describe does not move"}
    moved_pair = pair_bytes - 2 * 2 * sum(px)             # without the blur's read + write of every level
    out["end_to_end"]["moved_bytes_per_pair"] = int(moved_pair)
    out["end_to_end"]["frac_on_moved_bytes"] = round(value / world * moved_pair / 1e9 / HBM_PEAK_GBS, 5)
    if copy_peak is not None and copy_peak.get("GBps"):
        out["end_to_end"]["peak_measured_copy"] = copy_peak["GBps"]
        out["end_to_end"]["frac_of_measured"] = round(value / world * pair_bytes / 1e9 / copy_peak["GBps"], 5)
        out["end_to_end"]["frac_of_measured_on_moved_bytes"] = round(value / world * moved_pair / 1e9 / copy_peak["GBps"], 5)
    if world == 1 and args.pcie_seconds > 0:
        out["pcie_inclusive"] = pcie_inclusive(HS, exs[0], sp, pairs, min(B, 16), args.pcie_seconds)      # 16 pairs per ticket (two tickets in flight), whatever the step's batch
    if world == 1 and args.call_site:
        cs = call_site()
        if cs is not None:
            out["call_site"] = cs
    if world == 1 and args.cpu_seconds > 0 and not under_profiler():
        out["cpu_baseline"] = cpu_baseline(pairs, args.cpu_seconds)
    if world > 1:
        rccl_probe_guarded(out, ex, rank, world, dev)      # prints the line (with the `rccl` block) itself
        return
    print(json.dumps(out), flush=True)


def rccl_probe_guarded(out, ex, rank, world, dev, timeout_s=60.0):
    """N > 1, AFTER everything else: evidence for the first multi-GPU run — a throw-away RCCL communicator over all ranks through the C ABI (hs_comm_*;
    C2 itself needs no collective), reporting what RCCL says about it (ncclCommCount per rank, version) and that a 16-byte all-gather moved bytes.
    The measurement is complete when this runs, so it must not be able to take the line down: ncclCommInitRank has never run with more than one rank
    anywhere (no multi-GPU node was available to this build), and if it blocks, a watchdog prints the line with an error note and ends every rank with
    exit code 0.  Rank 0 passes the finished line as `out` and prints it; the other ranks pass None."""
    import threading
    lock, state = threading.Lock(), {"printed": False}

    def emit(info):
        with lock:
            if state["printed"]:
                return
            state["printed"] = True
            if out is not None:
                out["rccl"] = info
                out["rccl_probe_ok"] = "error" not in info      # a hung or failed probe is visible at the top level of the line (the exit code stays 0: the measurement is complete)
                print(json.dumps(out), flush=True)

    def watchdog():
        time.sleep(timeout_s)
        with lock:
            late = not state["printed"]
        if late:
            emit({"error": "the RCCL probe did not finish within %.0f s (the timed measurement above is complete)" % timeout_s})
            sys.stdout.flush()
            os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        from hyslam_amd.distributed import rccl_probe
        info = rccl_probe(ex, rank, world, dev)
    except Exception as e:
        info = {"error": str(e)[:200]}
    emit(info)


def parity_checksum(rank, args, N, cap, kL, dL, nL, kR, dR, nR, uR, depth):
    """sha256 of EVERY pair's outputs in the timed loop's buffers (keypoints L, descriptors L, keypoints R, descriptors R, uRight, depth — the valid
    prefixes) against the committed oracle checksums of the rank's distinct seeded pairs (pair j of a step = distinct pair j % distinct;
    tests/golden/bench_c2_seed1000.json holds 4 per rank).  ok = True / False, or None when no committed checksum applies (--density 3, rank > 7);
    `pairs_checked` = how many of the step's pairs were compared, `pairs_failed` = their indices."""
    import hashlib
    B = int(nL.numel())
    nd = max(1, min(args.distinct, B))
    res = {"ok": None, "fixture": "tests/golden/bench_c2_seed1000.json", "rank0_outputs_sha256": None, "pairs_per_pass": B, "pairs_checked": 0, "pairs_failed": []}
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_c2_seed1000.json")))["ranks"].get(str(rank))
    except Exception:
        gold = None
    applies = gold is not None and args.density == 1 and (W, H, NFEAT) == (1920, 1080, 2000)
    gp = (gold.get("pairs") or [gold]) if gold is not None else []
    kb = N.KP_DTYPE.itemsize
    n_l, n_r = nL.cpu().numpy(), nR.cpu().numpy()
    h_kL, h_kR = kL.cpu().numpy().reshape(B, cap * kb), kR.cpu().numpy().reshape(B, cap * kb)
    h_dL, h_dR = dL.cpu().numpy().reshape(B, cap * 32), dR.cpu().numpy().reshape(B, cap * 32)
    h_uR, h_dp = uR.cpu().numpy().reshape(B, cap), depth.cpu().numpy().reshape(B, cap)
    distinct_hashes = {}
    for j in range(B):
        a, b = int(n_l[j]), int(n_r[j])
        h = hashlib.sha256()
        if 0 <= a <= cap and 0 <= b <= cap:
            for row, n, item in ((h_kL[j], a, kb), (h_dL[j], a, 32), (h_kR[j], b, kb), (h_dR[j], b, 32)):
                h.update(row[:n * item].tobytes())
            h.update(h_uR[j, :a].tobytes())
            h.update(h_dp[j, :a].tobytes())
        hx = h.hexdigest()
        if j == 0:
            res["rank0_outputs_sha256"] = hx
        distinct_hashes.setdefault(j % nd, set()).add(hx)
        if applies and (j % nd) < len(gp):
            g = gp[j % nd]
            res["pairs_checked"] += 1
            if not (hx == g["outputs_sha256"] and a == g["nL"] and b == g["nR"]):
                res["pairs_failed"].append(j)
    res["copies_identical"] = all(len(v) == 1 for v in distinct_hashes.values())      # every copy of a distinct pair gave the same bytes (holds with or without a fixture)
    if applies:
        res["ok"] = res["pairs_checked"] > 0 and not res["pairs_failed"] and res["copies_identical"]
        res["expected_sha256"] = gp[0]["outputs_sha256"]
    return res


def measured_copy_peak(ex, torch, dev, gib=2.0, reps=6):
    """the HBM roofline is quoted against the vendor peak AND against what a device-to-device copy reaches on THIS box, timed in this
    run: `gib` GiB (>= 2: eight times the 256 MiB Infinity Cache) copied (a) by hipMemcpyAsync device-to-device and (b) by the library's 16-byte-per-lane
    grid-stride copy kernel (hs_debug_stream_copy: the access width of this library's own streaming loads).  GB/s counts read + written bytes.
    torch.cuda.Event on torch's current stream, on which both copies are enqueued."""
    import ctypes as C
    try:
        n = int(gib * (1 << 30)) // 4096 * 4096
        src = torch.empty(n, dtype=torch.uint8, device=dev)
        dst = torch.empty(n, dtype=torch.uint8, device=dev)
        src.view(torch.int32).random_(0, 1 << 30)         # not zeros: zero pages can be special-cased by the memory system
        dst.zero_()
        st = torch.cuda.Stream()                           # an explicit stream: both copies and both events are enqueued on it (stream 0 would mean "the handle's own stream" to the library)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

        def run(fn):
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st)
            e1.synchronize()
            return 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9

        def memcpy():
            rc = hip.hipMemcpyAsync(dst.data_ptr(), src.data_ptr(), n, 3, st.cuda_stream)        # 3 = hipMemcpyDeviceToDevice
            if rc:
                raise RuntimeError("hipMemcpyAsync -> %d" % rc)

        def kernel():
            ex.debug_stream_copy(dst.data_ptr(), src.data_ptr(), n, 16, st.cuda_stream)

        def kernel4():
            ex.debug_stream_copy(dst.data_ptr(), src.data_ptr(), n, 64, st.cuda_stream)

        torch.cuda.synchronize()
        g_memcpy = run(memcpy)
        dst.zero_()
        torch.cuda.synchronize()
        g_kernel = run(kernel)
        dst.zero_()
        torch.cuda.synchronize()
        g_kernel4 = run(kernel4)
        torch.cuda.synchronize()
        same = bool(torch.equal(dst[:1 << 20], src[:1 << 20]) and torch.equal(dst[-(1 << 20):], src[-(1 << 20):]))
        del src, dst
        torch.cuda.empty_cache()
        return {"GBps": round(max(g_memcpy, g_kernel, g_kernel4), 1), "hipMemcpyAsync_d2d_GBps": round(g_memcpy, 1), "copy_kernel_16B_per_lane_GBps": round(g_kernel, 1),
                "copy_kernel_4x16B_per_lane_nt_GBps": round(g_kernel4, 1),
                "bytes_copied": n, "reps": reps, "counts": "read + written bytes", "copy_verified": same}
    except Exception as e:      # a secondary figure must never take the headline down
        return {"GBps": None, "error": str(e)[:200]}


def one_pair_latency(HS, N, torch, dev, local_rank, pair, sp, calls=400):
    """What a SLAM front end lives on: ONE 1080p pair per call of hs_stereo_frontend_batch_device on a handle of its own (the small-batch launch plan).
    `back_to_back_ms` = calls enqueued without waiting (the device time a pair occupies the GPU: 1 / throughput at one pair per call);
    `synchronous_ms` = enqueue + hs_orb_synchronize per call (what a caller that needs the result before its next step waits for; host launch cost included)."""
    try:
        ex1 = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NFEAT), device=local_rank)
        cap = ex1.max_keypoints()
        kb = N.KP_DTYPE.itemsize
        L, R = torch.from_numpy(pair[0]).to(dev), torch.from_numpy(pair[1]).to(dev)
        mk = lambda n, dt=torch.uint8: torch.zeros(n, dtype=dt, device=dev)
        kL, kR, dL, dR, nL, nR = mk(cap * kb), mk(cap * kb), mk(cap * 32), mk(cap * 32), mk(1, torch.int32), mk(1, torch.int32)
        uR, depth = mk(cap, torch.float32), mk(cap, torch.float32)
        ex1.reserve(W, H, 2)

        def call():
            ex1.stereo_frontend_batch_device(L.data_ptr(), R.data_ptr(), 1, W, H, W, W * H, kL.data_ptr(), dL.data_ptr(), nL.data_ptr(),
                                             kR.data_ptr(), dR.data_ptr(), nR.data_ptr(), cap, sp, uR.data_ptr(), depth.data_ptr(), 0)
        for _ in range(20):
            call()
        ex1.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            call()
        ex1.synchronize()
        b2b = (time.perf_counter() - t0) / calls * 1e3
        t0 = time.perf_counter()
        for _ in range(calls):
            call()
            ex1.synchronize()
        syn = (time.perf_counter() - t0) / calls * 1e3
        return {"back_to_back_ms": round(b2b, 4), "synchronous_ms": round(syn, 4), "calls": calls, "keypoints_left": int(nL.item()),
                "launches_per_call": sum(max(1, ex1._lib.hs_orb_stage_launches(ex1._h, st)) for st in range(6))}
    except Exception as e:
        return {"back_to_back_ms": None, "error": str(e)[:200]}


def under_profiler():
    """True when a profiler's preloaded library rides along (rocprofv3 sets LD_PRELOAD / ROCPROF* / ROCP_*): it initialises the GPU in every child
    process it is inherited by, so the secondary legs that start children (call_site, cpu_baseline) are skipped under it."""
    e = os.environ
    pre = e.get("LD_PRELOAD", "")
    return ("rocprof" in pre or "roctracer" in pre or "rocprofiler" in pre
            or any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER_")) for k in e))


def call_site():
    """What hySLAM's call sites would see through the drop-in C++ classes (tests/cpp/bench_adaptor.cpp compiled against hyslam_amd/host/*.h by
    __graft_entry__.build()): ImageProcessing::ProcessStereoImage with two HipORBExtractor threads + HipStereomatcher, and the
    optional one-call HipStereoFrontend, in ms per 1080p pair, split into gather / C ABI / scatter.  A child process, run after the timed loop."""
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "bench_adaptor")
    if not os.path.exists(exe) or under_profiler():
        return None
    import tempfile
    from hyslam_amd.synth import synth_stereo_pair
    try:
        with tempfile.TemporaryDirectory() as td:
            L, R = synth_stereo_pair(1000, W, H)
            fl, fr = os.path.join(td, "L.raw"), os.path.join(td, "R.raw")
            L.tofile(fl); R.tofile(fr)
            r = subprocess.run([exe, str(W), str(H), fl, fr, "20", "50000"], capture_output=True, timeout=180)
            # the same with the whole child confined to ONE last-level-cache domain: on a multi-socket host the application's own threads (hySLAM creates its
            # left-extractor thread per frame) otherwise wander between the sockets (profiles/r05_late_experiments.txt); a deployment choice, reported beside
            confined = None
            try:
                cpus = open("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % sorted(os.sched_getaffinity(0))[0]).read().strip()
                rc = subprocess.run([exe, str(W), str(H), fl, fr, "20", "50000", cpus], capture_output=True, timeout=180)      # the child confines ITSELF (sched_setaffinity before its first HIP call): no re-exec'ing launcher in between
                if rc.returncode == 0:
                    dc = json.loads(rc.stdout.decode())
                    confined = {"cpus": cpus, "how": "sched_setaffinity at the top of the child's main (before any HIP call)", "TrackLocalMap_ms": dc["TrackLocalMap_SearchByProjection_ms"]["total"],
                                "ProcessStereoImage_ms_per_pair": dc["ProcessStereoImage_ms"]["total"], "HipStereoFrontend_pipelined_ms_per_pair": dc["HipStereoFrontend_ms"]["pipelined_per_pair"]}
            except Exception:
                confined = None
        if r.returncode != 0:
            return {"error": (r.stdout + r.stderr).decode(errors="replace")[-200:]}
        d = json.loads(r.stdout.decode())
        p, f, t = d["ProcessStereoImage_ms"], d["HipStereoFrontend_ms"], d["TrackLocalMap_SearchByProjection_ms"]
        return {"source": "tests/cpp/bench_adaptor: the C++ adaptors compiled against host/cv_compat.h's STAND-INS for cv::Mat / FeatureDescriptor / FeatureViews / Frame / MapPoint "
                          "(OpenCV and hySLAM are not in this image): allocation costs are glibc malloc's, not OpenCV's allocator's",
                "TrackLocalMap_ms": t["total"],
                "TrackLocalMap_split_ms": {"landmarks": t["landmarks"], "matches": t["matches"], "gather": t["gather"], "c_abi": t["c_abi"], "scatter": t["scatter"],
                                           "associateLandMark_calls": t.get("associateLandMark_calls"), "of_full_replay": t.get("of_full_replay"),
                                           "frame_on_device": t.get("frame_on_device")},
                "ProcessStereoImage_ms_per_pair": p["total"], "ProcessStereoImage_pairs_per_s": round(1e3 / p["total"], 1) if p["total"] > 0 else None,
                "ProcessStereoImage_split_ms": {"extract_LR_threads": p["extract_LR_threads"], "of_which_c_abi": p["extract_c_abi"], "of_which_scatter": p["extract_scatter"],
                                                "FeatureViews_ctor": p["FeatureViews_ctor"], "stereo_gather": p["stereo_gather"], "stereo_c_abi": p["stereo_c_abi"], "getData": p["getData"],
                                                "stereo_frames_on_device": p.get("stereo_frames_on_device")},
                "HipStereoFrontend_ms_per_pair": f["process_total"], "HipStereoFrontend_split_ms": {"submit_plus_wait": f["submit_plus_wait"], "FeatureViews_build": f["FeatureViews_build"]},
                "HipStereoFrontend_pipelined_ms_per_pair": f["pipelined_per_pair"], "keypoints": d["keypoints"], "stereo_matches": d["stereo_matches"],
                "thread_placement": "free (the scheduler's); the adaptor's own helper threads follow their caller's L3 domain", "confined_to_one_L3_domain": confined}
    except Exception as e:      # a secondary figure must never take the headline down
        return {"error": str(e)[:200]}


def pcie_inclusive(HS, ex, sp, pairs, B, seconds):
    """Secondary figure, never `value`: the same workload fed from HOST memory through the pipelined ingest of the C ABI (hs_orb_submit_batch /
    hs_orb_wait, two tickets in flight: the H2D copy of batch i+1 runs under the kernels of batch i, results come back into page-locked memory
    on a third stream) — frames in page-locked host memory (hs_host_alloc), keypoints / descriptors / uRight / depth delivered to host arrays.
    The reference is fed the same way: host frames through a bounded queue (System.cc:194-196, ImageProcessing.cpp:69-116)."""
    try:
        pin = ex.pinned_frames(2 * B, H, W)
        for i in range(B):
            pin[i], pin[B + i] = pairs[i % len(pairs)]
        imgs = [pin[i] for i in range(2 * B)]
        t = [ex.submit_batch(imgs, sp), ex.submit_batch(imgs, sp)]
        outs = [ex.wait(t[0]), ex.wait(t[1])]
        t = [ex.submit_batch(imgs, sp), ex.submit_batch(imgs, sp)]
        n, k, t0 = 0, 0, time.perf_counter()
        while True:
            outs[k] = ex.wait(t[k], outs[k])
            n += 1
            if time.perf_counter() - t0 >= seconds and n >= 4:
                break
            t[k] = ex.submit_batch(imgs, sp)
            k ^= 1
        ex.wait(t[k ^ 1], outs[k ^ 1])
        n += 1
        dt = time.perf_counter() - t0
        return {"value": round(n * B / dt, 1), "unit": "stereo_pairs/s", "GBps_h2d": round(n * 2 * B * W * H / dt / 1e9, 2), "pairs_per_ticket": B,
                "tickets_in_flight": 2, "host_memory": "page-locked (hs_host_alloc)",
                "note": "frames in host memory -> results in host memory; H2D of ticket i+1 under the kernels of ticket i; python binding included"}
    except Exception as e:      # the secondary figure must never take the headline down
        return {"value": None, "error": str(e)[:200]}


# ------------------------------------------------------------------------------------------------ C3
def run_c3(args, rank, world, local_rank, dev, torch, dist, HS, N):
    """BASELINE config 3: 64 synthetic 1920x1080 mono frames per step (8 distinct tiled), extract only; per-kernel achieved GB/s of
    algorithmic bytes (the 'HBM roofline run')."""
    from hyslam_amd.synth import synth_image
    B, nd = 64, 8
    imgs = [synth_image(100 + 64 * rank + i, W, H) for i in range(nd)]
    frames = torch.from_numpy(np.stack([imgs[i % nd] for i in range(B)])).to(dev)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NFEAT), device=local_rank)
    cap = ex.max_keypoints()
    kps = torch.empty(B * cap * N.KP_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    desc = torch.empty(B * cap * 32, dtype=torch.uint8, device=dev)
    n = torch.zeros(B, dtype=torch.int32, device=dev)
    ex.reserve(W, H, B)

    def step():
        ex.extract_batch_device(frames.data_ptr(), B, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), n.data_ptr(), cap, 0)

    elapsed, prof = timed(step, args, fence_fn(torch, dist, world), torch, dist, world, dev, ex.profile_begin, ex.profile_end, ex.profile_pause)
    if rank != 0:
        return
    per_stage, per_frame = algorithmic_bytes(pyramid_pixels(ex, W, H), NFEAT)
    stage_ms = {s: (ms / max(c, 1)) for s, (ms, c) in prof.items() if c}
    gbs = {s: round(per_stage[s] * B / (stage_ms[s] * 1e-3) / 1e9, 1) for s in ("pyramid", "fast_cells", "describe") if s in stage_ms}
    value = world * B * args.steps * args.inner_repeats / elapsed
    out = base_line("mono frames/sec ORB extract, 1920x1080 @2000 feat, batch 64", value, "frames/s", args, world, elapsed)
    out["config"] = {"workload": "C3: 64 x 1920x1080 mono frames per step, 2000 features, extract only", "distinct_frames": nd,
                     "keypoints_frame0": int(n[0].item())}
    out["stage_ms_per_step"] = {s: round(v, 5) for s, v in stage_ms.items()}
    out["profiled_steps"] = profiled_steps(args)      # the first of the timed steps; the others record no events
    out["per_kernel_algorithmic_GBps"] = gbs
    out["end_to_end"] = {"algorithmic_bytes_per_frame": int(per_frame), "achieved_GBps": round(value / world * per_frame / 1e9, 1),
                         "frac_of_hbm_peak": round(value / world * per_frame / 1e9 / HBM_PEAK_GBS, 5)}
    print(json.dumps(out), flush=True)


# ------------------------------------------------------------------------------------------------ C4
def run_c4(args, rank, world, local_rank, dev, torch, dist, HS, N):
    """BASELINE config 4: dual-camera stream + tracking search.  Per step: one 1920x1080 stereo pair (extract L+R + stereo match), one
    4000x3000 documentation-camera frame with the reference's 'Imaging' settings (3000 features, scale 1.4; config/slam_feature_config.yaml:22-29)
    on a second handle / stream, and SearchByProjection (local-map variant, th = 5, nnratio 0.8; slam_tracking_config.yaml:101-103) of a
    50 000-landmark local map against the stereo frame's OWN device-resident outputs (hs_search_by_projection_device).  Harness mirrored:
    ImageProcessing::ProcessStereoImage / ProcessMonoImage (src/main/ImageProcessing.cpp:41-116) + TrackLocalMap (TrackLocalMap.cpp:55-78)."""
    import ctypes as C
    from hyslam_amd.synth import synth_image, synth_stereo_pair, synth_local_map
    IW, IH, IFEAT, L = 4000, 3000, 3000, 50000
    Limg, Rimg = synth_stereo_pair(2 + 31 * rank, W, H)
    left, right = torch.from_numpy(Limg).to(dev), torch.from_numpy(Rimg).to(dev)
    big = torch.from_numpy(synth_image(3 + 31 * rank, IW, IH)).to(dev)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NFEAT), device=local_rank)
    exi = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=IFEAT, fScaleFactor=1.4), device=local_rank)
    ex.reserve(W, H, 2)
    exi.reserve(IW, IH, 1)
    if args.c4_split:                # two cameras share the GPU: split both launch sequences so that their kernels interleave (hs_orb_set_split)
        ex.set_split(1)
        exi.set_split(1)
    cap, icap = ex.max_keypoints(), exi.max_keypoints()
    kb = N.KP_DTYPE.itemsize
    mk = lambda n, dt=torch.uint8: torch.zeros(n, dtype=dt, device=dev)
    kL, kR, dL, dR, nL, nR = mk(cap * kb), mk(cap * kb), mk(cap * 32), mk(cap * 32), mk(1, torch.int32), mk(1, torch.int32)
    uR, depth = mk(cap, torch.float32), mk(cap, torch.float32)
    ik, idesc, inn = mk(icap * kb), mk(icap * 32), mk(1, torch.int32)
    fx = 1050.0
    sp = HS.stereo_params(HS.Camera(fx=fx, mbf=fx * 0.12, mnMaxY=float(H)))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def frontend():
        ex.stereo_frontend_batch_device(left.data_ptr(), right.data_ptr(), 1, W, H, W, W * H, kL.data_ptr(), dL.data_ptr(), nL.data_ptr(),
                                        kR.data_ptr(), dR.data_ptr(), nR.data_ptr(), cap, sp, uR.data_ptr(), depth.data_ptr(), s1.cuda_stream)

    # the local map: landmarks back-projected from the frame's own features (the bench cannot call the oracle; the parity test of this shape
    # is tests/test_gpu_matchers.py::test_c4_projection_50k_landmarks_1080p)
    frontend()
    torch.cuda.synchronize()
    n0 = int(nL.item())
    kps0 = kL.cpu().numpy().view(N.KP_DTYPE)[:n0]
    lms = synth_local_map(kps0, dL.cpu().numpy().reshape(-1, 32)[:n0], depth.cpu().numpy()[:n0], L, 77 + rank, fx, fx, W / 2 - 0.5, H / 2 - 0.5)
    d_lms = torch.from_numpy(lms.view(np.uint8).reshape(-1)).to(dev)
    obs = torch.full((cap,), -1, dtype=torch.int32, device=dev)
    F = N.FrameView()
    for i, v in enumerate(np.eye(3, dtype=np.float32).reshape(-1)):
        F.Rcw[i] = float(v)
    F.fx, F.fy, F.cx, F.cy, F.mbf, F.sensor = fx, fx, W / 2 - 0.5, H / 2 - 0.5, fx * 0.12, 1
    F.min_x, F.max_x, F.min_y, F.max_y, F.size_ref, F.n = 0.0, float(W), 0.0, float(H), 31.0, n0
    F.kps, F.desc, F.uR, F.kp_lm_obs = kL.data_ptr(), dL.data_ptr(), uR.data_ptr(), obs.data_ptr()
    pp = N.ProjParams(5.0, 100.0, 0.8, 0.5, 1.5, 1, 1, 0)
    midx, mdist, nm = mk(L, torch.int32), mk(L, torch.float32), mk(1, torch.int32)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    times = {"stereo_frontend": 0.0, "projection_50k": 0.0, "imaging_extract": 0.0}
    timing = {"on": False}

    def step():
        if timing["on"]:
            ev[0].record(s1)
        frontend()
        if timing["on"]:
            ev[1].record(s1)
        N.check(ex._h, ex._lib.hs_search_by_projection_device(ex._h, C.byref(F), d_lms.data_ptr(), L, C.byref(pp), midx.data_ptr(), mdist.data_ptr(),
                                                              nm.data_ptr(), s1.cuda_stream))
        if timing["on"]:
            ev[2].record(s1)
            ev[3].record(s2)
        exi.extract_batch_device(big.data_ptr(), 1, IW, IH, IW, IW * IH, ik.data_ptr(), idesc.data_ptr(), inn.data_ptr(), icap, s2.cuda_stream)

    elapsed, _ = timed(step, args, fence_fn(torch, dist, world), torch, dist, world, dev)
    # per-stage device times from a few serialised extra steps (events on the launch streams; outside the timed region)
    timing["on"] = True
    reps = 10
    for _ in range(reps):
        torch.cuda.synchronize()
        step()
        e_end = torch.cuda.Event(enable_timing=True)
        e_end.record(s2)
        torch.cuda.synchronize()
        times["stereo_frontend"] += ev[0].elapsed_time(ev[1]) / reps
        times["projection_50k"] += ev[1].elapsed_time(ev[2]) / reps
        times["imaging_extract"] += ev[3].elapsed_time(e_end) / reps
    if rank != 0:
        return
    _, pair_frame = algorithmic_bytes(pyramid_pixels(ex, W, H), NFEAT)
    _, img_frame = algorithmic_bytes(pyramid_pixels(exi, IW, IH), IFEAT)
    proj_bytes = L * lms.dtype.itemsize + n0 * (24 + 32 + 4)
    step_bytes = 2 * pair_frame + img_frame + proj_bytes
    value = world * args.steps * args.inner_repeats / elapsed
    out = base_line("dual-camera steps/sec: 1080p stereo extract+match + 4000x3000 extract + 50k-landmark SearchByProjection", value, "steps/s", args, world, elapsed)
    out["config"] = {"workload": "C4: per step one 1920x1080 stereo pair (2000 feat @1.2) + one 4000x3000 frame (3000 feat @1.4) + SearchByProjection of a "
                                 "50 000-landmark local map (th 5, nnratio 0.8) on the stereo frame's device-resident outputs",
                     "keypoints_stereo_left": n0, "keypoints_imaging": int(inn.item()), "projection_matches": int(nm.item()), "landmarks": L,
                     "split_launch_sequences": bool(args.c4_split)}
    out["stage_ms"] = {k: round(v, 4) for k, v in times.items()}
    dom = max(times, key=times.get)
    dom_bytes = {"stereo_frontend": 2 * pair_frame, "projection_50k": proj_bytes, "imaging_extract": img_frame}[dom]
    ach = dom_bytes / (times[dom] * 1e-3) / 1e9
    out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                       "traffic": None, "algorithmic_bytes_per_launch": int(dom_bytes), "launch_ms": round(times[dom], 4),
                       "note": "stage = the whole launch sequence of that stream (batch 1: latency-bound, see DESIGN.md)"}
    out["end_to_end"] = {"algorithmic_bytes_per_step": int(step_bytes), "achieved_GBps": round(value / world * step_bytes / 1e9, 2),
                         "frac_of_hbm_peak": round(value / world * step_bytes / 1e9 / HBM_PEAK_GBS, 5)}
    print(json.dumps(out), flush=True)


# ------------------------------------------------------------------------------------------------ C5
def run_c5(args, rank, world, local_rank, dev, torch, dist, HS, N):
    """BASELINE config 5: one 1920x1080 mono stream per GPU; per step every rank extracts its frame straight into the all-gather
    record, one RCCL all-gather moves all records, then each rank matches its descriptors against every peer's: brute-force 2-NN
    (one launch, counts read from the record headers on the device) or vocabulary-grouped BoW matching.  Nothing synchronises with the
    host inside a step.  value = frames/s over all ranks."""
    from hyslam_amd import distributed as D
    from hyslam_amd.synth import synth_rig
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NFEAT), device=local_rank)
    cap = ex.max_keypoints()
    # an 8-camera rig over one wide scene, neighbours overlapping by three quarters; two instants per camera, alternated
    # step by step, so that a step that reads a buffer another stage is still writing shows up as a wrong count instead of passing silently
    rig = max(world, 8)
    rig_frames = [synth_rig(200 + 10 * t, rig, W, H, cams=(list(range(8)) if world == 1 else [rank % rig])) for t in range(2)]      # (one panorama render per instant)
    frames = [torch.from_numpy(rig_frames[t][0]).to(dev) for t in range(2)]
    rb = D.record_bytes(cap)
    # ONE rank: the exchange cannot run, but the step should still time what a rank of the 8-camera rig does after it — its own extraction and the
    # match against SEVEN real peers.  The peers' records (cameras 1..7 of the same rig, both instants) are extracted ONCE up front into two local
    # gathered buffers (one per instant); a step extracts this rank's frame into record 0 of the instant's buffer and matches it against the 7 others.
    emu = 8 if world == 1 else 0
    nrec = emu or world
    gathered2 = [torch.zeros((nrec, rb), dtype=torch.uint8, device=dev) for _ in range(2 if emu else 1)]
    gathered = gathered2[0]
    o_n, o_k, o_d = D.record_offsets(cap)
    ex.reserve(W, H, 1)
    # ONE explicit stream carries extraction, all-gather and matcher of a step (torch's current stream during the step, so that
    # torch.distributed's collective is ordered on it too): the three stages are serial by construction
    s = torch.cuda.Stream()
    stream = s.cuda_stream
    outs = tuple(torch.zeros((nrec, cap), dtype=torch.int32, device=dev) for _ in range(3))
    if emu:
        for t in range(2):
            for cam in range(1, emu):
                f = torch.from_numpy(rig_frames[t][cam]).to(dev)
                r_ = gathered2[t][cam]
                ex.extract_batch_device(f.data_ptr(), 1, W, H, W, W * H, r_.data_ptr() + o_k, r_.data_ptr() + o_d, r_.data_ptr() + o_n, cap, stream)
        torch.cuda.synchronize()
    xc = None
    exchange_note = None
    rccl = None
    if world > 1 and args.c5_comm == "hs":
        # hs_comm_create blocks in ncclCommInitRank until ALL ranks arrive: ask every rank first (non-collective probe) and fall back TOGETHER
        ok, why = D.comm_available()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            args.c5_comm = "torch"
            exchange_note = "hs_comm unavailable on at least one rank (%s): fell back to torch.distributed.all_gather_into_tensor before any collective" % (why or "another rank")
    if world > 1 and args.c5_comm == "hs":
        ident = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            ident.copy_(torch.frombuffer(bytearray(D.RecordExchange.unique_id()), dtype=torch.uint8))
        dist.broadcast(ident, 0)
        xc = D.RecordExchange(ex, bytes(ident.cpu().numpy().tobytes()), world, rank)
        rccl = xc.rccl_info()                 # what RCCL itself says about the communicator the timed steps use
    # hs path: the extractor writes straight into this rank's slot of the gathered buffer and the all-gather runs in place;
    # torch path: a separate send buffer (all_gather_into_tensor does not promise in-place operation)
    rec = gathered[rank] if (world == 1 or xc is not None) else torch.zeros(rb, dtype=torch.uint8, device=dev)
    bow = None
    if args.c5_match == "bow":
        if not hasattr(D, "BowCrossCamera"):
            sys.exit("bench.py: --c5-match bow needs the device-resident vocabulary path (hyslam_amd.distributed.BowCrossCamera)")
        bow = D.BowCrossCamera(ex, nrec, cap, seed=17)
    tick = {"n": 0}

    def step():
        t = tick["n"] & 1
        frame = frames[t]
        tick["n"] += 1
        with torch.cuda.stream(s):
            if emu:
                g = gathered2[t]
                r0 = g[0]
                ex.extract_batch_device(frame.data_ptr(), 1, W, H, W, W * H, r0.data_ptr() + o_k, r0.data_ptr() + o_d, r0.data_ptr() + o_n, cap, stream)
                if bow is not None:
                    bow.match(g, 0, stream)
                else:
                    D.cross_camera_knn2(ex, g, 0, cap, stream, out=outs)
                return
            ex.extract_batch_device(frame.data_ptr(), 1, W, H, W, W * H, rec.data_ptr() + o_k, rec.data_ptr() + o_d, rec.data_ptr() + o_n, cap, stream)
            if xc is not None:
                xc.allgather(rec.data_ptr(), gathered.data_ptr(), rb, stream)
            elif world > 1:
                dist.all_gather_into_tensor(gathered.view(-1), rec)
            if bow is not None:
                bow.match(gathered, rank, stream)
            else:
                D.cross_camera_knn2(ex, gathered, rank, cap, stream, out=outs)

    elapsed, _ = timed(step, args, fence_fn(torch, dist, world), torch, dist, world, dev)
    if emu:
        gathered = gathered2[(tick["n"] - 1) & 1]                  # the buffer the LAST step matched in
    counts = gathered[:, :4].view(torch.int32)[:, 0].cpu().tolist()
    # every rank must hold the same gathered bytes, and record r must be rank r's own frame (hyslam_amd.distributed.verify_exchange)
    if emu:
        check = {"ranks_consistent": True, "note": "one rank: nothing was exchanged (8 local records)", "record_counts": counts}
    else:
        check = D.verify_exchange(gathered, counts[rank], rank, world, match_outputs=(outs if bow is None else (bow.match12, bow.n_matches)))
    if xc is not None:
        xc.close()
    good = 0
    if bow is None:
        n = counts[rank]
        for peer in range(nrec):
            if peer != rank:
                bd, sd = outs[1][peer, :n], outs[2][peer, :n]
                good += int(((bd < 50) & (bd.float() < 0.8 * sd.float())).sum().item())
    else:
        good = bow.total_matches(rank)
    if rank != 0:
        if not check["ranks_consistent"]:
            sys.exit(4)                      # every rank holds the same verdict (it was all-gathered): the whole job fails, not just rank 0
        return
    out = base_line("mono frames/sec ORB extract + all-gather + cross-camera match, 1920x1080 @2000 feat", world * args.steps * args.inner_repeats / elapsed, "frames/s",
                    args, world, elapsed)
    out["config"] = {"workload": "C5: one 1920x1080 mono stream per GPU, 2000 features, all-gather of %d-byte records, %s vs every peer"
                                 % (rb, "brute-force Hamming 2-NN" if bow is None else "vocabulary transform + BoW-grouped match (synthetic 10-ary vocabulary)"),
                     "exchange": ("emulated (8 local records): one rank, no collective — the step = this rank's extraction + the match against 7 pre-extracted peers of the same rig" if emu
                                  else ("hs_comm_allgather_records (RCCL through the C ABI), in place" if xc is not None else "torch.distributed.all_gather_into_tensor")),
                     "records": nrec, "keypoints_rank0": counts[0], "matches_rank0": good}
    if rccl is not None:
        out["rccl"] = rccl
    if exchange_note:
        out["config"]["exchange_note"] = exchange_note
    out["ranks_consistent"] = check["ranks_consistent"]
    out["exchange_check"] = check
    print(json.dumps(out), flush=True)
    if not check["ranks_consistent"]:
        sys.stderr.write("bench.py: the ranks disagree on the gathered records (%s)\n" % json.dumps(check))
        sys.exit(4)


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(pairs, budget_s):
    """The oracle in the reference's structure (ImageProcessing::ProcessStereoImage, src/main/ImageProcessing.cpp:69-116): left frame on a
    spawned thread, right on the caller, then the stereo matcher.  Two figures: `cpu_ref_structure` = one instance
    (2 host threads per pair), `cpu_all_cores` = nproc/2 such instances side by side.
    Both legs run in worker PROCESSES (bench.py --cpu-worker i: numpy + the oracle, no torch, no GPU): round 4 ran the all-cores leg as 128 Python
    THREADS of this process and measured 72.7 pairs/s on 256 cores against 14.6 on two — every oracle call allocates and first-touches ~20 MB of
    pyramid, and 128 threads of ONE process serialise on its address-space lock (page faults, mmap / munmap).  Separate processes do not, and
    MALLOC_*_THRESHOLD_ keeps a worker's buffers mapped between calls (for the single instance too: same code, same environment).  The workers of
    the all-cores leg are pinned to two logical CPUs each."""
    import tempfile
    nproc = os.cpu_count() or 2

    def leg(workers, pin, seconds):
        total, el_all, per_worker, note = 0, 0.0, [], None
        try:
            with tempfile.TemporaryDirectory() as td:
                fpath = os.path.join(td, "frames.npy")
                np.save(fpath, np.stack([np.stack(pr) for pr in pairs]))
                env = dict(os.environ, MALLOC_MMAP_THRESHOLD_=str(1 << 30), MALLOC_TRIM_THRESHOLD_=str(1 << 30), MALLOC_TOP_PAD_=str(64 << 20), OMP_NUM_THREADS="1")
                go = time.time() + 3.0 + 0.01 * workers               # common start: the workers import numpy and warm up first, then wait for `go`
                cmd = [sys.executable, os.path.abspath(__file__), "--cpu-frames", fpath, "--cpu-seconds", str(seconds), "--cpu-workers", str(workers if pin else 0)]
                procs = [subprocess.Popen(cmd + ["--cpu-worker", str(i)], env=dict(env, HS_CPU_GO=repr(go)), stdout=subprocess.PIPE) for i in range(workers)]
                for pr_ in procs:
                    o, _ = pr_.communicate(timeout=seconds * 6 + 120)
                    try:
                        r = json.loads(o.decode().strip().splitlines()[-1])
                        per_worker.append(r["pairs"] / r["seconds"]); total += r["pairs"]; el_all = max(el_all, r["seconds"])
                    except Exception:
                        note = "a worker process returned nothing"
        except Exception as e:
            note = str(e)[:200]
        return total, el_all, per_worker, note

    n, el, one, note1 = leg(1, False, budget_s)
    ref = one[0] if one else None
    # How many cores does this job really have?  os.cpu_count() says what the MACHINE has (256 logical cores on the GPU box), not what the job may
    # use: the box gives one GPU's job a CPU share (a cgroup quota) of a fraction of that, and 128 workers on a 16-CPU share is what round 4 measured
    # as "72.7 pairs/s on all 256 cores".  So: the affinity mask and the cgroup quota when they are visible, and in any case a short scaling probe
    # (2-s legs, the worker count doubling while the throughput still grows by >= 25 %) — the all-cores figure is the best point of that curve,
    # measured again over the full budget.
    cores_hint, how = effective_cpus()
    probe, w = {}, max(1, min(4, nproc // 2))
    best_w, best_v = 1, ref or 0.0
    while w <= max(1, nproc // 2):
        t_, e_, pw, _ = leg(w, True, min(2.0, budget_s))
        v = sum(pw) if pw else 0.0
        probe[str(w)] = round(v, 2)
        if v > 1.1 * best_v:                                          # more workers only when they buy >= 10 %: past the job's CPU share they only time-slice
            best_w, best_v = w, v
        if v < 1.25 * probe.get(str(w // 2), 0.0) or w == nproc // 2:
            break
        w = min(2 * w, max(1, nproc // 2))
    workers = best_w
    total, el_all, per_worker, note = leg(workers, True, budget_s)
    final_v = sum(per_worker) if per_worker else None
    probe_v = probe.get(str(workers))
    # the all-cores figure is the BEST the box showed at this worker count (the 2-s probe leg or the full-budget leg: the job's CPU quota is throttled in
    # bursts, and in round 5 the two disagreed by 25 % at the same worker count); both are printed, with their spread
    best_v = max([v for v in (final_v, probe_v) if v is not None], default=None)
    all_cores = {"value": round(best_v, 2) if best_v is not None else None, "full_budget_leg": round(final_v, 2) if final_v is not None else None, "probe_leg": probe_v,
                 "spread_between_legs": round(abs(final_v - probe_v) / max(final_v, probe_v), 3) if (final_v and probe_v) else None,
                 "threads": 2 * workers, "nproc": nproc, "workers": workers,
                 "cpus_available_to_this_job": cores_hint, "cpus_available_how": how, "scaling_probe_pairs_per_s_by_workers": probe,
                 "kind": "worker processes, 2 threads each, pinned to 2 logical CPUs each; worker count = the best point of the scaling probe",
                 "per_worker_pairs_per_s": {"min": round(min(per_worker), 3), "median": round(float(np.median(per_worker)), 3), "max": round(max(per_worker), 3)} if per_worker else None,
                 "sample": "%d pairs in %.1f s" % (total, el_all)}
    if best_v and ref:
        all_cores["speedup_over_ref_structure"] = round(best_v / ref, 1)
    if note:
        all_cores["note"] = note
    out = {"value": None if ref is None else round(ref, 3), "unit": "stereo_pairs/s", "cores": 2, "kind": "port",
           "cpu_ref_structure": {"value": None if ref is None else round(ref, 3), "threads": 2, "kind": "one worker process, 2 threads, not pinned"},
           "cpu_all_cores": all_cores,
           "sample": "%d synthetic 1920x1080 pairs in %.1f s; oracle/ C++ restatement (left||right threads + stereo match), "
                     "omits the reference's cv::Mat/FeatureDescriptor allocation overheads (an optimistic stand-in); host has %d logical cores, "
                     "this job may use %s" % (n, el, nproc, cores_hint if cores_hint else "an unknown share")}
    if note1:
        out["note"] = note1
    return out


def effective_cpus():
    """(cpus this job may use, how it is known): the affinity mask capped by the cgroup CPU quota (v2: cpu.max, v1: cpu.cfs_quota_us / cpu.cfs_period_us)"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    how = "affinity mask"
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            c = int(np.ceil(float(q[0]) / float(q[1])))
            if c < n:
                n, how = c, "cgroup v2 cpu.max"
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0 and int(np.ceil(quota / period)) < n:
                n, how = int(np.ceil(quota / period)), "cgroup v1 cfs quota"
        except Exception:
            pass
    return n, how


def cpu_worker(args):
    """One worker process of cpu_baseline()'s all-cores leg: the oracle's stereo front end (2 threads) on the frames of --cpu-frames until the budget is
    spent.  Pinned to two logical CPUs (worker i -> CPUs 2i, 2i+1 of the allowed set).  Prints {"pairs", "seconds"}.  Touches neither torch nor the GPU."""
    i = args.cpu_worker
    try:
        cpus = sorted(os.sched_getaffinity(0))
        if len(cpus) >= 2 and args.cpu_workers > 0:
            os.sched_setaffinity(0, {cpus[(2 * i) % len(cpus)], cpus[(2 * i + 1) % len(cpus)]})
    except Exception:
        pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    frames = np.load(args.cpu_frames)
    p = oracle.default_params(NFEAT)
    sp = oracle.stereo_params(fx=1050.0, mbf=1050.0 * 0.12, n_rows=H)
    oracle.stereo_frontend(p, sp, frames[0][0], frames[0][1])      # warm-up: the library is paged in, the heap has its size
    go = float(os.environ.get("HS_CPU_GO", "0"))
    while time.time() < go:
        time.sleep(0.005)
    done, t0 = 0, time.perf_counter()
    deadline = t0 + args.cpu_seconds
    while done < 1 or time.perf_counter() < deadline:
        L, R = frames[(i + done) % len(frames)]
        oracle.stereo_frontend(p, sp, L, R)
        done += 1
    print(json.dumps({"pairs": done, "seconds": time.perf_counter() - t0}), flush=True)
    return 0


def main():
    args = parse_args()
    if args.cpu_worker >= 0:
        sys.exit(cpu_worker(args))                 # a worker of the CPU baseline: no torch, no GPU
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))                # before torch / HIP are touched
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or without it to let "
                 "bench.py spawn the ranks)" % (args.gpus, world, args.gpus))
    if args.dry_run:
        sys.exit(dry_run(args, rank, world))

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    import hyslam_amd as HS
    from hyslam_amd import _native as N
    {"c2": run_c2, "c3": run_c3, "c4": run_c4, "c5": run_c5}[args.config](args, rank, world, local_rank, dev, torch, dist, HS, N)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
