#!/usr/bin/env python3
"""bench.py — stereo frames/s of ORB extract + left<->right match at 1920x1080 / 2000 features (BASELINE.json C2).

A *step* is one pass of the hot path (pyramid -> FAST/NMS cells -> quadtree distribution -> blur+orientation+rBRIEF for
left and right frames, then the stereo matcher) over one batch of `--pairs` synthetic stereo pairs that are already
resident in HBM.  One process per GPU; frames shard across ranks with no data-path collective (weak scaling).

Prints ONE JSON line on rank 0:
  value     whole-job stereo pairs per second (all ranks)
  roofline  the dominant kernel of the step: algorithmic bytes per launch / its HIP-event-measured duration vs 8 TB/s
  cpu_baseline  the oracle ("port" of the reference's CPU path, left||right on 2 threads) timed on this box, N=1 only
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
W, H, NFEAT = 1920, 1080, 2000


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
        "describe": 2 * P + k * 1369 + k * 60,   # blur read+write, 37x37 window per keypoint, record out
    }
    return per_stage, sum(per_stage.values())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=16, help="stereo pairs per step per GPU")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic pairs generated per rank (tiled to --pairs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--handles", type=int, default=2,
                    help="extractor handles per GPU; the pairs of a step are dealt over them and each runs on its own stream (the reference also "
                         "keeps separate extractor instances side by side, ImageProcessing.cpp:31-32).  Two independent launch sequences overlap one "
                         "sequence's latency-bound kernels (quadtree, stereo, small pyramid levels) with the other's issue-bound ones: +10 %")
    ap.add_argument("--lanes", type=int, default=1, choices=[1, 2],
                    help="launch sequences INSIDE one handle (hs_orb_set_lanes): same effect for callers that own a single handle")
    ap.add_argument("--config", choices=["c2", "c3", "c5"], default="c2",
                    help="c2 = stereo extract+match (the headline metric); c3 = batched 64 mono frames, extract only (per-kernel GB/s); "
                         "c5 = one mono stream per GPU + all-gather + cross-camera 2-NN")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    import hyslam_amd as HS
    from hyslam_amd import _native as N
    from hyslam_amd.synth import synth_stereo_pair

    if args.config == "c5":
        return run_c5(args, rank, world, local_rank, dev, torch, dist, HS, N)
    if args.config == "c3":
        return run_c3(args, rank, world, local_rank, dev, torch, dist, HS, N)

    # ---- synthetic input, resident in HBM before the timed region
    B = args.pairs
    nd = max(1, min(args.distinct, B))
    pairs = [synth_stereo_pair(1000 + 97 * rank + i, W, H) for i in range(nd)]
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
    from hyslam_amd.distributed import shard_range
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

    def profile_end():
        tot = {}
        for e in exs:
            for k, (ms, c) in e.profile_end().items():
                m0, c0 = tot.get(k, (0.0, 0))
                tot[k] = (m0 + ms, c0 + c)
        return tot

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    profile_begin()             # warm-up also pre-creates part of the event pool
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    profile_end()

    fence()
    profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t1 = time.perf_counter()
    prof = profile_end()

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    total_pairs = world * B * args.steps
    value = total_pairs / elapsed

    n_left = nL.cpu().numpy()
    n_match = int((depth.view(B, cap)[0] > 0).sum().item())

    out = None
    if rank == 0:
        px = pyramid_pixels(ex, W, H)
        per_stage, per_frame = algorithmic_bytes(px, NFEAT)
        frames_per_launch = 2 * max(b - a for a, b in parts) // lanes     # every launch sequence covers its own share of the pairs
        stage_ms = {s: (ms / max(c, 1)) for s, (ms, c) in prof.items()}
        # dominant kernel among those with an HBM-byte model (the quadtree is a latency-bound LDS kernel: a host stage in the reference,
        # no compulsory HBM bytes in SURVEY.md's accounting; its time is still listed in stage_ms_per_step)
        launches = {"pyramid": max(ex.GetLevels() - 1, 1), "fast_cells": 1, "describe": 1}   # the pyramid stage is one launch of k_resize_level per level
        launch_ms = {s: stage_ms[s] / launches[s] for s in launches}
        dom = max(launches, key=lambda s: launch_ms[s])             # the kernel with the longest launch
        dom_bytes = per_stage[dom] * frames_per_launch / launches[dom]
        achieved = dom_bytes / (launch_ms[dom] * 1e-3) / 1e9 if launch_ms[dom] > 0 else 0.0
        pair_bytes = 2 * per_frame
        traffic, traffic_src = measured_traffic(dom, B, frames_per_launch)
        out = {
            "metric": "stereo frames/sec ORB extract+match, 1920x1080 @2000 feat",
            "value": round(value, 2), "unit": "stereo_pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "C2: 1920x1080 stereo pair, 2000 features/frame, 8 levels @1.2, extract L+R + stereo match",
                       "pairs_per_step_per_gpu": B, "handles": nh, "lanes_per_handle": lanes, "distinct_pairs": nd, "sharding": "frames round-robin, no collective",
                       "keypoints_left_frame0": int(n_left[0]), "stereo_matches_frame0": n_match},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(dom_bytes), "launch_ms": round(launch_ms[dom], 5),
                         "frames_per_launch": frames_per_launch},
            "stage_ms_per_step": {s: round(v, 5) for s, v in stage_ms.items()},
            "end_to_end": {"algorithmic_bytes_per_pair": int(pair_bytes),
                           "achieved_GBps": round(value / world * pair_bytes / 1e9, 2),
                           "frac_of_hbm_peak": round(value / world * pair_bytes / 1e9 / HBM_PEAK_GBS, 5)},
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(pairs, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_c3(args, rank, world, local_rank, dev, torch, dist, HS, N):
    """BASELINE config 3: 64 synthetic 1920x1080 mono frames per step (seeds 100..163, 8 distinct tiled), extract only; per-kernel
    achieved GB/s of algorithmic bytes (the 'HBM roofline run')."""
    from hyslam_amd.synth import synth_image
    B = 64
    nd = 8
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

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ex.profile_begin()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ex.profile_end()
    fence()
    ex.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t1 = time.perf_counter()
    prof = ex.profile_end()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    if rank == 0:
        per_stage, per_frame = algorithmic_bytes(pyramid_pixels(ex, W, H), NFEAT)
        stage_ms = {s: (ms / max(c, 1)) for s, (ms, c) in prof.items() if c}
        gbs = {s: round(per_stage[s] * B / (stage_ms[s] * 1e-3) / 1e9, 1) for s in ("pyramid", "fast_cells", "describe") if s in stage_ms}
        value = world * B * args.steps / elapsed
        print(json.dumps({
            "metric": "mono frames/sec ORB extract, 1920x1080 @2000 feat, batch 64", "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "C3: 64 x 1920x1080 mono frames per step, 2000 features, extract only", "keypoints_frame0": int(n[0].item())},
            "stage_ms_per_step": {s: round(v, 5) for s, v in stage_ms.items()},
            "per_kernel_algorithmic_GBps": gbs,
            "end_to_end": {"algorithmic_bytes_per_frame": int(per_frame), "achieved_GBps": round(value / world * per_frame / 1e9, 1),
                           "frac_of_hbm_peak": round(value / world * per_frame / 1e9 / HBM_PEAK_GBS, 5)}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_c5(args, rank, world, local_rank, dev, torch, dist, HS, N):
    """BASELINE config 5: one 1920x1080 mono stream per GPU; per step every rank extracts its frame straight into the
    all-gather record, one RCCL all-gather moves all records, then each rank runs the Hamming 2-NN of its descriptors against
    every peer's.  value = frames/s over all ranks."""
    from hyslam_amd import distributed as D
    from hyslam_amd.synth import synth_image
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NFEAT), device=local_rank)
    cap = ex.max_keypoints()
    frame = torch.from_numpy(synth_image(200 + rank, W, H)).to(dev)
    rec = torch.zeros(D.record_bytes(cap), dtype=torch.uint8, device=dev)
    o_n, o_k, o_d = D.record_offsets(cap)
    ex.reserve(W, H, 1)
    stream = torch.cuda.current_stream().cuda_stream
    n_matches = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        ex.extract_batch_device(frame.data_ptr(), 1, W, H, W, W * H, rec.data_ptr() + o_k, rec.data_ptr() + o_d, rec.data_ptr() + o_n, cap, stream)
        g = D.all_gather_records(rec) if world > 1 else rec.unsqueeze(0)
        res, counts = D.cross_camera_knn2(ex, g, rank, cap, stream)
        return res, counts

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, counts = step()
    fence()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    good = 0
    for peer, (bi, bd, sd) in res.items():
        n = counts[rank]
        good += int(((bd[:n] < 50) & (bd[:n].float() < 0.8 * sd[:n].float())).sum().item())
    if rank == 0:
        print(json.dumps({
            "metric": "mono frames/sec ORB extract + all-gather + cross-camera 2-NN, 1920x1080 @2000 feat", "value": round(world * args.steps / elapsed, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "C5: one 1920x1080 mono stream per GPU, 2000 features, all-gather of %d-byte records, brute-force Hamming 2-NN vs every peer"
                                   % D.record_bytes(cap), "keypoints_rank0": counts[0], "ratio_test_matches_rank0": good}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(kernel, pairs_per_step, frames_per_launch):
    """HBM bytes per launch of `kernel` from the committed PMC pass (profiles/r01_k_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs of this workload, FETCH_SIZE doubled per the gfx950 calibration).  Counters cannot be read from
    inside this process, so the value is only reported when the profiled batch size matches; otherwise null."""
    path = os.path.join(ROOT, "profiles", "r01_k_hbm_traffic.json")
    try:
        t = json.load(open(path))
        k = {"fast_cells": "k_fast_rows", "pyramid": "k_resize_level_lds", "describe": "k_describe", "quadtree": "k_quadtree"}[kernel]
        e = t["kernels"][k]
        if t["pairs_per_step"] != pairs_per_step or e["frames_per_launch"] != frames_per_launch:
            return None, None
        return int((e["read_MB"] + e["written_MB"]) * 1e6), "profiles/r01_k_hbm_traffic.json"
    except Exception:
        return None, None


def cpu_baseline(pairs, budget_s):
    """The oracle in the reference's structure (ImageProcessing::ProcessStereoImage, src/main/ImageProcessing.cpp:69-116):
    left frame on a spawned thread, right on the caller, then the stereo matcher — 2 host cores per pair."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    p = oracle.default_params(NFEAT)
    sp = oracle.stereo_params(fx=1050.0, mbf=1050.0 * 0.12, n_rows=H)
    oracle.stereo_frontend(p, sp, pairs[0][0], pairs[0][1])      # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        L, R = pairs[n % len(pairs)]
        oracle.stereo_frontend(p, sp, L, R)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s and n >= 4:
            break
    return {"value": round(n / el, 3), "unit": "stereo_pairs/s", "cores": 2, "kind": "port",
            "sample": "%d synthetic 1920x1080 pairs in %.1f s; oracle/ C++ restatement (left||right threads + stereo match), "
                      "omits the reference's cv::Mat/FeatureDescriptor allocation overheads; host has %d logical cores"
                      % (n, el, os.cpu_count())}


if __name__ == "__main__":
    main()
