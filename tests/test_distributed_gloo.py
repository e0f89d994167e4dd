"""The N>1 path on CPU: two `gloo` ranks shard frames, all-gather the per-frame records and must end up with identical
gathered bytes and identical cross-camera match lists (the matcher used here is the oracle — test infrastructure; the
product's GPU 2-NN is covered in tests/test_gpu_matchers.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashlib
    import torch
    import torch.distributed as dist
    import oracle
    from hyslam_amd import distributed as D
    from hyslam_amd.synth import synth_image
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cap = 1000 + 4 * 8 + 4
    lo, hi = D.shard_range(5, rank, world)                       # 5 camera frames over 2 ranks: 3 + 2
    p = oracle.default_params(1000)
    frames = [synth_image(200 + i, 320, 240) for i in range(lo, hi)]
    k, d = oracle.extract(p, frames[0])                          # each rank publishes its first frame
    rec = torch.from_numpy(D.pack_record(k, d, cap))
    g = D.all_gather_records(rec)
    assert g.shape == (world, D.record_bytes(cap))
    mine = D.unpack_record(g[rank].numpy(), cap)
    assert mine[0].tobytes() == k.tobytes() and np.array_equal(mine[1], d)
    matches = {}
    for a in range(world):
        for b in range(world):
            if a != b:
                ka, da = D.unpack_record(g[a].numpy(), cap)
                kb, db = D.unpack_record(g[b].numpy(), cap)
                bi, bd, sd = oracle.hamming_knn2(da, db)
                matches[(a, b)] = hashlib.sha256(bi.tobytes() + bd.tobytes() + sd.tobytes()).hexdigest()
    q.put((rank, (lo, hi), hashlib.sha256(g.numpy().tobytes()).hexdigest(), matches, len(k)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgather_and_identical_match_lists():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [(0, 3), (3, 5)]
    assert res[0][2] == res[1][2], "ranks disagree on the gathered bytes"
    assert res[0][3] == res[1][3], "ranks disagree on the cross-camera match lists"
    assert res[0][4] > 100 and res[1][4] > 100


def test_record_layout_and_sharding():
    from hyslam_amd import distributed as D
    cap = 50
    assert D.record_bytes(cap) == 16 + cap * 56 and D.record_offsets(cap) == (0, 16, 16 + cap * 24)           # an even cap needs no padding
    assert D.record_offsets(51) == (0, 16, (16 + 51 * 24 + 15) & ~15) and D.record_offsets(51)[2] % 16 == 0 and D.record_bytes(51) == D.record_offsets(51)[2] + 51 * 32
    rng = np.random.default_rng(0)
    from hyslam_amd._native import KP_DTYPE
    k = np.zeros(7, KP_DTYPE); k["x"] = rng.random(7); k["octave"] = np.arange(7)
    d = rng.integers(0, 256, (7, 32), dtype=np.uint8)
    rec = D.pack_record(k, d, cap)
    k2, d2 = D.unpack_record(rec, cap)
    assert k2.tobytes() == k.tobytes() and np.array_equal(d, d2)
    e = D.unpack_record(D.pack_record(k[:0], d[:0], cap), cap)
    assert len(e[0]) == 0 and e[1].shape == (0, 32)
    with pytest.raises(ValueError):
        D.pack_record(np.zeros(cap + 1, KP_DTYPE), np.zeros((cap + 1, 32), np.uint8), cap)
    bad = rec.copy(); bad[:4] = np.frombuffer(np.int32(cap + 1).tobytes(), np.uint8)
    with pytest.raises(ValueError):
        D.unpack_record(bad, cap)
    for n in (0, 1, 7, 8, 9, 64):
        for w in (1, 2, 3, 8):
            r = [D.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_bench_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` (no torchrun): the parent spawns two ranks with RANK / WORLD_SIZE / MASTER_* set, relays rank 0's line and
    fails when WORLD_SIZE disagrees with --gpus.  --dry-run keeps the ranks off the GPU (gloo rendezvous only)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["steps"] == 3
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env2, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_spawned_bench_fails_fast_when_a_rank_dies():
    """`python bench.py --gpus 2` must not report success (or hang until the process-group timeout) when one of the ranks it started dies:
    the parent polls all children, stops the survivors and exits non-zero"""
    import time
    env = dict(os.environ, HS_BENCH_TEST_FAIL_RANK="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "a rank failed" in r.stderr, r.stdout + r.stderr
    assert time.time() - t0 < 120


def test_c5_self_check_over_gloo_and_a_corrupted_rank():
    """bench.py --config c5 verifies what it measured: every rank checksums every gathered record, the checksums are all-gathered and the job
    fails unless all ranks hold the same bytes and record r carries rank r's own count (hyslam_amd.distributed.verify_exchange).  Here over gloo
    at world 2 through bench.py's own spawn path (--dry-run: stand-in records, the real all-gather and check): consistent ranks pass, and when one
    rank's copy of the gathered buffer is corrupted the line says so and the job exits non-zero."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HS_BENCH_TEST_CORRUPT_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--config", "c5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ranks_consistent"] is True and line["exchange_check"]["record_counts"] == [40, 41] == line["exchange_check"]["rank_counts"]
    for bad in ("0", "1"):
        r = subprocess.run(cmd, env=dict(env, HS_BENCH_TEST_CORRUPT_RANK=bad), capture_output=True, text=True, timeout=300)
        assert r.returncode != 0, r.stdout + r.stderr
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["ranks_consistent"] is False and line["exchange_check"]["records_identical"] is False
        assert line["exchange_check"]["ranks_that_differ_from_rank0"] == [1]          # two ranks: whichever copy is corrupt, rank 1's differs from rank 0's


def _verify_worker(rank, world, port, q, lie):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from hyslam_amd import distributed as D
    from hyslam_amd._native import KP_DTYPE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(9 + rank)
    n = 20 + 3 * rank
    k = np.zeros(n, KP_DTYPE); k["x"] = rng.random(n)
    g = D.all_gather_records(torch.from_numpy(D.pack_record(k, rng.integers(0, 256, (n, 32), dtype=np.uint8), 32)))
    # `lie`: rank 1 claims another local count than its record carries (a stale buffer: the step alternates two instants for exactly this)
    q.put((rank, D.verify_exchange(g, n + (lie if rank == 1 else 0), rank, world, match_outputs=(torch.arange(4) + rank,))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lie", [0, 1])
def test_verify_exchange_counts(lie):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, 2, port, q, lie)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1], "the verdict itself must be the same on every rank"
    assert res[0]["records_identical"] is True and res[0]["record_counts"] == [20, 23]
    assert res[0]["ranks_consistent"] is (lie == 0) and res[0]["counts_match"] is (lie == 0)
    assert res[0]["match_checksums"][0] != res[0]["match_checksums"][1]


@pytest.mark.parametrize("mode", ["ok", "raises", "hangs"])
def test_the_rccl_probe_cannot_take_the_bench_line_down(mode):
    """bench.py at N > 1 ends with a throw-away RCCL communicator (evidence of how many ranks RCCL itself saw).  ncclCommInitRank has never run with more
    than one rank anywhere, so the probe is guarded: whatever it does — returns, raises, or blocks for ever — rank 0 prints its finished line exactly once
    and the process ends with exit code 0 (rccl_probe_guarded: a watchdog prints the line with an error note and leaves)."""
    code = r'''
import importlib.util, json, sys, time
spec = importlib.util.spec_from_file_location("bench", %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import hyslam_amd.distributed as D
mode = %r
def fake(ex, rank, world, dev):
    if mode == "raises": raise RuntimeError("ncclCommInitRank: unhandled system error")
    if mode == "hangs": time.sleep(3600)
    return {"version": 22203, "ranks_seen_by_rccl": [2, 2], "consistent": True}
D.rccl_probe = fake
b.rccl_probe_guarded({"metric": "m", "value": 1.0}, None, 0, 2, None, timeout_s=1.5)
print("after", flush=True)
''' % (os.path.join(ROOT, "bench.py"), mode)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    import json
    d = json.loads(lines[0])
    assert d["value"] == 1.0
    if mode == "ok":
        assert d["rccl"]["ranks_seen_by_rccl"] == [2, 2] and "after" in r.stdout
    elif mode == "raises":
        assert "unhandled system error" in d["rccl"]["error"] and "after" in r.stdout
    else:
        assert "did not finish" in d["rccl"]["error"] and "after" not in r.stdout
