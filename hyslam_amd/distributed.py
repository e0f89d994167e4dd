"""Frame-per-GPU sharding and the one collective of the path (BASELINE config 5).

Frames are independent, so extraction and stereo matching shard across ranks with no data-path collective.  Only
cross-camera matching needs an exchange: an all-gather of one fixed-size record per rank,

    [ int32 count | 12 B pad | hs_keypoint[cap] (24 B each) | pad to 16 B | uint8 desc[cap][32] ]      (~113 KB for cap = 2012)

The extractor writes its outputs straight into that layout (the C ABI takes separate pointers, so `d_n`, `d_kps`, `d_desc`
simply point into one buffer): packing costs nothing and the exchange is ONE collective per step.  On the 8-GPU xGMI mesh
the payload is ~1 us of wire time per link, i.e. the step is latency-bound; RCCL's all-gather (backend "nccl" on ROCm)
moves it in one hop over the 7 direct links.  With backend "gloo" the same code runs on CPU tensors (world-size-2 tests).
"""
import numpy as np

from ._native import KP_DTYPE

HEADER = 16
KP_BYTES = KP_DTYPE.itemsize      # 24
DESC_BYTES = 32


def _off_desc(cap):
    return (HEADER + cap * KP_BYTES + 15) & ~15       # the descriptors start on a 16-byte boundary whatever the parity of cap (they are written with 16-byte vector stores)


def record_bytes(cap):
    """== hs_record_bytes(cap) of the C ABI (tests/test_abi.py checks the two agree)"""
    return _off_desc(cap) + cap * DESC_BYTES


def record_offsets(cap):
    """byte offsets of (count, keypoints, descriptors) inside one record"""
    return 0, HEADER, _off_desc(cap)


def pack_record(kps, desc, cap):
    """numpy -> one record (uint8[record_bytes(cap)]); padding beyond `count` is zero."""
    n = len(kps)
    if n > cap:
        raise ValueError("more keypoints than the record holds")
    rec = np.zeros(record_bytes(cap), np.uint8)
    rec[:4] = np.frombuffer(np.int32(n).tobytes(), np.uint8)
    o_n, o_k, o_d = record_offsets(cap)
    rec[o_k:o_k + n * KP_BYTES] = np.ascontiguousarray(kps, KP_DTYPE).view(np.uint8)
    rec[o_d:o_d + n * DESC_BYTES] = np.ascontiguousarray(desc, np.uint8).reshape(-1)
    return rec


def unpack_record(rec, cap):
    """one record (numpy uint8) -> (keypoints[count], descriptors[count,32]); counts beyond cap are rejected."""
    rec = np.ascontiguousarray(rec, np.uint8)
    n = int(rec[:4].view(np.int32)[0])
    if n < 0 or n > cap:
        raise ValueError("corrupt frame record: count %d, cap %d" % (n, cap))
    o_n, o_k, o_d = record_offsets(cap)
    kps = rec[o_k:o_k + n * KP_BYTES].view(KP_DTYPE).copy()
    desc = rec[o_d:o_d + n * DESC_BYTES].reshape(n, DESC_BYTES).copy()
    return kps, desc


def shard_range(n_items, rank, world):
    """contiguous block of the `n_items` frames owned by `rank` (sizes differ by at most one)"""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_records(record, group=None):
    """record: torch uint8 tensor [record_bytes] on this rank's device -> [world, record_bytes] on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world, record.numel()), dtype=torch.uint8, device=record.device)
    dist.all_gather_into_tensor(out.view(-1), record.contiguous().view(-1), group=group)
    return out


def checksum64(data):
    """64-bit checksum (blake2b-8, as a non-negative Python int < 2^63) of a numpy array / bytes: what the ranks exchange to compare what they hold"""
    import hashlib
    b = data if isinstance(data, (bytes, bytearray, memoryview)) else np.ascontiguousarray(data).tobytes()
    return int.from_bytes(hashlib.blake2b(b, digest_size=8).digest(), "little") >> 1


def verify_exchange(gathered, local_count, rank, world, match_outputs=None, group=None):
    """After a cross-camera step: do all ranks hold the SAME gathered records, and is record r really rank r's frame?

    gathered       torch uint8 [world, record_bytes] on this rank (any device)
    local_count    the number of keypoints this rank extracted in the step the records belong to
    match_outputs  optional tensors / arrays (this rank's matcher outputs): their checksum is gathered and reported per rank

    Every rank computes a 64-bit checksum of each gathered record, the checksums are all-gathered (torch.distributed, backend of the caller's
    process group: RCCL on GPUs, gloo in the CPU tests) and compared: `records_identical` = every rank saw the same bytes for every record,
    `counts_match` = record r's header count equals the count rank r reports for itself.  Returns a dict (the same on every rank);
    `ranks_consistent` is the conjunction.  SURVEY.md C5: "identical match lists on every rank" needs identical inputs on every rank first."""
    import torch
    import torch.distributed as dist
    g = gathered.detach().cpu().numpy()
    sums = [checksum64(g[r]) for r in range(world)]
    counts = [int(g[r, :4].view(np.int32)[0]) for r in range(world)]
    msum = 0
    if match_outputs is not None:
        import hashlib
        h = hashlib.blake2b(digest_size=8)
        for t in match_outputs:
            h.update(np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t).tobytes())
        msum = int.from_bytes(h.digest(), "little") >> 1
    mine = torch.tensor(sums + [int(local_count), msum], dtype=torch.int64)
    if world > 1:
        dev = gathered.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        allv = torch.empty((world, world + 2), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allv.view(-1), mine.to(dev), group=group)
        allv = allv.cpu()
    else:
        allv = mine.view(1, -1)
    rec = allv[:, :world]
    records_identical = bool((rec == rec[0:1]).all().item())
    own_counts = allv[:, world].tolist()
    counts_match = records_identical and counts == own_counts
    bad = [r for r in range(world) if not bool((rec[r] == rec[0]).all().item())]
    return {"ranks_consistent": bool(records_identical and counts_match), "records_identical": records_identical, "counts_match": bool(counts_match),
            "record_counts": counts, "rank_counts": own_counts, "ranks_that_differ_from_rank0": bad,
            "record_checksums": ["%016x" % v for v in rec[0].tolist()], "match_checksums": ["%016x" % v for v in allv[:, world + 1].tolist()]}


def comm_available():
    """(ok, reason): can this process create an hs_comm communicator (librccl loads)?  Non-collective — ask on every rank before RecordExchange."""
    from . import _native as N
    L = N.lib()
    ok = L.hs_comm_available() == N.HS_OK
    return ok, ("" if ok else L.hs_comm_unavailable_reason().decode())


class RecordExchange:
    """The exchange through the C ABI (hs_comm_*: RCCL's all-gather without torch).  One rank calls unique_id() and hands the 128 bytes to
    the others (any channel: a file, a socket, torch.distributed.broadcast); every rank then builds RecordExchange(extractor, id, world, rank),
    which blocks until all ranks arrived.  allgather() enqueues on `stream` (0 = the extractor handle's own stream), where the extraction
    before it and the matcher after it run too: no events, no host synchronisation inside a step."""

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _native as N
        ident = (C.c_uint8 * 128)()
        st = N.lib().hs_comm_get_unique_id(ident)
        if st != N.HS_OK:
            raise N.HsError(st, "hs_comm_get_unique_id: " + N.lib().hs_status_string(st).decode())
        return bytes(ident)

    def __init__(self, extractor, ident, world, rank):
        import ctypes as C
        from . import _native as N
        self._ex, self.world, self.rank = extractor, world, rank
        self._c = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(ident)
        N.check(extractor._h, extractor._lib.hs_comm_create(extractor._h, buf, world, rank, C.byref(self._c)))

    def allgather(self, d_record, d_gathered, record_bytes, stream=0):
        """device addresses (integers); in place when d_record == d_gathered + rank * record_bytes"""
        import ctypes as C
        from . import _native as N
        lib = self._ex._lib
        st = lib.hs_comm_allgather_records(self._c, C.c_void_p(d_record), C.c_void_p(d_gathered), record_bytes, C.c_void_p(stream) if stream else None)
        if st != N.HS_OK:
            raise N.HsError(st, "hs_comm_allgather_records: " + lib.hs_comm_last_error(self._c).decode())

    def rccl_info(self):
        """what RCCL itself reports for this communicator: {"version": ncclGetVersion, "ranks": ncclCommCount, "rank": ncclCommUserRank} (-1 = unknown)"""
        lib = self._ex._lib
        return {"version": int(lib.hs_comm_rccl_version()), "ranks": int(lib.hs_comm_rccl_ranks(self._c)), "rank": int(lib.hs_comm_rccl_rank(self._c))}

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self._ex._lib.hs_comm_destroy(self._c)
            self._c.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_probe(extractor, rank, world, dev):
    """A throw-away communicator over all ranks of the default torch.distributed group (the id travels by broadcast): returns RecordExchange.rccl_info()
    gathered from every rank as {"version", "ranks_seen_by_rccl": [per rank], "consistent"} — or {"error": ...}.  Collective; call it OUTSIDE timed regions.
    Every rank first answers whether hs_comm is available (non-collective) and the answers are MIN-reduced, so no rank blocks in ncclCommInitRank alone."""
    import torch
    import torch.distributed as dist
    ok, why = comm_available()
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        return {"error": "hs_comm unavailable on at least one rank (%s)" % (why or "another rank")}
    ident = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        ident.copy_(torch.frombuffer(bytearray(RecordExchange.unique_id()), dtype=torch.uint8))
    dist.broadcast(ident, 0)
    xc = RecordExchange(extractor, bytes(ident.cpu().numpy().tobytes()), world, rank)
    info = xc.rccl_info()
    # one tiny all-gather through the communicator, so that the probe has also MOVED bytes between the ranks
    send = torch.full((16,), rank + 1, dtype=torch.uint8, device=dev)
    recv = torch.zeros((world, 16), dtype=torch.uint8, device=dev)
    xc.allgather(send.data_ptr(), recv.data_ptr(), 16, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    moved = bool((recv[:, 0].cpu() == torch.arange(1, world + 1, dtype=torch.uint8)).all().item())
    xc.close()
    mine = torch.tensor([info["ranks"], info["rank"], 1 if moved else 0], dtype=torch.int64, device=dev)
    allv = torch.empty((world, 3), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allv.view(-1), mine)
    allv = allv.cpu()
    seen = allv[:, 0].tolist()
    return {"version": info["version"], "ranks_seen_by_rccl": seen, "rccl_rank_of_each_rank": allv[:, 1].tolist(), "allgather_moved_bytes": bool(allv[:, 2].min().item() == 1),
            "consistent": all(v == world for v in seen) and allv[:, 1].tolist() == list(range(world)) and bool(allv[:, 2].min().item() == 1)}


def records_knn2_device(extractor, d_records, record_stride, world, rank, cap, d_best_idx, d_best_dist, d_second_dist, stream=0):
    """hs_records_knn2_device on raw device addresses (integers): 2-NN of record `rank`'s descriptors against every other record's;
    outputs [world][cap] int32.  The counts are read from the record headers on the device — nothing synchronises with the host."""
    import ctypes as C
    from . import _native as N
    N.check(extractor._h, extractor._lib.hs_records_knn2_device(extractor._h, C.c_void_p(d_records), record_stride, world, rank, cap,
                                                                C.c_void_p(d_best_idx), C.c_void_p(d_best_dist), C.c_void_p(d_second_dist),
                                                                C.c_void_p(stream) if stream else None))


def cross_camera_knn2(extractor, gathered, rank, cap, stream=0, out=None):
    """Brute-force Hamming 2-NN of this rank's descriptors against every other rank's, ONE launch, no host round trip.
    gathered: torch uint8 [world, record_bytes(cap)] on the GPU.  Returns (best_idx, best_dist, second_dist) int32 tensors [world, cap]
    (row `rank` and entries beyond this rank's count are meaningless) and the device view of the per-record counts."""
    import torch
    world = gathered.shape[0]
    if out is None:
        out = tuple(torch.empty((world, cap), dtype=torch.int32, device=gathered.device) for _ in range(3))
    records_knn2_device(extractor, gathered.data_ptr(), gathered.shape[1], world, rank, cap,
                        out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), stream)
    counts = gathered[:, :4].view(torch.int32).view(-1) if gathered.shape[1] % 4 == 0 else gathered[:, :4].contiguous().view(torch.int32).view(-1)
    return out, counts


class DeviceVocabulary:
    """A vocabulary tree resident in HBM (hs_vocab_upload): Frame::ComputeBoW and the BoW matcher run on device-resident descriptors."""

    def __init__(self, extractor, tree, levelsup=4, keepalive=None):
        import ctypes as C
        from . import _native as N
        self._ex, self._keep, self.levelsup = extractor, keepalive, levelsup
        self._v = C.c_void_p()
        N.check(extractor._h, extractor._lib.hs_vocab_upload(extractor._h, C.byref(tree), levelsup, C.byref(self._v)))
        self.groups = extractor._lib.hs_vocab_dev_groups(self._v)

    def transform_device(self, d_desc, d_n, n_max, d_word, d_weight, d_node, stream=0):
        import ctypes as C
        from . import _native as N
        ex = self._ex
        N.check(ex._h, ex._lib.hs_bow_transform_device(ex._h, self._v, C.c_void_p(d_desc), C.c_void_p(d_n) if d_n else None, n_max,
                                                       C.c_void_p(d_word), C.c_void_p(d_weight), C.c_void_p(d_node), C.c_void_p(stream) if stream else None))

    def records_bow_match_device(self, d_records, record_stride, world, rank, cap, thr, ratio, check_rotation, d_match12, d_n_matches, stream=0):
        import ctypes as C
        from . import _native as N
        ex = self._ex
        N.check(ex._h, ex._lib.hs_records_bow_match_device(ex._h, self._v, C.c_void_p(d_records), record_stride, world, rank, cap, thr, ratio,
                                                           int(check_rotation), C.c_void_p(d_match12), C.c_void_p(d_n_matches),
                                                           C.c_void_p(stream) if stream else None))

    def close(self):
        if getattr(self, "_v", None) is not None and self._v.value:
            self._ex._lib.hs_vocab_dev_destroy(self._v)
            self._v.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BowCrossCamera:
    """bench.py --config c5 --c5-match bow: vocabulary transform + BoW-grouped matching of this rank's frame against every peer's, on the
    gathered records, without leaving the device.  The vocabulary is a seeded synthetic 10-ary tree (ORBvoc is not available)."""

    def __init__(self, extractor, world, cap, seed=17, levels=4, levelsup=2, thr=50.0, ratio=0.6):
        import torch
        from .synth import synth_vocab_tree
        tree, keep, self.n_words = synth_vocab_tree(10, levels, seed)
        self.voc = DeviceVocabulary(extractor, tree, levelsup, keep)
        dev = torch.device("cuda", extractor.device)
        self.match12 = torch.full((world, cap), -1, dtype=torch.int32, device=dev)
        self.n_matches = torch.zeros(world, dtype=torch.int32, device=dev)
        self.world, self.cap, self.thr, self.ratio = world, cap, thr, ratio

    def match(self, gathered, rank, stream=0):
        self.voc.records_bow_match_device(gathered.data_ptr(), gathered.shape[1], self.world, rank, self.cap, self.thr, self.ratio, True,
                                          self.match12.data_ptr(), self.n_matches.data_ptr(), stream)

    def total_matches(self, rank):
        return int(self.n_matches.sum().item())
