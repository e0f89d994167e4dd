"""DBoW2 vocabulary files (text and binary, as ORBVocabulary::ORBVocabulary loads them — src/features/low_level/ORBVocabulary.cpp:14-29,
tools/bin_vocabulary.cc) -> the flat tree of the C ABI.  ORBvoc.txt itself is not available (a missing blob of the reference), so the files
are synthetic vocabularies written here, by a writer that follows the published format independently of the loader.  Host code only: no GPU."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import oracle
from hyslam_amd import _native as N


def write_text(path, k, L, parents, leaf, desc, weight):
    """DBoW2 saveToTextFile: header 'k L scoring weighting', then per node (ids 1..): parent is_leaf 32 bytes weight"""
    with open(path, "w") as f:
        f.write("%d %d 0 0\n" % (k, L))
        for i in range(1, len(parents)):
            f.write("%d %d %s %.9g\n" % (parents[i], int(leaf[i]), " ".join(str(int(b)) for b in desc[i]), float(weight[i])))


def write_binary(path, k, L, parents, leaf, desc, weight, count_root=True):
    """DBoW2 saveToBinaryFile (the ORB-SLAM2 fork tools/bin_vocabulary.cc links): nb_nodes = m_nodes.size() COUNTS the root, the record loop
    starts at node 1, so the file holds nb_nodes - 1 records and the reader runs to EOF.  count_root=False writes the header round 2 of this
    library wrote (records only), which the loader still accepts."""
    with open(path, "wb") as f:
        f.write(struct.pack("<IIiiii", len(parents) if count_root else len(parents) - 1, 41, k, L, 0, 0))
        for i in range(1, len(parents)):
            f.write(struct.pack("<i", parents[i]) + bytes(bytearray(desc[i].tolist())) + struct.pack("<f", float(weight[i])) + bytes([int(leaf[i])]))


def dbow2_transform(parents, leaf, desc, weight, feats, L, levelsup):
    """the published algorithm on explicit children lists in DBoW2's own numbering (first minimum wins)"""
    n = len(parents)
    children = [[] for _ in range(n)]
    for i in range(1, n):
        children[parents[i]].append(i)
    words = {}
    for i in range(1, n):
        if leaf[i]:
            words[i] = len(words)
    out = []
    for f in feats:
        cur, lvl, nid = 0, 0, 0
        while True:
            lvl += 1
            best, bi = None, -1
            for c in children[cur]:
                d = int(np.unpackbits(np.bitwise_xor(desc[c], f)).sum())
                if best is None or d < best:
                    best, bi = d, c
            cur = bi
            if lvl == L - levelsup:
                nid = cur
            if leaf[cur]:
                break
        out.append((words[cur], np.float32(weight[cur]), nid))
    return out


def synthetic(k, L, seed, scramble=False):
    """a k-ary tree of depth L in DBoW2's creation order (children of a node contiguous), or with node ids scrambled"""
    rng = np.random.default_rng(seed)
    parents, leaf = [-1], [False]

    def grow(node, depth):
        kids = []
        for _ in range(k):
            parents.append(node); leaf.append(depth == L); kids.append(len(parents) - 1)
        if depth < L:
            for c in kids:
                grow(c, depth + 1)
    grow(0, 1)
    n = len(parents)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    weight = np.where(leaf, rng.uniform(0.3, 8.0, n), 0.0).astype(np.float32)
    parents, leaf = np.array(parents), np.array(leaf)
    if scramble:                                   # a permutation of ids 1..n-1 that keeps parents before children but interleaves siblings
        order = [0]
        frontier = [0]
        kids = {i: [j for j in range(1, n) if parents[j] == i] for i in range(n)}
        while frontier:                            # level order, siblings of different parents interleaved
            nxt = []
            lists = [list(kids[p]) for p in frontier]
            while any(lists):
                for l in lists:
                    if l:
                        c = l.pop(0); order.append(c); nxt.append(c)
            frontier = nxt
        new = np.zeros(n, np.int64); new[order] = np.arange(n)
        p2 = np.full(n, -1); l2 = np.zeros(n, bool); d2 = np.zeros_like(desc); w2 = np.zeros_like(weight)
        for old in range(n):
            p2[new[old]] = new[parents[old]] if old else -1; l2[new[old]] = leaf[old]; d2[new[old]] = desc[old]; w2[new[old]] = weight[old]
        parents, leaf, desc, weight = p2, l2, d2, w2
    return parents, leaf, desc, weight


def load(path):
    v = C.c_void_p()
    st = N.lib().hs_vocab_load(str(path).encode(), C.byref(v))
    if st != N.HS_OK:
        return st, None, None
    T = N.VocabTree()
    assert N.lib().hs_vocab_get_tree(v, C.byref(T)) == N.HS_OK
    return st, v, T


def as_oracle_tree(T):
    To = oracle.VocabTree(T.n_nodes, T.levels, T.child_begin, T.child_count, T.desc, T.word_id, T.weight, T.orig_id)
    return To


@pytest.mark.parametrize("fmt", ["txt", "bin"])
@pytest.mark.parametrize("scramble", [False, True])
def test_load_and_transform(tmp_path, fmt, scramble):
    k, L = 5, 3
    parents, leaf, desc, weight = synthetic(k, L, 5, scramble)
    path = tmp_path / ("voc." + fmt)
    (write_text if fmt == "txt" else write_binary)(path, k, L, parents, leaf, desc, weight)
    st, v, T = load(path)
    assert st == N.HS_OK
    info = [C.c_int32() for _ in range(6)]
    N.lib().hs_vocab_info(v, *(C.byref(x) for x in info))
    assert [x.value for x in info] == [k, L, len(parents), int(leaf.sum()), 0, 0]
    assert bool(T.orig_id) == scramble                     # DBoW2's own numbering is kept when it is already a flat tree
    cc = np.ctypeslib.as_array(C.cast(T.child_count, C.POINTER(C.c_int32)), shape=(T.n_nodes,))
    cb = np.ctypeslib.as_array(C.cast(T.child_begin, C.POINTER(C.c_int32)), shape=(T.n_nodes,))
    assert cc[0] == k and ((cc == 0) | (cc == k)).all() and (cb[cc > 0] > np.nonzero(cc > 0)[0]).all()
    rng = np.random.default_rng(8)
    feats = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    feats[:40] = desc[rng.integers(1, len(parents), 40)]
    for levelsup in (1, 2, 3, 4):
        w, wt, nd = oracle.bow_transform(as_oracle_tree(T), feats, levelsup)
        ref = dbow2_transform(parents, leaf, desc, weight, feats, L, levelsup)
        assert w.tolist() == [r[0] for r in ref] and nd.tolist() == [r[2] for r in ref]
        assert np.array_equal(wt, np.array([r[1] for r in ref], np.float32))
    # text <-> binary conversion (tools/bin_vocabulary.cc) round trip keeps every node
    other = tmp_path / ("conv." + ("bin" if fmt == "txt" else "txt"))
    assert N.lib().hs_vocab_save(v, str(other).encode()) == N.HS_OK
    st2, v2, T2 = load(other)
    assert st2 == N.HS_OK and T2.n_nodes == T.n_nodes
    w2, wt2, nd2 = oracle.bow_transform(as_oracle_tree(T2), feats, 2)
    w1, wt1, nd1 = oracle.bow_transform(as_oracle_tree(T), feats, 2)
    assert np.array_equal(w1, w2) and np.array_equal(wt1, wt2) and np.array_equal(nd1, nd2)
    if fmt == "txt":       # the binary file hs_vocab_save wrote counts the root in its header (DBoW2: nb_nodes = m_nodes.size()) and holds one record less
        raw = other.read_bytes()
        assert struct.unpack("<II", raw[:8]) == (len(parents), 41) and len(raw) == 24 + 41 * (len(parents) - 1)
    N.lib().hs_vocab_destroy(v); N.lib().hs_vocab_destroy(v2)


def test_binary_header_both_node_count_conventions(tmp_path):
    """DBoW2's writer announces m_nodes.size() (root included) and writes size - 1 records; files of this library's round 2 announced the
    record count.  The loader goes by the file length, as DBoW2's reader does, and accepts both."""
    k, L = 4, 3
    parents, leaf, desc, weight = synthetic(k, L, 9, False)
    trees = []
    for count_root in (True, False):
        p = tmp_path / ("v%d.bin" % count_root)
        write_binary(p, k, L, parents, leaf, desc, weight, count_root)
        st, v, T = load(p)
        assert st == N.HS_OK and T.n_nodes == len(parents)
        trees.append((v, T))
    feats = np.random.default_rng(2).integers(0, 256, (100, 32), dtype=np.uint8)
    a = oracle.bow_transform(as_oracle_tree(trees[0][1]), feats, 2); b = oracle.bow_transform(as_oracle_tree(trees[1][1]), feats, 2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    for v, _ in trees:
        N.lib().hs_vocab_destroy(v)


def test_from_tree_matches_make_vocab_tree():
    To, keep, n_words = oracle.make_vocab_tree(oracle.VocabTree, 10, 3, 17)
    Tn, keep2, _ = oracle.make_vocab_tree(N.VocabTree, 10, 3, 17)
    v = C.c_void_p()
    assert N.lib().hs_vocab_from_tree(C.byref(Tn), 10, C.byref(v)) == N.HS_OK
    T = N.VocabTree(); N.lib().hs_vocab_get_tree(v, C.byref(T))
    feats = np.random.default_rng(1).integers(0, 256, (200, 32), dtype=np.uint8)
    a = oracle.bow_transform(To, feats, 2); b = oracle.bow_transform(as_oracle_tree(T), feats, 2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    N.lib().hs_vocab_destroy(v)


def test_bad_files(tmp_path):
    assert load(tmp_path / "missing.txt")[0] == N.HS_ERR_INVALID           # the reference prints "Wrong path to vocabulary" and exits
    p = tmp_path / "bad.txt"
    p.write_text("40 6 0 0\n")                                              # k out of DBoW2's accepted range
    assert load(p)[0] == N.HS_ERR_INVALID
    p = tmp_path / "trunc.bin"
    p.write_bytes(struct.pack("<IIiiii", 100, 41, 10, 6, 0, 0) + b"\0" * 50)
    assert load(p)[0] == N.HS_ERR_INVALID
    p = tmp_path / "count.bin"                                             # 100 nodes announced, 3 whole records present: neither nb nor nb - 1
    p.write_bytes(struct.pack("<IIiiii", 100, 41, 10, 6, 0, 0) + b"\0" * (3 * 41))
    assert load(p)[0] == N.HS_ERR_INVALID
    p = tmp_path / "cycle.txt"
    p.write_text("2 2 0 0\n" + "2 0 " + "0 " * 32 + "0\n" + "1 1 " + "0 " * 32 + "1.0\n")       # node 1's parent is node 2 and vice versa: not a tree
    assert load(p)[0] == N.HS_ERR_INVALID
