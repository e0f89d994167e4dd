// hs_vocab.hip — vocabulary files -> the flat tree of hs_vocab_tree (host code only).
//
// Replaces what HYSLAM::ORBVocabulary::ORBVocabulary(vocab_file) obtains from DBoW2 (src/features/low_level/ORBVocabulary.cpp:14-29):
//   ".txt"  -> TemplatedVocabulary<FORB>::loadFromTextFile   (the text format of ORBvoc.txt)
//   else    -> TemplatedVocabulary<FORB>::loadFromBinaryFile (what tools/bin_vocabulary.cc writes)
// DBoW2 (a modified copy, per the reference's Dependencies.md) and the vocabulary blob itself are NOT in the reference tree; the two
// formats are restated from the published ORB-SLAM2 DBoW2 sources that define them:
//   text    line 1: "k L scoring weighting"; then one line per node in id order (ids 1.., 0 is the root):
//           "parent_id is_leaf b0 b1 ... b31 weight"   (32 descriptor bytes as decimal numbers)
//   binary  uint32 nb_nodes, uint32 size_node (= 4 + 32 + 4 + 1), int32 k, int32 L, int32 scoring, int32 weighting, then records
//           { uint32 parent; uint8 desc[32]; float weight; uint8 is_leaf } for nodes 1 .. nb_nodes-1 until the end of the file:
//           saveToBinaryFile writes nb_nodes = m_nodes.size(), i.e. the root is COUNTED but has no record (its loop starts at node 1), and
//           loadFromBinaryFile sizes m_nodes from the header and reads records until EOF.  The loader therefore trusts the FILE LENGTH and
//           accepts nb_nodes - 1 records (DBoW2's writer) or nb_nodes records (files written by round 2 of this library, which did not count
//           the root).  BINARY LAYOUT UNVERIFIED: restated from the ORB-SLAM2 DBoW2 fork's published source; neither that library nor a sample
//           .bin file is in the reference tree (tools/bin_vocabulary.cc only calls the two functions), so no fixture pins it.
// Word ids are assigned to the leaves in file order, children keep file order (DBoW2 walks them in that order and the first minimum wins).
// The flat tree keeps DBoW2's node numbering whenever the children of every node are contiguous in it (true for vocabularies made by DBoW2's
// hierarchical k-means, ORBvoc included); otherwise nodes are renumbered breadth first and `orig_id` maps back to the DBoW2 NodeId.
#include "hs_internal.h"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

struct hs_vocab {
    int k = 0, L = 0, scoring = 0, weighting = 0, n_words = 0, levels = 0;
    std::vector<int32_t> child_begin, child_count, word_id, orig_id;
    std::vector<uint8_t> desc;
    std::vector<float> weight;
    bool renumbered = false;
    std::string err;
};

namespace {

struct RawNode { int parent; bool leaf; uint8_t d[32]; float w; std::vector<int> children; };

bool has_suffix(const std::string& s, const std::string& suf) { return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0; }

int finish(hs_vocab* v, std::vector<RawNode>& nodes)
{
    const int n = (int)nodes.size();
    if (n < 2) { v->err = "vocabulary has no nodes"; return HS_ERR_INVALID; }
    for (int i = 1; i < n; i++) {
        const int p = nodes[i].parent;
        if (p < 0 || p >= n || p == i) { v->err = "node with an invalid parent id"; return HS_ERR_INVALID; }
        nodes[p].children.push_back(i);
    }
    // word ids: leaves in file order (DBoW2: `if (nIsLeaf > 0) { wid = m_words.size(); ... }`)
    std::vector<int> wid(n, -1);
    int words = 0;
    for (int i = 1; i < n; i++) if (nodes[i].leaf) wid[i] = words++;
    // does DBoW2's own numbering already satisfy the flat-tree contract (children contiguous, after the parent)?
    bool flat = !nodes[0].children.empty();
    for (int i = 0; i < n && flat; i++) {
        const std::vector<int>& c = nodes[i].children;
        if (nodes[i].leaf && !c.empty()) flat = false;
        for (size_t j = 0; j < c.size() && flat; j++) if (c[j] != c[0] + (int)j || c[j] <= i) flat = false;
    }
    std::vector<int> order(n), newid(n, -1);        // order[new] = old
    if (flat) { for (int i = 0; i < n; i++) { order[i] = i; newid[i] = i; } }
    else {                                          // breadth first: every node's children become contiguous
        int head = 0, tail = 0;
        order[tail++] = 0; newid[0] = 0;
        while (head < tail) {
            const int o = order[head++];
            for (int c : nodes[o].children) { if (newid[c] >= 0) { v->err = "vocabulary is not a tree"; return HS_ERR_INVALID; } newid[c] = tail; order[tail++] = c; }
        }
        if (tail != n) { v->err = "vocabulary has nodes that the root does not reach"; return HS_ERR_INVALID; }
    }
    v->renumbered = !flat;
    v->child_begin.assign(n, 0); v->child_count.assign(n, 0); v->word_id.assign(n, -1); v->weight.assign(n, 0.f); v->desc.assign((size_t)n * 32, 0);
    v->orig_id.resize(n);
    for (int i = 0; i < n; i++) {
        const RawNode& r = nodes[order[i]];
        v->orig_id[i] = order[i];
        v->child_count[i] = (int)r.children.size();
        v->child_begin[i] = r.children.empty() ? 0 : newid[r.children[0]];
        v->word_id[i] = wid[order[i]];
        v->weight[i] = r.w;
        memcpy(&v->desc[(size_t)i * 32], r.d, 32);
        if (r.children.empty() && !r.leaf && i != 0) { v->err = "inner node without children"; return HS_ERR_INVALID; }
    }
    if (v->child_count[0] < 1) { v->err = "the root has no children"; return HS_ERR_INVALID; }
    // depth of the tree (levels below the root): DBoW2 keeps m_L from the header; a file may be shallower, the walk stops at leaves anyway
    int depth = 0;
    { std::vector<int> lvl(n, 0); for (int i = 0; i < n; i++) for (int c = 0; c < v->child_count[i]; c++) { lvl[v->child_begin[i] + c] = lvl[i] + 1; depth = std::max(depth, lvl[i] + 1); } }
    v->levels = v->L > 0 ? v->L : depth;
    v->n_words = words;
    return HS_OK;
}

int load_text(hs_vocab* v, const char* path)
{
    std::ifstream f(path);
    if (!f) { v->err = "cannot open vocabulary file"; return HS_ERR_INVALID; }
    std::string line;
    if (!std::getline(f, line)) { v->err = "empty vocabulary file"; return HS_ERR_INVALID; }
    { std::stringstream ss(line); ss >> v->k >> v->L >> v->scoring >> v->weighting; if (ss.fail()) { v->err = "bad header line"; return HS_ERR_INVALID; } }
    if (v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || v->scoring < 0 || v->scoring > 5 || v->weighting < 0 || v->weighting > 3) {
        v->err = "vocabulary file: wrong vocabulary parameters"; return HS_ERR_INVALID;            // the same sanity check DBoW2 applies
    }
    std::vector<RawNode> nodes(1);
    nodes[0].parent = -1; nodes[0].leaf = false; nodes[0].w = 0; memset(nodes[0].d, 0, 32);
    while (std::getline(f, line)) {
        if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
        std::stringstream ss(line);
        RawNode r; int leaf = 0;
        ss >> r.parent >> leaf;
        for (int i = 0; i < 32; i++) { int b = 0; ss >> b; r.d[i] = (uint8_t)b; }
        ss >> r.w;
        if (ss.fail()) { v->err = "bad node line " + std::to_string(nodes.size()); return HS_ERR_INVALID; }
        r.leaf = leaf > 0;
        nodes.push_back(r);
    }
    return finish(v, nodes);
}

int load_binary(hs_vocab* v, const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { v->err = "cannot open vocabulary file"; return HS_ERR_INVALID; }
    uint32_t nb = 0, sz = 0; int32_t hdr[4];
    bool ok = fread(&nb, 4, 1, f) == 1 && fread(&sz, 4, 1, f) == 1 && fread(hdr, 4, 4, f) == 4;
    if (!ok || sz != 41 || nb < 1 || nb > (1u << 26)) { fclose(f); v->err = "bad binary vocabulary header"; return HS_ERR_INVALID; }
    v->k = hdr[0]; v->L = hdr[1]; v->scoring = hdr[2]; v->weighting = hdr[3];
    std::vector<RawNode> nodes(1);
    nodes[0].parent = -1; nodes[0].leaf = false; nodes[0].w = 0; memset(nodes[0].d, 0, 32);
    std::vector<uint8_t> buf;
    {   // DBoW2 reads 41-byte records until EOF
        uint8_t chunk[41 * 256]; size_t got;
        while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
    }
    fclose(f);
    if (buf.size() % 41 != 0) { v->err = "truncated binary vocabulary (the node records do not end on a record boundary)"; return HS_ERR_INVALID; }
    const uint32_t records = (uint32_t)(buf.size() / 41);
    if (records != nb && records + 1 != nb) { v->err = "binary vocabulary: header announces " + std::to_string(nb) + " nodes, file holds " + std::to_string(records) + " records"; return HS_ERR_INVALID; }
    if (records < 1) { v->err = "binary vocabulary without nodes"; return HS_ERR_INVALID; }
    nb = records;
    nodes.reserve(nb + 1);
    for (uint32_t i = 0; i < nb; i++) {
        const uint8_t* p = &buf[(size_t)i * 41];
        RawNode r; int32_t parent; memcpy(&parent, p, 4); r.parent = parent; memcpy(r.d, p + 4, 32); memcpy(&r.w, p + 36, 4); r.leaf = p[40] != 0;
        nodes.push_back(r);
    }
    return finish(v, nodes);
}

} // namespace

// why the last hs_vocab_load / hs_vocab_from_tree / hs_vocab_save of THIS thread failed (there is no handle to hang the text on when a load fails)
static std::string& vocab_error() { static thread_local std::string e; return e; }

extern "C" {

const char* hs_vocab_last_error(void) { return vocab_error().c_str(); }

int hs_vocab_load(const char* path, hs_vocab** out)
{
    if (!path || !out) return HS_ERR_INVALID;
    *out = nullptr;
    hs_vocab* v = new hs_vocab();
    const int rc = has_suffix(path, ".txt") ? load_text(v, path) : load_binary(v, path);      // ORBVocabulary.cpp:17-21
    if (rc != HS_OK) { vocab_error() = std::string("hs_vocab_load(") + path + "): " + v->err; delete v; return rc; }      // (no diagnostics on stderr from inside a library)
    vocab_error().clear();
    *out = v;
    return HS_OK;
}

int hs_vocab_from_tree(const hs_vocab_tree* T, int k, hs_vocab** out)
{
    if (!T || !out || T->n_nodes < 2 || !T->child_begin || !T->child_count || !T->desc || !T->word_id || !T->weight) return HS_ERR_INVALID;
    *out = nullptr;
    const int n = T->n_nodes;
    std::vector<RawNode> nodes(n);
    for (int i = 0; i < n; i++) { nodes[i].parent = -1; nodes[i].leaf = T->child_count[i] == 0; nodes[i].w = T->weight[i]; memcpy(nodes[i].d, T->desc + (size_t)i * 32, 32); }
    for (int i = 0; i < n; i++)
        for (int c = 0; c < T->child_count[i]; c++) {
            const long ch = (long)T->child_begin[i] + c;
            if (ch <= i || ch >= n || nodes[ch].parent >= 0) return HS_ERR_INVALID;
            nodes[ch].parent = i;
        }
    for (int i = 1; i < n; i++) if (nodes[i].parent < 0) return HS_ERR_INVALID;
    hs_vocab* v = new hs_vocab();
    v->k = k; v->L = T->levels;
    const int rc = finish(v, nodes);
    if (rc != HS_OK) { delete v; return rc; }
    // a caller-built tree may carry its own word ids: keep them
    for (int i = 0; i < n; i++) if (T->child_count[i] == 0) v->word_id[i] = T->word_id[i];
    *out = v;
    return HS_OK;
}

void hs_vocab_destroy(hs_vocab* v) { delete v; }

int hs_vocab_get_tree(const hs_vocab* v, hs_vocab_tree* out)
{
    if (!v || !out) return HS_ERR_INVALID;
    out->n_nodes = (int32_t)v->child_begin.size(); out->levels = v->levels;
    out->child_begin = v->child_begin.data(); out->child_count = v->child_count.data(); out->desc = v->desc.data();
    out->word_id = v->word_id.data(); out->weight = v->weight.data();
    out->orig_id = v->renumbered ? v->orig_id.data() : nullptr;
    return HS_OK;
}

int hs_vocab_info(const hs_vocab* v, int32_t* k, int32_t* L, int32_t* n_nodes, int32_t* n_words, int32_t* scoring, int32_t* weighting)
{
    if (!v) return HS_ERR_INVALID;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (n_nodes) *n_nodes = (int32_t)v->child_begin.size();
    if (n_words) *n_words = v->n_words;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    return HS_OK;
}

// the converter of tools/bin_vocabulary.cc (load_as_text + save_as_binary) and its inverse: writes the vocabulary in DBoW2's node numbering
int hs_vocab_save(const hs_vocab* v, const char* path)
{
    if (!v || !path) return HS_ERR_INVALID;
    const int n = (int)v->child_begin.size();
    // parent of every node, in DBoW2 numbering
    std::vector<int> parent(n, 0), by_orig(n, 0);
    for (int i = 0; i < n; i++) by_orig[v->orig_id[i]] = i;
    for (int i = 0; i < n; i++) for (int c = 0; c < v->child_count[i]; c++) parent[v->child_begin[i] + c] = i;
    if (has_suffix(path, ".txt")) {
        FILE* f = fopen(path, "w");
        if (!f) return HS_ERR_INVALID;
        fprintf(f, "%d %d %d %d\n", v->k, v->L, v->scoring, v->weighting);
        for (int o = 1; o < n; o++) {
            const int i = by_orig[o];
            fprintf(f, "%d %d ", v->orig_id[parent[i]], v->child_count[i] == 0 ? 1 : 0);
            for (int b = 0; b < 32; b++) fprintf(f, "%d ", (int)v->desc[(size_t)i * 32 + b]);
            fprintf(f, "%.9g\n", (double)v->weight[i]);
        }
        fclose(f);
        return HS_OK;
    }
    FILE* f = fopen(path, "wb");
    if (!f) return HS_ERR_INVALID;
    const uint32_t nb = (uint32_t)n /* m_nodes.size(): the root is counted, its record is not written */, sz = 41; const int32_t hdr[4] = { v->k, v->L, v->scoring, v->weighting };
    fwrite(&nb, 4, 1, f); fwrite(&sz, 4, 1, f); fwrite(hdr, 4, 4, f);
    for (int o = 1; o < n; o++) {
        const int i = by_orig[o];
        const int32_t p = v->orig_id[parent[i]]; const uint8_t leaf = v->child_count[i] == 0;
        fwrite(&p, 4, 1, f); fwrite(&v->desc[(size_t)i * 32], 1, 32, f); fwrite(&v->weight[i], 4, 1, f); fwrite(&leaf, 1, 1, f);
    }
    fclose(f);
    return HS_OK;
}

} // extern "C"
