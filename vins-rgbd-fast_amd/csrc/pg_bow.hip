// Place recognition of pose_graph (include/vio_posegraph.h, vio_pg_voc_*): the part of the vendored DBoW2 that PoseGraph::detectLoop uses
// (pose_graph/src/pose_graph/pose_graph.cpp:308-393; ThirdParty/DBoW/TemplatedVocabulary.h, TemplatedDatabase.h, BowVector.cpp;
// file format ThirdParty/VocabularyBinary.{hpp,cpp}).
//   pg_voc_transform_kernel   one wavefront per descriptor walks the vocabulary tree: the lanes stride over the children of the current node
//                             (256-bit Hamming distance = four v_bcnt), wave arg-min on (distance, child position) = the sequential scan's
//                             "first child with the smallest distance" (TemplatedVocabulary.h:1217-1260).  The tree lives in HBM as flat
//                             arrays (children in the file's order, 32 B descriptor per node: 35 MB for k = 10, L = 6).
// The bag-of-words vector, the inverted file and the L1 query are per-keyframe host code over flat arrays, with every floating-point sum in
// the reference's order (ascending word id, features in input order within a word, inverted-file rows in entry order).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vio_posegraph.h"

extern thread_local std::string g_err;   // vio_abi.hip

struct vio_pg_voc {
    int k = 0, L = 0, scoring = 0, weighting = 0, n_nodes = 0, n_words = 0, nentries = 0;
    // host copies (validation, info); the walk itself runs on the device
    std::vector<int> child_begin, children, node_word;
    std::vector<double> node_weight;
    // device tree
    int *d_child_begin = nullptr, *d_children = nullptr, *d_node_word = nullptr;
    double *d_node_weight = nullptr;
    uint64_t *d_node_desc = nullptr;
    // per-call device scratch (grown on demand)
    uint64_t *d_desc = nullptr;
    int *d_word = nullptr;
    double *d_w = nullptr;
    int cap = 0;
    std::vector<int> h_word;
    std::vector<double> h_w;
    // inverted file: word id -> (entry id, weight), rows in ascending entry order (TemplatedDatabase::add)
    std::vector<std::vector<std::pair<int, double>>> ifile;
    // query scratch
    std::vector<double> acc;
    std::vector<char> seen;
    std::vector<int> touched;
};

namespace {

#define BWCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return VIO_EDEVICE; } } while (0)

__global__ __launch_bounds__(256) void pg_voc_transform_kernel(const uint64_t *desc, int n, const int *child_begin, const int *children,
                                                               const uint64_t *ndesc, const int *node_word, const double *node_weight,
                                                               int *out_word, double *out_w) {
    const int wv = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wv >= n) return;
    const uint64_t f0 = desc[(size_t)wv * 4], f1 = desc[(size_t)wv * 4 + 1], f2 = desc[(size_t)wv * 4 + 2], f3 = desc[(size_t)wv * 4 + 3];
    int node = 0;
    for (;;) {
        const int b = child_begin[node], e = child_begin[node + 1];
        if (b == e) break;   // leaf = word
        unsigned best = 0xFFFFFFFFu;
        for (int c = b + lane; c < e; c += 64) {
            const uint64_t *d = ndesc + (size_t)children[c] * 4;
            const unsigned dist = (unsigned)(__popcll(f0 ^ d[0]) + __popcll(f1 ^ d[1]) + __popcll(f2 ^ d[2]) + __popcll(f3 ^ d[3]));
            best = min(best, (dist << 22) | (unsigned)(c - b));   // distance <= 256 (9 bits), child position < 2^22
        }
        for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o, 64));
        node = children[b + (int)(best & 0x3FFFFFu)];
    }
    if (lane == 0) { out_word[wv] = node_word[node]; out_w[wv] = node_weight[node]; }
}

void voc_free(vio_pg_voc *v) {
    if (!v) return;
    void *ps[] = {v->d_child_begin, v->d_children, v->d_node_word, v->d_node_weight, v->d_node_desc, v->d_desc, v->d_word, v->d_w};
    for (void *p : ps) if (p) (void)hipFree(p);
    delete v;
}

// word id and weight of every descriptor (device walk), into v->h_word / v->h_w
int transform_words(vio_pg_voc *v, const uint64_t *desc, int n) {
    if (n > v->cap) {
        const int cap = std::max(n, 2048);
        if (v->d_desc) (void)hipFree(v->d_desc);
        if (v->d_word) (void)hipFree(v->d_word);
        if (v->d_w) (void)hipFree(v->d_w);
        v->d_desc = nullptr; v->d_word = nullptr; v->d_w = nullptr; v->cap = 0;
        BWCHK(hipMalloc((void **)&v->d_desc, (size_t)cap * 32));
        BWCHK(hipMalloc((void **)&v->d_word, (size_t)cap * sizeof(int)));
        BWCHK(hipMalloc((void **)&v->d_w, (size_t)cap * sizeof(double)));
        v->cap = cap;
    }
    v->h_word.resize((size_t)n); v->h_w.resize((size_t)n);
    if (n == 0) return VIO_OK;
    BWCHK(hipMemcpy(v->d_desc, desc, (size_t)n * 32, hipMemcpyHostToDevice));
    pg_voc_transform_kernel<<<(n + 3) / 4, 256>>>(v->d_desc, n, v->d_child_begin, v->d_children, v->d_node_desc, v->d_node_word, v->d_node_weight,
                                                   v->d_word, v->d_w);
    BWCHK(hipGetLastError());
    BWCHK(hipMemcpy(v->h_word.data(), v->d_word, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    BWCHK(hipMemcpy(v->h_w.data(), v->d_w, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return VIO_OK;
}

// TemplatedVocabulary::transform(features, BowVector) for an L1-scored vocabulary: (word id ascending, L1-normalised value)
void bow_vector(const vio_pg_voc *v, int n, std::vector<std::pair<int, double>> &out) {
    const bool tf = v->weighting == 0 /*TF_IDF*/ || v->weighting == 1 /*TF*/;
    std::vector<int> idx;
    idx.reserve((size_t)n);
    for (int i = 0; i < n; i++) if (v->h_w[i] > 0) idx.push_back(i);   // stopped words carry weight 0
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return v->h_word[a] < v->h_word[b]; });
    out.clear();
    for (size_t q = 0; q < idx.size();) {
        const int w = v->h_word[idx[q]];
        double val = v->h_w[idx[q]];          // addWeight's first insert / addIfNotExist
        size_t r = q + 1;
        for (; r < idx.size() && v->h_word[idx[r]] == w; r++) if (tf) val += v->h_w[idx[r]];
        out.emplace_back(w, val);
        q = r;
    }
    double norm = 0.0;                       // BowVector::normalize(L1)
    for (auto &e : out) norm += std::fabs(e.second);
    if (norm > 0.0) for (auto &e : out) e.second /= norm;
}

int db_add(vio_pg_voc *v, const std::vector<std::pair<int, double>> &bv) {
    const int entry = v->nentries++;
    for (auto &e : bv) v->ifile[(size_t)e.first].emplace_back(entry, e.second);
    return entry;
}

// TemplatedDatabase::queryL1 (TemplatedDatabase.h): ties of the final sort are ordered by entry id (std::sort leaves them unspecified)
void db_query(vio_pg_voc *v, const std::vector<std::pair<int, double>> &bv, int max_results, int max_id, std::vector<std::pair<int, double>> &ret) {
    v->acc.resize((size_t)v->nentries); v->seen.assign((size_t)v->nentries, 0); v->touched.clear();
    for (auto &q : bv) {
        const double qvalue = q.second;
        for (auto &r : v->ifile[(size_t)q.first]) {
            const int entry = r.first;
            if (entry < max_id || max_id == -1 || entry == v->nentries - 1) {
                const double value = std::fabs(qvalue - r.second) - std::fabs(qvalue) - std::fabs(r.second);
                if (v->seen[entry]) v->acc[entry] += value;
                else { v->seen[entry] = 1; v->acc[entry] = value; v->touched.push_back(entry); }
            }
        }
    }
    std::sort(v->touched.begin(), v->touched.end());
    ret.clear();
    for (int e : v->touched) ret.emplace_back(e, v->acc[e]);
    std::stable_sort(ret.begin(), ret.end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.second < b.second; });
    if (max_results > 0 && (int)ret.size() > max_results) ret.resize((size_t)max_results);
    for (auto &r : ret) r.second = -r.second / 2.0;
}

}  // namespace

extern "C" {

vio_pg_voc *vio_pg_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *node_id, const int32_t *parent_id,
                              const double *weight, const uint64_t *desc, int n_words, const int32_t *word_node, const int32_t *word_id) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device: the vocabulary walk has no CPU fallback"; return nullptr; }
    if (n_nodes < 1 || n_words < 1 || !node_id || !parent_id || !weight || !desc || !word_node || !word_id) { g_err = "vio_pg_voc_create: bad argument"; return nullptr; }
    if (scoring != 0) { g_err = "vio_pg_voc_create: only L1_NORM scoring (the scoring of brief_k10L6.bin) is supported"; return nullptr; }
    if (weighting < 0 || weighting > 3) { g_err = "vio_pg_voc_create: unknown weighting type"; return nullptr; }
    vio_pg_voc *v = new vio_pg_voc();
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->n_nodes = n_nodes; v->n_words = n_words;
    const int NN = n_nodes + 1;   // + the root (TemplatedVocabulary.h:1526)
    std::vector<int> cnt((size_t)NN + 1, 0);
    std::vector<char> seen((size_t)NN, 0);
    for (int i = 0; i < n_nodes; i++) {
        if (node_id[i] <= 0 || node_id[i] > n_nodes || parent_id[i] < 0 || parent_id[i] > n_nodes) { g_err = "vio_pg_voc_create: node id out of range"; delete v; return nullptr; }
        // every node once and never its own parent: with one parent per node whatever the walk reaches from the root is a tree (no cycle to spin in)
        if (seen[node_id[i]] || parent_id[i] == node_id[i]) { g_err = "vio_pg_voc_create: node listed twice or its own parent"; delete v; return nullptr; }
        seen[node_id[i]] = 1;
        cnt[(size_t)parent_id[i] + 1]++;
    }
    v->child_begin.assign((size_t)NN + 1, 0);
    for (int i = 0; i < NN; i++) v->child_begin[(size_t)i + 1] = v->child_begin[i] + cnt[(size_t)i + 1];
    v->children.assign((size_t)n_nodes, 0);
    v->node_word.assign((size_t)NN, 0);
    v->node_weight.assign((size_t)NN, 0.0);
    std::vector<uint64_t> nd((size_t)NN * 4, 0);
    std::vector<int> fill(v->child_begin.begin(), v->child_begin.end() - 1);
    for (int i = 0; i < n_nodes; i++) {   // children in the file's order (m_nodes[pid].children.push_back(nid))
        v->children[(size_t)fill[parent_id[i]]++] = node_id[i];
        v->node_weight[node_id[i]] = weight[i];
        std::memcpy(&nd[(size_t)node_id[i] * 4], desc + (size_t)i * 4, 32);
    }
    for (int i = 0; i < n_words; i++) {
        if (word_id[i] < 0 || word_id[i] >= n_words || word_node[i] <= 0 || word_node[i] > n_nodes) { g_err = "vio_pg_voc_create: word id out of range"; delete v; return nullptr; }
        v->node_word[word_node[i]] = word_id[i];
    }
    if (v->child_begin[1] == 0) { g_err = "vio_pg_voc_create: the root has no children"; delete v; return nullptr; }
    v->ifile.assign((size_t)n_words, {});
    bool ok = hipMalloc((void **)&v->d_child_begin, ((size_t)NN + 1) * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&v->d_children, (size_t)n_nodes * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&v->d_node_word, (size_t)NN * sizeof(int)) == hipSuccess &&
              hipMalloc((void **)&v->d_node_weight, (size_t)NN * sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&v->d_node_desc, (size_t)NN * 32) == hipSuccess;
    ok = ok && hipMemcpy(v->d_child_begin, v->child_begin.data(), ((size_t)NN + 1) * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_children, v->children.data(), (size_t)n_nodes * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_node_word, v->node_word.data(), (size_t)NN * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_node_weight, v->node_weight.data(), (size_t)NN * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(v->d_node_desc, nd.data(), (size_t)NN * 32, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { g_err = "vio_pg_voc_create: device allocation / upload failed"; voc_free(v); return nullptr; }
    return v;
}

vio_pg_voc *vio_pg_voc_load(const char *path) {
    FILE *f = path ? std::fopen(path, "rb") : nullptr;
    if (!f) { g_err = std::string("vio_pg_voc_load: cannot open ") + (path ? path : "(null)"); return nullptr; }
    int32_t hdr[6];   // k, L, scoringType, weightingType, nNodes, nWords (VINSLoop::Vocabulary::staticDataSize)
    struct FNode { int32_t nodeId, parentId; double weight; uint64_t d[4]; };
    struct FWord { int32_t nodeId, wordId; };
    static_assert(sizeof(FNode) == 48 && sizeof(FWord) == 8, "VocabularyBinary.hpp layout");
    if (std::fread(hdr, 4, 6, f) != 6 || hdr[4] < 1 || hdr[5] < 1) { std::fclose(f); g_err = "vio_pg_voc_load: bad header"; return nullptr; }
    std::vector<FNode> fn((size_t)hdr[4]);
    std::vector<FWord> fw((size_t)hdr[5]);
    const bool ok = std::fread(fn.data(), sizeof(FNode), fn.size(), f) == fn.size() && std::fread(fw.data(), sizeof(FWord), fw.size(), f) == fw.size();
    std::fclose(f);
    if (!ok) { g_err = "vio_pg_voc_load: truncated file"; return nullptr; }
    std::vector<int32_t> nid(fn.size()), pid(fn.size()), wn(fw.size()), wi(fw.size());
    std::vector<double> w(fn.size());
    std::vector<uint64_t> d(fn.size() * 4);
    for (size_t i = 0; i < fn.size(); i++) { nid[i] = fn[i].nodeId; pid[i] = fn[i].parentId; w[i] = fn[i].weight; std::memcpy(&d[i * 4], fn[i].d, 32); }
    for (size_t i = 0; i < fw.size(); i++) { wn[i] = fw[i].nodeId; wi[i] = fw[i].wordId; }
    return vio_pg_voc_create(hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], nid.data(), pid.data(), w.data(), d.data(), hdr[5], wn.data(), wi.data());
}

void vio_pg_voc_destroy(vio_pg_voc *v) { voc_free(v); }

int vio_pg_voc_info(const vio_pg_voc *v, int32_t *out7) {
    if (!v || !out7) return VIO_EINVAL;
    out7[0] = v->k; out7[1] = v->L; out7[2] = v->scoring; out7[3] = v->weighting; out7[4] = v->n_nodes; out7[5] = v->n_words; out7[6] = v->nentries;
    return VIO_OK;
}

int vio_pg_voc_transform(vio_pg_voc *v, const uint64_t *desc, int n, int32_t *word_id, double *word_weight) {
    if (!v || n < 0 || (n > 0 && (!desc || !word_id || !word_weight))) return VIO_EINVAL;
    int rc = transform_words(v, desc, n);
    if (rc != VIO_OK) return rc;
    for (int i = 0; i < n; i++) { word_id[i] = v->h_word[i]; word_weight[i] = v->h_w[i]; }
    return VIO_OK;
}

int vio_pg_voc_bow(vio_pg_voc *v, const uint64_t *desc, int n, int cap, int32_t *word_id, double *value) {
    if (!v || n < 0 || cap < 0 || (n > 0 && !desc)) return VIO_EINVAL;
    int rc = transform_words(v, desc, n);
    if (rc != VIO_OK) return rc;
    std::vector<std::pair<int, double>> bv;
    bow_vector(v, n, bv);
    for (size_t i = 0; i < bv.size() && (int)i < cap; i++) { word_id[i] = bv[i].first; value[i] = bv[i].second; }
    return (int)bv.size();
}

int vio_pg_db_add(vio_pg_voc *v, const uint64_t *desc, int n) {
    if (!v || n < 0 || (n > 0 && !desc)) return VIO_EINVAL;
    int rc = transform_words(v, desc, n);
    if (rc != VIO_OK) return rc;
    std::vector<std::pair<int, double>> bv;
    bow_vector(v, n, bv);
    return db_add(v, bv);
}

int vio_pg_db_query(vio_pg_voc *v, const uint64_t *desc, int n, int max_results, int max_id, int32_t *ids, double *scores) {
    if (!v || n < 0 || (n > 0 && !desc) || !ids || !scores) return VIO_EINVAL;
    int rc = transform_words(v, desc, n);
    if (rc != VIO_OK) return rc;
    std::vector<std::pair<int, double>> bv, ret;
    bow_vector(v, n, bv);
    db_query(v, bv, max_results, max_id, ret);
    for (size_t i = 0; i < ret.size(); i++) { ids[i] = ret[i].first; scores[i] = ret[i].second; }
    return (int)ret.size();
}

int vio_pg_detect_loop(vio_pg_voc *v, const uint64_t *desc, int n, int frame_index, int32_t *loop_index, int32_t *ids4, double *scores4,
                       int32_t *n_ret) {
    if (!v || n < 0 || (n > 0 && !desc) || !loop_index) return VIO_EINVAL;
    *loop_index = -1;
    int rc = transform_words(v, desc, n);   // one walk serves the query and the add
    if (rc != VIO_OK) return rc;
    std::vector<std::pair<int, double>> bv, ret;
    bow_vector(v, n, bv);
    db_query(v, bv, 4, frame_index - 50, ret);   // first query, then add (pose_graph.cpp:320-328)
    db_add(v, bv);
    if (n_ret) *n_ret = (int)ret.size();
    for (size_t i = 0; i < ret.size(); i++) { if (ids4) ids4[i] = ret[i].first; if (scores4) scores4[i] = ret[i].second; }
    bool find_loop = false;
    if (ret.size() >= 1 && ret[0].second > 0.05)
        for (size_t i = 1; i < ret.size(); i++)
            if (ret[i].second > 0.015) find_loop = true;
    if (find_loop && frame_index > 50) {
        int min_index = -1;
        for (size_t i = 0; i < ret.size(); i++)
            if (min_index == -1 || (ret[i].first < min_index && ret[i].second > 0.015)) min_index = ret[i].first;
        *loop_index = min_index;
    }
    return VIO_OK;
}

}  // extern "C"
