// ORACLE (test infrastructure) -- CPU restatement of the place-recognition half of pose_graph, the part of the vendored
// DBoW2 that PoseGraph::detectLoop uses (BriefVocabulary = TemplatedVocabulary<FBrief::TDescriptor, FBrief>, BriefDatabase):
//   VINSLoop::Vocabulary::deserialize          pose_graph/src/ThirdParty/VocabularyBinary.{hpp,cpp}   (the file format of brief_k10L6.bin)
//   TemplatedVocabulary::loadBin               ThirdParty/DBoW/TemplatedVocabulary.h:1509-1561
//   TemplatedVocabulary::transform (feature)   TemplatedVocabulary.h:1217-1260   (descend the tree: FIRST child with the smallest Hamming distance)
//   TemplatedVocabulary::transform (features)  TemplatedVocabulary.h:1065-1122   (TF / TF_IDF: addWeight per feature; IDF / BINARY: addIfNotExist)
//   BowVector::addWeight / addIfNotExist / normalize   ThirdParty/DBoW/BowVector.cpp
//   TemplatedDatabase::add / query / queryL1   ThirdParty/DBoW/TemplatedDatabase.h:408-475, :560-640 (use_di = false: PoseGraph::loadVocabulary :44-47)
//   PoseGraph::detectLoop                      pose_graph/src/pose_graph/pose_graph.cpp:308-393;  addKeyFrameIntoVoc :395-408
// PINNED IN PART against reference code run in this container: DBoW2::BowVector (addWeight / addIfNotExist / normalize) is the one arithmetic
// translation unit of the reference that compiles here from its own source; `make ref` builds it where it lies into _ref/ behind a C shim
// (ref_bowvector.cpp) and tests/test_oracle_bow_cpu.py requires the bag-of-words vectors of this file to equal the class's bit for bit.  The
// tree walk, the inverted file and queryL1 (templates over OpenCV / boost headers) remain restatements checked by definition tests.
// std::map containers are kept so that every floating-point sum runs in the reference's order.  One pinned choice: queryL1 sorts its results
// with std::sort (not stable); ties in the score are ordered by entry id here.  The vocabulary blob itself (support_files/brief_k10L6.bin) is
// missing from the reference tree: the tests use synthetic vocabularies written in the same file format.  Only tests/ may use this file.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

#include "oracle.h"

namespace ovio {

bool BowVoc::load_bin(const char *path) {   // VINSLoop::Vocabulary::deserialize + loadBin
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    int32_t hdr[6];
    if (std::fread(hdr, 4, 6, f) != 6) { std::fclose(f); return false; }
    const int nn = hdr[4], nw = hdr[5];
    if (nn < 0 || nw < 0) { std::fclose(f); return false; }
    struct FNode { int32_t nodeId, parentId; double weight; uint64_t d[4]; };
    struct FWord { int32_t nodeId, wordId; };
    static_assert(sizeof(FNode) == 48 && sizeof(FWord) == 8, "VocabularyBinary.hpp layout");
    std::vector<FNode> fn((size_t)nn);
    std::vector<FWord> fw((size_t)nw);
    bool ok = std::fread(fn.data(), sizeof(FNode), (size_t)nn, f) == (size_t)nn && std::fread(fw.data(), sizeof(FWord), (size_t)nw, f) == (size_t)nw;
    std::fclose(f);
    if (!ok) return false;
    std::vector<int32_t> nid(nn), pid(nn), wn(nw), wi(nw);
    std::vector<double> w(nn);
    std::vector<uint64_t> d((size_t)nn * 4);
    for (int i = 0; i < nn; i++) { nid[i] = fn[i].nodeId; pid[i] = fn[i].parentId; w[i] = fn[i].weight; std::memcpy(&d[(size_t)i * 4], fn[i].d, 32); }
    for (int i = 0; i < nw; i++) { wn[i] = fw[i].nodeId; wi[i] = fw[i].wordId; }
    return build(hdr[0], hdr[1], hdr[2], hdr[3], nn, nid.data(), pid.data(), w.data(), d.data(), nw, wn.data(), wi.data());
}

bool BowVoc::build(int k_, int L_, int scoring_, int weighting_, int nn, const int32_t *nid, const int32_t *pid, const double *w, const uint64_t *d,
                   int nw, const int32_t *wn, const int32_t *wi) {
    k = k_; L = L_; scoring = scoring_; weighting = weighting_;
    nodes.assign((size_t)nn + 1, Node());   // +1 to include the root (TemplatedVocabulary.h:1526)
    for (int i = 0; i < nn; i++) {
        if (nid[i] <= 0 || nid[i] > nn || pid[i] < 0 || pid[i] > nn) return false;
        Node &n = nodes[nid[i]];
        n.parent = pid[i]; n.weight = w[i];
        std::memcpy(n.desc, d + (size_t)i * 4, 32);
        nodes[pid[i]].children.push_back(nid[i]);
    }
    words.assign((size_t)nw, 0);
    for (int i = 0; i < nw; i++) {
        if (wi[i] < 0 || wi[i] >= nw || wn[i] <= 0 || wn[i] > nn) return false;
        nodes[wn[i]].word_id = wi[i];
        words[wi[i]] = wn[i];
    }
    ifile.assign((size_t)nw, {});
    nentries = 0;
    return !nodes[0].children.empty();
}

static inline int hamming256(const uint64_t *a, const uint64_t *b) {   // FBrief::distance: (a ^ b).count()
    return __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) + __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
}

void BowVoc::transform_one(const uint64_t *f, int &word_id, double &weight) const {   // TemplatedVocabulary.h:1217-1260
    int final_id = 0;
    do {
        const std::vector<int> &ch = nodes[final_id].children;
        final_id = ch[0];
        double best_d = hamming256(f, nodes[final_id].desc);
        for (size_t c = 1; c < ch.size(); c++) {
            double dd = hamming256(f, nodes[ch[c]].desc);
            if (dd < best_d) { best_d = dd; final_id = ch[c]; }
        }
    } while (!nodes[final_id].children.empty());
    word_id = nodes[final_id].word_id;
    weight = nodes[final_id].weight;
}

void BowVoc::transform(const uint64_t *desc, int n, std::map<int, double> &v) const {   // TemplatedVocabulary.h:1065-1122, L1 scoring: must normalise
    v.clear();
    const bool tf = weighting == 0 /*TF_IDF*/ || weighting == 1 /*TF*/;   // enum WeightingType { TF_IDF, TF, IDF, BINARY }
    for (int i = 0; i < n; i++) {
        int id; double w;
        transform_one(desc + (size_t)i * 4, id, w);
        if (!(w > 0)) continue;   // stopped word
        auto it = v.lower_bound(id);
        if (it != v.end() && it->first == id) { if (tf) it->second += w; }   // addWeight / addIfNotExist
        else v.insert(it, std::make_pair(id, w));
    }
    double norm = 0.0;   // BowVector::normalize(L1)
    for (auto &e : v) norm += std::fabs(e.second);
    if (norm > 0.0) for (auto &e : v) e.second /= norm;
}

int BowVoc::add(const uint64_t *desc, int n) {   // TemplatedDatabase::add, use_di = false
    std::map<int, double> v;
    transform(desc, n, v);
    const int entry = nentries++;
    for (auto &e : v) ifile[e.first].push_back(std::make_pair(entry, e.second));
    return entry;
}

int BowVoc::query(const uint64_t *desc, int n, int max_results, int max_id, std::vector<std::pair<int, double>> &ret) const {   // queryL1
    std::map<int, double> vec;
    transform(desc, n, vec);
    std::map<int, double> pairs;
    for (auto &q : vec) {
        const double qvalue = q.second;
        for (auto &r : ifile[q.first]) {
            const int entry = r.first;
            const double dvalue = r.second;
            if (entry < max_id || max_id == -1 || entry == nentries - 1) {
                const double value = std::fabs(qvalue - dvalue) - std::fabs(qvalue) - std::fabs(dvalue);
                auto it = pairs.lower_bound(entry);
                if (it != pairs.end() && it->first == entry) it->second += value;
                else pairs.insert(it, std::make_pair(entry, value));
            }
        }
    }
    ret.assign(pairs.begin(), pairs.end());
    std::stable_sort(ret.begin(), ret.end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.second < b.second; });
    if (max_results > 0 && (int)ret.size() > max_results) ret.resize(max_results);
    for (auto &r : ret) r.second = -r.second / 2.0;
    return (int)ret.size();
}

int BowVoc::detect_loop(const uint64_t *desc, int n, int frame_index) {   // pose_graph.cpp:308-393: first query, then add
    std::vector<std::pair<int, double>> ret;
    query(desc, n, 4, frame_index - 50, ret);
    add(desc, n);
    bool find_loop = false;
    if (ret.size() >= 1 && ret[0].second > 0.05)
        for (size_t i = 1; i < ret.size(); i++)
            if (ret[i].second > 0.015) find_loop = true;
    if (find_loop && frame_index > 50) {
        int min_index = -1;
        for (size_t i = 0; i < ret.size(); i++)
            if (min_index == -1 || (ret[i].first < min_index && ret[i].second > 0.015)) min_index = ret[i].first;
        return min_index;
    }
    return -1;
}

}  // namespace ovio
