"""Synthetic DBoW2 vocabularies and keyframe descriptor sets for the place-recognition tests (the real brief_k10L6.bin is missing from the
reference tree), plus the ctypes face of the oracle's restatement (oracle/bow.cpp)."""
import ctypes as C

import numpy as np

import vio_ct


def _flip(rng, d, nbits):
    """d: uint64[4]; returns a copy with nbits random bits toggled"""
    out = d.copy()
    for b in rng.choice(256, size=nbits, replace=False):
        out[b >> 6] ^= np.uint64(1) << np.uint64(b & 63)
    return out


def make_vocabulary(k, L, seed, irregular=False, weighting=0, stop_fraction=0.02):
    """Arrays in the file order of VINSLoop::Vocabulary (breadth first, children of a node consecutive): a random hierarchical tree whose
    children are their parent's descriptor with 96 >> level bits toggled.  irregular: 1..k children per node and leaves above level L."""
    rng = np.random.default_rng(seed)
    node_id, parent_id, desc, level = [], [], [], []
    root = rng.integers(0, 2 ** 63, 4, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 4, dtype=np.uint64)
    frontier = [(0, root, 0)]
    nxt = 1
    is_leaf = {}
    while frontier:
        new = []
        for (pid, pd, lv) in frontier:
            nk = int(rng.integers(1, k + 1)) if irregular else k
            for _ in range(nk):
                d = _flip(rng, pd, max(96 >> lv, 6))
                node_id.append(nxt); parent_id.append(pid); desc.append(d); level.append(lv + 1)
                leaf = (lv + 1 == L) or (irregular and lv + 1 >= 2 and rng.random() < 0.15)
                is_leaf[nxt] = leaf
                if not leaf:
                    new.append((nxt, d, lv + 1))
                nxt += 1
        frontier = new
    node_id, parent_id = np.asarray(node_id, np.int32), np.asarray(parent_id, np.int32)
    desc = np.asarray(desc, np.uint64).reshape(-1, 4)
    word_node = np.asarray([n for n in node_id if is_leaf[int(n)]], np.int32)
    word_id = np.arange(len(word_node), dtype=np.int32)
    weight = np.zeros(len(node_id))
    idf = rng.uniform(0.3, 9.0, len(word_node)) if weighting in (0, 2) else np.ones(len(word_node))
    idf[rng.random(len(word_node)) < stop_fraction] = 0.0          # stopped words
    weight[word_node - 1] = idf
    return dict(k=k, L=L, scoring=0, weighting=weighting, node_id=node_id, parent_id=parent_id, weight=weight, desc=desc, word_node=word_node,
                word_id=word_id)


def place_descriptors(voc, seed, n=400, word_range=None):
    """descriptors of one 'place': the leaf descriptors of n random words (out of word_range = (lo, hi) if given)"""
    rng = np.random.default_rng(seed)
    lo, hi = word_range if word_range is not None else (0, len(voc["word_node"]))
    leaves = voc["word_node"][rng.integers(lo, hi, n)]
    return voc["desc"][leaves - 1].copy()


def view_of(place, seed, keep=0.8, noise_bits=12, extra=60):
    """one keyframe's view of a place: a subset of its descriptors with noise_bits toggled each, plus unrelated descriptors"""
    rng = np.random.default_rng(seed)
    sel = place[rng.random(len(place)) < keep]
    out = np.array([_flip(rng, d, noise_bits) for d in sel], np.uint64).reshape(-1, 4)
    junk = rng.integers(0, 2 ** 63, (extra, 4), dtype=np.uint64) * np.uint64(2)
    out = np.concatenate([out, junk])
    return np.ascontiguousarray(out[rng.permutation(len(out))])


def bind_oracle():
    L = vio_ct.oracle()
    if not getattr(L, "_bow_bound", False):
        L.ovio_bow_load.argtypes = [C.c_char_p]
        L.ovio_bow_load.restype = C.c_void_p
        L.ovio_bow_create.argtypes = [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 2
        L.ovio_bow_create.restype = C.c_void_p
        L.ovio_bow_destroy.argtypes = [C.c_void_p]
        L.ovio_bow_destroy.restype = None
        L.ovio_bow_info.argtypes = [C.c_void_p, C.c_void_p]
        L.ovio_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ovio_bow_vector.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ovio_bow_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ovio_bow_query.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ovio_bow_detect_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L._bow_bound = True
    return L


class OracleVoc:
    def __init__(self, voc=None, path=None):
        self.L = bind_oracle()
        if path is not None:
            self.h = self.L.ovio_bow_load(path.encode())
        else:
            a = [np.ascontiguousarray(voc[k]) for k in ("node_id", "parent_id", "weight", "desc", "word_node", "word_id")]
            self.h = self.L.ovio_bow_create(voc["k"], voc["L"], voc["scoring"], voc["weighting"], len(a[0]), a[0].ctypes.data, a[1].ctypes.data,
                                            a[2].ctypes.data, a[3].ctypes.data, len(a[4]), a[4].ctypes.data, a[5].ctypes.data)
        assert self.h, "oracle vocabulary failed to build"

    def close(self):
        if self.h:
            self.L.ovio_bow_destroy(self.h)
            self.h = None

    def info(self):
        o = np.zeros(6, np.int32)
        self.L.ovio_bow_info(self.h, o.ctypes.data)
        return [int(x) for x in o]

    def transform(self, desc):
        d = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
        w, wt = np.zeros(len(d), np.int32), np.zeros(len(d))
        self.L.ovio_bow_transform(self.h, d.ctypes.data, len(d), w.ctypes.data, wt.ctypes.data)
        return w, wt

    def bow(self, desc):
        d = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
        w, v = np.zeros(max(len(d), 1), np.int32), np.zeros(max(len(d), 1))
        m = self.L.ovio_bow_vector(self.h, d.ctypes.data, len(d), len(w), w.ctypes.data, v.ctypes.data)
        return w[:m], v[:m]

    def add(self, desc):
        d = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
        return self.L.ovio_bow_add(self.h, d.ctypes.data, len(d))

    def query(self, desc, max_results=4, max_id=-1, cap=4096):
        d = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
        ids, sc = np.zeros(cap, np.int32), np.zeros(cap)
        m = self.L.ovio_bow_query(self.h, d.ctypes.data, len(d), max_results, max_id, ids.ctypes.data, sc.ctypes.data)
        return ids[:m], sc[:m]

    def detect_loop(self, desc, frame_index):
        d = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
        return self.L.ovio_bow_detect_loop(self.h, d.ctypes.data, len(d), int(frame_index))


def hamming(a, b):
    return int(sum(bin(int(x) ^ int(y)).count("1") for x, y in zip(a, b)))


def reference_walk(voc, f):
    """pure-Python walk of the tree (children in file order, first smallest distance): (word id, weight)"""
    children = {}
    for nid, pid in zip(voc["node_id"], voc["parent_id"]):
        children.setdefault(int(pid), []).append(int(nid))
    word_of = {int(n): int(w) for n, w in zip(voc["word_node"], voc["word_id"])}
    node = 0
    while node in children:
        best, bd = None, None
        for c in children[node]:
            d = hamming(f, voc["desc"][c - 1])
            if bd is None or d < bd:
                best, bd = c, d
        node = best
    return word_of[node], float(voc["weight"][node - 1])


_POP8 = np.array([bin(i).count("1") for i in range(256)], np.uint16)


def _dist(desc, centre):
    """Hamming distances of desc[M][4] (uint64) to one centre"""
    x = np.bitwise_xor(desc, centre[None, :]).view(np.uint8)
    return _POP8[x].sum(1)


def train_vocabulary(desc, img, k, L, seed):
    """A vocabulary fitted to real descriptors the way DBoW2's trainer starts (hierarchical clustering, here: k random seeds per node and one
    assignment pass, no Lloyd iterations): desc[M][4], img[M] = image of each descriptor (for the idf weights log(N / N_i)).  Returns the
    arrays make_vocabulary returns (TF_IDF weighting, L1 scoring)."""
    rng = np.random.default_rng(seed)
    desc = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
    n_img = int(img.max()) + 1
    node_id, parent_id, ndesc, weight, leaves = [], [], [], [], []
    frontier = [(0, np.arange(len(desc)), 0)]
    nxt = 1
    while frontier:
        new = []
        for (pid, members, lv) in frontier:
            uniq = np.unique(desc[members], axis=0)
            kk = min(k, len(uniq))
            centres = uniq[rng.choice(len(uniq), kk, replace=False)]
            d = np.stack([_dist(desc[members], c) for c in centres], 1)
            lab = d.argmin(1)                                   # first smallest, like the walk
            for c in range(kk):
                mem = members[lab == c]
                nid = nxt; nxt += 1
                node_id.append(nid); parent_id.append(pid); ndesc.append(centres[c])
                if lv + 1 == L or len(np.unique(desc[mem], axis=0)) <= 1:
                    ni = len(np.unique(img[mem])) if len(mem) else 0
                    weight.append(np.log(n_img / ni) if ni else 0.0)
                    leaves.append(nid)
                else:
                    weight.append(0.0)
                    new.append((nid, mem, lv + 1))
        frontier = new
    return dict(k=k, L=L, scoring=0, weighting=0, node_id=np.asarray(node_id, np.int32), parent_id=np.asarray(parent_id, np.int32),
                weight=np.asarray(weight), desc=np.asarray(ndesc, np.uint64).reshape(-1, 4), word_node=np.asarray(leaves, np.int32),
                word_id=np.arange(len(leaves), dtype=np.int32))
