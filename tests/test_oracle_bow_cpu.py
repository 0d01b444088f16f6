"""The oracle's restatement of the DBoW2 slice PoseGraph::detectLoop uses (oracle/bow.cpp) against independent Python restatements and
hand-computed cases.  No GPU, no vocabulary blob (missing upstream): synthetic vocabularies in the reference's file format."""
import importlib

import numpy as np
import pytest

import bow_util
import vio_ct


def test_tree_walk_and_bow_vector_follow_the_definition():
    voc = bow_util.make_vocabulary(6, 3, 3, irregular=True, stop_fraction=0.2)
    o = bow_util.OracleVoc(voc)
    assert o.info()[:2] == [6, 3] and o.info()[4] == len(voc["node_id"]) and o.info()[5] == len(voc["word_node"])
    rng = np.random.default_rng(0)
    feats = np.concatenate([bow_util.view_of(bow_util.place_descriptors(voc, 1, 120), 2), rng.integers(0, 2 ** 63, (40, 4), dtype=np.uint64)])
    w, wt = o.transform(feats)
    for i in range(0, len(feats), 7):
        assert (int(w[i]), float(wt[i])) == bow_util.reference_walk(voc, feats[i])
    # BowVector: TF_IDF accumulates the idf per occurrence in feature order, stopped words (weight 0) are skipped, then L1 normalisation
    acc = {}
    for wi, x in zip(w, wt):
        if x > 0:
            acc[int(wi)] = acc.get(int(wi), 0.0) + float(x)
    ids = sorted(acc)
    norm = 0.0
    for i in ids:
        norm += abs(acc[i])
    bw, bv = o.bow(feats)
    assert list(bw) == ids
    assert np.array_equal(bv, np.array([acc[i] / norm for i in ids]))
    assert abs(bv.sum() - 1.0) < 1e-12 and (wt == 0).any()
    o.close()


def test_hand_made_vocabulary_known_answers():
    """k = 2, L = 2: four words.  Descriptor = all zeros except word 0; distances and scores by hand."""
    Z, F = [0, 0, 0, 0], [2 ** 64 - 1] * 4
    node_id, parent_id = [1, 2, 3, 4, 5, 6], [0, 0, 1, 1, 2, 2]
    desc = np.array([Z, F, Z, [0xFF, 0, 0, 0], F, [2 ** 64 - 1, 2 ** 64 - 1, 2 ** 64 - 1, 0]], np.uint64)
    voc = dict(k=2, L=2, scoring=0, weighting=0, node_id=np.array(node_id, np.int32), parent_id=np.array(parent_id, np.int32),
               weight=np.array([0, 0, 1.0, 2.0, 4.0, 0.5]), desc=desc, word_node=np.array([3, 4, 5, 6], np.int32), word_id=np.array([0, 1, 2, 3], np.int32))
    o = bow_util.OracleVoc(voc)
    f = np.array([Z, [0xFF, 0, 0, 0], [0x0F, 0, 0, 0], F, [2 ** 64 - 1, 2 ** 64 - 1, 0, 0]], np.uint64)
    w, wt = o.transform(f)
    # [0x0F]: distance 4 to both children of node 1 -> the FIRST (word 0); the half-set descriptor ties at the root (128 / 128) -> the first
    # child (node 1), below it 128 to node 3 against 120 to node 4 -> word 1
    assert list(w) == [0, 1, 0, 2, 1] and list(wt) == [1.0, 2.0, 1.0, 4.0, 2.0]
    bw, bv = o.bow(f)
    assert list(bw) == [0, 1, 2] and np.allclose(bv, np.array([2.0, 4.0, 4.0]) / 10.0, rtol=0, atol=1e-16)
    # database: entry 0 = {w0}, entry 1 = {w1, w2}; query {w0, w1}: L1 score = 1 - 0.5 * |v - w|_1
    assert o.add(f[[0]]) == 0 and o.add(f[[1, 3]]) == 1
    ids, sc = o.query(f[[0, 0, 1]], 4, -1)                 # (w0, w1) = (2, 2) / 4
    e0 = 1 - 0.5 * (abs(0.5 - 1.0) + 0.5)
    e1 = 1 - 0.5 * (0.5 + abs(0.5 - 2 / 6) + 4 / 6)
    assert list(ids) == [0, 1] and np.allclose(sc, [e0, e1], atol=1e-15)
    # max_id = 0 hides entry 0, the newest entry is always considered (TemplatedDatabase.h queryL1)
    ids, sc = o.query(f[[0, 0, 1]], 4, 0)
    assert list(ids) == [1]
    o.close()


def test_file_format_round_trip(tmp_path):
    pg = importlib.import_module("vins-rgbd-fast_amd.posegraph")
    voc = bow_util.make_vocabulary(5, 3, 9)
    path = str(tmp_path / "voc.bin")
    pg.write_vocabulary(path, voc["k"], voc["L"], 0, 0, voc["node_id"], voc["parent_id"], voc["weight"], voc["desc"], voc["word_node"], voc["word_id"])
    import os
    assert os.path.getsize(path) == 24 + 48 * len(voc["node_id"]) + 8 * len(voc["word_node"])      # VocabularyBinary.hpp layout
    a, b = bow_util.OracleVoc(voc), bow_util.OracleVoc(path=path)
    feats = bow_util.view_of(bow_util.place_descriptors(voc, 4, 200), 5)
    assert a.info() == b.info()
    assert all(np.array_equal(x, y) for x, y in zip(a.transform(feats), b.transform(feats)))
    a.close(); b.close()


def test_detect_loop_finds_the_revisited_place():
    """PoseGraph::detectLoop (pose_graph.cpp:308-393): 40 places visited for 3 keyframes each, then places 2 and 5 revisited.  The database
    is queried BEFORE the keyframe is added; candidates must be 50 keyframes old (the newest entry is always scored: it is ret[0], the
    "neighbour"); the EARLIEST candidate above 0.015 is returned.  k = 10, L = 5 (1e5 words), every place in its own part of the tree."""
    voc = bow_util.make_vocabulary(10, 5, 21)
    o = bow_util.OracleVoc(voc)
    places = [bow_util.place_descriptors(voc, 100 + p, 60, (2000 * p, 2000 * p + 2000)) for p in range(40)]   # disjoint parts of the tree
    seq = [p for p in range(40) for _ in range(3)] + [2, 2, 2, 5, 5, 5]
    found = {}
    for idx, p in enumerate(seq):
        r = o.detect_loop(bow_util.view_of(places[p], 1000 + idx, noise_bits=4, extra=0), idx)
        if r != -1:
            found[idx] = r
    assert found == {120: 6, 121: 6, 122: 6, 123: 15, 124: 15, 125: 15}, found
    # frame_index <= 50 never reports a loop, whatever the scores (:384)
    o2 = bow_util.OracleVoc(voc)
    for idx in range(6):
        assert o2.detect_loop(bow_util.view_of(places[0], 2000 + idx, noise_bits=4, extra=0), idx) == -1
    o.close(); o2.close()


def _ref_bowvector():
    """the reference's own DBoW2::BowVector compiled from where it lies (oracle/_ref, `make -C oracle ref`); None where it cannot be built"""
    import ctypes as C
    import os
    import subprocess
    so = os.path.join(vio_ct.ROOT, "oracle", "_ref", "libdbow_bowvector_ref.so")
    if not os.path.exists(so):
        if not os.path.isdir("/root/reference/pose_graph/src/ThirdParty/DBoW"):
            return None
        subprocess.run(["make", "-C", os.path.join(vio_ct.ROOT, "oracle"), "ref"], check=True, capture_output=True)
    L = C.CDLL(so)
    L.oref_bow_vector.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return L


@pytest.mark.parametrize("weighting", [0, 1, 2, 3])
def test_bow_vector_against_the_reference_class_itself(weighting):
    """The one piece of reference arithmetic that compiles in this image from its own source: DBoW2::BowVector (ThirdParty/DBoW/BowVector.cpp).
    The restatement's bag-of-words vector -- accumulation per feature for TF-IDF / TF, first occurrence for IDF / BINARY, stopped words,
    L1 normalisation -- must equal, bit for bit, what the reference's class makes of the same (word, weight) sequence."""
    R = _ref_bowvector()
    if R is None:
        pytest.skip("reference tree not present and oracle/_ref not prebuilt")
    voc = bow_util.make_vocabulary(7, 3, 40 + weighting, irregular=True, weighting=weighting, stop_fraction=0.1)
    o = bow_util.OracleVoc(voc)
    for seed in range(4):
        feats = bow_util.view_of(bow_util.place_descriptors(voc, seed, 700), 50 + seed, noise_bits=30, extra=200)
        w, wt = o.transform(feats)
        ids, val = np.zeros(len(w), np.int32), np.zeros(len(w))
        w32 = np.ascontiguousarray(w, np.int32)
        m = R.oref_bow_vector(len(w), w32.ctypes.data, wt.ctypes.data, weighting, len(w), ids.ctypes.data, val.ctypes.data)
        bw, bv = o.bow(feats)
        assert m == len(bw) > 20 and np.array_equal(ids[:m], bw) and np.array_equal(val[:m], bv)
        assert (np.bincount(w32).max() > 1)          # repeated words: the accumulation order matters
    o.close()
