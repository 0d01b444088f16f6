"""Place recognition on the GPU (vio_pg_voc_* in include/vio_posegraph.h) against the oracle's restatement of the vendored DBoW2 (oracle/bow.cpp):
tree walk bit for bit, bag-of-words vectors, database queries and PoseGraph::detectLoop decisions identical (same summation orders)."""
import importlib

import numpy as np
import pytest

import bow_util
import vio_ct

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    vio_ct.pkg()
    return importlib.import_module("vins-rgbd-fast_amd.posegraph")


def _hip(pg, voc):
    return pg.Vocabulary.from_arrays(voc["k"], voc["L"], voc["scoring"], voc["weighting"], voc["node_id"], voc["parent_id"], voc["weight"], voc["desc"],
                                     voc["word_node"], voc["word_id"])


@pytest.mark.parametrize("k,L,irregular,weighting", [(10, 4, False, 0), (12, 4, True, 0), (70, 2, False, 0), (6, 3, True, 1), (6, 3, True, 2), (5, 3, False, 3)])
def test_tree_walk_and_bow_vector_bit_exact(pg, k, L, irregular, weighting):
    voc = bow_util.make_vocabulary(k, L, 7 + k, irregular=irregular, weighting=weighting, stop_fraction=0.05)
    o, h = bow_util.OracleVoc(voc), _hip(pg, voc)
    info = h.info()
    assert [info[x] for x in ("k", "L", "scoring", "weighting", "nodes", "words")] == o.info()
    rng = np.random.default_rng(k)
    feats = np.concatenate([bow_util.view_of(bow_util.place_descriptors(voc, 1, 900), 2, noise_bits=20, extra=300),
                            voc["desc"][rng.integers(0, len(voc["desc"]), 64)],                                       # exact node descriptors: ties
                            np.zeros((3, 4), np.uint64), np.full((2, 4), 2 ** 64 - 1, np.uint64)])
    wo, wto = o.transform(feats)
    wh, wth = h.transform(feats)
    assert np.array_equal(wo, wh) and np.array_equal(wto, wth)
    for a, b in zip(o.bow(feats), h.bow(feats)):
        assert np.array_equal(a, b)
    assert len(h.bow(feats[:0])[0]) == 0                  # no features: empty vector
    o.close(); h.close()


def test_database_and_detect_loop_follow_the_oracle(pg, tmp_path):
    voc = bow_util.make_vocabulary(10, 4, 33)
    path = str(tmp_path / "voc.bin")
    pg.write_vocabulary(path, voc["k"], voc["L"], 0, 0, voc["node_id"], voc["parent_id"], voc["weight"], voc["desc"], voc["word_node"], voc["word_id"])
    o, h = bow_util.OracleVoc(path=path), pg.Vocabulary.load(path)        # PoseGraph::loadVocabulary on the reference's file format
    places = [bow_util.place_descriptors(voc, 100 + p, 150) for p in range(30)]
    seq = [p for p in range(30) for _ in range(3)] + [3, 3, 7, 7, 7, 11, 3]
    loops = 0
    for idx, p in enumerate(seq):
        d = bow_util.view_of(places[p], 500 + idx, noise_bits=8, extra=40)
        if idx % 9 == 4:   # free queries at several cut-offs before the keyframe enters the database
            for (mr, mid) in ((4, -1), (0, -1), (10, idx - 20), (3, 0)):
                io, so = o.query(d, mr, mid)
                ih, sh = h.query(d, mr, mid)
                assert np.array_equal(io, ih) and np.array_equal(so, sh), (idx, mr, mid)
        lo = o.detect_loop(d, idx)
        lh, ids, sc = h.detectLoop(d, idx, with_results=True)
        assert lo == lh, (idx, lo, lh)
        assert len(ids) <= 4 and (len(sc) < 2 or (np.diff(sc) <= 0).all())
        loops += int(lh != -1)
    assert loops >= 5 and h.info()["entries"] == len(seq)
    # addKeyFrameIntoVoc (pose_graph.cpp:395-408): plain add
    assert o.add(places[0]) == h.add(places[0]) == len(seq)
    o.close(); h.close()


def test_vocabulary_errors_are_reported(pg, tmp_path):
    P = vio_ct.pkg()
    with pytest.raises(P.VioError):
        pg.Vocabulary.load(str(tmp_path / "missing.bin"))
    voc = bow_util.make_vocabulary(3, 2, 1)
    with pytest.raises(P.VioError):   # only L1 scoring
        pg.Vocabulary.from_arrays(3, 2, 1, 0, voc["node_id"], voc["parent_id"], voc["weight"], voc["desc"], voc["word_node"], voc["word_id"])
    (tmp_path / "short.bin").write_bytes(b"\\x03\\x00\\x00\\x00" * 7)
    with pytest.raises(P.VioError):
        pg.Vocabulary.load(str(tmp_path / "short.bin"))
