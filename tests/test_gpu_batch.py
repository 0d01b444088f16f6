"""Batch semantics on the GPU: a sequence gives bit-identical results whether it runs alone or inside a batch (no cross-sequence
state), and the marginalisation prior handed to the next frame agrees with the oracle's as a quadratic form."""
import numpy as np
import pytest

import vio_ct

pytestmark = pytest.mark.gpu


def _drive(P, cfg, sc, seqs, n):
    syn = P.Synth(sc)
    S = len(seqs)
    b = P.VioBatch(cfg, S)
    imu = [syn.imu(s, int(n / sc.cam_rate * sc.imu_rate) + 64) for s in seqs]
    k = [0] * S
    for f, tf in enumerate(vio_ct.frame_times(sc, n)):
        for i in range(S):
            ti, ai, gi = imu[i]
            k2 = vio_ct.imu_until(ti, k[i], tf, sc.imu_rate)
            if k2 > k[i]:
                b.push_imu(i, ti[k[i]:k2], ai[k[i]:k2], gi[k[i]:k2])
            k[i] = k2
        fr = [syn.render_host(s, float(tf)) for s in seqs]
        b.feed(np.stack([x[0] for x in fr]), np.stack([x[1] for x in fr]), [tf] * S)
    return b


def test_sequence_is_independent_of_its_batch(P):
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    n = 26
    batch = _drive(P, cfg, sc, list(range(20, 30)), n)      # 10 sequences in one handle
    for pos, s in ((2, 22), (9, 29)):
        alone = _drive(P, cfg, sc, [s], n)
        wa, wb = alone.window(0), batch.window(pos)
        assert np.array_equal(wa, wb), (s, float(np.abs(wa - wb).max()))          # bit-identical window state
        ta, tb = alone.tracks(0), batch.tracks(pos)
        assert all(np.array_equal(x, y) for x, y in zip(ta, tb))
        la, lb = alone.landmarks(0), batch.landmarks(pos)
        assert np.array_equal(la, lb)
        pa, pb = alone.prior(0), batch.prior(pos)
        assert (pa is None) == (pb is None)
        if pa is not None:
            assert all(np.array_equal(x, y) for x, y in zip(pa, pb))


def test_prior_matches_oracle_as_a_quadratic_form(P):
    """The solver consumes the prior only through J^T J, J^T r and |r|^2 (J itself is defined up to an orthogonal factor and the
    sign of each eigenvector): compare those, in the canonical layout, right after initialisation and a few frames later."""
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    seq = 7
    for n in (30, 42):
        ref = vio_ct.run_oracle_sequence(cfg, sc, seq, n)
        b = _drive(P, cfg, sc, [seq], n)
        po, ph = ref["oracle"].prior(), b.prior(0)
        assert po is not None and ph is not None
        Jo, ro, xo, preso = po
        Jh, rh, xh, presh = ph
        assert np.array_equal(preso, presh)
        assert np.abs(xo - xh).max() < 1e-6  # linearisation point = window states: parity of the pipeline itself
        Ho, Hh = Jo.T @ Jo, Jh.T @ Jh
        scale = np.abs(Ho).max()
        assert scale > 1e6  # an informative prior (right after initialisation it is numerically empty)
        assert np.abs(Ho - Hh).max() < 1e-5 * scale, float(np.abs(Ho - Hh).max() / scale)
        go, gh = Jo.T @ ro, Jh.T @ rh
        assert np.abs(go - gh).max() < 1e-5 * max(1.0, np.abs(go).max())
        assert abs(ro @ ro - rh @ rh) < 1e-5 * max(1.0, ro @ ro)
