"""Ceres' projected Armijo line search on a bounds-constrained program (VERDICT r5 "missing" 1; estimator.cpp:1282-1297 bounds the inverse
depth of depth-less landmarks, which makes TrustRegionMinimizer run ArmijoLineSearch along every trust-region step).  CPU only: the scalar
machinery (oracle/om.h LS1D block = csrc/dmath.h, text equality asserted) against an independent numpy restatement, and the oracle's solves
on a scene whose steps violate the bound, every recorded search replayed by that numpy reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import vio_ct


@pytest.fixture(scope="module")
def orc():
    L = vio_ct.oracle()
    L.ovio_ls_next_step.restype = C.c_double
    L.ovio_ls_next_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    L.ovio_poly_roots_real.restype = C.c_int
    L.ovio_poly_roots_real.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    return L


@pytest.fixture(scope="module")
def P():
    return vio_ct.pkg()


# ---- the numpy reference: Ceres' polynomial.cc / line_search.cc restated independently (np.linalg.solve, np.roots) ----
def np_fit(samples):
    """Hermite interpolation through (x, value, gradient) samples: coefficients, highest degree first (FindInterpolatingPolynomial)"""
    n = 2 * len(samples)
    A, b = np.zeros((n, n)), np.zeros(n)
    for i, (x, v, g) in enumerate(samples):
        for j in range(n):
            A[2 * i, j] = x ** (n - 1 - j)
            A[2 * i + 1, j] = (n - 1 - j) * x ** (n - 2 - j) if j < n - 1 else 0.0
        b[2 * i], b[2 * i + 1] = v, g
    return np.linalg.solve(A, b)


def np_next_step(lower, previous, current, lo, hi):
    if not current[3]:
        return min(max(current[0] * 0.5, lo), hi)
    smp = [lower[:3], current[:3]] + ([previous[:3]] if previous[3] else [])
    c = np_fit(smp)
    cand = [(lo + hi) / 2.0, lo, hi]
    d = np.polyder(c)
    d = np.trim_zeros(d, "f")
    if len(d) > 1:
        cand += [float(r.real) for r in np.roots(d) if lo <= r.real <= hi]
    cand += [s[0] for s in smp if lo <= s[0] <= hi]
    vals = [np.polyval(c, x) for x in cand]
    return cand[int(np.argmin(vals))], min(vals), c


def np_armijo(f, cost, g0, dmax):
    """ArmijoLineSearch::DoSearch with the Solver::Options defaults; f(alpha) -> (value, gradient, valid).  Returns (alpha, success, trials)"""
    lower, previous = (0.0, cost, g0, 1), (0.0, 0.0, 0.0, 0)
    current = (1.0,) + tuple(f(1.0))
    trials, it = [current], 0
    while (not current[3]) or current[1] > cost + 1e-4 * g0 * current[0]:
        it += 1
        if it >= 20:
            return 1.0, False, trials
        r = np_next_step(lower, previous, current, 1e-3 * current[0], 0.6 * current[0])
        a = r if not isinstance(r, tuple) else r[0]
        if a * dmax < 1e-9:
            return 1.0, False, trials
        previous = current
        current = (a,) + tuple(f(a))
        trials.append(current)
    return current[0], True, trials


def test_the_line_search_text_is_shared(P):
    def body(path):
        t = open(path).read()
        m = re.search(r"// ---- LS1D:.*?// ---- end LS1D[^\n]*\n", t, re.S)
        assert m, path
        return m.group(0)
    a = body(os.path.join(vio_ct.ROOT, "oracle", "om.h"))
    assert a == body(os.path.join(vio_ct.ROOT, "vins-rgbd-fast_amd", "csrc", "dmath.h")) and "ls_next_step" in a


def test_polynomial_roots_against_numpy(orc):
    rng = np.random.default_rng(3)
    worst = 0.0
    for trial in range(400):
        deg = int(rng.integers(1, 5))
        if trial % 3 == 0:   # from chosen roots: clusters, conjugate pairs
            roots = []
            while len(roots) < deg:
                if deg - len(roots) >= 2 and rng.random() < 0.5:
                    z = complex(rng.normal(0, 2), abs(rng.normal(0, 1)) + 0.05)
                    roots += [z, z.conjugate()]
                else:
                    roots.append(complex(rng.normal(0, 2), 0.0))
            c = np.real(np.poly(roots)) * rng.uniform(0.5, 3.0) * rng.choice([-1, 1])
        else:
            c = rng.normal(0, 1, deg + 1) * 10.0 ** rng.uniform(-2, 2, deg + 1)
        if trial % 7 == 0:
            c = np.r_[0.0, c]   # a leading zero is removed first
        c = np.ascontiguousarray(c, np.float64)
        out = np.zeros(4)
        n = orc.ovio_poly_roots_real(c.ctypes.data, len(c), out.ctypes.data)
        ref = np.roots(np.trim_zeros(c, "f"))
        assert n == len(ref)
        got, want = np.sort(out[:n]), np.sort(ref.real)
        scale = 1.0 + np.abs(ref).max()
        # (a double real root splits into a pair whose real parts carry sqrt(eps): the interpolation only ever compares polynomial VALUES there)
        tol = 1e-6 * scale if trial % 3 == 0 else 1e-9 * scale
        assert np.abs(got - want).max() < tol, (c, got, want)
        worst = max(worst, float(np.abs(got - want).max() / scale))
    assert worst < 1e-6


def test_next_step_against_numpy(orc):
    rng = np.random.default_rng(11)
    for trial in range(300):
        cost = rng.uniform(1, 100)
        g0 = -rng.uniform(0.1, 50)
        lower = np.array([0.0, cost, g0, 1.0])
        xc = rng.uniform(0.01, 1.0)
        current = np.array([xc, cost + rng.normal(0.5, 1.0) * abs(g0) * xc, rng.normal(0, 2) * abs(g0), 1.0])
        if trial % 2:
            xp = xc / rng.uniform(0.05, 0.6)
            previous = np.array([xp, cost + rng.normal(1.0, 1.0) * abs(g0) * xp, rng.normal(0, 2) * abs(g0), 1.0])
        else:
            previous = np.zeros(4)
        if trial % 17 == 0:
            current[3] = 0.0
        lo, hi = 1e-3 * xc, 0.6 * xc
        got = orc.ovio_ls_next_step(lower.ctypes.data, previous.ctypes.data, current.ctypes.data, lo, hi)
        ref = np_next_step(tuple(lower), tuple(previous), tuple(current), lo, hi)
        assert lo <= got <= hi
        if not isinstance(ref, tuple):
            assert got == ref
            continue
        x_ref, v_ref, c = ref
        # the same minimum of the same interpolating polynomial (two candidates may tie to round-off: compare the polynomial's value too)
        assert abs(got - x_ref) <= 1e-7 * hi or abs(np.polyval(c, got) - v_ref) <= 1e-9 * (abs(v_ref) + abs(cost)), (trial, got, x_ref)


def _blind_frames(P, sc, seq, n, blind_mm):
    syn = P.Synth(sc)
    out = []
    for t in vio_ct.frame_times(sc, n):
        g, d = syn.render_host(seq, float(t))
        d = d.copy()
        d[d > blind_mm] = 0
        out.append((g, d))
    return out


def _read_searches(path):
    rows = np.fromfile(path, np.float64).reshape(-1, 5)
    out, cur = [], None
    for r in rows:
        if r[0] == -1.0:
            cur = dict(cost=r[1], g0=r[2], dmax=r[3], clamped=int(r[4]), trials=[])
        elif r[0] == -2.0:
            cur.update(alpha=r[1], success=bool(r[2]))
            out.append(cur)
        else:
            cur["trials"].append((r[0], r[1], r[2], int(r[3])))
    return out


def test_constrained_solves_run_the_projected_armijo_search(P, tmp_path, monkeypatch):
    """A sensor declared to reach 10 m but blind beyond 2.5 m (the scene of test_inverse_depth_bound_engages_identically): parallax-only
    landmarks at 2.5 .. 5 m sit on / beyond the bound 2 / DEPTH_MAX_DIST, the dogleg step pushes them outside and the projected step no longer
    decreases the cost as predicted.  Every search the oracle ran is replayed by the numpy Armijo reference on the recorded function samples:
    same trial steps, same outcome; the accepted step satisfies the sufficient-decrease condition, the rejected ones do not."""
    cfg = P.canonical_config(depth_max=10.0)
    sc = vio_ct.synth_like(cfg)
    seq, n = 2, 40
    frames = _blind_frames(P, sc, seq, n, 2500)
    dump = str(tmp_path / "ls.bin")
    monkeypatch.setenv("OVIO_DUMP_LS", dump)
    o = vio_ct.run_oracle_sequence(cfg, sc, seq, n, frames=frames)
    monkeypatch.delenv("OVIO_DUMP_LS")
    evals, contractions = o["oracle"].line_search_stats()
    clamps, bounded = o["oracle"].bound_stats()
    S = _read_searches(dump)
    assert len(S) > 100 and evals == sum(len(s["trials"]) for s in S) and contractions == evals - len(S)
    assert bounded > 1000 and clamps > 1000
    violating = [s for s in S if s["clamped"] > 0]
    shortened = [s for s in S if s["success"] and s["alpha"] < 1.0]
    assert len(violating) > 50 and len(shortened) > 20 and any(s["clamped"] > 0 for s in shortened)
    # with landmarks ON the bound whose gradient points outwards the projected step can be an ascent direction however short it is: the search
    # then ends on "step size too small" (alpha * |delta|_inf < 1e-9), the step is left as it was and the trust-region test rejects it
    failed = [s for s in S if not s["success"]]
    assert failed and all(s["alpha"] == 1.0 and s["trials"][-1][0] * s["dmax"] >= 1e-9 for s in failed)
    assert any(len(s["trials"]) >= 3 for s in S)     # the three-sample (quintic) interpolation is exercised
    for s in S:
        assert s["g0"] < 0 and s["trials"][0][0] == 1.0
        table = {t[0]: t[1:] for t in s["trials"]}

        def f(a):
            k = min(table, key=lambda x: abs(x - a))
            # the reference asks for the step the oracle tried (to the conditioning of the 6 x 6 Hermite system of a quintic through
            # samples a decade apart: two LU variants agree to ~1e-9 there)
            assert abs(k - a) <= 1e-6 * max(a, 1e-30), (a, k)
            return table[k]
        alpha, ok, trials = np_armijo(f, s["cost"], s["g0"], s["dmax"])
        assert ok == s["success"] and len(trials) == len(s["trials"])
        assert abs(alpha - s["alpha"]) <= 1e-6 * alpha
        for t in s["trials"][:-1]:
            assert (not t[3]) or t[1] > s["cost"] + 1e-4 * s["g0"] * t[0]
        if ok:
            t = s["trials"][-1]
            assert t[1] <= s["cost"] + 1e-4 * s["g0"] * t[0]
    # the estimate is better for it (the clamp-only treatment of rounds 1 - 5 is kept behind reference_quirks bit 3 for this comparison)
    cfg2 = P.canonical_config(depth_max=10.0, reference_quirks=8)
    o2 = vio_ct.run_oracle_sequence(cfg2, sc, seq, n, frames=frames)
    assert o2["oracle"].line_search_stats() == (0, 0)
    ate = lambda oo: vio_ct.ate_rmse(np.array([x[1] for x in oo["traj"]]), np.array(oo["gt"]))
    assert ate(o) < ate(o2)


def test_an_unconstrained_program_never_enters_the_search(P):
    """No bounded landmark in the problem (the canonical workload: every pixel has depth) -> Program::IsBoundsConstrained() is false, no
    projection of x0, no line search: bit-identical to the clamp-only build."""
    sc = None
    res = []
    for quirks in (0, 8):
        cfg = P.canonical_config(reference_quirks=quirks)
        sc = vio_ct.synth_like(cfg)
        o = vio_ct.run_oracle_sequence(cfg, sc, 3, 40)
        assert o["oracle"].bound_stats() == (0, 0) and o["oracle"].line_search_stats() == (0, 0)
        res.append(np.array([x[1] for x in o["traj"]]))
    assert np.array_equal(res[0], res[1])
