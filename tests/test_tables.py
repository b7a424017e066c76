"""Collocation tables: product (libmpx via mp.Collocation*) and oracle
against the reference's golden vectors, the reference's own known-answer tests, and mpmath."""
import os

import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp
from oracle import mpopt_oracle as npo

T = np.load(os.path.join(os.path.dirname(__file__), "golden", "tables.npz"))
SCHEMES = ["LGR", "LGL", "CGL"]
RANGES = {"tm1_1": (-1, 1), "t0_1": (0, 1)}


@pytest.fixture(autouse=True)
def _restore_class_state():
    yield
    mp.CollocationRoots._TAU_MIN, mp.CollocationRoots._TAU_MAX = -1, 1
    mp.Collocation.D_MATRIX_METHOD = "symbolic"


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())


@pytest.mark.parametrize("tag", list(RANGES))
@pytest.mark.parametrize("scheme", SCHEMES + ["LG"])
def test_roots_match_reference(tag, scheme):
    a, b = RANGES[tag]
    mp.CollocationRoots._TAU_MIN, mp.CollocationRoots._TAU_MAX = a, b
    fn = mp.CollocationRoots(scheme)._taus_fn
    for deg in [1, 2, 3, 4, 5, 6, 8, 10, 15, 20, 30]:
        key = f"{tag}/{scheme}/{deg}/roots"
        if key not in T:
            continue
        r = fn(deg)
        assert r.shape == T[key].shape
        assert np.abs(r - T[key]).max() < 5e-16 * 4, (scheme, deg)
        assert np.all(np.diff(r) > 0)
        assert np.abs(npo.roots(scheme, deg, a, b) - T[key]).max() == 0.0  # oracle: same arithmetic as the reference


@pytest.mark.parametrize("tag", list(RANGES))
@pytest.mark.parametrize("scheme", SCHEMES)
def test_tables_match_reference_numerical_backend(tag, scheme):
    """D (order 1, 2), w, interpolation and off-node matrices, degrees where the reference's
    coefficient arithmetic is itself accurate (<= 10): 1e-10 relative."""
    a, b = RANGES[tag]
    mp.CollocationRoots._TAU_MIN, mp.CollocationRoots._TAU_MAX = a, b
    for deg in [1, 2, 3, 4, 5, 6, 8, 10]:
        k = f"{tag}/{scheme}/{deg}"
        col = mp.Collocation([deg], scheme)
        # the golden itself (np.poly1d coefficient products) is only ~5e-10 accurate at p=10 on [0,1]
        TOL = 1e-10 if deg <= 8 else 5e-9
        assert rel(col.get_diff_matrix(deg).full(), T[k + "/D1"]) < TOL
        assert rel(col.get_diff_matrix(deg, order=2).full(), T[k + "/D2"]) < TOL
        assert rel(col.get_quadrature_weights(deg).full().ravel(), T[k + "/w"]) < TOL
        mids = T[k + "/mids"]
        assert rel(col.get_interpolation_matrix(mids, deg), T[k + "/C_mid"]) < TOL
        assert rel(col.get_diff_matrix(deg, taus=mids), T[k + "/D1_mid"]) < TOL
        assert rel(col.get_diff_matrix(deg, taus=mids, order=2), T[k + "/D2_mid"]) < TOL
        ends = np.array([col.tau0, col.tau1], dtype=float)
        assert rel(col.get_diff_matrix(deg, taus=ends), T[k + "/D1_ends"]) < TOL
        ab = T[k + "/w_sub_ab"]
        assert rel(col.get_quadrature_weights(deg, tau0=ab[0], tau1=ab[1]).full().ravel(), T[k + "/w_sub"]) < TOL
        # oracle = the reference's arithmetic restated
        x = npo.roots(scheme, deg, a, b)
        assert rel(npo.diff_matrix(x), T[k + "/D1"]) < 1e-13
        assert rel(npo.diff_matrix(x, order=2), T[k + "/D2"]) < 1e-13
        assert rel(npo.quad_weights(x, a, b), T[k + "/w"]) < 1e-13
        assert rel(npo.interp_matrix(x, mids), T[k + "/C_mid"]) < 1e-13


@pytest.mark.parametrize("tag", list(RANGES))
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("grid", ["20x3", "4x5", "3_10_3", "2_4_3"])
def test_composites_match_reference(tag, scheme, grid):
    a, b = RANGES[tag]
    mp.CollocationRoots._TAU_MIN, mp.CollocationRoots._TAU_MAX = a, b
    k = f"{tag}/{scheme}/comp_{grid}"
    orders = [int(v) for v in T[k + "/orders"]]
    col = mp.Collocation(orders, scheme)
    TOL = 1e-10 if max(orders) <= 8 else 5e-9  # accuracy of the golden itself, see above
    assert rel(col.get_composite_differentiation_matrix().full(), T[k + "/compD"]) < TOL
    assert rel(col.get_composite_quadrature_weights().full().ravel(), T[k + "/compW"]) < TOL
    taus_mid = [list((col._taus_fn(d)[:-1] + col._taus_fn(d)[1:]) / 2.0) for d in orders]
    assert rel(col.get_composite_interpolation_matrix(taus_mid, orders), T[k + "/compI_mid"]) < TOL
    taus_end = [np.array([col.tau0, col.tau1]) for _ in orders]
    assert rel(col.get_composite_interpolation_Dmatrix_at(taus_end, orders, order=1), T[k + "/compDat_ends"]) < TOL
    G = npo.Grid(orders, scheme, a, b)
    assert rel(G.comp_D(), T[k + "/compD"]) < 1e-13 and rel(G.comp_W(), T[k + "/compW"]) < 1e-13
    assert rel(G.comp_I_mid(), T[k + "/compI_mid"]) < 1e-13


def test_symbolic_backend_agrees_within_reference_tolerance():
    """tests/test_mpopt.py:612-624 of the reference: |symbolic - numerical| < 1e-5 (p=3; p=5 in
    examples/feature-demos/compare_symbolic_vs_numerical_approximation.py:28-51)."""
    for scheme in SCHEMES:
        for deg in (3, 5):
            col = mp.Collocation([deg], scheme)
            assert np.abs(col.get_composite_differentiation_matrix().full() - T[f"symbolic/{scheme}/{deg}/compD"]).max() < 1e-5
            assert np.abs(col.get_composite_quadrature_weights().full().ravel() - T[f"symbolic/{scheme}/{deg}/compW"]).max() < 1e-5


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("tmin", [-1, 0])
def test_degree_one_known_answers(scheme, tmin):
    """Reference tests/test_mpopt.py:927-1086: nodes = [tau_min, tau_max], cardinal basis,
    D = [[-1/h, 1/h], [-1/h, 1/h]], second-order D = 0."""
    mp.CollocationRoots._TAU_MIN, mp.CollocationRoots._TAU_MAX = tmin, 1
    col = mp.Collocation([1], scheme)
    taus = col.roots[1]
    assert taus[0] == tmin and taus[-1] == 1 and col.tau0 == taus[0] and col.tau1 == taus[-1]
    h = 1 - tmin
    D = col.get_diff_matrix(1).full()
    assert np.abs(D - np.array([[-1 / h, 1 / h], [-1 / h, 1 / h]])).max() < 1e-6
    assert np.abs(col.get_diff_matrix(1, order=2).full()).max() < 1e-6
    for j, pj in enumerate(col.polys[1]):
        for i, t in enumerate(taus):
            assert abs(pj(t) - (1.0 if i == j else 0.0)) < 1e-12


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("deg", [15, 20, 30, 50])
def test_high_degree_against_mpmath(scheme, deg):
    """At high degree the reference's np.poly1d arithmetic loses digits (4e-4 at p=30, see
    DESIGN.md); truth is a 50-digit mpmath evaluation of the same definitions."""
    import mpmath as mpm

    mpm.mp.dps = 50
    x = mp.CollocationRoots(scheme)._taus_fn(deg)
    col = mp.Collocation([deg], scheme)
    D = col.get_diff_matrix(deg).full()
    w = col.get_quadrature_weights(deg).full().ravel()
    xs = [mpm.mpf(float(v)) for v in x]
    n = len(xs)
    lam = [1 / mpm.fprod([xs[j] - xs[m] for m in range(n) if m != j]) for j in range(n)]
    Dt = np.zeros((n, n))
    for i in range(n):
        s = mpm.mpf(0)
        for j in range(n):
            if i != j:
                v = (lam[j] / lam[i]) / (xs[i] - xs[j])
                Dt[i, j] = float(v)
                s += v
        Dt[i, i] = float(-s)
    assert np.abs(D - Dt).max() / np.abs(Dt).max() < 1e-12
    # D annihilates constants and differentiates x exactly; w integrates the basis
    assert np.abs(D @ np.ones(n)).max() < 1e-9 * np.abs(D).max()
    assert np.abs(D @ x - 1).max() < 1e-9 * np.abs(D).max()
    assert abs(w.sum() - 2) < 1e-13 and abs(w @ x) < 1e-13 and abs(w @ x ** 2 - 2 / 3) < 1e-13
    if scheme == "LGR":
        assert abs(w[0]) < 1e-14  # flipped-Radau rule is exact: the extra left node carries no weight
    # roots against scipy (the reference's source of nodes)
    assert np.abs(x - npo.roots(scheme, deg)).max() < 1e-14


def test_unknown_scheme_and_lg_quirks():
    """mpopt.py:4182-4188: unknown scheme -> linspace with `degree` points; LG has p nodes (a4)."""
    r = mp.CollocationRoots("nope")._taus_fn(5)
    assert np.allclose(r, np.linspace(-1, 1, 5))
    assert len(mp.CollocationRoots("LG")._taus_fn(4)) == 4
    with pytest.raises(M.MpxError):
        M.NlpFunctions(mp.OCP(), 1, [3], "LG", with_device=False)


@pytest.mark.parametrize("S,po", [(3, 3), (1, [4]), (2, [2, 5])])
def test_lg_scheme_fails_where_and_how_the_reference_fails(S, po):
    """SURVEY row a4, closed as "matches the reference's failure": with scheme "LG" the reference CONSTRUCTS the optimizer and raises
    ValueError from create_nlp (compute_numerical_approximation -> get_composite_differentiation_matrix, mpopt.py:99 -> 4032-4038:
    "cannot reshape array of size p^2 into shape (p+1, p+1)") -- for every grid and both D_MATRIX_METHODs, checked against the
    imported reference in the build container.  Same here: the constructor succeeds, the node sets are served
    (p nodes), create_nlp raises an error that IS a ValueError (and an MpxError)."""
    import problems

    mpo = mp.mpopt(problems.moon_lander(mp, M.math), S, po, "LG")
    assert len(mp.CollocationRoots("LG")._taus_fn(5)) == 5
    with pytest.raises(ValueError) as e:
        mpo.create_nlp()
    assert isinstance(e.value, M.MpxError) and "LG" in str(e.value)
