"""Resident kernel of single evaluations (mpx_kernels.h: resident_loop; opt-in, MPX_RESIDENT=1 -- it measured no faster than the
launched kernels): the regime an NLP solver drives -- one evaluation point per call through host pointers in
page-locked memory -- without a kernel launch or a stream synchronisation per call.  Results must be the bits of the launched
kernels (MPX_NO_RESIDENT=1); the kernel leaves when idle and comes back on demand; mpx_destroy ends it.
Call mix of an interior-point iteration: docs/source/notebooks/moon_lander.ipynb:192-198."""
import ctypes
import os
import time

import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp, _lib
import problems

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _resident_on(monkeypatch):
    monkeypatch.setenv("MPX_RESIDENT", "1")  # (read per call by libmpx)

CASES = {"moon_lander_20x3": (problems.moon_lander, 20, 3, "LGR"), "moon_lander_1000x5": problems.BENCH_CASES[0],
         "hyper_sensitive_4000x3": problems.BENCH_CASES[3], "dae_vdp_9x7": (problems.dae_vdp, 9, 7, "LGL")}


def build(case):
    builder, S, po, scheme = case
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    return ocp, mpo, mpo.create_nlp()[0]["oracle"]


@pytest.mark.parametrize("name", list(CASES))
def test_resident_equals_launched_kernels(name):
    ocp, mpo, o = build(CASES[name])
    rng = np.random.default_rng(1)
    z0 = mpo.initialize_solution()
    p = np.full(o.n_p, 1.0 / o.n_segments)
    what_sets = (["f"], ["g"], ["f", "g"], ["f", "grad_f"], ["g", "jac_g"], ["f", "g", "grad_f", "jac_g"], ["hess_l"], ["f", "g", "grad_f", "jac_g", "hess_l"])
    for k in range(3):
        z = z0 * (1 + 0.02 * rng.uniform(-1, 1, o.n_z)) + 0.02 * rng.uniform(-1, 1, o.n_z)
        lam, sig = rng.standard_normal(o.n_g), float(rng.uniform(0.5, 1.5))
        if k == 2:
            w = rng.uniform(0.5, 1.5, o.n_p)
            p = w / w.sum() * ocp.n_phases  # the widths change: prefix sums are recomputed
        for ccs in (False, True):
            for what in what_sets:
                a = o.eval(what, z, p, lam_g=lam, sigma=sig, pinned=True, ccs_order=ccs)
                a = {q: np.array(v, copy=True) for q, v in a.items()}
                os.environ["MPX_NO_RESIDENT"] = "1"
                try:
                    b = o.eval(what, z, p, lam_g=lam, sigma=sig, pinned=True, ccs_order=ccs)
                finally:
                    del os.environ["MPX_NO_RESIDENT"]
                for q in what:
                    assert np.array_equal(a[q], b[q]), (name, k, ccs, what, q)
    o.close()


def test_resident_kernel_leaves_when_idle_and_comes_back():
    ocp, mpo, o = build(CASES["moon_lander_20x3"])
    z, p = mpo.initialize_solution(), np.full(o.n_p, 1.0 / 20)
    r0 = np.array(o.eval(["g"], z, p, pinned=True)["g"], copy=True)
    for pause in (0.0, 0.06, 0.0, 0.06):  # the idle limit is 20 ms: the kernel has left after 60 ms
        time.sleep(pause)
        assert np.array_equal(o.eval(["g"], z, p, pinned=True)["g"], r0)
    # many calls back to back: the regime itself
    t0 = time.perf_counter()
    for _ in range(2000):
        o.eval_pinned_again = o._L.mpx_eval  # (attribute access only: keep the loop honest about Python overhead)
        r = o.eval(["g"], z, p, pinned=True)
    dt = (time.perf_counter() - t0) / 2000
    assert np.array_equal(r["g"], r0)
    print(f"[resident] moon lander 20x3: {dt * 1e6:.1f} us per o.eval(['g'], pinned=True) incl. Python")
    o.close()  # (mpx_destroy right after a request: the kernel is asked to stop, its stream is drained)
    ocp, mpo, o = build(CASES["moon_lander_20x3"])
    assert np.array_equal(o.eval(["g"], z, p, pinned=True)["g"], r0)
    time.sleep(0.06)
    o.close()  # (and after it has left by itself)
