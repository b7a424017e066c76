"""Row spans of mixed-degree grids (heavy passes: the largest bucket's tiles write whole g / grad_f row spans from LDS, DESIGN.md
section 4) up to the real LDS size of a compute unit, and the planner's notes when a grid falls back to the staging block + unpack
pass (mpx_get_notes).  Reference rows: mpopt.py:458 (row order), 227-232 (defects)."""
import os

import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp
import problems

mixed = lambda S: [30 if s % 3 == 1 else 3 for s in range(S)]


def test_two_halves_grid_falls_back_and_says_so():
    """80 segments of degree 30 followed by 800 of degree 3: the last degree-30 tile would have to absorb 2400 foreign nodes."""
    o = M.NlpFunctions(problems.van_der_pol(mp, M.math), 880, [30] * 80 + [3] * 800, "CGL", with_device=False)
    first, length, foreign = o.tile_spans()
    assert not length.any()
    notes = o.notes()
    assert len(notes) == 1 and "unpack pass" in notes[0] and "2400" in notes[0]
    o.close()
    o = M.NlpFunctions(problems.van_der_pol(mp, M.math), 48, mixed(48), "CGL", with_device=False)
    assert o.tile_spans()[1].any() and o.notes() == []
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["kitchen_sink", "staged_ascent"])
def test_row_spans_beyond_64_kb_of_lds(case):
    """Kitchen sink (3 states, 2 controls, path rows, DU rows: 14 row slots per node) on config 3's degree pattern: span rows +
    the degree-30 kernel's own LDS exceed the 64 KB a launch gets by default -- the library raises the kernels' dynamic shared
    memory limit instead of falling back.  Bit-identical to the unpack path."""
    if case == "kitchen_sink":
        ocp, S, po = problems.kitchen_sink(mp, M.math), 48, mixed(48)
    else:  # 7 states + 3 controls (examples/Multi-phase launch-vehicle family): the grid tests/test_host.py plans on the host
        ocp, S, po = problems.staged_ascent(mp, M.math), 30, [3, 4, 3] * 10
    mpo = mp.mpopt(ocp, S, po, "LGR")
    o = mpo.create_nlp()[0]["oracle"]
    assert o.tile_spans()[1].any() and o.notes() == []
    rng = np.random.default_rng(2)
    B = 5
    Z = mpo.initialize_solution()[None, :] + 0.05 * rng.standard_normal((B, o.n_z))
    w = rng.uniform(0.5, 1.5, (ocp.n_phases, S))
    p = (w / w.sum(1, keepdims=True)).ravel()
    a = o.eval(["f", "g", "grad_f", "jac_g"], Z, p)
    os.environ["MPX_NO_ABSORB"] = "1"  # read at context creation: staging block + unpack pass
    try:
        o2 = mp.mpopt(ocp, S, po, "LGR").create_nlp()[0]["oracle"]
    finally:
        del os.environ["MPX_NO_ABSORB"]
    assert not o2.tile_spans()[1].any()
    b = o2.eval(["f", "g", "grad_f", "jac_g"], Z, p)
    for k in ("f", "g", "grad_f", "jac_g"):
        assert np.array_equal(a[k], b[k]), k
    o.close(), o2.close()
