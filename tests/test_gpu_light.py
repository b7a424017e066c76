"""Light passes (f, g, grad_f without the Jacobian values) of grids with a high degree: the mpx_light_* kernels (contractions on the
matrix cores, span-coalesced I/O; mpx_kernels.h: light_body) against the node kernels they replace for these passes
(MPX_NO_LIGHT=1): g and the node entries of grad_f bit for bit, f and the (t0, tf, a) entries of grad_f to rounding; against the
numpy oracle; batch-split invariance.  What nlp_g / nlp_f compute: reference mpopt.py:227-232, 455."""
import os

import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp
import problems
from helpers import assert_entries, border_columns

mixed = lambda S: [30 if s % 3 == 1 else 3 for s in range(S)]
CASES = {
    "config3_pattern_48": (problems.van_der_pol, 48, mixed(48), "CGL"),          # 16 high-degree segments: one full group
    "config3_pattern_100": (problems.van_der_pol, 100, mixed(100), "CGL"),       # 33 of them: ragged last group
    "high_first_and_last": (problems.van_der_pol, 9, [30, 3, 3, 30, 5, 30, 3, 2, 30], "LGL"),  # node 0 on the matrix core; ragged degrees
    "dae_vdp_9x17": (problems.dae_vdp, 9, 17, "LGL"),                            # regular grid, path row, parameter: 4 rows of g per node
    "dae_vdp_40x13": (problems.dae_vdp, 40, 13, "LGR"),                          # lowest degree of the scheme, several groups
    "vdp_3x31": (problems.van_der_pol, 3, 31, "CGL"),                            # highest degree: both M tiles full
    "kitchen_sink_mixed": (problems.kitchen_sink, 12, [20, 3, 20, 5] * 3, "LGR"),  # two phases, DU rows, time dependence, parameters
    "time_dependent_mixed": (problems.time_dependent, 30, [3, 16, 4] * 10, "CGL"),
}
NO_PLAN = {
    "two_high_degrees": (problems.van_der_pol, 8, [20, 16] * 4, "CGL"),
    "low_degrees_only": (problems.van_der_pol, 8, [3, 5] * 4, "CGL"),
    "degree_above_31": (problems.van_der_pol, 3, [40, 3, 3], "CGL"),
}
# single-degree grids of degree <= 12: the mpx_lightlow_* kernels (spans of 64 * CHL consecutive nodes per wavefront)
LOW_CASES = {
    "moon_lander_20x3": (problems.moon_lander, 20, 3, "LGR"),                     # one short span
    "moon_lander_300x5": (problems.moon_lander, 300, 5, "LGR"),                   # three spans, segments straddle the span ends
    "schwartz_2x200x3": (problems.two_phase_schwartz, 200, 3, "LGL"),             # two phases
    "kitchen_sink_40x5": (problems.kitchen_sink, 40, 5, "LGR"),                   # five inputs per node: shorter spans; DU rows, parameters
    "dae_vdp_37x12": (problems.dae_vdp, 37, 12, "CGL"),                           # highest degree of the family
    "time_dependent_400x3": (problems.time_dependent, 400, 3, "LGR"),
    "hyper_sensitive_700x1": (problems.hyper_sensitive, 700, 1, "LGR"),           # degree 1: every node its own segment
    "van_der_pol_513x2": (problems.van_der_pol, 513, 2, "CGL"),                   # N = 1027: the last span holds three nodes
}


def build(case):
    builder, S, po, scheme = case
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    return ocp, mpo, mpo.create_nlp()[0]["oracle"]


@pytest.mark.parametrize("name", list(LOW_CASES))
def test_low_degree_plan_structure(name):
    """Single-degree grids of degree <= 12: spans of 64 * CHL nodes (CHL from the LDS budget of the span rows), one partial-sum
    slot per span -- never more spans than tiles in a phase."""
    builder, S, po, scheme = LOW_CASES[name]
    ocp = builder(mp, M.math)
    o = M.NlpFunctions(ocp, S, [po] * S, scheme, with_device=False)
    deg, n_groups, span, n_low = o.light_plan()
    nin = ocp.nx + ocp.nu
    chl = min(12, max(1, (53248 // (32 * nin) - 2 * po - 8) // 64))
    assert deg == po and n_low == 0 and span == (64 * chl + 2 * po + 8 + 1) // 2 * 2
    assert n_groups == -(-o.n_nodes // (64 * chl))
    o.close()


@pytest.mark.parametrize("name", list(CASES) + list(NO_PLAN))
def test_light_plan_structure(name):
    """The plan is host arithmetic: it exists exactly for grids with one degree in 13..31 and otherwise degrees <= 12."""
    builder, S, po, scheme = (CASES.get(name) or NO_PLAN[name])
    ocp = builder(mp, M.math)
    orders = [po] * S if isinstance(po, int) else list(po)
    o = M.NlpFunctions(ocp, S, orders, scheme, with_device=False)
    deg, n_groups, span, n_low = o.light_plan()
    if name in NO_PLAN:
        assert (deg, n_groups) == (0, 0)
    else:
        high = [d for d in orders if d > 12]
        assert deg == high[0] and n_groups >= -(-len(high) // 16) and span <= 640
        assert n_low == sum(d for d in orders if d <= 12) + (1 if orders[0] <= 12 else 0)
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES) + list(LOW_CASES))
def test_light_kernels_against_node_kernels_and_oracle(name):
    from oracle.mpopt_oracle import OracleNLP

    case = CASES.get(name) or LOW_CASES[name]
    ocp, mpo, o = build(case)
    assert (o.light_plan()[0] > 12) == (name in CASES) and o.light_plan()[1] > 0
    rng = np.random.default_rng(3)
    node = np.ones(o.n_z, bool)
    node[border_columns(o)] = False
    z0 = mpo.initialize_solution()
    for B in (1, 16, 37):
        Z = z0[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))) + 0.02 * rng.uniform(-1, 1, (B, o.n_z))
        w = rng.uniform(0.5, 1.5, (ocp.n_phases, o.n_segments))
        p = (w / w.sum(1, keepdims=True)).ravel()
        masks = (["f", "g"], ["g"], ["f"], ["f", "grad_f"], ["f", "g", "grad_f"], ["grad_f"])
        light = [o.eval(m, Z, p) for m in masks]
        os.environ["MPX_NO_LIGHT"] = "1"
        try:
            heavy = [o.eval(m, Z, p) for m in masks]
        finally:
            del os.environ["MPX_NO_LIGHT"]
        for m, a, b in zip(masks, light, heavy):
            if "g" in m:
                assert np.array_equal(a["g"], b["g"]), (name, B, m)
            if "grad_f" in m:
                assert np.array_equal(a["grad_f"][:, node], b["grad_f"][:, node]), (name, B, m)
                assert_entries(a["grad_f"][:, ~node], b["grad_f"][:, ~node], 1e-12, what=f"{name} B={B} grad_f (t0, tf, a) entries, light vs node kernels")
            if "f" in m:
                assert np.abs(a["f"] - b["f"]).max() <= 1e-13 * max(1.0, np.abs(b["f"]).max()), (name, B, m)
                assert np.array_equal(a["f"], light[0]["f"]), (name, B, m)  # every light pass sums f in the same order
        # batch-split invariance of the light passes: a batch equals its single evaluations bit for bit
        for b_ in (0, B - 1):
            r1 = o.eval(["f", "g", "grad_f"], Z[b_], p)
            assert r1["f"] == light[4]["f"][b_] and np.array_equal(r1["g"], light[4]["g"][b_]) and np.array_equal(r1["grad_f"], light[4]["grad_f"][b_])
        # per-point widths
        P2 = np.stack([np.roll(p.reshape(ocp.n_phases, -1), b_, axis=1).ravel() for b_ in range(B)])
        lp = o.eval(["f", "g"], Z, P2)
        r1 = o.eval(["f", "g"], Z[B - 1], P2[B - 1])
        assert r1["f"] == lp["f"][B - 1] and np.array_equal(r1["g"], lp["g"][B - 1])
    builder, S, po, scheme = case
    O = OracleNLP(ocp, S, po, scheme)
    r = o.eval(["f", "g", "grad_f"], Z[0], p)
    assert abs(r["f"] - O.f(Z[0], p)) <= 1e-10 * max(1.0, abs(O.f(Z[0], p)))
    go = O.g(Z[0], p)
    assert np.abs(r["g"] - go).max() <= 1e-10 * max(1.0, np.abs(go).max())
    assert_entries(r["grad_f"], O.grad_f(Z[0], p), 1e-10, what=f"{name} grad_f (light) vs numpy oracle")
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(LOW_CASES))
def test_short_and_long_spans_give_the_same_bits(name, monkeypatch):
    """Small batches run one 64-node chunk per wavefront (mpx_lightlows_*), large ones spans of up to 12 chunks (mpx_lightlow_*); the
    partial-sum slots are per chunk in both, so f and the (t0, tf, a) entries of grad_f -- the sums -- agree bit for bit: a single
    evaluation with the long spans forced (MPX_LIGHT_LONG_SPANS=1), and a point inside a batch large enough to take them by itself."""
    ocp, mpo, o = build(LOW_CASES[name])
    rng = np.random.default_rng(9)
    n_groups = o.light_plan()[1]
    B = -(-1100 // n_groups)  # (4 * 256 compute units / n_groups is where the library switches)
    z0 = mpo.initialize_solution()
    Z = z0[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))) + 0.02 * rng.uniform(-1, 1, (B, o.n_z))
    w = rng.uniform(0.5, 1.5, (ocp.n_phases, o.n_segments))
    p = (w / w.sum(1, keepdims=True)).ravel()
    for mask in (["f"], ["f", "g"], ["f", "grad_f"], ["f", "g", "grad_f"]):
        big = o.eval(mask, Z, p)
        for b_ in (0, B // 2, B - 1):
            one = o.eval(mask, Z[b_], p)
            monkeypatch.setenv("MPX_LIGHT_LONG_SPANS", "1")
            forced = o.eval(mask, Z[b_], p)
            monkeypatch.delenv("MPX_LIGHT_LONG_SPANS")
            for k in mask:
                assert np.array_equal(np.asarray(one[k]), np.asarray(big[k][b_])), (name, mask, k, b_)
                assert np.array_equal(np.asarray(one[k]), np.asarray(forced[k])), (name, mask, k, b_, "forced long spans")
    o.close()


@pytest.mark.gpu
def test_light_pass_of_a_single_degree_grid_on_poisoned_device_memory():
    """Round 6 (found by a soak of random grids, one placement in some thousand): a group of light_body without low-degree nodes -- every
    group of a single-degree grid of degree 13 ... 31 -- fetched the descriptor `foreign[f_first]` all the same and took a segment index
    from it for its width loads; the list was empty and its one-element allocation uninitialised, so the index was whatever the memory
    held: a memory access fault when it was large.  The list now ends with a zero descriptor and empty uploads are zeroed.  Here the
    device allocator's small blocks are filled with 0x7f bytes and freed before the context is created."""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes, hip.hipMemset.argtypes, hip.hipFree.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t], [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t], [ctypes.c_void_p]
    blocks = []
    for size in (8, 16, 24, 32, 48, 64, 128, 256, 512, 1024, 4096) * 24:
        ptr = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(ptr), size) == 0
        assert hip.hipMemset(ptr, 0x7F, size) == 0
        blocks.append(ptr)
    assert hip.hipDeviceSynchronize() == 0
    for ptr in blocks:
        assert hip.hipFree(ptr) == 0
    ocp = problems.van_der_pol(mp, M.math)
    S, P = 98, 20
    mpo = mp.mpopt(ocp, S, P, "CGL")
    o = mpo.create_nlp()[0]["oracle"]
    assert o.light_plan()[1] > 0
    rng = np.random.default_rng(50)
    z = mpo.initialize_solution() + 0.05 * rng.standard_normal(o.n_z)
    w = rng.uniform(0.4, 1.6, S)
    p = w / w.sum()
    for B in (1, 5):
        Z = np.stack([z] * B)
        light = o.eval(["f", "g"], Z if B > 1 else z, p)
        full = o.eval(["f", "g", "grad_f", "jac_g"], Z if B > 1 else z, p)
        assert np.array_equal(np.asarray(light["g"]), np.asarray(full["g"]))
    from oracle.mpopt_oracle import OracleNLP

    go = OracleNLP(ocp, S, P, "CGL").g(z, p)
    assert np.abs(np.asarray(light["g"])[0] - go).max() < 1e-10 * max(1.0, np.abs(go).max())
    o.close()
