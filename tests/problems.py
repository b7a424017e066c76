"""OCP input definitions shared by the golden-vector generator, the oracle tests and the
GPU parity tests.

Every builder takes the *module that provides ``OCP``* (``mp``) and a math namespace ``fn``
(``sqrt/exp/sin/cos/log``), so the very same problem statement can be handed to

* the reference (``/root/reference/mpopt/mpopt.py``, imported by ``tests/golden/make_golden.py``),
* the product (``mpopt_amd``), and
* the oracle (``oracle/``).

The problem statements restate the reference's own inputs:
moon lander  examples/singlephase/moon_lander.py:30-63 (tests/test_mpopt.py:114-144),
Van der Pol  tests/test_mpopt.py:206-227, with parameter + path row examples/singlephase/dae_vdp.py:28-60,
hypersensitive  examples/singlephase/hyper_sensitive.py:31-41,
two-phase Schwartz  tests/test_mpopt.py:165-202 (examples/Multi-phase/tpschwartz.py:30-69),
generic 2-phase fixture  tests/test_mpopt.py:89-110.
``kitchen_sink`` is synthetic: it exercises every feature flag of the transcription at once.
"""
import os

import numpy as np


def moon_lander(mp, fn=None):
    ocp = mp.OCP(n_states=2, n_controls=1)
    ocp.dynamics[0] = lambda x, u, t: [x[1], u[0] - 1.5]
    ocp.running_costs[0] = lambda x, u, t: u[0]
    ocp.terminal_constraints[0] = lambda xf, tf, x0, t0: [xf[0], xf[1]]
    ocp.tf0[0] = 4.0
    ocp.x00[0] = [10.0, -2.0]
    ocp.lbx[0] = [0.0, -20.0]
    ocp.ubx[0] = [20.0, 20.0]
    ocp.lbu[0] = 0
    ocp.ubu[0] = 3
    ocp.lbtf[0], ocp.ubtf[0] = 3, 5
    ocp.validate()
    return ocp


def van_der_pol(mp, fn=None):
    ocp = mp.OCP(n_states=2, n_controls=1)
    ocp.dynamics[0] = lambda x, u, t: [(1 - x[1] * x[1]) * x[0] - x[1] + u[0], x[0]]
    ocp.running_costs[0] = lambda x, u, t: x[0] * x[0] + x[1] * x[1] + u[0] * u[0]
    ocp.x00[0] = [0, 1]
    ocp.lbu[0] = -1.0
    ocp.ubu[0] = 1.0
    ocp.lbx[0][1] = -0.25
    ocp.lbtf[0] = 10.0
    ocp.ubtf[0] = 10.0
    ocp.validate()
    return ocp


def dae_vdp(mp, fn=None):
    ocp = mp.OCP(n_states=2, n_controls=1, n_params=1)
    ocp.dynamics[0] = lambda x, u, t, a: [(1 - x[1] * x[1]) * x[0] - x[1] + u[0], x[0]]
    ocp.running_costs[0] = lambda x, u, t, a: x[0] * x[0] + x[1] * x[1] + u[0] * u[0]
    ocp.path_constraints[0] = lambda x, u, t, a: [a[0] - x[1]]
    ocp.x00[0] = [0, 1]
    ocp.lbu[0] = -1.0
    ocp.ubu[0] = 1.0
    ocp.lba[0] = 0.25
    ocp.uba[0] = 0.5
    ocp.lbx[0][1] = -0.25
    ocp.lbtf[0] = 10.0
    ocp.ubtf[0] = 10.0
    ocp.validate()
    return ocp


def hyper_sensitive(mp, fn=None):
    ocp = mp.OCP(n_states=1, n_controls=1, n_phases=1)
    ocp.dynamics[0] = lambda x, u, t: [-x[0] * x[0] * x[0] + u[0]]
    ocp.running_costs[0] = lambda x, u, t: 0.5 * (x[0] * x[0] + u[0] * u[0])
    ocp.terminal_constraints[0] = lambda xf, tf, x0, t0: [xf[0] - 1.0]
    ocp.x00[0] = 1
    ocp.lbtf[0] = ocp.ubtf[0] = 1000.0
    ocp.scale_t = 1 / 1000.0
    ocp.validate()
    return ocp


def two_phase_schwartz(mp, fn=None):
    ocp = mp.OCP(n_states=2, n_controls=1, n_phases=2)

    def dynamics0(x, u, t):
        return [x[1], u[0] - 0.1 * (1.0 + 2.0 * x[0] * x[0]) * x[1]]

    ocp.dynamics = [dynamics0, dynamics0]

    def path_constraints0(x, u, t):
        return [1.0 - 9.0 * (x[0] - 1) * (x[0] - 1) - (x[1] - 0.4) * (x[1] - 0.4) / (0.3 * 0.3)]

    ocp.path_constraints[0] = path_constraints0
    ocp.terminal_costs[1] = lambda xf, tf, x0, t0: 5 * (xf[0] * xf[0] + xf[1] * xf[1])
    ocp.x00[0] = [1, 1]
    ocp.x00[1] = [1, 1]
    ocp.xf0[0] = [1, 1]
    ocp.xf0[1] = [0, 0]
    ocp.lbx[0][1] = -0.8
    ocp.lbu[0], ocp.ubu[0] = -1, 1
    ocp.lbt0[0], ocp.ubt0[0] = 0, 0
    ocp.lbtf[0], ocp.ubtf[0] = 1, 1
    ocp.lbtf[1], ocp.ubtf[1] = 2.9, 2.9
    ocp.validate()
    return ocp


def analytic_solution(mp, fn=None):
    """Chachuat Ex. 3.10 (reference tests/test_mpopt.py:1090-1112): x(t) = -2t^2 + 6t + 1, u(t) = 2(t - 1)."""
    ocp = mp.OCP(n_states=1, n_controls=1)
    ocp.dynamics[0] = lambda x, u, t: [2 * (1 - u[0])]
    ocp.running_costs[0] = lambda x, u, t: 0.5 * u[0] * u[0] - x[0]
    ocp.x00[0] = [1.0]
    ocp.lbtf[0] = 1.0
    ocp.ubtf[0] = 1.0
    ocp.validate()
    return ocp


def generic_two_phase(mp, fn=None):
    """tests/test_mpopt.py:89-110 plus the optional row blocks switched on."""
    ocp = mp.OCP(n_states=2, n_controls=2, n_phases=2)
    dynamics = lambda x, u, t: [u[0], u[0]]
    path_constraints = lambda x, u, t: [x[0] + 1, u[0]]
    running_costs = lambda x, u, t: u[0]
    terminal_constraints = lambda xf, tf, x0, t0: [-xf[0]]
    terminal_costs = lambda xf, tf, x0, t0: tf
    ocp.dynamics = [dynamics] * ocp.n_phases
    ocp.path_constraints = [path_constraints] * ocp.n_phases
    ocp.running_costs = [running_costs] * ocp.n_phases
    ocp.terminal_constraints = [terminal_constraints] * ocp.n_phases
    ocp.terminal_costs = [terminal_costs] * ocp.n_phases
    for phase in range(ocp.n_phases):
        ocp.lbu[phase], ocp.ubu[phase] = -1.0, 1.0
        ocp.lbtf[phase], ocp.ubtf[phase] = 1.0, 1.0
    ocp.diff_u[0] = 1
    ocp.du_continuity[1] = 1
    ocp.validate()
    return ocp


def kitchen_sink(mp, fn):
    """Synthetic: 3 states, 2 controls, 2 parameters, 2 phases, explicit time dependence,
    transcendental functions, non-unit scaling and every optional constraint block."""
    ocp = mp.OCP(n_states=3, n_controls=2, n_phases=2, n_params=2)

    def dyn0(x, u, t, a):
        return [
            x[1] * fn.cos(x[2]) + a[0] * u[0],
            -x[0] * x[1] + u[1] * fn.exp(-0.1 * t) + a[1],
            u[0] * x[2] / (1.0 + x[0] * x[0]) - 0.3 * t,
        ]

    def dyn1(x, u, t, a):
        return [
            x[1] + 0.5 * t * u[0],
            fn.sin(x[0]) * u[1] - a[0] * x[1] * x[1],
            fn.sqrt(1.0 + x[2] * x[2]) * a[1] - u[0] * u[1],
        ]

    ocp.dynamics = [dyn0, dyn1]
    ocp.path_constraints[0] = lambda x, u, t, a: [x[0] * x[0] + u[0] * u[0] - 4.0 - a[0], x[1] * t - 3.0]
    ocp.path_constraints[1] = lambda x, u, t, a: [u[1] * x[2] - 2.0 - 0.2 * t * a[1]]
    ocp.running_costs[0] = lambda x, u, t, a: u[0] * u[0] + 0.5 * u[1] * u[1] + 0.1 * x[0] * t + a[0] * a[0]
    ocp.running_costs[1] = lambda x, u, t, a: fn.exp(0.1 * x[1]) + u[0] * u[0] * (1.0 + a[1] * a[1])
    ocp.terminal_costs[0] = lambda xf, tf, x0, t0, a: 0.3 * xf[0] * x0[1] + 0.2 * tf * a[0]
    ocp.terminal_costs[1] = lambda xf, tf, x0, t0, a: xf[2] * xf[2] + (tf - t0) * (tf - t0) * 0.05 + a[1] * xf[0]
    ocp.terminal_constraints[0] = lambda xf, tf, x0, t0, a: [xf[1] * tf - x0[0] * a[1]]
    ocp.terminal_constraints[1] = lambda xf, tf, x0, t0, a: [xf[0] - 1.0 + 0.1 * t0, xf[1] * xf[2] - a[0]]
    ocp.scale_x = np.array([0.5, 2.0, 1.25])
    ocp.scale_u = np.array([4.0, 0.25])
    ocp.scale_a = np.array([2.0, 0.5])
    ocp.scale_t = 0.1
    ocp.x00[0] = [1.0, -0.5, 0.2]
    ocp.xf0[0] = [0.5, 0.5, 0.4]
    ocp.x00[1] = [0.5, 0.5, 0.4]
    ocp.xf0[1] = [1.0, 0.0, -0.3]
    ocp.u00[0] = [0.1, -0.2]
    ocp.uf0[0] = [0.3, 0.2]
    ocp.u00[1] = [0.3, 0.2]
    ocp.uf0[1] = [-0.1, 0.1]
    ocp.a0[0] = [0.3, 0.7]
    ocp.a0[1] = [0.6, -0.2]
    ocp.t00[0], ocp.tf0[0] = 0.0, 2.0
    ocp.t00[1], ocp.tf0[1] = 2.0, 5.0
    ocp.lbx[0] = [-5.0, -6.0, -7.0]
    ocp.ubx[0] = [5.0, 6.0, 7.0]
    ocp.lbu[0] = [-2.0, -3.0]
    ocp.ubu[0] = [2.0, 3.0]
    ocp.lbu[1] = [-1.0, -np.inf]
    ocp.ubu[1] = [np.inf, 1.5]
    ocp.lba[0] = [-1.0, -1.0]
    ocp.uba[0] = [1.0, 2.0]
    ocp.lbtf[0], ocp.ubtf[0] = 1.0, 3.0
    ocp.lbt0[1], ocp.ubt0[1] = 1.0, 3.0
    ocp.lbtf[1], ocp.ubtf[1] = 4.0, 6.0
    ocp.lbe[0] = [-0.1, 0.0, 0.0]
    ocp.ube[0] = [0.1, 0.0, 0.2]
    ocp.diff_u[0] = 1
    ocp.diff_u[1] = 1
    ocp.lbdu[1], ocp.ubdu[1] = -7, 9
    ocp.du_continuity[0] = 1
    ocp.du_continuity[1] = 1
    ocp.midu[1] = 0
    ocp.validate()
    return ocp


def time_dependent(mp, fn):
    """Synthetic, single phase: explicit time dependence in dynamics, path row, running cost; terminal cost and constraint in
    (xf, tf, x0, t0, a); one parameter; non-unit scaling everywhere.  Hand-written counterpart with first and second
    derivatives: `time_dependent` in oracle/mpopt_oracle.c (full-size parity of the d/dt chain rule, tests/test_gpu_parity.py)."""
    ocp = mp.OCP(n_states=2, n_controls=1, n_params=1)
    ocp.dynamics[0] = lambda x, u, t, a: [x[1] * fn.cos(0.3 * t) + a[0] * u[0], -x[0] * x[1] + u[0] * fn.exp(-0.1 * t) - 0.3 * t * x[0]]
    ocp.path_constraints[0] = lambda x, u, t, a: [x[0] * t - 3.0 - a[0] * x[1]]
    ocp.running_costs[0] = lambda x, u, t, a: u[0] * u[0] + 0.1 * x[0] * t + a[0] * a[0] * x[1] * x[1]
    ocp.terminal_costs[0] = lambda xf, tf, x0, t0, a: 0.3 * xf[0] * x0[1] + 0.2 * tf * a[0] + 0.05 * (tf - t0) * (tf - t0)
    ocp.terminal_constraints[0] = lambda xf, tf, x0, t0, a: [xf[1] * tf - x0[0] * a[0]]
    ocp.scale_x = np.array([0.5, 2.0])
    ocp.scale_u = np.array([4.0])
    ocp.scale_a = np.array([2.0])
    ocp.scale_t = 0.1
    ocp.x00[0] = [1.0, -0.5]
    ocp.xf0[0] = [0.5, 0.5]
    ocp.u00[0], ocp.uf0[0] = [0.1], [0.3]
    ocp.a0[0] = [0.3]
    ocp.t00[0], ocp.tf0[0] = 0.5, 2.5
    ocp.lbx[0], ocp.ubx[0] = [-5.0, -6.0], [5.0, 6.0]
    ocp.lbu[0], ocp.ubu[0] = [-2.0], [2.0]
    ocp.lba[0], ocp.uba[0] = [-1.0], [1.0]
    ocp.lbt0[0], ocp.ubt0[0] = 0.0, 1.0
    ocp.lbtf[0], ocp.ubtf[0] = 2.0, 3.0
    ocp.validate()
    return ocp


#: name -> (builder, n_segments, poly_orders, scheme).  These are the parity cases for which
#: tests/golden/ holds vectors produced by the reference's own transcription code.
GOLDEN_CASES = {
    "moon_lander_20x3_LGR": (moon_lander, 20, 3, "LGR"),            # BASELINE.json configs[0]
    "moon_lander_10x6_LGR": (moon_lander, 10, 6, "LGR"),            # the grid of the reference's published per-call table (moon_lander.ipynb:171-210)
    "moon_lander_mixed_LGL": (moon_lander, 3, [2, 4, 3], "LGL"),
    "van_der_pol_4x3_CGL": (van_der_pol, 4, 3, "CGL"),
    "dae_vdp_mixed_CGL": (dae_vdp, 3, [3, 6, 3], "CGL"),
    "hyper_sensitive_5x3_LGR": (hyper_sensitive, 5, 3, "LGR"),
    "schwartz_4x3_LGL": (two_phase_schwartz, 4, 3, "LGL"),
    "generic_two_phase_LGR": (generic_two_phase, 2, [2, 3], "LGR"),
    "kitchen_sink_mixed_CGL": (kitchen_sink, 3, [2, 4, 3], "CGL"),
    "kitchen_sink_1x5_LGR": (kitchen_sink, 1, [5], "LGR"),
}


#: mpopt_adaptive (segment widths as decision variables)
ADAPTIVE_CASES = {
    "adaptive_moon_lander_3x2_LGR": (moon_lander, 3, [2, 2, 2], "LGR"),      # the reference's docstring example
    "adaptive_van_der_pol_mixed_CGL": (van_der_pol, 3, [2, 4, 3], "CGL"),
    "adaptive_hyper_sensitive_4x3_LGL": (hyper_sensitive, 4, 3, "LGL"),
    "adaptive_generic_two_phase_LGR": (generic_two_phase, 2, [2, 3], "LGR"),
    "adaptive_kitchen_sink_mixed_LGR": (kitchen_sink, 2, [3, 2], "LGR"),      # time-dependent, 2 phases, parameters
}


#: (builder, n_segments, poly_orders, scheme) of the BASELINE.json configurations at full size
BENCH_CASES = [
    (moon_lander, 1000, 5, "LGR"),                                           # configs[1] (the metric's config)
    (van_der_pol, 2000, [30 if s % 3 == 1 else 3 for s in range(2000)], "CGL"),  # configs[2]
    (two_phase_schwartz, 500, 3, "LGL"),                                     # configs[3]
    (hyper_sensitive, 4000, 3, "LGR"),                                       # configs[4]
]


#: full-size parity cases beyond the BASELINE configurations (tests/test_gpu_parity.py FULL): the configs[2] variant with a
#: parameter column and a path row (examples/singlephase/dae_vdp.py:28-60), and the explicitly time-dependent problem
FULL_EXTRA_CASES = [
    (dae_vdp, 2000, [30 if s % 3 == 1 else 3 for s in range(2000)], "CGL"),
    (time_dependent, 4000, 3, "LGR"),
    (time_dependent, 2000, [30 if s % 3 == 1 else 3 for s in range(2000)], "CGL"),
    (time_dependent, 8000, 3, "LGR"),  # stress: twice config 5's nodes
]


#: seeds of the random mixed-degree grids of tests/test_gpu_parity.py::test_random_mixed_degree_grids (round 2's tools/span_soak.py
#: as a test); __graft_entry__.build() compiles their kernels so that the GPU box finds them in the cache
# The grids the reference PUBLISHES per-call oracle timings for (CasADi's timing table recorded in the
# documentation notebooks, unknown CPU): name -> (builder, n_segments, degree, scheme, C-oracle problem names, scale_t, midu,
# published (nlp_f, nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l) in us per call, citation under /root/reference/docs/source/notebooks/)
PUBLISHED_GRIDS = {
    "moon_lander_10x6_LGR": (moon_lander, 10, 6, "LGR", ["moon_lander"], 1.0, [1], (4.39, 23.46, 6.24, 30.44, 8.44), "moon_lander.ipynb:171-210"),
    "moon_lander_2x30_CGL": (moon_lander, 2, 30, "CGL", ["moon_lander"], 1.0, [1], (4.38, 59.19, 6.20, 88.47, 8.11), "moon_lander.ipynb:280-319"),
    "moon_lander_2x30_LGL": (moon_lander, 2, 30, "LGL", ["moon_lander"], 1.0, [1], (4.22, 60.86, 6.28, 89.37, 8.16), "moon_lander.ipynb:373-412"),
    "hyper_sensitive_5x50_LGR": (hyper_sensitive, 5, 50, "LGR", ["hyper_sensitive"], 1e-3, [0], (11.35, 113.76, 23.05, 196.68, 49.66), "hypersensitive.ipynb:165-204"),
    "hyper_sensitive_5x50_CGL": (hyper_sensitive, 5, 50, "CGL", ["hyper_sensitive"], 1e-3, [0], (11.31, 113.88, 23.22, 201.06, 51.43), "hypersensitive.ipynb:267-306"),
    "hyper_sensitive_5x50_LGL": (hyper_sensitive, 5, 50, "LGL", ["hyper_sensitive"], 1e-3, [0], (11.98, 124.86, 23.43, 201.54, 50.91), "hypersensitive.ipynb:360-399"),
    "van_der_pol_1x25_LGR": (van_der_pol, 1, 25, "LGR", ["van_der_pol"], 1.0, [1], (4.39, 26.99, 5.92, 39.66, 10.50), "vanderpol.ipynb:177-216"),
    "schwartz_1x20_LGR": (two_phase_schwartz, 1, 20, "LGR", ["schwartz_phase0", "schwartz_phase1"], 1.0, [1, 0], (3.22, 30.29, 4.03, 46.51, 13.69), "twophaseschwartz.ipynb:195-234"),
}

# degrees above the LDS tables (round 6: streamed tables, mpx_kernels.h TAB_GLB; tests/test_gpu_high_degree.py)
HIGH_DEGREE_CASES = {
    "moon_lander_1x100_LGR": (moon_lander, 1, 100, "LGR"),     # the reference's documented grid (docs/source/notebooks/getting_started.ipynb:743)
    "van_der_pol_2x69_CGL": (van_der_pol, 2, 69, "CGL"),       # first streamed degree
    "dae_vdp_3_100_3_LGL": (dae_vdp, 3, [3, 100, 3], "LGL"),   # mixed: a streamed bucket between register-table buckets, parameter + path row
    "hyper_sensitive_4x128_LGR": (hyper_sensitive, 4, 128, "LGR"),  # two segments per tile, tile boundary inside the phase
    "kitchen_sink_95_71": (kitchen_sink, 2, [95, 71], "LGR"),  # two phases, control-slope rows (D.U), time dependence; P + 1 = 96 = 0 mod 4
    "moon_lander_1x255_CGL": (moon_lander, 1, 255, "CGL"),     # the largest degree the library takes
}

SOAK_SEEDS = [int(x) for x in os.environ["MPX_SOAK_SEEDS"].split(",")] if os.environ.get("MPX_SOAK_SEEDS") else [3, 11, 12, 22, 29, 40]  # (env: one-off wider soaks)


def soak_case(seed):
    """(builder, n_segments, poly_orders, scheme) of a random mixed-degree grid: 2-3 distinct degrees in runs of random length."""
    rng = np.random.default_rng(seed)
    degs = rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 13, 16, 20, 30], size=int(rng.integers(2, 4)), replace=False)
    S = int(rng.integers(2, 400))
    runs = rng.integers(1, int(rng.integers(2, 40)), size=S)
    po = np.repeat(rng.choice(degs, size=S), runs)[:S].tolist()
    builder = [van_der_pol, dae_vdp, kitchen_sink, two_phase_schwartz, hyper_sensitive][seed % 5]
    return builder, S, po, ["LGR", "LGL", "CGL"][seed % 3]


def lane_soak_case(seed):
    """(builder, n_segments, poly_orders, scheme) of a random small mixed-degree mpopt_adaptive grid with time-independent dynamics
    (the problems whose point tasks fall into groups: mpopt_amd/assembly_lanes.py)."""
    rng = np.random.default_rng(seed)
    builder = [van_der_pol, dae_vdp, two_phase_schwartz, hyper_sensitive, moon_lander, generic_two_phase][seed % 6]
    S = int(rng.integers(4, 26))
    return builder, S, [int(x) for x in rng.integers(1, 7, size=S)], ["LGR", "LGL", "CGL"][seed % 3]


def sample_point(name, n_z, n_p, n_g, z0, lbx, ubx):
    """Deterministic evaluation point: Z0 + seeded perturbation clipped
    to the bounds, non-uniform positive widths summing to one per phase, N(0,1) multipliers."""
    seed = 20260928 + sum(ord(c) for c in name)
    rng = np.random.default_rng(seed)
    xi = rng.uniform(-1, 1, n_z)
    xi2 = rng.uniform(-1, 1, n_z)
    z = z0 + 0.05 * np.abs(z0) * xi + 0.01 * xi2
    z = np.minimum(np.maximum(z, lbx), ubx)
    lam = rng.standard_normal(n_g)
    sigma = 0.75
    return z, lam, sigma, rng


def sample_widths(rng, n_segments, n_phases):
    w = rng.uniform(0.5, 1.5, (n_phases, n_segments))
    w = w / w.sum(axis=1, keepdims=True)
    return w.reshape(-1)


def ascent_numpy_style(mp, fn=None, use_numpy=True):
    """Point mass in a central gravity field with exponential-atmosphere drag, written the way the reference's
    launch-vehicle examples are (examples/Multi-phase/multistage_launch_vehicle.py): numpy functions applied to the
    symbols (np.sqrt, np.dot, np.exp, np.cos, np.sin, arrays of symbols).  ``use_numpy=False`` writes the same model with the
    ``fn`` namespace, for comparison."""
    ocp = mp.OCP(n_states=7, n_controls=3, n_phases=1)
    mu, Re, h0, cd_a, thrust, isp_g0 = 1.0, 1.0, 0.05, 0.3, 0.6, 2.5
    sqrt, exp, cos, sin = (np.sqrt, np.exp, np.cos, np.sin) if use_numpy else (fn.sqrt, fn.exp, fn.cos, fn.sin)

    def dynamics(x, u, t):
        r, v, m = x[0:3], x[3:6], x[6]
        if use_numpy:  # slices of the state vector are vectors: scalar * r, r + v, np.dot(r, r) work as on CasADi columns
            uu = u
            rn = sqrt(np.dot(r, r))
            vn = sqrt(np.dot(v, v) + 1e-6)
            drag = -0.5 * cd_a * exp(-(rn - Re) / h0) * vn * v
            acc = -mu / (rn * rn * rn) * r + (thrust * uu + drag) / m
            acc = list(acc)
        else:
            rn = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
            vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + 1e-6)
            k = -0.5 * cd_a * exp(-(rn - Re) / h0) * vn
            acc = [-mu / (rn * rn * rn) * r[i] + (thrust * u[i] + k * v[i]) / m for i in range(3)]
        return [v[0], v[1], v[2], acc[0], acc[1], acc[2], -thrust / isp_g0 * (1.0 + 0.1 * cos(t) * sin(t))]

    ocp.dynamics[0] = dynamics
    ocp.path_constraints[0] = lambda x, u, t: [u[0] * u[0] + u[1] * u[1] + u[2] * u[2] - 1.0]
    ocp.running_costs[0] = lambda x, u, t: 0.01 * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2])
    ocp.terminal_costs[0] = lambda xf, tf, x0, t0: -xf[6]
    ocp.terminal_constraints[0] = (lambda xf, tf, x0, t0: [sqrt(xf[0] * xf[0] + xf[1] * xf[1] + xf[2] * xf[2]) - 1.2])
    ocp.x00[0] = [1.0, 0.0, 0.0, 0.0, 0.3, 0.1, 1.0]
    ocp.xf0[0] = [0.9, 0.7, 0.2, -0.3, 0.6, 0.1, 0.6]
    ocp.u00[0], ocp.uf0[0] = [0.2, 0.9, 0.1], [0.0, 1.0, 0.0]
    ocp.lbu[0], ocp.ubu[0] = [-1, -1, -1], [1, 1, 1]
    ocp.lbx[0][6] = 0.1
    ocp.lbtf[0], ocp.ubtf[0] = 0.5, 2.0
    ocp.tf0[0] = 1.0
    return ocp


def staged_ascent(mp, fn):
    """Two-stage ascent written with the constructs of the reference's flagship example (examples/Multi-phase/
    multistage_launch_vehicle.py:70-150): slices of the state (``x[:3]``), ``vertcat`` of symbols, scalar * vector products
    indexed afterwards, one dynamics function specialised per phase through default arguments, ``xf[-1]``."""
    ocp = mp.OCP(n_states=7, n_controls=3, n_phases=2)
    mu, Re, omega, rho0, h_scale, sa_cd = 1.0, 1.0, 0.07, 1.2, 0.08, 0.4
    thrust, mdot = [0.9, 0.5], [0.25, 0.12]

    def dynamics(x, u, t, param=0, T=0.0, md=0.0):
        r, v, m = x[:3], x[3:6], x[6]
        r_mag = fn.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
        v_rel = fn.vertcat(v[0] + r[1] * omega, v[1] - r[0] * omega, v[2])
        v_rel_mag = fn.sqrt(v_rel[0] * v_rel[0] + v_rel[1] * v_rel[1] + v_rel[2] * v_rel[2] + 1e-8)
        rho = rho0 * fn.exp(-(r_mag - Re) / h_scale)
        D = -rho / (2 * m) * sa_cd * v_rel_mag * v_rel
        g = -mu / (r_mag * r_mag * r_mag) * r
        return [x[3], x[4], x[5], T / m * u[0] + param * D[0] + g[0], T / m * u[1] + param * D[1] + g[1],
                T / m * u[2] + param * D[2] + g[2], -md]

    ocp.dynamics = [lambda x, u, t: dynamics(x, u, t, param=1, T=thrust[0], md=mdot[0]),
                    lambda x, u, t: dynamics(x, u, t, param=1, T=thrust[1], md=mdot[1])]
    path = lambda x, u, t: [u[0] * u[0] + u[1] * u[1] + u[2] * u[2] - 1, -u[0] * u[0] - u[1] * u[1] - u[2] * u[2] + 1,
                            -fn.sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) / Re + 1]
    ocp.path_constraints = [path] * 2
    ocp.terminal_costs[1] = lambda xf, tf, x0, t0: -xf[-1]

    def terminal_constraints1(x, t, x0, t0):
        h = fn.vertcat(x[1] * x[5] - x[4] * x[2], x[3] * x[2] - x[0] * x[5], x[0] * x[4] - x[1] * x[3])  # r x v
        return [h[0] * h[0] + h[1] * h[1] + h[2] * h[2] - 1.3, x[0] * x[3] + x[1] * x[4] + x[2] * x[5]]

    ocp.terminal_constraints[1] = terminal_constraints1
    ocp.x00 = np.array([[1.0, 0.0, 0.02, 0.0, 0.07, 0.0, 1.0], [0.95, 0.4, 0.05, -0.3, 0.8, 0.05, 0.55]])
    ocp.xf0 = np.array([[0.95, 0.4, 0.05, -0.3, 0.8, 0.05, 0.6], [0.2, 1.1, 0.1, -1.0, 0.2, 0.0, 0.3]])
    ocp.u00 = np.array([[1, 0, 0], [0, 1, 0]])
    ocp.uf0 = np.array([[0, 1, 0], [0, 1, 0]])
    ocp.t00, ocp.tf0 = np.array([[0.0], [1.5]]), np.array([[1.5], [3.5]])
    ocp.lbu, ocp.ubu = np.array([[-1.0] * 3] * 2), np.array([[1.0] * 3] * 2)
    ocp.lbtf, ocp.ubtf = np.array([[1.5], [3.0]]), np.array([[1.5], [4.0]])
    ocp.lbt0, ocp.ubt0 = np.array([[0.0], [1.5]]), np.array([[0.0], [1.5]])
    ocp.lbe, ocp.ube = np.array([[0, 0, 0, 0, 0, 0, -0.05]]), np.array([[0, 0, 0, 0, 0, 0, -0.05]])  # stage mass drop
    ocp.scale_x = np.array([1.0, 1.0, 1.0, 2.0, 2.0, 2.0, 1.5])
    ocp.validate()
    return ocp


# ---- random OCPs (round 6): the user callables themselves drawn at random -----------------------------------------------------------
# Every other problem here is hand-written; the product's tracer, symbolic differentiation and code generator (mpopt_amd/expr.py,
# codegen.py) take the place of CasADi's SX graph + AD behind `ca.nlpsol` (mpopt.py:757), and the numpy / sympy oracle differentiates the
# same callables by another route (sympy).  A seed draws: phases, states, controls, parameters, every optional row block, the scalings,
# the callables as random expression trees over (x, u, t, a) / (xf, tf, x0, t0, a), and a grid.
RANDOM_OCP_SEEDS = [int(x) for x in os.environ["MPX_RANDOM_OCP_SEEDS"].split(",")] if os.environ.get("MPX_RANDOM_OCP_SEEDS") else list(range(1, 13))
RANDOM_OCP_WIDE_SEEDS = [int(x) for x in os.environ["MPX_RANDOM_OCP_WIDE_SEEDS"].split(",")] if os.environ.get("MPX_RANDOM_OCP_WIDE_SEEDS") else list(range(1, 9))
#: (seed, wide) of the random OCPs of the suite; wide = callables drawn from the whole math namespace (tan, asin, acos, sinh, cosh, atan2, fabs, fmax, fmin, sign, x ** y)
RANDOM_OCPS = [(s, False) for s in RANDOM_OCP_SEEDS] + [(s, True) for s in RANDOM_OCP_WIDE_SEEDS]


_TREE_OPS = ["add", "sub", "mul", "mul", "div1", "sin", "cos", "exp", "sqrt1", "tanh", "log2", "pow2", "pow3", "atan", "powf", "scale"]
# the rest of the math namespace (mpopt_amd/expr.py: _Math), arguments mapped into each function's domain; the piecewise ones
# (fabs, fmax, fmin, sign) are differentiable at the random evaluation points with probability one
_TREE_OPS_WIDE = _TREE_OPS + ["tan", "asin", "acos", "sinh", "cosh", "atan2", "fabs", "fmax", "fmin", "sign", "powg", "rdiv", "neg"]
_TREE_BINARY = ("add", "sub", "mul", "div1", "atan2", "fmax", "fmin", "powg")


def _random_tree(rng, leaves, depth, ops=_TREE_OPS):
    """A random expression tree (nested tuples) over `leaves`; every operation is smooth and bounded-ish on bounded arguments."""
    if depth == 0 or rng.random() < 0.15:
        if rng.random() < 0.2:
            return ("const", float(np.round(rng.uniform(-2, 2), 3)))
        return ("leaf", leaves[int(rng.integers(len(leaves)))])
    op = ops[int(rng.integers(len(ops)))]
    if op in _TREE_BINARY:
        return (op, _random_tree(rng, leaves, depth - 1, ops), _random_tree(rng, leaves, depth - 1, ops))
    if op == "scale":
        return (op, float(np.round(rng.uniform(-1.5, 1.5), 3)), _random_tree(rng, leaves, depth - 1, ops))
    return (op, _random_tree(rng, leaves, depth - 1, ops))


def _eval_tree(tr, env, fn):
    op = tr[0]
    if op == "const":
        return tr[1]
    if op == "leaf":
        return env[tr[1]]
    if op == "scale":
        return tr[1] * _eval_tree(tr[2], env, fn)
    a = _eval_tree(tr[1], env, fn)
    if op in _TREE_BINARY:
        b = _eval_tree(tr[2], env, fn)
        if op == "atan2":
            return fn.atan2(a, 1.5 + b * b)
        if op == "fmax":
            return fn.fmax(a, b)
        if op == "fmin":
            return fn.fmin(a, 0.5 * b + 0.1)
        if op == "powg":
            return fn.power(1.5 + a * a, 0.3 * b)
        return a + b if op == "add" else a - b if op == "sub" else a * b if op == "mul" else a / (1.5 + b * b)
    if op == "tan":
        return fn.tan(0.4 * fn.tanh(a))
    if op == "asin":
        return fn.asin(0.9 * fn.tanh(a))
    if op == "acos":
        return fn.acos(0.9 * fn.sin(a))
    if op == "sinh":
        return fn.sinh(0.3 * a)
    if op == "cosh":
        return fn.cosh(0.3 * a)
    if op == "fabs":
        return fn.fabs(a - 0.05)
    if op == "sign":
        return fn.sign(a - 0.05) * a
    if op == "rdiv":
        return 1.0 / (2.0 + a * a)
    if op == "neg":
        return -a
    if op == "sin":
        return fn.sin(a)
    if op == "cos":
        return fn.cos(a)
    if op == "exp":
        return fn.exp(0.2 * a)
    if op == "sqrt1":
        return fn.sqrt(1.0 + a * a)
    if op == "tanh":
        return fn.tanh(a)
    if op == "log2":
        return fn.log(2.0 + a * a)
    if op == "pow2":
        return a * a
    if op == "pow3":
        return a ** 3
    if op == "atan":
        return fn.arctan(a)
    if op == "powf":
        return (1.5 + a * a) ** 0.7
    raise ValueError(op)


def random_ocp_case(seed, wide=False):
    """(builder, n_segments, poly_orders, scheme) of the random OCP of `seed`; wide: the callables draw from the whole math namespace."""
    g = np.random.default_rng((5000 if wide else 1000) + seed)
    ops = _TREE_OPS_WIDE if wide else _TREE_OPS
    nph, nx, nu, na = int(g.integers(1, 3)), int(g.integers(1, 5)), int(g.integers(1, 3)), int(g.integers(0, 3))
    timed = bool(g.random() < 0.6)
    node_leaves = [("x", i) for i in range(nx)] + [("u", i) for i in range(nu)] + [("a", i) for i in range(na)] + ([("t", 0)] * 2 if timed else [])
    term_leaves = [("xf", i) for i in range(nx)] + [("x0", i) for i in range(nx)] + [("tf", 0), ("t0", 0)] + [("a", i) for i in range(na)]
    spec = []
    for ph in range(nph):
        spec.append(dict(
            dyn=[_random_tree(g, node_leaves, int(g.integers(1, 4)), ops) for _ in range(nx)],
            path=[_random_tree(g, node_leaves, int(g.integers(1, 3)), ops) for _ in range(int(g.integers(0, 3)))],
            run=_random_tree(g, node_leaves, int(g.integers(1, 4)), ops),
            tcost=_random_tree(g, term_leaves, int(g.integers(0, 3)), ops),
            tcon=[_random_tree(g, term_leaves, int(g.integers(1, 3)), ops) for _ in range(int(g.integers(0, 3)))],
        ))
    sx, su, sa, st = g.choice([0.5, 1.0, 2.0, 1.25], nx), g.choice([0.25, 1.0, 4.0], nu), g.choice([0.5, 1.0, 2.0], max(na, 1))[:na], float(g.choice([0.1, 1.0, 2.0]))
    guess = dict(x00=g.uniform(-1, 1, (nph, nx)), xf0=g.uniform(-1, 1, (nph, nx)), u00=g.uniform(-1, 1, (nph, nu)), uf0=g.uniform(-1, 1, (nph, nu)),
                 a0=g.uniform(-1, 1, (nph, max(na, 1)))[:, :na], t=np.cumsum(g.uniform(0.5, 2.0, nph + 1)))
    flags = dict(diff_u=g.integers(0, 2, nph), du_cont=g.integers(0, 2, nph), midu=g.integers(0, 2, nph), ubu_inf=g.random(nph) < 0.3)
    S = int(g.integers(1, 7))
    degs = g.choice([2, 3, 4, 5, 7, 9, 13, 16, 21], size=int(g.integers(1, 4)), replace=False)
    po = [int(x) for x in g.choice(degs, size=S)]
    scheme = ["LGR", "LGL", "CGL"][int(g.integers(3))]

    def builder(mp, fn):
        ocp = mp.OCP(n_states=nx, n_controls=nu, n_phases=nph, n_params=na)

        def node_env(x, u, t, a):
            env = {("x", i): x[i] for i in range(nx)}
            env.update({("u", i): u[i] for i in range(nu)})
            env.update({("a", i): a[i] for i in range(na)})
            env[("t", 0)] = t
            return env

        def term_env(xf, tf, x0, t0, a):
            env = {("xf", i): xf[i] for i in range(nx)}
            env.update({("x0", i): x0[i] for i in range(nx)})
            env.update({("a", i): a[i] for i in range(na)})
            env[("tf", 0)], env[("t0", 0)] = tf, t0
            return env

        def node_fn(trees, scalar=False):
            if na:
                f = lambda x, u, t, a: [_eval_tree(tr, node_env(x, u, t, a), fn) for tr in trees]
                return (lambda x, u, t, a: f(x, u, t, a)[0]) if scalar else f
            f = lambda x, u, t: [_eval_tree(tr, node_env(x, u, t, None), fn) for tr in trees]
            return (lambda x, u, t: f(x, u, t)[0]) if scalar else f

        def term_fn(trees, scalar=False):
            if na:
                f = lambda xf, tf, x0, t0, a: [_eval_tree(tr, term_env(xf, tf, x0, t0, a), fn) for tr in trees]
                return (lambda xf, tf, x0, t0, a: f(xf, tf, x0, t0, a)[0]) if scalar else f
            f = lambda xf, tf, x0, t0: [_eval_tree(tr, term_env(xf, tf, x0, t0, None), fn) for tr in trees]
            return (lambda xf, tf, x0, t0: f(xf, tf, x0, t0)[0]) if scalar else f

        for ph in range(nph):
            sp_ = spec[ph]
            ocp.dynamics[ph] = node_fn(sp_["dyn"])
            if sp_["path"]:
                ocp.path_constraints[ph] = node_fn(sp_["path"])
            ocp.running_costs[ph] = node_fn([sp_["run"]], scalar=True)
            ocp.terminal_costs[ph] = term_fn([sp_["tcost"]], scalar=True)
            if sp_["tcon"]:
                ocp.terminal_constraints[ph] = term_fn(sp_["tcon"])
            ocp.x00[ph], ocp.xf0[ph] = guess["x00"][ph].tolist(), guess["xf0"][ph].tolist()
            ocp.u00[ph], ocp.uf0[ph] = guess["u00"][ph].tolist(), guess["uf0"][ph].tolist()
            if na:
                ocp.a0[ph] = guess["a0"][ph].tolist()
                ocp.lba[ph], ocp.uba[ph] = [-2.0] * na, [2.0] * na
            ocp.t00[ph], ocp.tf0[ph] = float(guess["t"][ph]), float(guess["t"][ph + 1])
            ocp.lbx[ph], ocp.ubx[ph] = [-3.0] * nx, [3.0] * nx
            ocp.lbu[ph] = [-2.0] * nu
            ocp.ubu[ph] = [np.inf if flags["ubu_inf"][ph] else 2.0] * nu
            ocp.lbtf[ph], ocp.ubtf[ph] = float(guess["t"][ph + 1]) - 0.4, float(guess["t"][ph + 1]) + 0.4
            if ph:
                ocp.lbt0[ph], ocp.ubt0[ph] = float(guess["t"][ph]) - 0.4, float(guess["t"][ph]) + 0.4
            ocp.diff_u[ph], ocp.du_continuity[ph], ocp.midu[ph] = int(flags["diff_u"][ph]), int(flags["du_cont"][ph]), int(flags["midu"][ph])
        ocp.scale_x, ocp.scale_u, ocp.scale_t = np.array(sx, float), np.array(su, float), st
        if na:
            ocp.scale_a = np.array(sa, float)
        ocp.validate()
        return ocp

    return builder, S, po, scheme
