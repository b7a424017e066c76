"""Moon lander (minimum-fuel soft landing): the getting-started problem of the reference's documentation
(docs/source/notebooks/getting_started.ipynb), solved through ``mpopt_amd.mp`` with every NLP oracle call on the GPU.

    python examples/moon_lander.py [--plot out.png]
"""
import sys

from mpopt_amd import mp

ocp = mp.OCP(n_states=2, n_controls=1)
ocp.dynamics[0] = lambda x, u, t: [x[1], u[0] - 1.5]          # height, velocity; thrust u, lunar gravity 1.5
ocp.running_costs[0] = lambda x, u, t: u[0]                   # fuel
ocp.terminal_constraints[0] = lambda xf, tf, x0, t0: [xf[0], xf[1]]
ocp.x00[0] = [10.0, -2.0]
ocp.lbu[0], ocp.ubu[0] = 0, 3
ocp.lbtf[0], ocp.ubtf[0] = 3, 5
ocp.validate()

mpo, post = mp.solve(ocp, n_segments=20, poly_orders=3, scheme="LGR", plot=False)
x, u, t, _ = post.get_data()
print(f"optimal fuel J = {float(post.solution['f']):.5f}   (documentation: 8.24677)")
print(f"touch-down at t = {t[-1, 0]:.4f}, final state {x[-1]}")
print(f"solver: {mpo.nlp_solver.stats['return_status']}, {mpo.nlp_solver.stats['iter_count']} iterations")
if "--plot" in sys.argv:
    fig, axs = post.plot_phases()
    fig.savefig(sys.argv[sys.argv.index("--plot") + 1])
