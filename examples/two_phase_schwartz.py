"""Two-phase Schwartz problem: a path constraint in the first phase only, phases linked by continuity events."""
from mpopt_amd import mp

ocp = mp.OCP(n_states=2, n_controls=1, n_phases=2)
dyn = lambda x, u, t: [x[1], u[0] - 0.1 * (1.0 + 2.0 * x[0] * x[0]) * x[1]]
ocp.dynamics = [dyn, dyn]
ocp.path_constraints[0] = lambda x, u, t: [1.0 - 9.0 * (x[0] - 1) ** 2 - (x[1] - 0.4) ** 2 / 0.09]
ocp.terminal_costs[1] = lambda xf, tf, x0, t0: 5 * (xf[0] * xf[0] + xf[1] * xf[1])
ocp.x00[0] = ocp.x00[1] = [1, 1]
ocp.xf0[0], ocp.xf0[1] = [1, 1], [0, 0]
ocp.lbx[0][1] = -0.8
ocp.lbu[0], ocp.ubu[0] = -1, 1
ocp.lbt0[0], ocp.ubt0[0] = 0, 0
ocp.lbtf[0], ocp.ubtf[0] = 1, 1
ocp.lbtf[1], ocp.ubtf[1] = 2.9, 2.9
ocp.validate()

mpo, post = mp.solve(ocp, n_segments=4, poly_orders=5, scheme="LGL", plot=False)
x, u, t, _ = post.get_data()                                  # both phases stacked
print(f"J = {float(post.solution['f']):.3e}; {x.shape[0]} nodes over t in [{t[0, 0]:.2f}, {t[-1, 0]:.2f}]")
print("solver:", mpo.nlp_solver.stats["return_status"])
