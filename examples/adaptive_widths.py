"""Segment widths as decision variables (``mp.mpopt_adaptive``): three segments of degree 2 are enough for the moon
lander, because the optimiser moves a segment boundary onto the bang-bang switch."""
from mpopt_amd import mp

ocp = mp.OCP(n_states=2, n_controls=1, n_phases=1)
ocp.dynamics[0] = lambda x, u, t: [x[1], u[0] - 1.5]
ocp.running_costs[0] = lambda x, u, t: u[0]
ocp.terminal_constraints[0] = lambda xf, tf, x0, t0: [xf[0], xf[1]]
ocp.x00[0] = [10, -2]
ocp.lbu[0], ocp.ubu[0] = 0, 3
ocp.lbtf[0], ocp.ubtf[0] = 3, 5

opt = mp.mpopt_adaptive(ocp, n_segments=3, poly_orders=[2] * 3)
solution = opt.solve()
post = opt.process_results(solution, plot=False)
x, u, t, _ = post.get_data()
print(f"J = {float(solution['f']):.6f}  (analytic optimum 8.24621, 20 x 3 fixed grid 8.24677)")
print("thrust at the nodes:", [round(float(v), 3) for v in u[:, 0]])
