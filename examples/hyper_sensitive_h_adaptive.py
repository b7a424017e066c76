"""Hyper-sensitive problem with iterative refinement of the segment widths (``mp.mpopt_h_adaptive``): the NLP structure
never changes, only the parameter vector of widths -- one GPU context serves the whole loop, the dynamics residuals that
drive the refinement come from the GPU interpolation kernel."""
from mpopt_amd import mp

ocp = mp.OCP(n_states=1, n_controls=1)
ocp.dynamics[0] = lambda x, u, t: [-x[0] * x[0] * x[0] + u[0]]
ocp.running_costs[0] = lambda x, u, t: 0.5 * (x[0] * x[0] + u[0] * u[0])
ocp.terminal_constraints[0] = lambda xf, tf, x0, t0: [xf[0] - 1.0]
ocp.x00[0] = 1.0
ocp.lbtf[0] = ocp.ubtf[0] = 1000.0
ocp.scale_t = 1 / 1000.0
ocp.validate()

mpo = mp.mpopt_h_adaptive(ocp, n_segments=15, poly_orders=4, scheme="LGR")
sol = mpo.solve(max_iter=4, mpopt_options={"method": "residual", "sub_method": "merge_split"})
post = mpo.process_results(sol, plot=False)
x, u, t, _ = post.get_data()
print(f"J = {float(sol['f']):.6f} after {mpo.iter_count} refinements; max residual per iteration: "
      + ", ".join(f"{k}: {v:.2e}" for k, v in mpo.iter_info.items()))
print("segment width fractions:", [round(float(w), 4) for w in mpo._nlp_sw_params])
