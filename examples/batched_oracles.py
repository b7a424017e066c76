"""What the GPU adds to the reference's workflow: the NLP oracles of one transcription evaluated for thousands of points
per launch, inputs and outputs resident in HBM (torch tensors), e.g. for multi-start, sampling or parameter sweeps.
BASELINE configs[1]: moon lander, 1000 segments of degree 5."""
import time

import numpy as np
import torch

from mpopt_amd import mp
from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_JAC

ocp = mp.OCP(n_states=2, n_controls=1)
ocp.dynamics[0] = lambda x, u, t: [x[1], u[0] - 1.5]
ocp.running_costs[0] = lambda x, u, t: u[0]
ocp.terminal_constraints[0] = lambda xf, tf, x0, t0: [xf[0], xf[1]]
ocp.x00[0] = [10.0, -2.0]
ocp.lbu[0], ocp.ubu[0] = 0, 3
ocp.lbtf[0], ocp.ubtf[0] = 3, 5

mpo = mp.mpopt(ocp, n_segments=1000, poly_orders=5, scheme="LGR")
nlp, bounds = mpo.create_nlp()
o = nlp["oracle"]                                            # f, g, grad_f, jac_g, hess_l on the GPU
B, dev = 1024, torch.device("cuda", 0)
rng = np.random.default_rng(0)
Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.01 * rng.standard_normal((B, o.n_z)), device=dev)
p = torch.full((o.n_p,), 1.0 / 1000, dtype=torch.float64, device=dev)
f, g = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
grad, jac = torch.empty(B, o.n_z, dtype=torch.float64, device=dev), torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
mask = MPX_F | MPX_G | MPX_GRAD | MPX_JAC
o.eval_device(mask, B, Z, p, 0, None, None, f, g, grad, jac, None)
o.sync()
t0 = time.perf_counter()
for _ in range(20):
    o.eval_device(mask, B, Z, p, 0, None, None, f, g, grad, jac, None)
o.sync()
dt = (time.perf_counter() - t0) / 20
print(f"n_z = {o.n_z}, n_g = {o.n_g}, nnz(jac_g) = {o.nnz_jac}: {B} evaluations of f, g, grad_f, jac_g in {dt * 1e3:.3f} ms "
      f"= {B / dt / 1e6:.2f} M evaluations/s; max |g| of the first point {float(g[0].abs().max()):.3f}")
