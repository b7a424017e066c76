"""mpopt_amd -- MI355X-native collocation hot path behind mpopt's public interface.

    from mpopt_amd import mp
    ocp = mp.OCP(n_states=2, n_controls=1)
    ...
    mpo = mp.mpopt(ocp, n_segments=1000, poly_orders=5, scheme="LGR")
    nlp, bounds = mpo.create_nlp()
    oracle = nlp["oracle"]            # f, g, grad_f, jac_g, hess_l on the GPU (include/mpx.h)

``mp`` mirrors ``mpopt.mp`` of the reference (mpopt/__init__.py:20) for the classes on the path.
"""
from . import mpopt as mp  # noqa: F401
from .expr import math_ns as math  # noqa: F401
from .mpopt import OCP, Collocation, CollocationRoots, mpopt, mpopt_adaptive, mpopt_h_adaptive, mpopt_ph_adaptive, solve  # noqa: F401
from .nlp import NlpFunctions  # noqa: F401
from ._lib import MpxError  # noqa: F401
