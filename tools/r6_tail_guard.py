"""Out-of-bounds probe (round 6): every input and output array of an evaluation sits at the very END of its own device allocation (hipMalloc of a
multiple of 2 MB, the array in its last bytes), so that a kernel reading or writing past the end of an array leaves the allocation -- a memory
access fault where the next addresses are unmapped -- instead of landing in a neighbour's bytes.  Results are compared with the same evaluation
on ordinary arrays (bit for bit).  One process per grid would isolate a fault; the tool prints each step before it runs it.
    python tools/r6_tail_guard.py [case ...]"""
import ctypes
import os
import sys

os.environ.setdefault("MPX_ENV_DYNAMIC", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import mpopt_amd as M
from mpopt_amd import mp
from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_HESS, MPX_JAC
import problems

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
GRAN = 2 << 20
HEAD = bool(os.environ.get("GUARD_HEAD"))  # GUARD_HEAD=1: the arrays at the START of their allocations instead (reads / writes BEFORE an array)


class Tail:
    """A float64 device array of `n` entries in the last 8 n bytes of its own allocation."""

    def __init__(self, n, host=None):
        self.n = int(n)
        nb = max(8, 8 * self.n)
        size = (nb + GRAN - 1) // GRAN * GRAN
        base = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(base), size) == 0
        assert hip.hipMemset(base, 0x7F, size) == 0
        self.base, self.ptr = base, (base.value if HEAD else base.value + size - 8 * self.n)
        if host is not None:
            h = np.ascontiguousarray(host, dtype=np.float64)
            assert h.size == self.n
            assert hip.hipMemcpy(ctypes.c_void_p(self.ptr), h.ctypes.data_as(ctypes.c_void_p), 8 * self.n, 1) == 0

    def get(self, shape):
        out = np.empty(self.n)
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.ptr), 8 * self.n, 2) == 0
        return out.reshape(shape)

    def free(self):
        hip.hipFree(self.base)


CASES = {
    "vdp_98x20_CGL": (problems.van_der_pol, 98, 20, "CGL"),                       # light_body, single degree, last group of 2 segments
    "vdp_3_30_3_CGL": (problems.van_der_pol, 47, [30 if s % 3 == 1 else 3 for s in range(47)], "CGL"),   # configs[2]'s pattern, ends on a degree-30 segment
    "vdp_3_30_3_3_CGL": (problems.van_der_pol, 48, [30 if s % 3 == 1 else 3 for s in range(48)], "CGL"),
    "moon_lander_333x5_LGR": (problems.moon_lander, 333, 5, "LGR"),              # light_low, N = 1666
    "hyper_sensitive_1001x3_LGR": (problems.hyper_sensitive, 1001, 3, "LGR"),
    "schwartz_77x3_LGL": (problems.two_phase_schwartz, 77, 3, "LGL"),            # two phases, all-phases kernels
    "kitchen_sink_mixed": (problems.kitchen_sink, 41, [2, 5, 3, 4] * 10 + [13], "LGR"),
    "time_dependent_50x7": (problems.time_dependent, 50, 7, "LGR"),
    "moon_lander_3x100_LGR": (problems.moon_lander, 3, 100, "LGR"),              # streamed tables + light_high
    "dae_vdp_3_100_3_LGL": (problems.dae_vdp, 5, [3, 100, 3, 3, 69], "LGL"),
    "hyper_sensitive_1x255_CGL": (problems.hyper_sensitive, 1, 255, "CGL"),
    "moon_lander_20x3_LGR": (problems.moon_lander, 20, 3, "LGR"),                # configs[0]
    # assembled contexts (mpopt_adaptive): point + gather kernels (B = 1, 3), fused kernels, lane-per-point kernels (B = 70)
    "adaptive_moon_lander_20x5": (problems.moon_lander, 20, 5, "LGR"),
    "adaptive_kitchen_sink_mixed": (problems.kitchen_sink, 6, [3, 2, 4, 3, 2, 5], "LGR"),
    "adaptive_van_der_pol_mixed": (problems.van_der_pol, 9, [2, 4, 3] * 3, "CGL"),
}


def main():
    names = sys.argv[1:] or list(CASES)
    for name in names:
        builder, S, po, scheme = CASES[name]
        ocp = builder(mp, M.math)
        adaptive = name.startswith("adaptive_")
        mpo = (mp.mpopt_adaptive if adaptive else mp.mpopt)(ocp, S, po, scheme)
        o = mpo.create_nlp()[0]["oracle"]
        rng = np.random.default_rng(1)
        for B in (1, 3, 70) if adaptive else (1, 3):
            Zh = mpo.initialize_solution()[None, :] + 0.05 * rng.standard_normal((B, o.n_z))
            w = rng.uniform(0.4, 1.6, (ocp.n_phases, S))
            ph = (w / w.sum(axis=1, keepdims=True)).ravel()
            lamh, sigh = rng.standard_normal((B, o.n_g)), rng.uniform(0.5, 1.5, B)
            if adaptive:
                ph = np.zeros(0)
            ref = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], Zh, None if adaptive else ph, lam_g=lamh, sigma=sigh)
            refq = o.eval_grad_gamma(Zh, None if adaptive else ph, lamh, sigh)
            Z, P, L, Sg = Tail(Zh.size, Zh), Tail(max(1, ph.size), ph if ph.size else np.zeros(1)), Tail(lamh.size, lamh), Tail(B, sigh)
            if adaptive:
                P.ptr = None
            for mask in (MPX_F | MPX_G | MPX_GRAD | MPX_JAC, MPX_F | MPX_G, MPX_F, MPX_G, MPX_GRAD, MPX_F | MPX_GRAD, MPX_G | MPX_JAC, MPX_HESS, 31):
                print(name, "B", B, "mask", mask, "...", end="", flush=True)
                f, g, q, jv, hv = Tail(B), Tail(B * o.n_g), Tail(B * o.n_z), Tail(B * o.nnz_jac), Tail(B * o.nnz_hess)
                o.eval_device(mask, B, Z.ptr, P.ptr, 0, L.ptr if mask & MPX_HESS else None, Sg.ptr if mask & MPX_HESS else None,
                              f.ptr if mask & MPX_F else None, g.ptr if mask & MPX_G else None, q.ptr if mask & MPX_GRAD else None,
                              jv.ptr if mask & MPX_JAC else None, hv.ptr if mask & MPX_HESS else None)
                o.sync()
                if mask & MPX_G:
                    assert np.array_equal(g.get((B, o.n_g)), np.asarray(ref["g"]).reshape(B, -1)), "g"
                if mask & MPX_JAC:
                    assert np.array_equal(jv.get((B, o.nnz_jac)), np.asarray(ref["jac_g"]).reshape(B, -1)), "jac"
                if mask & MPX_HESS:
                    assert np.array_equal(hv.get((B, o.nnz_hess)), np.asarray(ref["hess_l"]).reshape(B, -1)), "hess"
                if mask & MPX_F:
                    assert np.allclose(f.get((B,)), np.asarray(ref["f"]).reshape(B), rtol=1e-13, atol=1e-13), "f"
                if mask & MPX_GRAD:
                    assert np.allclose(q.get((B, o.n_z)), np.asarray(ref["grad_f"]).reshape(B, -1), rtol=1e-12, atol=1e-13), "grad_f"
                for t in (f, g, q, jv, hv):
                    t.free()
                print("ok", flush=True)
            print(name, "B", B, "nlp_grad ...", end="", flush=True)
            gx, gp = Tail(B * o.n_z), Tail(max(1, B * o.n_p))
            o.eval_grad_gamma_device(B, Z.ptr, P.ptr, L.ptr, Sg.ptr, gx.ptr, gp.ptr if o.n_p else None)
            o.sync()
            assert np.array_equal(gx.get((B, o.n_z)), np.asarray(refq["grad_gamma_x"]).reshape(B, -1))
            print("ok", flush=True)
            for t in (Z, P, L, Sg, gx, gp):
                t.free()
        o.close()
    print(("head" if HEAD else "tail") + " guard: all cases ok")


if __name__ == "__main__":
    main()
