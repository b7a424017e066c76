"""ctypes wrapper of oracle/mpopt_oracle.c (TEST INFRASTRUCTURE ONLY; see the header of the C file)."""
import ctypes
import os
import subprocess

import numpy as np

from . import mpopt_oracle as npo

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mpopt_oracle.c")
LIB = os.path.join(HERE, "liborc.so")
LIB_LD = os.path.join(HERE, "liborc_ld.so")  # the same source with -DORC_LONG_DOUBLE: every double an 80-bit long double (the arbiter)
_lib = None
_lib_ld = None


def build(force=False):
    """Both builds of mpopt_oracle.c: liborc.so (binary64, also the timed cpu_baseline) and liborc_ld.so (long-double arbiter)."""
    for path, extra in ((LIB, []), (LIB_LD, ["-DORC_LONG_DOUBLE"])):
        if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(SRC):
            subprocess.check_call(["gcc", "-O2", "-std=c11", "-fopenmp", "-fPIC", "-shared", "-o", path + ".tmp", SRC, "-lm"] + extra)
            os.replace(path + ".tmp", path)
    return LIB


def lib(long_double=False):
    global _lib, _lib_ld
    if long_double:
        if _lib_ld is None:
            build()
            _lib_ld = _bind(ctypes.CDLL(LIB_LD), ctypes.c_longdouble)
        return _lib_ld
    if _lib is None:
        _lib = _bind(ctypes.CDLL(build()), ctypes.c_double)
    return _lib


def _bind(L, c_real):
    if True:
        dp, ip = ctypes.POINTER(c_real), ctypes.POINTER(ctypes.c_int)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ip, ctypes.c_int, ip, dp,
                                 c_real, c_real, dp, dp, dp, c_real, ip]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        for n in ("orc_n_z", "orc_n_g", "orc_nnz"):
            getattr(L, n).restype = ctypes.c_int64
            getattr(L, n).argtypes = [ctypes.c_void_p]
        L.orc_set_table.restype = ctypes.c_int
        L.orc_set_table.argtypes = [ctypes.c_void_p, ctypes.c_int, dp, dp, dp]
        L.orc_get_table.restype = ctypes.c_int
        L.orc_get_table.argtypes = [ctypes.c_void_p, ctypes.c_int, dp, dp, dp]
        L.orc_eval.restype = ctypes.c_int64
        L.orc_eval.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 8
        L.orc_eval_many_omp.restype = ctypes.c_int
        L.orc_eval_many_omp.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_hess_capacity.restype = ctypes.c_int64
        L.orc_hess_capacity.argtypes = [ctypes.c_void_p]
        L.orc_hess.restype = ctypes.c_int64
        L.orc_hess.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_real] + [ctypes.c_void_p] * 4
        L.orc_grad_gamma.restype = ctypes.c_int
        L.orc_grad_gamma.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_real] + [ctypes.c_void_p] * 3
        L.orc_ipopt_mix.restype = None
        L.orc_ipopt_mix.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                    c_real] + [ctypes.c_void_p] * 7
        L.orc_eval_fn.restype = None
        L.orc_eval_fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                  c_real] + [ctypes.c_void_p] * 7
        L.orc_eval_many.restype = None
        L.orc_eval_many.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 6
    return L


class COracle:
    """``names``: one C problem name per phase (see PROBLEMS[] in mpopt_oracle.c)."""

    def __init__(self, names, n_segments, poly_orders, scheme="LGR", tau0=-1.0, tau1=1.0, scale_x=None, scale_u=None,
                 scale_a=None, scale_t=1.0, midu=None, long_double=False):
        """``long_double=True``: the -DORC_LONG_DOUBLE build (80-bit arithmetic throughout) on the SAME binary64 inputs and the SAME
        tables as the double build -- the arbiter of the parity tests.  eval / hess / grad_gamma work there; results come back
        rounded to binary64 (1.1e-16 relative: five orders below the tolerance they arbitrate)."""
        self._ld = bool(long_double)
        self._real = np.longdouble if long_double else float
        c_real = ctypes.c_longdouble if long_double else ctypes.c_double
        float_ = self._creal = c_real  # (by-value reals of the C interface)
        L = lib(long_double)
        orders = np.ascontiguousarray([poly_orders] * n_segments if np.isscalar(poly_orders) else poly_orders, dtype=np.intc)
        degs = np.ascontiguousarray(sorted(set(int(d) for d in orders)), dtype=np.intc)
        taus = np.ascontiguousarray(np.concatenate([npo.roots(scheme, int(d), tau0, tau1) for d in degs]), dtype=self._real)
        arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        sx = np.ascontiguousarray(scale_x if scale_x is not None else np.ones(8), dtype=self._real)
        su = np.ascontiguousarray(scale_u if scale_u is not None else np.ones(8), dtype=self._real)
        sa = np.ascontiguousarray(scale_a if scale_a is not None else np.ones(8), dtype=self._real)
        midu = np.ascontiguousarray(midu if midu is not None else [1] * len(names), dtype=np.intc)
        dp, ip = ctypes.POINTER(c_real), ctypes.POINTER(ctypes.c_int)
        self._h = L.orc_create(len(names), arr, int(n_segments), orders.ctypes.data_as(ip), len(degs), degs.ctypes.data_as(ip),
                               taus.ctypes.data_as(dp), float_(tau0), float_(tau1), sx.ctypes.data_as(dp), su.ctypes.data_as(dp),
                               sa.ctypes.data_as(dp), float_(scale_t), midu.ctypes.data_as(ip))
        if not self._h:
            raise ValueError(f"unknown C oracle problem in {names}")
        self.n_z, self.n_g, self.nnz = L.orc_n_z(self._h), L.orc_n_g(self._h), L.orc_nnz(self._h)
        twin = None
        if long_double:  # the double build with the same arguments: ITS tables (binary64) are the problem both builds evaluate
            twin = COracle(names, n_segments, poly_orders, scheme, tau0, tau1, scale_x, scale_u, scale_a, scale_t, midu)
        for d in degs:
            if twin is not None:
                n = int(d) + 1
                D, w, Cm = np.zeros(n * n), np.zeros(n), np.zeros(max(int(d) * n, 1))
                dpd = ctypes.POINTER(ctypes.c_double)
                assert lib().orc_get_table(twin._h, int(d), D.ctypes.data_as(dpd), w.ctypes.data_as(dpd), Cm.ctypes.data_as(dpd)) == 0
                D, w, Cm = (np.ascontiguousarray(a_, dtype=np.longdouble) for a_ in (D, w, Cm))
                assert L.orc_set_table(self._h, int(d), D.ctypes.data_as(dp), w.ctypes.data_as(dp), Cm.ctypes.data_as(dp)) == 0
            elif d > 10:  # see "exact_tables" in mpopt_oracle.py
                x = npo.roots(scheme, int(d), tau0, tau1)
                D = np.ascontiguousarray(npo.exact_tables(x, x, 1))
                w = np.ascontiguousarray(npo.exact_tables(x, None, "w", tau0, tau1))
                Cm = np.ascontiguousarray(npo.exact_tables(x, (x[:-1] + x[1:]) / 2.0, 0))
                assert L.orc_set_table(self._h, int(d), D.ctypes.data_as(dp), w.ctypes.data_as(dp), Cm.ctypes.data_as(dp)) == 0

    def eval(self, z, p):
        L = lib(self._ld)
        z, p = np.ascontiguousarray(z, self._real), np.ascontiguousarray(p, self._real)
        f, g, grad = np.zeros(1, self._real), np.zeros(self.n_g, self._real), np.zeros(self.n_z, self._real)
        rows, cols, vals = np.zeros(self.nnz, np.int32), np.zeros(self.nnz, np.int32), np.zeros(self.nnz, self._real)
        n = L.orc_eval(self._h, z.ctypes.data, p.ctypes.data, f.ctypes.data, g.ctypes.data, grad.ctypes.data, rows.ctypes.data,
                       cols.ctypes.data, vals.ctypes.data)
        assert n == self.nnz, (n, self.nnz)
        return dict(f=float(f[0]), g=g.astype(float), grad_f=grad.astype(float), jac_row=rows, jac_col=cols, jac_val=vals.astype(float))

    def hess(self, z, p, sigma, lam_g):
        """hess_l as COO triplets of the upper triangle (duplicates add up): dict hess_row, hess_col, hess_val."""
        L = lib(self._ld)
        z, p, lam = np.ascontiguousarray(z, self._real), np.ascontiguousarray(p, self._real), np.ascontiguousarray(lam_g, self._real)
        cap = L.orc_hess_capacity(self._h)
        rows, cols, vals = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, self._real)
        n = L.orc_hess(self._h, z.ctypes.data, p.ctypes.data, self._creal(sigma), lam.ctypes.data, rows.ctypes.data, cols.ctypes.data,
                       vals.ctypes.data)
        return dict(hess_row=rows[:n].copy(), hess_col=cols[:n].copy(), hess_val=vals[:n].astype(float))

    def grad_gamma(self, z, p, sigma, lam_g):
        """``nlp_grad``: (grad_gamma_x [n_z], grad_gamma_p [n_p]) of gamma = sigma*f + lam_g^T g."""
        L = lib(self._ld)
        z, p, lam = np.ascontiguousarray(z, self._real), np.ascontiguousarray(p, self._real), np.ascontiguousarray(lam_g, self._real)
        ggx, ggp = np.zeros(self.n_z, self._real), np.zeros(len(p), self._real)
        assert L.orc_grad_gamma(self._h, z.ctypes.data, p.ctypes.data, self._creal(sigma), lam.ctypes.data, ggx.ctypes.data, ggp.ctypes.data) == 0
        return ggx.astype(float), ggp.astype(float)

    def hess_matrix(self, z, p, sigma, lam_g):
        """Upper triangle of hess_l as a scipy CSR matrix (duplicates summed)."""
        import scipy.sparse as sp

        h = self.hess(z, p, sigma, lam_g)
        return sp.coo_matrix((h["hess_val"], (h["hess_row"], h["hess_col"])), shape=(self.n_z, self.n_z)).tocsr()

    def time_ipopt_mix(self, Z, p, sigma, lam_g, reps, n_g_calls=1):
        """Wall seconds for ``reps`` passes over the points Z of: n_g_calls x nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l
        (separate functions, one core)."""
        import time

        L = lib()
        Z, p, lam = np.ascontiguousarray(Z, float), np.ascontiguousarray(p, float), np.ascontiguousarray(lam_g, float)
        g, grad, vals = np.zeros(self.n_g), np.zeros(self.n_z), np.zeros(self.nnz)
        cap = L.orc_hess_capacity(self._h)
        hr, hc, hv = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
        args = (Z.ctypes.data, p.ctypes.data, float(sigma), lam.ctypes.data, g.ctypes.data, grad.ctypes.data, vals.ctypes.data,
                hr.ctypes.data, hc.ctypes.data, hv.ctypes.data)
        L.orc_ipopt_mix(self._h, Z.shape[0], 1, int(n_g_calls), *args)
        t = time.perf_counter()
        L.orc_ipopt_mix(self._h, Z.shape[0], int(reps), int(n_g_calls), *args)
        return time.perf_counter() - t

    def time_fn(self, which, Z, p, sigma, lam_g, seconds=1.0):
        """Seconds per call of ONE oracle function on one core: which in {"nlp_g", "nlp_grad_f", "nlp_jac_g", "nlp_hess_l", "nlp_f"},
        cycling over the points Z, for about ``seconds`` of wall time."""
        import time

        L = lib()
        code = {"nlp_g": 0, "nlp_grad_f": 1, "nlp_jac_g": 2, "nlp_hess_l": 3, "nlp_f": 4}[which]
        Z, p, lam = np.ascontiguousarray(Z, float).reshape(-1, self.n_z), np.ascontiguousarray(p, float), np.ascontiguousarray(lam_g, float)
        g, grad, vals = np.zeros(self.n_g), np.zeros(self.n_z), np.zeros(self.nnz)
        cap = L.orc_hess_capacity(self._h)
        hr, hc, hv = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
        args = (Z.ctypes.data, p.ctypes.data, float(sigma), lam.ctypes.data, g.ctypes.data, grad.ctypes.data, vals.ctypes.data,
                hr.ctypes.data, hc.ctypes.data, hv.ctypes.data)
        n = Z.shape[0]
        L.orc_eval_fn(self._h, code, n, 1, *args)
        t = time.perf_counter()
        L.orc_eval_fn(self._h, code, n, 1, *args)
        one = max(time.perf_counter() - t, 1e-7)
        reps = max(1, int(seconds / one))
        t = time.perf_counter()
        L.orc_eval_fn(self._h, code, n, reps, *args)
        return (time.perf_counter() - t) / (reps * n)

    def time_many(self, Z, p, reps):
        """Wall seconds for ``reps`` passes of f+g+grad_f+jac_g over the points Z (values only)."""
        import time

        L = lib()
        Z, p = np.ascontiguousarray(Z, float), np.ascontiguousarray(p, float)
        B = Z.shape[0]
        f, g, grad, vals = np.zeros(B), np.zeros(self.n_g), np.zeros(self.n_z), np.zeros(self.nnz)
        L.orc_eval_many(self._h, B, 1, Z.ctypes.data, p.ctypes.data, f.ctypes.data, g.ctypes.data, grad.ctypes.data, vals.ctypes.data)
        t = time.perf_counter()
        L.orc_eval_many(self._h, B, int(reps), Z.ctypes.data, p.ctypes.data, f.ctypes.data, g.ctypes.data, grad.ctypes.data, vals.ctypes.data)
        return time.perf_counter() - t

    def time_many_all_cores(self, Z, p, reps):
        """(wall seconds, threads) for ``reps`` passes over Z with OpenMP over the points."""
        import time

        L = lib()
        Z, p = np.ascontiguousarray(Z, float), np.ascontiguousarray(p, float)
        f = np.zeros(Z.shape[0])
        L.orc_eval_many_omp(self._h, Z.shape[0], 1, Z.ctypes.data, p.ctypes.data, f.ctypes.data)
        t = time.perf_counter()
        n = L.orc_eval_many_omp(self._h, Z.shape[0], int(reps), Z.ctypes.data, p.ctypes.data, f.ctypes.data)
        return time.perf_counter() - t, n

    def __del__(self):
        try:
            lib(self._ld).orc_destroy(self._h)
        except Exception:
            pass
