"""Assembled NLP oracles: point functions + constant sparse maps -> libmpx assembled context.

A transcription is handed over as

    f(z) = sum_k  fw_k . phi_k(L_k z),        g(z) = G_z z + g_0 + sum_k G_k phi_k(L_k z)

where every ``phi_k`` is a *point function* (the user's dynamics / path / cost callables traced at a
collocation node, a mid-point or a phase end), evaluated at ``n_k`` points whose local variables are
linear in ``z`` (``L_k``: selections and interpolation rows).  From this description the module

  * differentiates each point function symbolically (``mpopt_amd.expr``: structural zeros, like
    CasADi's SX that the reference relies on, mpopt.py:757) and emits it as HIP device code;
  * expands  jac_g = G_z + sum_k G_k (dphi_k) L_k,  grad_f = sum_k fw_k (dphi_k) L_k  and
    hess_l = sum_k L_k^T (sum_r mu_r d2phi_k,r) L_k  ONCE into gather rows over the raw point
    derivatives (sparsity patterns included), in a fixed term order;
  * creates the context with ``mpx_create_assembled`` (include/mpx.h).  All arithmetic of an
    evaluation then happens on the GPU: generated point kernels + one gather kernel.

Used by ``mpopt_adaptive`` (mpopt_amd/adaptive.py).  Nothing here evaluates the NLP on the CPU.
"""
import ctypes

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import MpxError, mpx_assembly, mpx_point_set
from .expr import Tracer
from .nlp import NlpFunctions

SRC_ONE = -1  # gather source of the constant 1


def src_z(col):
    """Gather source index of z[col]."""
    return -2 - np.asarray(col, dtype=np.int64)


class PointFunction:
    """``build(loc, cst) -> [out expressions]`` traced once; first derivatives of every output and second
    derivatives of ``sum_r mu_r out_r`` with respect to the local variables, structural entries only."""

    def __init__(self, n_loc, n_cst, build):
        tr = self.tr = Tracer()
        self.n_loc, self.n_cst = int(n_loc), int(n_cst)
        loc = [tr.var(f"loc[{i}]") for i in range(self.n_loc)]
        cst = [tr.var(f"cst[{i}]") for i in range(self.n_cst)]
        self.out = [tr.wrap(e) for e in build(loc, cst)]
        self.n_out = len(self.out)
        memo = [dict() for _ in range(self.n_loc)]
        self.J = []  # (r, v, expr)
        for r, e in enumerate(self.out):
            for v in range(self.n_loc):
                d = tr.diff(e, loc[v], memo[v])
                if not d.is_zero:
                    self.J.append((r, v, d))
        mu = [tr.var(f"mu[{r}]") for r in range(self.n_out)]
        lag = tr.zero
        for r, e in enumerate(self.out):
            lag = lag + mu[r] * e
        self.H = []  # (v1, v2, expr), v1 <= v2
        for v1 in range(self.n_loc):
            g1 = tr.diff(lag, loc[v1], memo[v1])
            if g1.is_zero:
                continue
            for v2 in range(v1, self.n_loc):
                d2 = tr.diff(g1, loc[v2], memo[v2])
                if not d2.is_zero:
                    self.H.append((v1, v2, d2))
        self.n_jac, self.n_hess = len(self.J), len(self.H)

    def source(self, fid):
        tr = self.tr
        names = {f"loc[{i}]": f"loc[{i}]" for i in range(self.n_loc)}
        names.update({f"cst[{i}]": f"cst[{i}]" for i in range(self.n_cst)})
        names.update({f"mu[{r}]": f"mu[{r}]" for r in range(self.n_out)})
        sig = "const double* __restrict__ loc, const double* __restrict__ cst"
        # (#pragma clang fp contract(off): every kernel that inlines these bodies -- two-pass and fused -- rounds each operation
        # the same way, so their results agree bit for bit whatever the surrounding code looks like)
        o = [f"template <> struct Pt<{fid}> {{",
             f"  static constexpr int NLOC = {self.n_loc}, NCST = {self.n_cst}, NOUT = {self.n_out}, NJ = {self.n_jac}, NH = {self.n_hess};",
             f"  __device__ static __forceinline__ void val({sig}, double* out) {{", "#pragma clang fp contract(off)"]
        o += tr.emit([(f"out[{r}]", e) for r, e in enumerate(self.out)], names, "    ")
        o += ["  }", f"  __device__ static __forceinline__ void jac({sig}, double* out, double* J) {{", "#pragma clang fp contract(off)"]
        o += tr.emit([(f"out[{r}]", e) for r, e in enumerate(self.out)] + [(f"J[{q}]", s[2]) for q, s in enumerate(self.J)], names, "    ")
        o += ["  }", f"  __device__ static __forceinline__ void hes({sig}, const double* __restrict__ mu, double* H) {{", "#pragma clang fp contract(off)"]
        o += tr.emit([(f"H[{q}]", s[2]) for q, s in enumerate(self.H)], names, "    ")
        o += ["  }", "};"]
        return "\n".join(o)


class PointSet:
    """``n`` points of one point function.

    ``L``   : sparse (n*n_loc, n_z), row p*n_loc + v = local variable v of point p
    ``cst`` : (n, n_cst) per-point constants
    ``G``   : sparse (n_g, n*n_out), column p*n_out + r = coefficient of output r of point p in g
    ``fw``  : (n, n_out) coefficients of the outputs in the objective"""

    def __init__(self, fn, n, L, cst, G, fw):
        self.fn, self.n = fn, int(n)
        self.L = sp.csr_matrix(L)
        self.L.sum_duplicates()
        self.cst = np.ascontiguousarray(np.asarray(cst, dtype=np.float64).reshape(self.n, fn.n_cst))
        self.G = sp.csc_matrix(G)
        self.G.sum_duplicates()
        self.fw = np.ascontiguousarray(np.asarray(fw, dtype=np.float64).reshape(self.n, fn.n_out))
        assert self.L.shape[0] == self.n * fn.n_loc and self.G.shape[1] == self.n * fn.n_out

    def _rows(self, v):
        """CSR slices of the local-variable rows v of all points: (counts, flat point index, col, coef)."""
        rows = np.arange(self.n) * self.fn.n_loc + v
        lo, hi = self.L.indptr[rows], self.L.indptr[rows + 1]
        cnt = hi - lo
        take = np.concatenate([np.arange(a, b) for a, b in zip(lo, hi)]) if cnt.sum() else np.zeros(0, np.int64)
        return cnt, np.repeat(np.arange(self.n), cnt), self.L.indices[take].astype(np.int64), self.L.data[take]

    def _gcols(self, r):
        """Entries of G acting on output r of all points: (point, g row, coef)."""
        cols = np.arange(self.n) * self.fn.n_out + r
        lo, hi = self.G.indptr[cols], self.G.indptr[cols + 1]
        cnt = hi - lo
        take = np.concatenate([np.arange(a, b) for a, b in zip(lo, hi)]) if cnt.sum() else np.zeros(0, np.int64)
        return np.repeat(np.arange(self.n), cnt), self.G.indices[take].astype(np.int64), self.G.data[take]

    @staticmethod
    def _ell(n, n_var, entries):
        """entries[v] = (point, index, coef) -> (nterm[v], idx[(toff_v + t) * n + p], coef[...]); padding: index 0, coef 0."""
        nterm, idx, coef = [], [], []
        for v in range(n_var):
            pt, ix, cf = entries[v]
            cnt = np.bincount(pt, minlength=n) if len(pt) else np.zeros(n, np.int64)
            T = int(cnt.max()) if n else 0
            I, C = np.zeros((T, n), np.int32), np.zeros((T, n))
            if len(pt):
                order = np.argsort(pt, kind="stable")
                pt, ix, cf = pt[order], ix[order], cf[order]
                first = np.concatenate([[0], np.cumsum(cnt)[:-1]])
                t = np.arange(len(pt)) - first[pt]
                I[t, pt], C[t, pt] = ix, cf
            nterm.append(T), idx.append(I.ravel()), coef.append(C.ravel())
        cat = lambda parts, dt: np.ascontiguousarray(np.concatenate(parts) if parts and sum(len(a) for a in parts) else np.zeros(1), dtype=dt)
        return np.ascontiguousarray(nterm if nterm else [0], dtype=np.int32), cat(idx, np.int32), cat(coef, np.float64)


def _csr_from_terms(n_rows, rows, src, coef):
    rows = np.asarray(rows, dtype=np.int64)
    order = np.argsort(rows, kind="stable")  # stable: the stored order of a row's terms is the generation order
    ptr = np.zeros(n_rows + 1, np.int64)
    np.add.at(ptr, rows + 1, 1)
    return np.cumsum(ptr), np.ascontiguousarray(np.asarray(src, dtype=np.int64)[order], dtype=np.int32), np.ascontiguousarray(np.asarray(coef, dtype=np.float64)[order])


def _cat(parts, dtype):
    parts = [np.asarray(a, dtype=dtype).ravel() for a in parts]
    return np.concatenate(parts) if parts else np.zeros(0, dtype)


class AssembledNlpFunctions(NlpFunctions):
    """``NlpFunctions`` over an assembled context: same evaluation / pattern / CasADi-symbol surface, no NLP
    parameters (``p`` is ignored), no tiles."""

    def __init__(self, n_z, n_g, sets, Gz, g0, device=0, with_device=None, verbose=False):
        self.n_z_, self.n_g_ = int(n_z), int(n_g)
        self.sets = list(sets)
        Gz = sp.coo_matrix(Gz)
        Gz.sum_duplicates()
        g0 = np.asarray(g0, dtype=np.float64).reshape(self.n_g_)
        funcs = []  # distinct point functions -> kernel ids
        for s in self.sets:
            if s.fn not in funcs:
                funcs.append(s.fn)
        self.functions = funcs
        self._expand(Gz, g0)
        import os
        if os.environ.get("MPX_ASM_CLASS_MAJOR", "0") == "1":  # (A/B of round 6, a negative result: profiles/r6_asm_pair/README.md)
            self._class_major_jac()
        sizes = dict(RAW_N=self.raw_n, RAWH_N=self.rawh_n, NZ=self.n_z_, NG=self.n_g_, NNZJ=self.nnz_jac_, NNZH=self.nnz_hess_)
        # ELL tables of the local variables and of the multipliers of every set (what _create hands to libmpx), and the number of
        # distinct coefficients (bit patterns, padding included) in each family: the dictionaries of the packed tables of the fused
        # kernels (mpx_assembly_fused.h: one 32-bit entry per term = index | code << 16; libmpx builds the same dictionaries)
        self._ell = []
        for s in self.sets:
            fn = s.fn
            loc = PointSet._ell(s.n, fn.n_loc, [s._rows(v)[1:] for v in range(fn.n_loc)])
            mu = []
            for r in range(fn.n_out):
                gp, gi, gc = s._gcols(r)
                nzp = np.nonzero(s.fw[:, r])[0]
                mu.append((np.concatenate([gp, nzp]), np.concatenate([gi, np.full(len(nzp), self.n_g_, np.int64)]), np.concatenate([gc, s.fw[nzp, r]])))
            self._ell.append((loc, PointSet._ell(s.n, fn.n_out, mu)))

        def _ndistinct(arrs):
            v = np.concatenate([np.ascontiguousarray(a, dtype=np.float64).ravel() for a in arrs]) if arrs else np.zeros(0)
            return len(np.unique(v.view(np.int64))) if len(v) else 0

        sizes["NDICT_LOC"] = _ndistinct([e[0][2] for e in self._ell]) if self.n_z_ + 1 < 65536 else 0
        sizes["NDICT_MU"] = _ndistinct([e[1][2] for e in self._ell]) if self.n_g_ + 1 < 65536 else 0
        for tag, (ptr, _, _) in (("FGJ", self.fgj), ("HES", self.hess)):  # shape of the multi-term and long rows of each pass
            nt = np.diff(ptr)
            multi, longr = nt[(nt >= 2) & (nt <= 24)], nt[nt > 24]  # 24 = MPX_GATHER_LONG (mt below: _ell_width)
            # ELL width of the multi-term rows kept in registers: the smallest width <= 12 that leaves at most 16 wider rows to
            # the one-wavefront-per-row path
            mt = 0
            if len(multi):
                mt = next((w for w in range(2, 13) if (multi > w).sum() <= 16), 12)
            sizes["MT_" + tag], sizes["NMULTI_" + tag] = int(mt), int(((multi <= mt)).sum()) if len(multi) else 0
            # rows past the ELL width are the "long" rows of the fused kernels (one wavefront per row; libmpx uses the same threshold:
            # MT, or MPX_GATHER_LONG = 24 where no row has 2 .. 24 terms)
            longr = nt[nt > (mt if mt >= 2 else 24)]
            sizes["NLONG_" + tag], sizes["LT_" + tag] = len(longr), int(longr.max()) if len(longr) else 0
            # distinct coefficients (bit patterns) of the rows with at most one term: the dictionary of the packed row registers
            # of the fused kernels (mpx_assembly_fused.h, RowRegsPacked; libmpx builds the same dictionary and checks the count)
            cf = np.ascontiguousarray(self.fgj[2] if tag == "FGJ" else self.hess[2], dtype=np.float64)
            first = cf[ptr[:-1][nt == 1]] if (nt == 1).any() else np.zeros(0)
            rows_m = np.flatnonzero((nt >= 2) & (nt <= mt)) if mt else np.zeros(0, dtype=np.int64)
            terms_m = np.concatenate([cf[ptr[r]:ptr[r + 1]] for r in rows_m]) if len(rows_m) else np.zeros(0)  # ELL table of the multi-term rows
            vals = np.concatenate([first, np.zeros(1 if (nt == 0).any() else 0), terms_m])
            sizes["NDICT_" + tag] = len(np.unique(vals.view(np.int64))) if len(vals) else 0
        # per set: function id and the running term offsets of its local variables / multipliers -- compile-time constants of the
        # fused kernels (mpx_assembly_fused.h, mpxgen::SetT; the host computes the same offsets from loc_nterm / mu_nterm)
        self._set_consts = [(self.functions.index(s.fn), np.concatenate([[0], np.cumsum(np.asarray(e[0][0], np.int64))]).tolist(),
                             np.concatenate([[0], np.cumsum(np.asarray(e[1][0], np.int64))]).tolist()) for s, e in zip(self.sets, self._ell)]
        # batches: lane <-> evaluation point, the tables as straight-line code (assembly_lanes.py; no plan: no useful grouping of the
        # point tasks, the fused kernels keep that pass).  With a plan the entries of hess_l / jac_g are ordered group by group.  Same
        # long-row thresholds as libmpx (mpx_assembly.cpp: thr_hes, thr_fgj).  MPX_NO_LANES_CODE=1: contexts without it (A/B, tests).
        from . import assembly_lanes
        import os
        self.lanes_plan = self.lanes_plan_fgj = None
        self.lanes_source, self._lanes_attached = None, False
        if not os.environ.get("MPX_NO_LANES_CODE"):
            self.lanes_plan = assembly_lanes.plan_pass(self, "hes")
            # (the first-order pass this way is an opt-in, MPX_LANES_FGJ=1: bit-identical, but the pass is 96 % output and the fused kernel
            # already runs at the write-stream ceiling -- 100 against 58 us at moon lander 20x5, profiles/r5_lanes)
            self.lanes_plan_fgj = assembly_lanes.plan_pass(self, "fgj") if os.environ.get("MPX_LANES_FGJ", "0") == "1" else None
            # Only the PLAN and the entry order it implies are fixed here (the public order of hess_l / jac_g depends on it: use
            # jac_pattern / hess_pattern / ccs_perm, never an assumed order); the source text of the kernels is generated, compiled and
            # attached at the first batch (attach_lane_kernels) -- a solve through single evaluations never pays for any of it.
            self._lanes_todo = []
            for plan, tag in ((self.lanes_plan, "HES"), (self.lanes_plan_fgj, "FGJ")):
                if plan is not None:
                    assembly_lanes.group_major(self, plan)
                    mt = sizes["MT_" + tag]
                    self._lanes_todo.append((plan, mt if mt >= 2 else 24))
            self._lanes_funcs = funcs
            if self._lanes_todo:
                self.lanes_source = True  # (generated on demand: lanes_source_text)
        self.source = self._source(funcs, sizes, self._set_consts)
        if with_device is None:
            with_device = _lib.gpu_available()
        self.code_object = None
        if with_device:
            self.code_object, self.code_object_path = _lib.compile_kernels(self.source, verbose=verbose)
        self._create(device)

    def attach_lane_kernels(self, verbose=False):
        """Compile (cached) and attach the lane-per-evaluation-point kernels of this transcription (mpx_assembled_attach_kernels).
        Called by the evaluation methods at the first batch of >= 64 points; idempotent."""
        if self._lanes_attached or self.lanes_source is None or self.code_object is None:
            return
        # Whatever goes wrong here -- a source too large to be worth compiling, hipcc failing, the library refusing the object -- the
        # fused and two-pass kernels serve every call: the failure is recorded ONCE (a warning, `lanes_error`) and never retried.
        self._lanes_attached = True
        try:
            src = self.lanes_source_text()
            if len(src) > self.LANES_MAX_SOURCE_BYTES:
                raise _lib.MpxError(f"generated source of {len(src) >> 10} KB exceeds LANES_MAX_SOURCE_BYTES ({self.LANES_MAX_SOURCE_BYTES >> 10} KB)")
            co, _ = _lib.compile_kernels(src, verbose=verbose)
            self._lanes_co_buf = ctypes.create_string_buffer(co, len(co))  # (kept like the context's own code object)
            _lib.check(self._L.mpx_assembled_attach_kernels(self._ctx, ctypes.cast(self._lanes_co_buf, ctypes.c_void_p), len(co)), self._ctx)
        except _lib.MpxError as e:
            import warnings

            self.lanes_error, self.lanes_source = str(e), None
            warnings.warn(f"mpopt_amd: lane-per-point kernels not attached ({str(e)[:300]}); the fused kernels serve the batches", RuntimeWarning)

    # (one body per group SHAPE keeps real grids at 100-500 KB of source: moon lander 200 x 3 -> 125 KB, 1.5 s of hipcc; a transcription
    # whose every group is its own shape would grow with the grid as it did up to round 5 -- 3 MB and 107 s at 80 x 5 -- and is refused)
    LANES_MAX_SOURCE_BYTES = 2 << 20
    lanes_error = None

    def lanes_source_text(self):
        """The translation unit of the lane-per-point kernels (generated on first use, kept)."""
        if getattr(self, "_lanes_text", None) is None:
            from . import assembly_lanes

            parts = [assembly_lanes.pass_source(self, thr, plan) for plan, thr in self._lanes_todo]
            self._lanes_text = "\n".join(["// generated by mpopt_amd.assembly_lanes -- do not edit", "#include <hip/hip_runtime.h>", "namespace mpxgen {",
                                          "template <int FID> struct Pt;"] + [f.source(k) for k, f in enumerate(self._lanes_funcs)] +
                                         ["}  // namespace mpxgen", assembly_lanes.common_source(self)] + parts) + "\n"
        return self._lanes_text

    def _wants_lanes(self, mask, batch):
        """A batch of a pass that has lane kernels (hess_l; the first-order pass only as the MPX_LANES_FGJ opt-in)?"""
        return (batch >= 64 and not self._lanes_attached and self.lanes_source is not None and
                ((mask & _lib.MPX_HESS and self.lanes_plan is not None) or (mask & 15) == 15 and self.lanes_plan_fgj is not None))

    def eval_device(self, mask, batch, *args, **kwargs):
        if self._wants_lanes(int(mask), int(batch)):
            self.attach_lane_kernels()
        return super().eval_device(mask, batch, *args, **kwargs)

    def eval(self, what, z, *args, **kwargs):
        what = list(what)  # (iterated twice)
        if np.ndim(z) == 2:
            names = set(what)
            mask = (_lib.MPX_HESS if "hess_l" in names else 0) | (15 if {"f", "g", "grad_f", "jac_g"} <= names else 0)
            if self._wants_lanes(mask, np.shape(z)[0]):
                self.attach_lane_kernels()
        return super().eval(what, z, *args, **kwargs)

    def batched_plan(self):
        """(lanes per workgroup of the fused kernels, groups of the lane-per-point hess_l kernel, groups of the lane-per-point first-order
        kernel) the context carries (mpx_get_assembled_plan; attaches the lane kernels if that has not happened yet); 0: absent."""
        self.attach_lane_kernels()
        a, b, d = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(self._L.mpx_get_assembled_plan(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(d)), self._ctx)
        return a.value, b.value, d.value

    # -- generated source ---------------------------------------------------------------------------
    @staticmethod
    def _source(funcs, sizes, set_consts=()):
        parts = ["// generated by mpopt_amd.assembly -- do not edit", "#include <hip/hip_runtime.h>", "namespace mpxgen {",
                 "template <int FID> struct Pt;"]
        parts += [f.source(k) for k, f in enumerate(funcs)]
        # evaluation points a lane takes through the gathers together (mpx_assembly_kernels.h): 4 for small point functions
        # (MI355X, moon lander 20x5 at B = 4096: 1.05-1.45x on f+g+grad_f+jac_g, 1.1-2.1x on hess_l depending on the box); the
        # generated code is replicated per point, so large functions (kitchen sink: 3x the compile time, no gain) keep 1.
        n_stmt = sum(p.count(";") for p in parts)
        parts += ["}  // namespace mpxgen", "#ifndef MPX_PTS_UNROLL  // (-DMPX_PTS_UNROLL=n in MPX_HIPCC_FLAGS overrides)", f"#define MPX_PTS_UNROLL {4 if n_stmt <= 1000 else 1}", "#endif", '#include "mpx_assembly_kernels.h"']
        parts.append(f"MPX_INSTANTIATE_POINTS({len(funcs)})")
        # fused persistent kernels for batches (mpx_assembly_fused.h): the sizes of this problem are compile-time constants there
        # (row loops unroll, a lane's share of the row table lives in registers)
        parts += [f"#define MPX_FUSE_{k} {int(v)}" for k, v in sizes.items()]
        if 0 < len(set_consts) <= 16:
            arr = lambda v: "{" + ", ".join(str(int(x)) for x in v) + "}"
            parts += ["namespace mpxgen {", "template <int SET> struct SetT;"]
            for k, (fid, lt, mt) in enumerate(set_consts):
                parts.append(f"template <> struct SetT<{k}> {{\n  static constexpr int FID = {fid};\n"
                             f"  __host__ __device__ static constexpr int lt(int v) {{ constexpr int a[] = {arr(lt)}; return a[v]; }}\n"
                             f"  __host__ __device__ static constexpr int mt(int r) {{ constexpr int a[] = {arr(mt)}; return a[r]; }}\n}};")
            parts += ["}  // namespace mpxgen", f"#define MPX_FUSE_SETS {len(set_consts)}"]
        parts += ['#include "mpx_assembly_fused.h"', f"MPX_INSTANTIATE_FUSED({len(funcs)})"]
        return "\n".join(parts) + "\n"

    # -- expansion of the chain rule into gather rows -------------------------------------------------
    @staticmethod
    def _ell_width(nt):
        """ELL width of the multi-term rows of a pass (the smallest width <= 12 that leaves at most 16 wider rows to the
        one-wavefront-per-row path; 0: no multi-term rows) -- the threshold between 'multi' and 'long' rows is this, or 24 without it."""
        multi = nt[(nt >= 2) & (nt <= 24)]
        return next((w for w in range(2, 13) if (multi > w).sum() <= 16), 12) if len(multi) else 0

    def _class_major_jac(self):
        """Order the entries of jac_g by ROW CLASS (round 6): first the rows with at most one term -- constants of the linear part and
        single products, 85-95 % of a Jacobian --, then the rows with 2 .. MT terms, then the long rows.  The order of a pattern is the
        context's to choose (jac_pattern / ccs_perm report it).  Why: the fused kernel keeps a lane's single-term rows in registers and
        stores them 8 bytes per lane; with the classes interleaved a lane could not own two ADJACENT single-term rows, so pairing rows
        for 16-byte stores (MPX_FUSE_PAIR_ROWS) met a multi-term row in every few pairs and lost (60.2 against 56.8 us, round 4).  With
        the classes contiguous every pair of the first region is complete.  MEASURED (round 6, tools/r6_asm_pair_ab.sh, moon lander 20x5,
        B = 4096): no gain either -- 57.0-59.5 us as it was, 58.3-60.4 with this order, 60.5-62.5 with this order and paired rows:
        the pass is not bound by the width of its stores.  Off by default (MPX_ASM_CLASS_MAJOR=1)."""
        ptr, src, coef = self.fgj
        first = 1 + self.n_g_ + self.n_z_
        nt = np.diff(ptr)
        mt = self._ell_width(nt)
        thr = mt if mt >= 2 else 24
        ntj = nt[first:]
        cls = np.where(ntj <= 1, 0, np.where(ntj <= thr, 1, 2))
        order = np.argsort(cls, kind="stable")  # (inside a class the column-major order of the pattern stays)
        if np.array_equal(order, np.arange(len(order))):
            return
        full = np.concatenate([np.arange(first), first + order])
        new_ptr = np.concatenate([[0], np.cumsum(nt[full])]).astype(ptr.dtype)
        take = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in full]) if len(src) else np.zeros(0, np.int64)
        self.fgj = (new_ptr, np.ascontiguousarray(src[take]), np.ascontiguousarray(coef[take]))
        self.jrow, self.jcol = np.ascontiguousarray(self.jrow[order]), np.ascontiguousarray(self.jcol[order])

    def _expand(self, Gz, g0):
        n_z, n_g = self.n_z_, self.n_g_
        off = offh = 0
        self.raw_off, self.rawh_off = [], []
        for s in self.sets:
            self.raw_off.append(off), self.rawh_off.append(offh)
            off += s.n * (s.fn.n_out + s.fn.n_jac)
            offh += s.n * s.fn.n_hess
        R, S, C = [], [], []  # rows f, g, grad of the first-order gather
        JK, JS, JC = [Gz.row.astype(np.int64) * n_z + Gz.col], [np.full(Gz.nnz, SRC_ONE, np.int64)], [Gz.data]
        R.append(1 + Gz.row); S.append(src_z(Gz.col)); C.append(Gz.data)
        nz0 = np.nonzero(g0)[0]
        R.append(1 + nz0); S.append(np.full(len(nz0), SRC_ONE)); C.append(g0[nz0])
        HK, HS, HC = [], [], []
        for k, s in enumerate(self.sets):
            fn, n, o = s.fn, s.n, self.raw_off[k]
            for r in range(fn.n_out):
                w = s.fw[:, r]
                nzp = np.nonzero(w)[0]
                R.append(np.zeros(len(nzp), np.int64)); S.append(o + r * n + nzp); C.append(w[nzp])
                gp, gi, gc = s._gcols(r)
                R.append(1 + gi); S.append(o + r * n + gp); C.append(gc)
            for q, (r, v, _) in enumerate(fn.J):
                slot = o + (fn.n_out + q) * n
                cnt, lp, lc, lv = s._rows(v)  # terms of the local variable, grouped by point
                # gradient:  fw[p, r] * dphi_r/dv * L[v, c]
                w = s.fw[lp, r]
                m = w != 0
                R.append(1 + n_g + lc[m]); S.append(slot + lp[m]); C.append(w[m] * lv[m])
                # Jacobian:  G[i, (p, r)] * dphi_r/dv * L[(p, v), c]   (join on the point)
                gp, gi, gc = s._gcols(r)
                if len(gp) and len(lp):
                    first = np.concatenate([[0], np.cumsum(cnt)[:-1]])
                    rep = cnt[gp]
                    gidx = np.repeat(np.arange(len(gp)), rep)
                    within = np.arange(rep.sum()) - np.repeat(np.concatenate([[0], np.cumsum(rep)[:-1]]), rep)
                    li = first[gp][gidx] + within
                    JK.append(gi[gidx] * n_z + lc[li]); JS.append(slot + gp[gidx]); JC.append(gc[gidx] * lv[li])
            oh = self.rawh_off[k]
            for q, (v1, v2, _) in enumerate(fn.H):
                slot = oh + q * n
                c1, p1, i1, a1 = s._rows(v1)
                c2, p2, i2, a2 = s._rows(v2)
                if not (len(p1) and len(p2)):
                    continue
                f2 = np.concatenate([[0], np.cumsum(c2)[:-1]])
                rep = c2[p1]  # every term of v1 at point p pairs with all terms of v2 at p
                t1 = np.repeat(np.arange(len(p1)), rep)
                within = np.arange(rep.sum()) - np.repeat(np.concatenate([[0], np.cumsum(rep)[:-1]]), rep)
                t2 = f2[p1][t1] + within
                ca, cb, w = i1[t1], i2[t2], a1[t1] * a2[t2]
                if v1 == v2:
                    keep = ca <= cb
                    ca, cb, w, pp = ca[keep], cb[keep], w[keep], p1[t1][keep]
                else:
                    w = np.where(ca == cb, 2.0 * w, w)
                    ca, cb, pp = np.minimum(ca, cb), np.maximum(ca, cb), p1[t1]
                HK.append(ca * n_z + cb); HS.append(slot + pp); HC.append(w)
        # Jacobian / Hessian patterns: distinct (row, col) keys.  Entry ORDER is ours to choose (mpx_ccs_perm maps it to
        # CasADi's compressed-column order): entries are sorted by the raw slot their first term reads -- set-major,
        # slot-major, point-minor, exactly the layout the point kernels write -- so consecutive lanes of the gather kernel
        # read consecutive raw values and write consecutive outputs; constants (no raw source) come first.
        def pattern(K, S_, C_):
            K, S_, C_ = _cat(K, np.int64), _cat(S_, np.int64), _cat(C_, np.float64)
            rows, cols = K // n_z, K % n_z
            colmajor = cols * (max(n_g, n_z) + 1) + rows
            uniq, first, inv = np.unique(colmajor, return_index=True, return_inverse=True)
            src0 = np.full(len(uniq), np.iinfo(np.int64).max)
            np.minimum.at(src0, inv, np.where(S_ >= 0, S_, -1))  # smallest source of the entry (-1: has a constant / z term)
            order = np.lexsort((uniq, src0))                      # by source, ties in column-major order
            rank = np.empty(len(uniq), np.int64)
            rank[order] = np.arange(len(uniq))
            return rows[first][order].astype(np.int32), cols[first][order].astype(np.int32), rank[inv], S_, C_

        self.jrow, self.jcol, jinv, JS, JC = pattern(JK, JS, JC)
        self.hrow, self.hcol, hinv, HS, HC = pattern(HK, HS, HC)
        self.nnz_jac_, self.nnz_hess_ = len(self.jrow), len(self.hrow)
        rows = np.concatenate([_cat(R, np.int64), 1 + n_g + n_z + jinv])
        self.fgj = _csr_from_terms(1 + n_g + n_z + self.nnz_jac_, rows, np.concatenate([_cat(S, np.int64), JS]), np.concatenate([_cat(C, np.float64), JC]))
        self.hess = _csr_from_terms(self.nnz_hess_, hinv, HS, HC)
        self.raw_n, self.rawh_n = off, offh

    # -- context ----------------------------------------------------------------------------------------
    def _create(self, device):
        L = _lib.lib()
        keep = self._keep = []  # arrays referenced by the descriptor during mpx_create_assembled

        def i32(a):
            a = np.ascontiguousarray(a, dtype=np.int32)
            keep.append(a)
            return a.ctypes.data_as(_lib.c_int32_p)

        def f64(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data_as(_lib.c_double_p)

        def i64(a):
            a = np.ascontiguousarray(a, dtype=np.int64)
            keep.append(a)
            return a.ctypes.data_as(_lib.c_int64_p)

        arr = (mpx_point_set * len(self.sets))()
        for k, s in enumerate(self.sets):
            fn, d = s.fn, arr[k]
            d.fid, d.n_points = self.functions.index(fn), s.n
            d.n_loc, d.n_cst, d.n_out, d.n_jac, d.n_hess = fn.n_loc, fn.n_cst, fn.n_out, fn.n_jac, fn.n_hess
            (nt, ix, cf), (mnt, mix, mcf) = self._ell[k]
            d.loc_nterm, d.loc_idx, d.loc_coef = i32(nt), i32(ix), f64(cf)
            d.cst = f64(s.cst.T.copy() if s.cst.size else np.zeros(1))
            d.mu_nterm, d.mu_idx, d.mu_coef = i32(mnt), i32(mix), f64(mcf)
        D = mpx_assembly()
        D.version = 1
        D.n_z, D.n_g, D.nnz_jac, D.nnz_hess = self.n_z_, self.n_g_, self.nnz_jac_, self.nnz_hess_
        D.n_sets, D.sets = len(self.sets), arr
        for dst, (ptr, src, coef) in ((D.fgj, self.fgj), (D.hess, self.hess)):
            dst.n_rows, dst.ptr, dst.src, dst.coef = len(ptr) - 1, i64(ptr), i32(src if len(src) else np.zeros(1)), f64(coef if len(coef) else np.zeros(1))
        D.jac_row, D.jac_col, D.hess_row, D.hess_col = i32(self.jrow), i32(self.jcol), i32(self.hrow), i32(self.hcol)
        if self.code_object is not None:
            self._co_buf = ctypes.create_string_buffer(self.code_object, len(self.code_object))
            D.code_object = ctypes.cast(self._co_buf, ctypes.c_void_p)
            D.code_object_size = len(self.code_object)
        D.device = int(device)
        ctx = ctypes.c_void_p()
        rc = L.mpx_create_assembled(ctypes.byref(D), ctypes.byref(ctx))
        if rc != 0:
            raise MpxError(f"mpx_create_assembled failed ({rc}): {L.mpx_last_error(None).decode()}")
        self._adopt(ctx, L)
        self._keep = None
