"""``NlpFunctions``: the NLP oracle functions of a transcribed OCP, evaluated on the GPU.

This object stands where ``ca.nlpsol`` keeps ``nlp_f, nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l, nlp_grad``
after differentiating mpopt's ``{"f","x","g","p"}`` dict (mpopt.py:757).  It owns one ``mpx_ctx``
(include/mpx.h) and adds a batch dimension: every call evaluates ``B`` points in one launch.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import MPX_CCS_ORDER, MPX_F, MPX_G, MPX_GRAD, MPX_JAC, MPX_HESS, MpxError, mpx_problem, mpx_sizes
from .codegen import ProblemProgram


def _ptr(x):
    """void* of a numpy array, a torch tensor, an int device pointer or None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if isinstance(x, int):
        return x
    return x.data_ptr()  # torch tensor


class ResidualPlan:
    """Target points of one phase for the off-node evaluation (include/mpx.h, mpx_resid_*): what
    ``mpopt.interpolate_single_phase`` / ``get_dynamics_residuals_single_phase`` compute in the reference
    (mpopt.py:1428-1543), on the GPU and batched."""

    FIELDS = ("ti", "xi", "ui", "dxi", "dui", "dyn", "resid")

    def __init__(self, oracle, phase, taus_per_segment, deriv_order=1):
        """``deriv_order=2``: ``dxi`` / ``dui`` are the second derivatives of the interpolating polynomials
        (mpopt.py:1316-1321); ``dyn`` / ``resid`` are then meaningless."""
        self.oracle, self.phase, self.deriv_order = oracle, int(phase), int(deriv_order)
        taus = [np.ascontiguousarray(np.asarray(t, dtype=np.float64).ravel()) for t in taus_per_segment]
        if len(taus) != oracle.n_segments:
            raise ValueError("one array of target points per segment is required")
        self.seg_ptr = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(t) for t in taus])]), dtype=np.int64)
        self.taus = np.ascontiguousarray(np.concatenate(taus) if len(taus) else np.zeros(0), dtype=np.float64)
        if self.taus.size == 0:
            self.taus = np.zeros(1)
        self.n_pts = int(self.seg_ptr[-1])
        h = ctypes.c_void_p()
        _lib.check(oracle._L.mpx_resid_plan_create_order(oracle._ctx, self.phase, self.seg_ptr.ctypes.data_as(_lib.c_int64_p),
                                                         _lib.dptr(self.taus), self.deriv_order, ctypes.byref(h)), oracle._ctx)
        self._h = h
        o = oracle.ocp
        self.widths = {"ti": 0, "xi": o.nx, "ui": o.nu, "dxi": o.nx, "dui": o.nu, "dyn": o.nx, "resid": o.nx}

    def eval(self, z, p, what=FIELDS):
        """Host arrays; z (n_z,) or (B, n_z).  Returns dict name -> array (B?, n_pts[, width])."""
        orc = self.oracle
        z = np.ascontiguousarray(z, dtype=np.float64)
        single = z.ndim == 1
        z = z.reshape(-1, orc.n_z)
        B = z.shape[0]
        p = np.ascontiguousarray(p, dtype=np.float64)
        per_point = int(p.size == B * orc.n_p and B > 1)
        out, ptrs = {}, []
        for name in self.FIELDS:
            if name in what and (self.widths[name] or name == "ti"):
                shape = (B, self.n_pts) if name == "ti" else (B, self.n_pts, self.widths[name])
                out[name] = np.zeros(shape)
                ptrs.append(out[name].ctypes.data if out[name].size else None)
            else:
                ptrs.append(None)
        _lib.check(orc._L.mpx_resid_eval(orc._ctx, self._h, B, z.ctypes.data, p.ctypes.data, per_point, *ptrs), orc._ctx)
        return {k: (v[0] if single else v) for k, v in out.items()}

    def eval_device(self, batch, z, p, p_per_point=0, **outs):
        orc = self.oracle
        ptrs = [_ptr(outs.get(name)) for name in self.FIELDS]
        _lib.check(orc._L.mpx_resid_eval_device(orc._ctx, self._h, int(batch), _ptr(z), _ptr(p), int(p_per_point), *ptrs), orc._ctx)

    def split(self, arr):
        """Per-segment list (``None`` for empty segments), like the reference's return values."""
        return [arr[self.seg_ptr[s]:self.seg_ptr[s + 1]] if self.seg_ptr[s + 1] > self.seg_ptr[s] else None
                for s in range(len(self.seg_ptr) - 1)]

    def close(self):
        if getattr(self, "_h", None):
            self.oracle._L.mpx_resid_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            if getattr(self.oracle, "_ctx", None):
                self.close()
        except Exception:
            pass


class NlpFunctions:
    def __init__(self, ocp, n_segments, poly_orders, scheme="LGR", tau0=-1.0, tau1=1.0, midu_rows=None,
                 device=0, with_device=None, verbose=False):
        L = _lib.lib()
        self.ocp = ocp
        self.n_segments = int(n_segments)
        self.poly_orders = np.ascontiguousarray(poly_orders, dtype=np.int32)
        if len(self.poly_orders) != self.n_segments:
            raise ValueError("poly_orders must have one entry per segment")
        if scheme not in _lib.SCHEMES or scheme == "LG":
            raise _lib.MpxSchemeError(f"scheme {scheme!r} is not usable for transcription (needs degree+1 nodes; LGR, LGL, CGL) -- the reference "
                                      "raises ValueError here as well: cannot reshape array of size p^2 into shape (p+1, p+1), mpopt.py:4032-4038")
        self.scheme = scheme
        if midu_rows is None:
            midu_rows = [bool(ocp.midu[ph]) and bool((np.asarray(ocp.lbu[ph]) > -np.inf).any()
                                                     or (np.asarray(ocp.ubu[ph]) < np.inf).any())
                         for ph in range(ocp.n_phases)]
        self.program = ProblemProgram(ocp, self.poly_orders, midu_rows)
        self.structure = np.ascontiguousarray(self.program.structure(), dtype=np.int32)
        self.source = self.program.source()
        if with_device is None:
            with_device = _lib.gpu_available()
        self.code_object = None
        if with_device:
            self.code_object, self.code_object_path = _lib.compile_kernels(self.source, verbose=verbose)
        links = np.ascontiguousarray(np.asarray(ocp.phase_links, dtype=np.int32).reshape(-1))
        prob = mpx_problem()
        prob.version = 1
        prob.n_phases, prob.nx, prob.nu, prob.na = ocp.n_phases, ocp.nx, ocp.nu, ocp.na
        prob.n_segments = self.n_segments
        prob.poly_orders = self.poly_orders.ctypes.data_as(_lib.c_int32_p)
        prob.scheme = _lib.SCHEMES[scheme]
        prob.tau0, prob.tau1 = float(tau0), float(tau1)
        prob.n_links = len(links) // 2
        prob.links = links.ctypes.data_as(_lib.c_int32_p)
        prob.structure = self.structure.ctypes.data_as(_lib.c_int32_p)
        prob.structure_len = len(self.structure)
        if self.code_object is not None:
            self._co_buf = ctypes.create_string_buffer(self.code_object, len(self.code_object))
            prob.code_object = ctypes.cast(self._co_buf, ctypes.c_void_p)
            prob.code_object_size = len(self.code_object)
        prob.device = int(device)
        ctx = ctypes.c_void_p()
        rc = L.mpx_create(ctypes.byref(prob), ctypes.byref(ctx))
        if rc != 0:
            raise MpxError(f"mpx_create failed ({rc}): {L.mpx_last_error(None).decode()}")
        self._adopt(ctx, L)

    def _adopt(self, ctx, L):
        """Take ownership of a created context and read its sizes."""
        self._ctx = ctx
        self._L = L
        s = mpx_sizes()
        _lib.check(L.mpx_get_sizes(ctx, ctypes.byref(s)), ctx)
        self.sizes = s
        self.n_z, self.n_p, self.n_g = s.n_z, s.n_p, s.n_g
        self.nnz_jac, self.nnz_hess, self.n_nodes, self.n_tiles = s.nnz_jac, s.nnz_hess, s.n_nodes, s.n_tiles
        self.bytes_fgj, self.bytes_hess = s.bytes_fgj, s.bytes_hess
        self._jac_pat = self._hess_pat = None
        self._pinned = {}

    def pinned_empty(self, shape):
        """float64 numpy array over page-locked host memory owned by this context (mpx_host_alloc)."""
        n = int(np.prod(shape))
        ptr = ctypes.c_void_p()
        _lib.check(self._L.mpx_host_alloc(self._ctx, max(n, 1) * 8, ctypes.byref(ptr)), self._ctx)
        buf = (ctypes.c_double * max(n, 1)).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=np.float64, count=n).reshape(shape)
        self._pinned.setdefault("_owned", []).append(ptr)
        return arr

    def _pinned_buf(self, key, shape):
        cur = self._pinned.get(key)
        if cur is None or cur.shape != tuple(shape):
            cur = self._pinned[key] = self.pinned_empty(tuple(shape))
        return cur

    def make_current(self):
        """Select this context for the CasADi-external entry points (nlp_f ... nlp_hess_l in libmpx.so):
        ``ca.nlpsol("solver", "ipopt", mpopt_amd._lib.LIB_PATH, opts)`` then evaluates on the GPU."""
        _lib.check(self._L.mpx_set_current(self._ctx), self._ctx)
        NlpFunctions._current = self

    def close(self):
        if getattr(NlpFunctions, "_current", None) is self:
            self._L.mpx_set_current(None)
            NlpFunctions._current = None
        if getattr(self, "_ctx", None):
            for ptr in self._pinned.pop("_owned", []):
                self._L.mpx_host_free(self._ctx, ptr)
            self._pinned = {}
            self._L.mpx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- structure -------------------------------------------------------------------------
    @property
    def has_device(self):
        return self.code_object is not None

    def jac_pattern(self):
        if self._jac_pat is None:
            r, c = np.empty(self.nnz_jac, np.int32), np.empty(self.nnz_jac, np.int32)
            _lib.check(self._L.mpx_pattern_jac(self._ctx, r.ctypes.data_as(_lib.c_int32_p), c.ctypes.data_as(_lib.c_int32_p)), self._ctx)
            self._jac_pat = (r, c)
        return self._jac_pat

    def hess_pattern(self):
        if self._hess_pat is None:
            r, c = np.empty(self.nnz_hess, np.int32), np.empty(self.nnz_hess, np.int32)
            _lib.check(self._L.mpx_pattern_hess(self._ctx, r.ctypes.data_as(_lib.c_int32_p), c.ctypes.data_as(_lib.c_int32_p)), self._ctx)
            self._hess_pat = (r, c)
        return self._hess_pat

    def ccs_perm(self, which="jac"):
        nnz = self.nnz_jac if which == "jac" else self.nnz_hess
        perm, colind = np.empty(nnz, np.int64), np.empty(self.n_z + 1, np.int64)
        _lib.check(self._L.mpx_ccs_perm(self._ctx, MPX_JAC if which == "jac" else MPX_HESS,
                                        perm.ctypes.data_as(_lib.c_int64_p), colind.ctypes.data_as(_lib.c_int64_p)), self._ctx)
        return perm, colind

    def comp_weights(self):
        w = np.empty(self.n_nodes)
        _lib.check(self._L.mpx_get_comp_weights(self._ctx, _lib.dptr(w)), self._ctx)
        return w

    def residual_plan(self, phase, taus_per_segment, deriv_order=1):
        return ResidualPlan(self, phase, taus_per_segment, deriv_order)

    # -- evaluation ------------------------------------------------------------------------
    def eval(self, what, z, p, lam_g=None, sigma=None, pinned=False, ccs_order=False):
        """Host arrays in, dict of host arrays out.  ``z``: (B, n_z) or (n_z,); ``p``: (n_p,) shared
        or (B, n_p).  ``what``: iterable of {"f","g","grad_f","jac_g","hess_l"}.
        ``pinned=True``: inputs are staged through, and outputs are *views of*, page-locked buffers owned
        by this object (true DMA transfers, about half the latency of a single evaluation); the returned
        arrays are overwritten by the next ``eval(..., pinned=True)`` of the same shape.
        ``ccs_order=True``: ``jac_g`` / ``hess_l`` values leave in compressed-column order (``ccs_perm``), permuted
        on the device (MPX_CCS_ORDER)."""
        mask = (MPX_CCS_ORDER if ccs_order else 0) + sum({"f": MPX_F, "g": MPX_G, "grad_f": MPX_GRAD, "jac_g": MPX_JAC, "hess_l": MPX_HESS}[w] for w in set(what))
        z = np.ascontiguousarray(z, dtype=np.float64)
        single = z.ndim == 1
        z = z.reshape(-1, self.n_z)
        B = z.shape[0]
        p = np.zeros(0) if (p is None or self.n_p == 0) else np.ascontiguousarray(p, dtype=np.float64)
        if p.size not in (self.n_p, B * self.n_p):
            raise ValueError(f"p has {p.size} values, expected {self.n_p} or {B}x{self.n_p}")
        per_point = int(p.size == B * self.n_p and B > 1)
        out = {}
        new = (lambda key, shape: self._pinned_buf(key, shape)) if pinned else (lambda key, shape: np.empty(shape))
        f = new("f", (B,)) if mask & MPX_F else None
        g = new("g", (B, self.n_g)) if mask & MPX_G else None
        gr = new("grad", (B, self.n_z)) if mask & MPX_GRAD else None
        jv = new("jac", (B, self.nnz_jac)) if mask & MPX_JAC else None
        hv = lam = sig = None
        if mask & MPX_HESS:
            lam = np.ascontiguousarray(np.broadcast_to(np.asarray(lam_g, dtype=np.float64).reshape(-1, self.n_g), (B, self.n_g)))
            sig = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, dtype=np.float64).reshape(-1), (B,)))
            hv = new("hess", (B, self.nnz_hess))
        if pinned:
            zin = self._pinned_buf("z", z.shape)
            zin[...] = z
            z = zin
            if lam is not None:
                lin = self._pinned_buf("lam", lam.shape)
                lin[...] = lam
                lam = lin
        rc = self._L.mpx_eval(self._ctx, mask, B, _ptr(z), _ptr(p), per_point, _ptr(lam), _ptr(sig), _ptr(f), _ptr(g),
                              _ptr(gr), _ptr(jv), _ptr(hv))
        _lib.check(rc, self._ctx)
        for k, v in (("f", f), ("g", g), ("grad_f", gr), ("jac_g", jv), ("hess_l", hv)):
            if v is not None:
                out[k] = v[0] if single else v
        return out

    def eval_device(self, mask, batch, z, p, p_per_point=0, lam_g=None, sigma=None, f=None, g=None, grad_f=None,
                    jac_val=None, hess_val=None):
        """Device pointers (torch tensors or ints) in and out, asynchronous on the context stream."""
        rc = self._L.mpx_eval_device(self._ctx, int(mask), int(batch), _ptr(z), _ptr(p), int(p_per_point), _ptr(lam_g),
                                     _ptr(sigma), _ptr(f), _ptr(g), _ptr(grad_f), _ptr(jac_val), _ptr(hess_val))
        _lib.check(rc, self._ctx)

    def eval_grad_gamma(self, z, p, lam_g, sigma, what=("grad_gamma_x", "grad_gamma_p")):
        """``nlp_grad``: gradient of gamma = sigma * f + lam_g^T g w.r.t. x and w.r.t. the parameters p (mpx_eval_grad_gamma).
        Host arrays; ``z`` (n_z,) or (B, n_z), ``lam_g`` / ``sigma`` broadcast over the batch.  Returns a dict."""
        z = np.ascontiguousarray(z, dtype=np.float64)
        single = z.ndim == 1
        z = z.reshape(-1, self.n_z)
        B = z.shape[0]
        p = np.zeros(0) if (p is None or self.n_p == 0) else np.ascontiguousarray(p, dtype=np.float64)
        if p.size not in (self.n_p, B * self.n_p):
            raise ValueError(f"p has {p.size} values, expected {self.n_p} or {B}x{self.n_p}")
        per_point = int(p.size == B * self.n_p and B > 1)
        lam = np.ascontiguousarray(np.broadcast_to(np.asarray(lam_g, dtype=np.float64).reshape(-1, self.n_g), (B, self.n_g)))
        sig = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, dtype=np.float64).reshape(-1), (B,)))
        gx = np.empty((B, self.n_z)) if "grad_gamma_x" in what else None
        gp = np.empty((B, self.n_p)) if "grad_gamma_p" in what else None
        rc = self._L.mpx_eval_grad_gamma(self._ctx, B, _ptr(z), _ptr(p), per_point, _ptr(lam), _ptr(sig), _ptr(gx),
                                         _ptr(gp) if self.n_p else None)
        _lib.check(rc, self._ctx)
        out = {}
        if gx is not None:
            out["grad_gamma_x"] = gx[0] if single else gx
        if gp is not None:
            out["grad_gamma_p"] = gp[0] if single else gp
        return out

    def eval_grad_gamma_device(self, batch, z, p, lam_g, sigma, grad_gamma_x=None, grad_gamma_p=None, p_per_point=0):
        """Device pointers (torch tensors or ints), asynchronous on the context stream (mpx_eval_grad_gamma_device)."""
        rc = self._L.mpx_eval_grad_gamma_device(self._ctx, int(batch), _ptr(z), _ptr(p), int(p_per_point), _ptr(lam_g), _ptr(sigma),
                                                _ptr(grad_gamma_x), _ptr(grad_gamma_p))
        _lib.check(rc, self._ctx)

    def alloc_outputs(self, mask, batch, z, p, p_per_point=0, lam_g=None, sigma=None, tries=16, target_us=None):
        """Output arrays for ``eval_device(mask, batch, ...)`` -- torch tensors f [batch], g [batch, n_g], grad_f [batch, n_z],
        jac_val [batch, nnz_jac], hess_val [batch, nnz_hess] for the outputs ``mask`` names, ``None`` for the others -- placed by
        MEASUREMENT: the node kernels stream into several GB of output per pass, and how fast the HBM controllers drain those
        writes depends on where the driver put the pages (the same kernel on the same box runs 850 us into
        one allocation and 1050-1100 us into the next; nothing user code can choose -- not the kind of allocation, not the offset
        inside a larger one: tools/alloc_kind_probe.py, alloc_slide_probe.py, alloc_order_probe.py).  So up to ``tries`` candidate
        sets are allocated (all held until the end: a freed slow placement would be handed out again), each is timed with the real
        inputs, the search stops early once a candidate is 8 % faster than the slowest seen (placements come in two states), the
        fastest is returned and the rest is freed.  A one-time cost of a few passes per candidate at set-up; results do not depend on it.
        ``target_us`` (round 5): the node-kernel time per pass the caller knows a well-placed set reaches (e.g. the algorithmic bytes of
        the pass over 0.95 of the measured HBM copy rate for the metric's kernel): the search then stops at the first candidate within
        3 % of it and otherwise keeps drawing up to ``tries`` (default 16 since round 5: with one fast placement in six -- the driver's
        round-4 box -- six draws missed it every third time).  Without a target the relative rule alone decides.
        Returns ``(outputs, report)``; report = node-kernel microseconds per pass of every candidate, the index kept, ``tries_used``,
        ``stopped_by`` ("target", "relative", "tries", "memory")."""
        import torch

        from ._lib import MPX_F, MPX_G, MPX_GRAD, MPX_HESS, MPX_JAC

        dev = z.device
        shapes = ((MPX_F, (batch,)), (MPX_G, (batch, self.n_g)), (MPX_GRAD, (batch, self.n_z)), (MPX_JAC, (batch, self.nnz_jac)),
                  (MPX_HESS, (batch, self.nnz_hess)))
        cands, times = [], []
        need = 8 * sum(int(np.prod(sh)) for bit, sh in shapes if mask & bit)
        stopped = "tries"
        for _ in range(max(1, int(tries))):
            if cands and torch.cuda.mem_get_info(dev)[0] < 1.25 * need:
                stopped = "memory"
                break  # (the candidates are all held until the end: never search the device out of memory)
            outs = [torch.empty(sh, dtype=torch.float64, device=dev) if mask & bit else None for bit, sh in shapes]
            for _ in range(8):  # (the first six passes into new arrays are the library's own geometry measurement, include/mpx.h)
                self.eval_device(mask, batch, z, p, p_per_point, lam_g, sigma, *outs)
            self.sync()
            self.profile(True)
            for _ in range(6):
                self.eval_device(mask, batch, z, p, p_per_point, lam_g, sigma, *outs)
            ms, n = self.profile_read()
            self.profile(False)
            cands.append(outs)
            times.append(ms * 1e3 / 6.0)  # node kernels of one pass (all degree buckets)
            if target_us is not None:
                if times[-1] <= 1.03 * target_us:
                    stopped = "target"
                    break
            elif len(times) > 1 and times[-1] <= 0.92 * max(times):
                stopped = "relative"
                break
        best = min(range(len(times)), key=times.__getitem__)
        keep = cands[best]
        del cands, outs
        torch.cuda.empty_cache()
        return keep, {"node_us_per_pass": [round(t, 1) for t in times], "kept": best, "tries_used": len(times), "stopped_by": stopped,
                      "target_us": None if target_us is None else round(float(target_us), 1)}

    def geometry_reset(self):
        """Void the launch-geometry measurements (call after re-allocating output arrays; include/mpx.h)."""
        _lib.check(self._L.mpx_geometry_reset(self._ctx), self._ctx)

    def set_mid_resid_output(self, resid):
        """Device array [batch][n_phases * (N - 1)][nx] the MPX_MID_RESID passes write (mpx_set_mid_resid_output); None: off."""
        _lib.check(self._L.mpx_set_mid_resid_output(self._ctx, ctypes.c_void_p(_ptr(resid))), self._ctx)

    def set_stream(self, stream):
        _lib.check(self._L.mpx_set_stream(self._ctx, ctypes.c_void_p(int(stream) if stream else None)), self._ctx)

    def sync(self):
        _lib.check(self._L.mpx_sync(self._ctx), self._ctx)

    def timer_start(self):
        _lib.check(self._L.mpx_timer_start(self._ctx), self._ctx)

    def timer_stop(self):
        ms = ctypes.c_double()
        _lib.check(self._L.mpx_timer_stop(self._ctx, ctypes.byref(ms)), self._ctx)
        return ms.value

    def profile(self, enable=True):
        _lib.check(self._L.mpx_profile(self._ctx, int(bool(enable))), self._ctx)

    def profile_read(self):
        """(summed node-kernel milliseconds, number of node-kernel launches) since the last read."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        _lib.check(self._L.mpx_profile_read(self._ctx, ctypes.byref(ms), ctypes.byref(n)), self._ctx)
        return ms.value, n.value

    def set_tile_range(self, begin, end, run_boundary=True):
        _lib.check(self._L.mpx_set_tile_range(self._ctx, int(begin), int(end), int(bool(run_boundary))), self._ctx)

    # -- segment sharding (include/mpx.h, mpx_shard_*) --------------------------------------
    def shard_setup(self, world, rank):
        """Put the context into segment-sharded mode for ``world`` ranks (``world == 1`` leaves it)."""
        _lib.check(self._L.mpx_shard_setup(self._ctx, int(world), int(rank)), self._ctx)
        self._shard_world = int(world)

    def shard_info(self, mask):
        """(rank_len, tile_cuts): padded per-rank, per-point length of the exchange buffer in doubles; tile ranges."""
        n, ne = ctypes.c_int64(), ctypes.c_int64()
        cuts = np.zeros(getattr(self, "_shard_world", 1) + 1, np.int64)
        _lib.check(self._L.mpx_shard_info(self._ctx, int(mask), ctypes.byref(n), ctypes.byref(ne), cuts.ctypes.data_as(_lib.c_int64_p)), self._ctx)
        return n.value, cuts

    def shard_table(self, mask):
        """int64 array [n_entries][6] = (rank, kind, offset, length, stride, packed_offset) of every owned run."""
        n, ne = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._L.mpx_shard_info(self._ctx, int(mask), ctypes.byref(n), ctypes.byref(ne), None), self._ctx)
        out = np.zeros((ne.value, 6), np.int64)
        _lib.check(self._L.mpx_shard_table(self._ctx, int(mask), out.ctypes.data_as(_lib.c_int64_p)), self._ctx)
        return out

    def shard_owned(self, which, rank):
        """Owner-resident sharding: int64 array [n_runs][2] of the (offset, length) runs of output ``which`` ("g", "grad_f",
        "jac_g", "hess_l") that ``rank`` owns after an evaluation (mpx_shard_owned); what no rank owns is written by the
        boundary pass on every rank."""
        w = {"g": MPX_G, "grad_f": MPX_GRAD, "jac_g": MPX_JAC, "hess_l": MPX_HESS}[which]
        n = ctypes.c_int64()
        _lib.check(self._L.mpx_shard_owned(self._ctx, w, int(rank), ctypes.byref(n), None), self._ctx)
        out = np.zeros((n.value, 2), np.int64)
        if n.value:
            _lib.check(self._L.mpx_shard_owned(self._ctx, w, int(rank), ctypes.byref(n), out.ctypes.data_as(_lib.c_int64_p)), self._ctx)
        return out

    def shard_pack(self, mask, batch, vals, send):
        _lib.check(self._L.mpx_shard_pack(self._ctx, int(mask), int(batch), _ptr(vals), _ptr(send)), self._ctx)

    def shard_unpack(self, mask, batch, recv, vals):
        _lib.check(self._L.mpx_shard_unpack(self._ctx, int(mask), int(batch), _ptr(recv), _ptr(vals)), self._ctx)

    def equal_area_widths_device(self, phase, batch, n_pts, resid, p_in, p_out, damping=0.4, p_in_per_point=0):
        """Device-side equal-area width update of the h-adaptive loop (mpx_equal_area_widths_device)."""
        _lib.check(self._L.mpx_equal_area_widths_device(self._ctx, int(phase), int(batch), int(n_pts), _ptr(resid), _ptr(p_in),
                                                        int(p_in_per_point), _ptr(p_out), float(damping)), self._ctx)

    def tile_weights(self):
        w = np.empty(self.n_tiles, np.int64)
        _lib.check(self._L.mpx_get_tile_weights(self._ctx, w.ctypes.data_as(_lib.c_int64_p)), self._ctx)
        return w

    def tile_spans(self):
        """Per tile: (first node, length, foreign nodes) of the g / grad_f row span the tile stores itself on mixed-degree grids
        (mpx_get_tile_spans); all zero where the scheme does not apply."""
        a = [np.zeros(self.n_tiles, np.int32) for _ in range(3)]
        _lib.check(self._L.mpx_get_tile_spans(self._ctx, *[v.ctypes.data_as(_lib.c_int32_p) for v in a]), self._ctx)
        return tuple(a)

    def notes(self):
        """Planner decisions worth knowing (mpx_get_notes), a list of lines."""
        t = self._L.mpx_get_notes(self._ctx)
        return [ln for ln in (t.decode() if t else "").splitlines() if ln]

    def light_plan(self):
        """(degree, n_groups, max_span_nodes, n_low_degree_nodes) of the light-pass plan (mpx_get_light_plan); degree 0: none."""
        d, g, sp, nf = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._L.mpx_get_light_plan(self._ctx, ctypes.byref(d), ctypes.byref(g), ctypes.byref(sp), ctypes.byref(nf)), self._ctx)
        return d.value, g.value, sp.value, nf.value

    def partials(self, batch):
        """(device pointer, element count) of the per-tile partial-sum buffer for ``batch`` points."""
        ptr, cnt = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(self._L.mpx_get_partials(self._ctx, int(batch), ctypes.byref(ptr), ctypes.byref(cnt)), self._ctx)
        return ptr.value, cnt.value

    def tile_jac_range(self, tile):
        b, e = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._L.mpx_get_tile_jac_range(self._ctx, int(tile), ctypes.byref(b), ctypes.byref(e)), self._ctx)
        return b.value, e.value

    # -- the five oracles with CasADi's names/signatures (single point, host arrays) --------
    def nlp_f(self, x, p):
        return self.eval(["f"], x, p)["f"]

    def nlp_g(self, x, p):
        return self.eval(["g"], x, p)["g"]

    def nlp_grad_f(self, x, p):
        r = self.eval(["f", "grad_f"], x, p)
        return r["f"], r["grad_f"]

    def nlp_jac_g(self, x, p):
        r = self.eval(["g", "jac_g"], x, p)
        return r["g"], r["jac_g"]

    def nlp_hess_l(self, x, p, lam_f, lam_g):
        return self.eval(["hess_l"], x, p, lam_g=lam_g, sigma=lam_f)["hess_l"]

    def nlp(self, x, p):
        r = self.eval(["f", "g"], x, p)
        return r["f"], r["g"]

    def nlp_grad(self, x, p, lam_f, lam_g):
        """(f, g, grad_gamma_x, grad_gamma_p) -- what CasADi's Nlpsol evaluates once after the last iterate; ``lam_p`` of its result
        is ``-grad_gamma_p`` at ``lam_f = 1``."""
        r = self.eval(["f", "g"], x, p)
        q = self.eval_grad_gamma(x, p, lam_g, lam_f)
        return r["f"], r["g"], q["grad_gamma_x"], q["grad_gamma_p"]
