"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"nlp_{name}.npz"))


def build_case(name, with_device=None):
    import mpopt_amd as M
    from mpopt_amd import mp
    import problems

    if name in problems.ADAPTIVE_CASES:  # widths as variables: assembled context (mpopt_amd/adaptive.py)
        builder, S, po, scheme = problems.ADAPTIVE_CASES[name]
        ocp = builder(mp, M.math)
        mpo = mp.mpopt_adaptive(ocp, S, po, scheme)
        return ocp, mpo, mpo.create_nlp()[0]["oracle"]
    builder, S, po, scheme = problems.GOLDEN_CASES[name]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    mpo.compute_numerical_approximation()
    oracle = M.NlpFunctions(ocp, S, mpo.poly_orders, scheme, tau0=mpo.tau0, tau1=mpo.tau1, with_device=with_device)
    return ocp, mpo, oracle


def coo_to_dict(rows, cols, vals):
    d = {}
    for r, c, v in zip(rows.tolist(), cols.tolist(), np.asarray(vals).tolist()):
        assert (r, c) not in d, f"duplicate entry {(r, c)}"
        d[(r, c)] = v
    return d


def assert_coo_close(rows, cols, vals, rrows, rcols, rvals, rtol=1e-10, what=""):
    """Compare triplets: every reference entry must be present and equal; extra entries of ours
    must be explicit zeros (structural over-approximation, e.g. exact-zero D entries)."""
    mine = coo_to_dict(rows, cols, vals)
    ref = coo_to_dict(rrows, rcols, rvals)
    scale = max(1.0, max((abs(v) for v in ref.values()), default=1.0))
    for k, v in ref.items():
        assert k in mine, f"{what}: entry {k} missing"
        assert abs(mine[k] - v) <= rtol * max(scale, 1.0), f"{what}: entry {k}: {mine[k]} vs {v}"
    for k, v in mine.items():
        if k not in ref:
            assert abs(v) <= 1e-13 * scale, f"{what}: extra entry {k} = {v} is not a structural zero"


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / max(1.0, np.abs(b).max()) if a.size else 0.0


def entry_errors(a, b, floor=None):
    """Per-ENTRY relative error |a - b| / max(|b|, floor) of two arrays.  ``floor`` (scalar or array): the magnitude below which an
    entry is compared absolutely -- the scale of the terms the entry is a sum of (an entry that is a difference of large terms
    cannot be relatively accurate with respect to ITSELF).  Default: the median magnitude of the non-zero reference entries, i.e.
    the typical size of an entry of this class -- NOT the largest entry of the array."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.shape != b.shape:
        raise AssertionError(f"shape {a.shape} vs {b.shape}")
    if a.size == 0:
        return np.zeros(0), 1.0
    if floor is None:
        nz = np.abs(b[b != 0])
        floor = float(np.median(nz)) if nz.size else 1.0
    return np.abs(a - b) / np.maximum(np.abs(b), floor), floor


ENTRY_LOG = []  # (what, n, worst, floor) of every assert_entries call: tests/conftest.py prints the per-class summary at the end of the run


def assert_entries(a, b, tol=1e-10, floor=None, what="", report=True):
    """Every entry within ``tol`` relative (north_star: "within 1e-10 relative for FP64 residuals/derivatives"), see entry_errors.
    Prints the worst per-entry relative error of the class so that it lands in the test log (-s / GPUTEST)."""
    e, fl = entry_errors(a, b, floor)
    worst = float(e.max()) if e.size else 0.0
    if report:
        print(f"[entries] {what}: n={e.size} worst per-entry rel err {worst:.2e} (floor {np.max(fl):.2e})")
        ENTRY_LOG.append((what, int(e.size), worst, float(np.max(fl))))
    if not worst <= tol:  # (also catches NaN)
        k = int(np.nanargmax(e)) if not np.isnan(e).all() else 0
        raise AssertionError(f"{what}: entry {np.unravel_index(k, e.shape)}: {np.asarray(a).ravel()[k]!r} vs {np.asarray(b).ravel()[k]!r}, "
                             f"rel err {worst:.3e} > {tol:.1e} (floor {np.max(fl):.3e})")
    return worst


def align_coo(rows, cols, rrows, rcols, rvals, what=""):
    """Reference triplets aligned onto the pattern (rows, cols): array of len(rows) with the reference value of every entry (0
    where the reference has none -- our explicit structural zeros); every reference entry must exist in the pattern."""
    pos = {(int(r), int(c)): k for k, (r, c) in enumerate(zip(rows.tolist(), cols.tolist()))}
    assert len(pos) == len(rows), f"{what}: duplicate entries in the pattern"
    out = np.zeros(len(rows))
    for r, c, v in zip(np.asarray(rrows).tolist(), np.asarray(rcols).tolist(), np.asarray(rvals).tolist()):
        assert (r, c) in pos, f"{what}: reference entry {(r, c)} missing from the pattern"
        out[pos[(r, c)]] += v
    return out


def border_columns(o):
    """Indices of the (t0, tf, a) variables of every phase in z (the dense border of jac_g / hess_l, mpopt.py:537-543)."""
    nzp = o.n_z // o.ocp.n_phases
    k = (o.ocp.nx + o.ocp.nu) * o.n_nodes
    return np.concatenate([np.arange(ph * nzp + k, ph * nzp + k + 2 + o.ocp.na) for ph in range(o.ocp.n_phases)])


def assert_by_class(a, b, classes, tol=1e-10, what="", floors=None):
    """Per-entry parity of aligned value arrays, one floor per ENTRY CLASS (north_star: 1e-10 relative for FP64 derivatives):
    ``classes`` maps a class name to a boolean mask; every entry must belong to exactly one class.  Floor of a class = the median
    magnitude of its non-zero reference entries unless ``floors[name]`` says otherwise.  Returns {class: worst per-entry rel err}."""
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    cover = np.zeros(a.size, int)
    worst = {}
    for name, m in classes.items():
        m = np.asarray(m, bool).ravel()
        cover += m
        if m.any():
            worst[name] = assert_entries(a[m], b[m], tol, floor=(floors or {}).get(name), what=f"{what} [{name}]")
    assert (cover == 1).all(), f"{what}: classes do not partition the entries"
    return worst


def jac_classes(o, rows, cols, vals_ref, vals_ref_other):
    """Entry classes of jac_g in a given (rows, cols) order: 'constant' = copies of differentiation / interpolation table entries
    and the +-1 of linking rows (identical at two different evaluation points of the ORACLE), 'border' = the (t0, tf, a) columns,
    'variable' = everything else the kernels compute from (z, p)."""
    border = np.isin(cols, border_columns(o))
    const = (np.asarray(vals_ref) == np.asarray(vals_ref_other)) & ~border
    return {"constant (D / interpolation copies)": const, "border columns (t0, tf, a)": border, "variable node entries": ~const & ~border}


def hess_classes(o, rows, cols):
    bc = border_columns(o)
    rb, cb = np.isin(rows, bc), np.isin(cols, bc)
    return {"node bands": ~rb & ~cb, "border (node x (t0, tf, a))": rb ^ cb, "corner ((t0, tf, a) x (t0, tf, a))": rb & cb}


def grad_classes(o):
    m = np.zeros(o.n_z, bool)
    m[border_columns(o)] = True
    return {"node entries": ~m, "(t0, tf, a) entries (sums over all nodes)": m}


def emulate_shard_exchange(tables, rank_len, world, batch, arrays_by_rank, all_gather):
    """TEST INFRASTRUCTURE: host-side restatement of mpx_shard_pack / all-gather / mpx_shard_unpack over numpy arrays, from the
    table ``mpx_shard_table`` reports (CPU tests of the N>1 choreography; the product path uses the device kernels).
    ``arrays_by_rank[kind]`` is this rank's flat array of that kind; ``all_gather(send) -> recv[world * len(send)]``."""
    import numpy as np

    me = arrays_by_rank["rank"]
    send = np.zeros(max(rank_len * batch, 2))
    for r, kind, off, ln, stride, dst in tables:
        if r == me:
            for b in range(batch):
                send[dst * batch + b * ln: dst * batch + (b + 1) * ln] = arrays_by_rank[int(kind)][off + b * stride: off + b * stride + ln]
    recv = all_gather(send)
    n = len(send)
    for r, kind, off, ln, stride, dst in tables:
        if r != me:
            for b in range(batch):
                arrays_by_rank[int(kind)][off + b * stride: off + b * stride + ln] = recv[r * n + dst * batch + b * ln: r * n + dst * batch + (b + 1) * ln]


class HostShardOracle:
    """TEST INFRASTRUCTURE (CPU tests of ``SegmentShardedEvaluator`` at world sizes no test box has the GPUs for): stands in for the
    DEVICE calls of ``NlpFunctions`` -- eval_device / shard_pack / shard_unpack / sync -- over host torch tensors, driven by the REAL
    shard tables and ownership runs of a structure-only libmpx context (``struct``).  The "node kernels" copy what this rank owns
    out of seeded truth arrays (identical on every rank) and leave everything else NaN; pack / unpack restate mpx_shard_copy_kernel
    (mpx_host.cpp) from ``mpx_shard_table``; the "boundary pass" sums the tile partials of every point in slot order into f -- NaN
    unless every rank's slots arrived.  The product path never sees this class."""

    def __init__(self, struct, batch, seed=0):
        import torch
        from mpopt_amd._lib import MPX_HESS, MPX_JAC

        self.s, self.B, self.torch = struct, int(batch), torch
        self.rank = 0
        rng = np.random.default_rng(seed)
        o = struct
        self.nnz = {0: o.nnz_jac, 1: o.nnz_hess}
        self.truth = {"g": rng.standard_normal((batch, o.n_g)), "grad_f": rng.standard_normal((batch, o.n_z)),
                      "jac_g": rng.standard_normal((batch, o.nnz_jac)), "hess_l": rng.standard_normal((batch, o.nnz_hess))}
        self._rng = rng
        self.staging = self.partial = None
        self.calls = []

    # ---- structure: straight from libmpx -------------------------------------------------------------------------------------
    def shard_setup(self, world, rank):
        self.s.shard_setup(world, rank)
        self.world, self.rank = int(world), int(rank)

    def shard_info(self, mask):
        return self.s.shard_info(mask)

    def shard_owned(self, which, rank):
        return self.s.shard_owned(which, rank)

    def sync(self):
        pass

    def set_stream(self, stream):
        raise AssertionError("host tensors: no stream to set")

    def _tab(self, mask):
        return self.s.shard_table(mask)

    def _strides(self, mask):
        from mpopt_amd._lib import MPX_OWNER_RESIDENT

        tab = self.s.shard_table(mask & ~MPX_OWNER_RESIDENT)
        st = {}
        for k in (1, 2):
            rows = tab[tab[:, 1] == k]
            st[k] = int(rows[0, 4]) if len(rows) else 0
        return st

    def _truth_kind(self, mask, kind):
        """Seeded truth of the staging block / the tile partials of this pass (the same stream on every rank)."""
        st = self._strides(mask)[kind]
        r = np.random.default_rng(1000 + 10 * kind + (1 if self._hess(mask) else 0))
        return r.standard_normal((self.B, st))

    @staticmethod
    def _hess(mask):
        from mpopt_amd._lib import MPX_HESS

        return bool(mask & MPX_HESS)

    # ---- the device calls ---------------------------------------------------------------------------------------------------
    def eval_device(self, mask, batch, z, p, ppp=0, lam_g=None, sigma=None, f=None, g=None, grad_f=None, jac_val=None, hess_val=None):
        from mpopt_amd._lib import MPX_BOUNDARY_ONLY, MPX_G, MPX_GRAD, MPX_OWNER_RESIDENT

        torch = self.torch
        assert batch == self.B
        hess, owner = self._hess(mask), bool(mask & MPX_OWNER_RESIDENT)
        vals = hess_val if hess else jac_val
        self.calls.append(("boundary" if mask & MPX_BOUNDARY_ONLY else "nodes", hess, owner))
        if mask & MPX_BOUNDARY_ONLY:
            if f is not None and not hess:  # fixed-order sum of every tile's slots: NaN unless all of them arrived
                f.copy_(self.partial.sum(dim=1))
            return
        tab = self._tab(mask & ~MPX_OWNER_RESIDENT)
        st = self._strides(mask)
        self.staging = torch.full((self.B, max(st[1], 1)), float("nan"), dtype=torch.float64)
        self.partial = torch.full((self.B, max(st[2], 1)), float("nan"), dtype=torch.float64)
        truth = {0: torch.tensor(self.truth["hess_l" if hess else "jac_g"]), 1: torch.tensor(self._truth_kind(mask, 1)) if st[1] else None,
                 2: torch.tensor(self._truth_kind(mask, 2))}
        dst = {0: vals, 1: self.staging, 2: self.partial}
        for rr, kind, off, ln, stride, _ in tab.tolist():
            if rr != self.rank or dst[kind] is None or (owner and kind == 1):
                continue
            dst[kind].view(self.B, -1)[:, off:off + ln] = truth[kind][:, off:off + ln]
        if owner:  # owner-resident: the node kernels store their own g / grad_f rows directly
            for name, arr in (("g", g), ("grad_f", grad_f)):
                if arr is not None and not hess:
                    for off, ln in self.shard_owned(name, self.rank).tolist():
                        arr[:, off:off + ln] = torch.tensor(self.truth[name][:, off:off + ln])

    def _copy(self, mask, vals, buf, unpack):
        from mpopt_amd._lib import MPX_OWNER_RESIDENT

        owner = bool(mask & MPX_OWNER_RESIDENT)
        rank_len, _ = self.s.shard_info(mask)
        n = max(rank_len * self.B, 2)
        src = {0: vals, 1: self.staging, 2: self.partial}
        for rr, kind, off, ln, stride, dst in self._tab(mask).tolist():
            if (rr == self.rank) == bool(unpack) or (owner and kind != 2) or src[kind] is None:
                continue
            a = src[kind].view(self.B, -1)
            assert kind == 0 or a.shape[1] == stride or a.shape[1] == 1
            base = (rr * n if unpack else 0) + dst * self.B
            for b in range(self.B):
                if unpack:
                    a[b, off:off + ln] = buf[base + b * ln: base + (b + 1) * ln]
                else:
                    buf[base + b * ln: base + (b + 1) * ln] = a[b, off:off + ln]

    def shard_pack(self, mask, batch, vals, send):
        send.fill_(float("nan"))
        self._copy(mask, vals, send, 0)

    def shard_unpack(self, mask, batch, recv, vals):
        self._copy(mask, vals, recv, 1)
