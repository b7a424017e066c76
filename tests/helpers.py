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
