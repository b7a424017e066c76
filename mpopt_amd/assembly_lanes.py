"""Lane-per-evaluation-point Hessian kernel of assembled contexts: plan + generated source (round 5).

The fused kernels of round 3 (``csrc/mpx_assembly_fused.h``) interpret the tables of an assembled context -- local variables of a
point, multipliers, the gather rows of hess_l -- with one LANE PER ROW and the evaluation points one after the other: every term is
a table entry fetched, decoded and followed to a value in LDS, and the pass sits at a quarter of the HBM roofline waiting on its own
dependent chains (profiles/r5_adaptive_hess).  For a BATCH the tables are the constant and the evaluation points the data, so
this module turns the pass the other way round:

  * a wavefront takes 64 evaluation points (lane <-> point) and one GROUP of point tasks -- for ``mpopt_adaptive`` a collocation
    segment: its nodes and mid-points (reference mpopt.py:3034-3124) -- and the table becomes straight-line code: every
    coefficient a literal, every index a register name, one ``v_fma_f64`` per term and 64 points, no decode, no dictionary;
  * z and lam_g enter through an LDS tile: the group's columns are a handful of contiguous runs of every point's row, loaded with
    consecutive lanes on consecutive addresses and read back transposed (lane = point); the group's rows of hess_l leave the same
    way;
  * a Hessian entry whose terms come from two neighbouring groups (the variables of a node shared by two segments) is computed by
    the later group, which re-evaluates the few point tasks of its neighbour it needs (a halo): no exchange between wavefronts, no
    second launch, no atomics.

Every sum keeps the term order and the fma chain of the two-pass kernels (``mpx_assembly_kernels.h`` point_eval,
``mpx_gather_kernel``; rows past the pass's long-row threshold: lane-strided partial sums and the pairwise tree of
``mpx_wave_total``), so the results are bit-identical to them and the host may choose by batch size.

A context whose rows couple everything with everything (time-dependent dynamics: every node time depends on every earlier width)
has no such groups; ``plan_hess`` returns None and the fused kernels keep the pass.
"""
import os

import numpy as np

from .expr import _cfloat

LDW = 65  # doubles per tile row: 64 evaluation points + one pad (rows are written with lanes across rows, read with lanes along one)


def _runs(idx):
    """Sorted unique indices -> [(start, length, first position)]."""
    out = []
    idx = np.asarray(idx, dtype=np.int64)
    k = 0
    while k < len(idx):
        j = k
        while j + 1 < len(idx) and idx[j + 1] == idx[j] + 1:
            j += 1
        out.append((int(idx[k]), j - k + 1, k))
        k = j + 1
    return out


def _hess_deps(fn):
    """Per structural Hessian entry q of a point function: the local variables and the multipliers its expression reads."""
    if not hasattr(fn, "_lane_deps"):
        deps = []
        for _, _, e in fn.H:
            names = [n.val for n in fn.tr.toposort([e]) if n.op == "var"]
            deps.append((frozenset(int(x[4:-1]) for x in names if x.startswith("loc[")), frozenset(int(x[3:-1]) for x in names if x.startswith("mu["))))
        fn._lane_deps = deps
    return fn._lane_deps


def _pieces(runs):
    """Contiguous runs -> power-of-two pieces [(start, log2 size, first position)]: a piece of 2^k columns is moved by 2^k
    instructions of 64 lanes, lane = (point within the instruction, column within the piece) by shift and mask."""
    out = []
    for a, n, e in runs:
        while n:
            k = min(n.bit_length() - 1, 6)  # (at most 64 columns: one evaluation point per instruction)
            out.append((a, k, e))
            a, n, e = a + (1 << k), n - (1 << k), e + (1 << k)
    return out


class HessLanePlan:
    """Groups of point tasks, the rows each group owns, its halo tasks and the columns of z / lam_g it reads."""

    def __init__(self, groups, n_tasks, n_rows):
        self.groups, self.n_tasks, self.n_rows = groups, n_tasks, n_rows
        self.ne_max = max(len(g["zcols"]) + len(g["lcols"]) for g in groups)
        self.nr_max = max(len(g["rows"]) for g in groups)
        self.halo_tasks = sum(len(g["tasks"]) for g in groups) - n_tasks


def plan_hess(o, min_tasks=8, max_tasks=40, max_raw=260, max_tile=300):
    """``o``: AssembledNlpFunctions after ``_expand``.  -> HessLanePlan or None (no useful grouping)."""
    ptr, src, _ = o.hess
    n_rows = len(ptr) - 1
    if n_rows == 0 or o.rawh_n == 0:
        return None
    # point tasks with Hessian entries; owner task of every raw slot
    tasks, task_of = [], np.full(o.rawh_n, -1, np.int64)
    for k, s in enumerate(o.sets):
        if s.fn.n_hess == 0:
            continue
        base = len(tasks)
        tasks += [(k, p) for p in range(s.n)]
        for q in range(s.fn.n_hess):
            task_of[o.rawh_off[k] + q * s.n:o.rawh_off[k] + (q + 1) * s.n] = base + np.arange(s.n)
    n_t = len(tasks)
    if n_t < 2 * min_tasks:
        return None
    # a task's position along z: the smallest column any of its local variables reads (X blocks come first in z, so this is the
    # node index for a node and the first node of the segment for a mid-point)
    key = np.empty(n_t, np.int64)
    for t, (k, p) in enumerate(tasks):
        s = o.sets[k]
        lo, hi = s.L.indptr[p * s.fn.n_loc], s.L.indptr[(p + 1) * s.fn.n_loc]
        cols = s.L.indices[lo:hi][s.L.data[lo:hi] != 0]
        key[t] = cols.min() if len(cols) else 0
    order = np.lexsort((np.arange(n_t), key))
    pos = np.empty(n_t, np.int64)
    pos[order] = np.arange(n_t)
    row_tasks = [np.unique(pos[task_of[src[ptr[r]:ptr[r + 1]]]]) for r in range(n_rows)]  # positions, ascending
    # Cuts of the ordered task list into groups.  A row belongs to the group of its LAST task; the tasks in front of the group's cut
    # that its rows read are the group's halo (re-evaluated there).  Dynamic programme over the cut positions: least sum of SQUARED
    # group weights (own + halo tasks) -- small halos, and many even groups rather than few large ones (a group is one wavefront: its
    # raw values live in registers, and a batch of 4096 points is only 64 blocks of 64) --, own tasks per group in [min_tasks, max_own].
    weight = np.array([1.0 + 0.1 * o.sets[k].fn.n_hess for k, _ in tasks])[order]  # by position
    rows_by_last = [[] for _ in range(n_t)]
    for r, rt in enumerate(row_tasks):
        if len(rt):
            rows_by_last[rt[-1]].append(rt)
    max_own = 2 * min_tasks
    INF = float("inf")
    best, back = np.full(n_t + 1, INF), np.full(n_t + 1, -1, np.int64)
    best[0] = 0.0
    for a in range(n_t):
        if best[a] == INF:
            continue
        halo, hw, own = set(), 0.0, 0.0
        for b in range(a + 1, min(n_t, a + max_own) + 1):
            own += weight[b - 1]
            for rt in rows_by_last[b - 1]:
                for t in rt:
                    if t >= a:
                        break
                    if t not in halo:
                        halo.add(int(t))
                        hw += weight[t]
            if b - a >= min_tasks or b == n_t:
                cand = best[a] + (own + hw) ** 2
                if cand < best[b]:
                    best[b], back[b] = cand, a
    if best[n_t] == INF:
        return None
    cuts = [n_t]
    while cuts[-1] > 0:
        cuts.append(int(back[cuts[-1]]))
    cuts = cuts[::-1]
    if len(cuts) < 3:
        return None
    grp_of_pos = np.zeros(n_t, np.int64)
    for g in range(len(cuts) - 1):
        grp_of_pos[cuts[g]:cuts[g + 1]] = g
    groups = [dict(rows=[], tasks=set()) for _ in range(len(cuts) - 1)]
    for r, rt in enumerate(row_tasks):
        g = groups[grp_of_pos[rt[-1]] if len(rt) else 0]
        g["rows"].append(r)
        g["tasks"].update(int(x) for x in rt)
    groups = [g for g in groups if g["rows"]]
    slot_q = np.zeros(o.rawh_n, np.int64)  # Hessian entry q of a raw slot
    for k, s in enumerate(o.sets):
        for q in range(s.fn.n_hess):
            slot_q[o.rawh_off[k] + q * s.n:o.rawh_off[k] + (q + 1) * s.n] = q
    for g in groups:
        g["tasks"] = [tasks[order[x]] for x in sorted(g["tasks"])]
        # what the group's rows read of each task: entries q -> the local variables / multipliers those depend on; only their
        # columns of z / lam_g are loaded (time-independent dynamics never read the running sum of the earlier widths)
        used = {kp: set() for kp in g["tasks"]}
        for r in g["rows"]:
            for e in range(ptr[r], ptr[r + 1]):
                used[tasks[task_of[src[e]]]].add(int(slot_q[src[e]]))
        g["use"] = {}
        zc, lc, raw = set(), set(), 0
        for k, p in g["tasks"]:
            s = o.sets[k]
            deps = _hess_deps(s.fn)
            lv = set().union(*[deps[q][0] for q in used[(k, p)]]) if used[(k, p)] else set()
            mv = set().union(*[deps[q][1] for q in used[(k, p)]]) if used[(k, p)] else set()
            g["use"][(k, p)] = (lv, mv)
            for v in lv:
                lo, hi = s.L.indptr[p * s.fn.n_loc + v], s.L.indptr[p * s.fn.n_loc + v + 1]
                zc.update(int(c) for c, d in zip(s.L.indices[lo:hi], s.L.data[lo:hi]) if d != 0)
            for r_ in mv:
                lo, hi = s.G.indptr[p * s.fn.n_out + r_], s.G.indptr[p * s.fn.n_out + r_ + 1]
                lc.update(int(c) for c, d in zip(s.G.indices[lo:hi], s.G.data[lo:hi]) if d != 0)
            raw += len(used[(k, p)])
        g["zcols"], g["lcols"], g["raw"] = sorted(zc), sorted(lc), raw
        if len(g["tasks"]) > max_tasks or raw > max_raw or len(zc) + len(lc) > max_tile or len(g["rows"]) > max_tile:
            if os.environ.get("MPX_LANES_VERBOSE"):
                print(f"assembly_lanes: no plan -- a group of {len(g['tasks'])} tasks, {raw} raw values, {len(zc) + len(lc)} columns, {len(g['rows'])} rows")
            return None
    return HessLanePlan(groups, n_t, n_rows)


def _chain(terms, acc):
    """Statements of  acc = fma(c_t, v_t, acc)  over ``terms`` = [(coef, value expression)] from acc = 0, in order.
    ``__builtin_fma``, not ``fma``: HIP's fma() is an OCML function whose llvm.fma call carries the ``contract`` flag whatever the
    pragma in scope says; with a LITERAL coefficient 1.0 the optimiser rewrites it into an fadd that inherits the flag and then
    fuses it with a product inside the (contract-off) point function -- one rounding less than the two-pass kernels, seen on
    hyper-sensitive 40x4 as last-bit differences in the (t0 | tf, width) entries.  The builtin takes its flags from the pragma."""
    if not terms:
        return [f"{acc} = 0.0;"]
    out = [f"{acc} = __builtin_fma({_cfloat(terms[0][0])}, {terms[0][1]}, 0.0);"]
    out += [f"{acc} = __builtin_fma({_cfloat(c)}, {v}, {acc});" for c, v in terms[1:]]
    return out


def _tree(names, lo, hi):
    """Pairwise tree of mpx_wave_total over lanes lo .. hi - 1; ``names[j]`` is None for a lane without terms (its partial sum is
    +0.0, and x + 0.0 == x for every x these sums can take: a partial sum starts from fma(c, v, +0.0) and is never -0.0)."""
    if hi - lo == 1:
        return names[lo]
    mid = (lo + hi) // 2
    a, b = _tree(names, lo, mid), _tree(names, mid, hi)
    if a is None:
        return b
    if b is None:
        return a
    return f"({b} + {a})"


def group_major(o, plan):
    """Reorder the entries of hess_l group by group (the order of the pattern is the context's to choose -- assembly.py: pattern --,
    ``mpx_ccs_perm`` maps any order to CasADi's): the rows a wavefront computes become ONE contiguous run of every evaluation
    point's hess_val instead of a dozen runs of five.  Inside a group the entries keep their relative order."""
    ptr, src, coef = o.hess
    order = np.concatenate([np.asarray(g["rows"], np.int64) for g in plan.groups])
    assert len(order) == len(ptr) - 1 and len(np.unique(order)) == len(order)
    nt = np.diff(ptr)[order]
    new_ptr = np.concatenate([[0], np.cumsum(nt)]).astype(np.int64)
    take = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in order]) if len(src) else np.zeros(0, np.int64)
    o.hess = (new_ptr, np.ascontiguousarray(src[take]), np.ascontiguousarray(coef[take]))
    o.hrow, o.hcol = np.ascontiguousarray(o.hrow[order]), np.ascontiguousarray(o.hcol[order])
    first = 0
    for g in plan.groups:
        g["rows"] = list(range(first, first + len(g["rows"])))
        first += len(g["rows"])


def hess_source(o, thr, plan):
    """-> source text to append to the generated translation unit (``plan``: plan_hess, after group_major)."""
    ptr, src, coef = o.hess
    n_g = o.n_g_
    slot_task = {}  # raw slot -> (k, p, q)
    for k, s in enumerate(o.sets):
        for q in range(s.fn.n_hess):
            for p in range(s.n):
                slot_task[o.rawh_off[k] + q * s.n + p] = (k, p, q)
    # (row strides of z, lam_g and hess_val: compile-time constants of the address arithmetic -- the host always passes dense arrays)
    parts = ["#ifndef MPX_LANE_LDW", f"#define MPX_LANE_LDW {LDW}", "#endif", f"#define MPX_LANE_ZS {o.n_z_}", f"#define MPX_LANE_LS {o.n_g_}",
             f"#define MPX_LANE_OS {plan.n_rows}", "namespace mpxgen {", "template <int G> struct LaneGrp;"]
    for gi, g in enumerate(plan.groups):
        zpos = {c: e for e, c in enumerate(g["zcols"])}
        nz = len(g["zcols"])
        lpos = {c: nz + e for e, c in enumerate(g["lcols"])}
        hname = {kp: f"H{i}" for i, kp in enumerate(g["tasks"])}
        body = []
        for (k, p), hn in hname.items():
            s = o.sets[k]
            fn, fid = s.fn, o.functions.index(s.fn)
            body.append(f"    double {hn}[{max(fn.n_hess, 1)}];")
            body.append("    {")
            body.append(f"      double loc[{max(fn.n_loc, 1)}], mu[{max(fn.n_out, 1)}];")
            lv, mv = g["use"][(k, p)]
            for v in range(fn.n_loc):
                if v not in lv:  # (no entry the group reads depends on it)
                    body.append(f"      loc[{v}] = 0.0;")
                    continue
                lo, hi = s.L.indptr[p * fn.n_loc + v], s.L.indptr[p * fn.n_loc + v + 1]
                terms = [(float(d), f"T[{zpos[int(c)]} * MPX_LANE_LDW]") for c, d in zip(s.L.indices[lo:hi], s.L.data[lo:hi]) if d != 0]
                body += ["      " + ln for ln in _chain(terms, f"loc[{v}]")]
            cst = ", ".join(_cfloat(float(x)) for x in s.cst[p]) if fn.n_cst else "0.0"
            body.append(f"      const double cst[{max(fn.n_cst, 1)}] = {{{cst}}};")
            for r in range(fn.n_out):
                if r not in mv:
                    body.append(f"      mu[{r}] = 0.0;")
                    continue
                lo, hi = s.G.indptr[p * fn.n_out + r], s.G.indptr[p * fn.n_out + r + 1]
                terms = [(float(d), f"T[{lpos[int(c)]} * MPX_LANE_LDW]") for c, d in zip(s.G.indices[lo:hi], s.G.data[lo:hi]) if d != 0]
                if s.fw[p, r] != 0:
                    terms.append((float(s.fw[p, r]), "sg"))
                body += ["      " + ln for ln in _chain(terms, f"mu[{r}]")]
            body.append(f"      Pt<{fid}>::hes(loc, cst, mu, {hn});")
            body.append("    }")
        for j, r in enumerate(g["rows"]):
            terms = []
            for e in range(ptr[r], ptr[r + 1]):
                k, p, q = slot_task[int(src[e])]
                terms.append((float(coef[e]), f"{hname[(k, p)]}[{q}]"))
            if len(terms) <= thr:
                body += ["    " + ln for ln in _chain(terms, f"R[{j}]")]
            else:  # a long row: lane l of the two-pass kernel sums terms l, l + 64, ... in order, then mpx_wave_total
                names = [None] * 64
                body.append("    {")
                for lane in range(min(64, len(terms))):
                    names[lane] = f"q{lane}"
                    body.append(f"      double q{lane};")
                    body += ["      " + ln for ln in _chain(terms[lane::64], f"q{lane}")]
                body.append(f"      R[{j}] = {_tree(names, 0, 64)};")
                body.append("    }")
        zr, lr, rr = _pieces(_runs(g["zcols"])), _pieces(_runs(g["lcols"])), _pieces(_runs(g["rows"]))
        ne = nz + len(g["lcols"])
        parts.append(f"template <> struct LaneGrp<{gi}> {{")
        parts.append(f"  static constexpr int NE = {ne}, NR = {len(g['rows'])};")
        for name, call in (("load", "ld"), ("fill", "put")):
            parts.append(f"  template <class IO> __device__ static __forceinline__ void {name}(IO& io) {{")
            parts += [f"    io.template {call}<0, {a}, {n}, {e}>();" for a, n, e in zr]
            parts += [f"    io.template {call}<1, {a}, {n}, {nz + e}>();" for a, n, e in lr]
            parts.append("  }")
        parts.append("  template <class IO> __device__ static __forceinline__ void store(IO& io) {")
        parts += [f"    io.template st<{a}, {n}, {e}>();" for a, n, e in rr]
        parts.append("  }")
        parts.append("  __device__ static __forceinline__ void compute(const double* __restrict__ T, const double sg, double* __restrict__ R) {")
        parts.append("#pragma clang fp contract(off)")
        parts += body
        parts.append("  }")
        parts.append("};")
    parts.append("}  // namespace mpxgen")
    parts.append(f"#define MPX_LANE_GROUPS {len(plan.groups)}")
    parts.append(f"#define MPX_LANE_NE_MAX {max(plan.ne_max, plan.nr_max)}")
    parts.append(f"#define MPX_LANE_NNZH {plan.n_rows}")
    parts.append('#include "mpx_assembly_lanes.h"')
    parts.append("MPX_INSTANTIATE_LANES_HESS")
    return "\n".join(parts)
