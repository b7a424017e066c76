"""Lane-per-evaluation-point kernels of assembled contexts: plan + generated source (round 5).

The fused kernels of round 3 (``csrc/mpx_assembly_fused.h``) interpret the tables of an assembled context -- local variables of a
point, multipliers, the gather rows of a pass -- with one LANE PER ROW and the evaluation points one after the other: every term is
a table entry fetched, decoded and followed to a value in LDS, and hess_l sat at a quarter of the HBM roofline waiting on its own
dependent chains (profiles/r5_adaptive_hess).  For a BATCH the tables are the constant and the evaluation points the data, so
this module turns a pass the other way round:

  * a wavefront takes 64 evaluation points (lane <-> point) and one GROUP of point tasks -- for ``mpopt_adaptive`` a collocation
    segment: its nodes and mid-points (reference mpopt.py:3034-3124) -- and the table becomes straight-line code: every
    coefficient a literal, every index a register name, one ``v_fma_f64`` per term and 64 points, no decode, no dictionary;
  * z and lam_g enter through an LDS tile: the group's columns are a handful of contiguous runs of every point's row, loaded with
    consecutive lanes on consecutive addresses and read back transposed (lane = point); the group's rows leave the same way, in
    chunks of the tile's size;
  * a row whose terms come from two neighbouring groups (the variables of a node shared by two segments) is computed by the later
    group, which re-evaluates the few point tasks of its neighbour it needs (a halo): no exchange between wavefronts, no atomics;
  * GLOBAL rows -- sums over (nearly) all point tasks: the objective, d f / d t0, d f / d tf; with time-dependent dynamics the
    Hessian entries of pairs of widths -- cannot belong to a group.  The raw values they read are written by the wavefront that OWNS
    the task to a scratch array [block][slot][lane], and a second, small kernel (``mpx_asml_*_global``, lanes <-> points again, its
    table a constant array of the code object) sums them in the canonical order;
  * rows without any point task are constants (Jacobian entries of the linear part: D blocks): dealt to the groups evenly -- they
    have to be written, nothing has to be computed.

Every sum keeps the term order and the fma chain of the two-pass kernels (``mpx_assembly_kernels.h`` point_eval,
``mpx_gather_kernel``; rows past the pass's long-row threshold: lane-strided partial sums and the pairwise tree of
``mpx_wave_total``), so the results are bit-identical to them and the host may choose by batch size.  The entries of hess_l /
jac_g are ordered group by group (``group_major``: the order of a pattern is the context's to choose, ``mpx_ccs_perm`` maps any
order to CasADi's), so that a wavefront writes contiguous runs.
"""
import os

import numpy as np

from .expr import _cfloat

LDW = 65   # doubles per tile row: 64 evaluation points + one pad (rows are written with lanes across rows, read with lanes along one)
CHUNK = 32  # rows of a group that leave together when the group has more rows than fit the tile beside its inputs
GCAP = 24   # a row that reads more point tasks than this is a global row


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


def _var_deps(fn, exprs):
    """Per expression: (local variables, multipliers) it reads."""
    deps = []
    for e in exprs:
        names = [n.val for n in fn.tr.toposort([e]) if n.op == "var"]
        deps.append((frozenset(int(x[4:-1]) for x in names if x.startswith("loc[")), frozenset(int(x[3:-1]) for x in names if x.startswith("mu["))))
    return deps


def _hess_deps(fn):
    """Per structural Hessian entry q of a point function: the local variables and the multipliers its expression reads."""
    if not hasattr(fn, "_lane_deps"):
        fn._lane_deps = _var_deps(fn, [e for _, _, e in fn.H])
    return fn._lane_deps


def _first_deps(fn):
    """The same for the first-order values of a point function, in raw-slot order: outputs, then Jacobian entries."""
    if not hasattr(fn, "_lane_deps1"):
        fn._lane_deps1 = _var_deps(fn, list(fn.out) + [e for _, _, e in fn.J])
    return fn._lane_deps1


class Pass:
    """What the planner and the emitter need to know about one pass ('hes': hess_l; 'fgj': f, g, grad_f, jac_g)."""

    def __init__(self, o, kind):
        self.o, self.kind = o, kind
        if kind == "hes":
            self.ptr, self.src, self.coef = o.hess
            self.arrays = [("h", 0, o.nnz_hess_)]  # (name, first row, rows): one output array
            self.raw_off, self.raw_n = o.rawh_off, o.rawh_n
            self.nval = lambda fn: fn.n_hess
            self.deps = _hess_deps
        else:
            self.ptr, self.src, self.coef = o.fgj
            n_g, n_z = o.n_g_, o.n_z_
            self.arrays = [("f", 0, 1), ("g", 1, n_g), ("q", 1 + n_g, n_z), ("j", 1 + n_g + n_z, o.nnz_jac_)]
            self.raw_off, self.raw_n = o.raw_off, o.raw_n
            self.nval = lambda fn: fn.n_out + fn.n_jac
            self.deps = _first_deps
        self.n_rows = len(self.ptr) - 1
        self.tasks, self.task_of = [], np.full(max(self.raw_n, 1), -1, np.int64)
        self.val_of = np.zeros(max(self.raw_n, 1), np.int64)  # index of a raw slot inside its task's value vector
        for k, s in enumerate(o.sets):
            nv = self.nval(s.fn)
            if nv == 0:
                continue
            base = len(self.tasks)
            self.tasks += [(k, p) for p in range(s.n)]
            for q in range(nv):
                a = self.raw_off[k] + q * s.n
                self.task_of[a:a + s.n] = base + np.arange(s.n)
                self.val_of[a:a + s.n] = q

    def array_of(self, r):
        for a, (_, first, n) in enumerate(self.arrays):
            if first <= r < first + n:
                return a, r - first
        raise IndexError(r)


class LanePlan:
    """Groups of point tasks, the rows each group owns (in chunks), its halo tasks, the columns of z / lam_g it reads; the global rows
    and the scratch slots they read."""

    def __init__(self, P, groups, global_rows, sid):
        self.kind, self.groups, self.global_rows, self.sid = P.kind, groups, global_rows, sid
        self.n_tasks, self.n_rows = len(P.tasks), P.n_rows
        self.ne_max = max(g["ne"] for g in groups)
        self.nr_max = max(len(g["rows"]) for g in groups)
        self.tile_rows = max(g["tile_rows"] for g in groups)
        self.halo_tasks = sum(len(g["tasks"]) for g in groups) - sum(1 for g in groups for t in g["tasks"] if t in g["own"])


def plan_pass(o, kind, min_tasks=8, max_tasks=40, max_raw=None, max_tile=300):
    """``o``: AssembledNlpFunctions after ``_expand``.  -> LanePlan or None (no useful grouping)."""
    P = Pass(o, kind)
    ptr, src = P.ptr, P.src
    n_rows, tasks, n_t = P.n_rows, P.tasks, len(P.tasks)
    verbose = os.environ.get("MPX_LANES_VERBOSE")
    min_tasks = int(os.environ.get("MPX_LANES_MIN_TASKS", min_tasks))  # (own tasks of a group: min_tasks .. 2 min_tasks; A/B)
    if max_raw is None:
        max_raw = int(os.environ.get("MPX_LANES_MAX_RAW", 400))  # (Van der Pol 10x6, 264 raw values per group: 30.5 against 180 us fused)
    if n_rows == 0 or P.raw_n == 0 or n_t < 2 * min_tasks:
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
    row_tasks, row_z = [], []
    for r in range(n_rows):
        sr = src[ptr[r]:ptr[r + 1]]
        row_tasks.append(np.unique(pos[P.task_of[sr[sr >= 0]]]))  # positions, ascending
        row_z.append(np.unique(-2 - sr[sr <= -2]))
    # global rows: too many tasks for any group (and, below: linear rows -- no task, columns of z -- that no group's tile covers
    # by half: the sum of the widths)
    is_global = np.array([len(row_tasks[r]) > GCAP for r in range(n_rows)])
    if kind == "fgj":
        is_global[0] = True  # (f: its own array)
    # Cuts of the ordered task list into groups.  A row belongs to the group of its LAST task; the tasks in front of the group's cut
    # that its rows read are the group's halo (re-evaluated there).  Dynamic programme over the cut positions: least sum of SQUARED
    # group weights (own + halo tasks) -- small halos, and many even groups rather than few large ones (a group is one wavefront: its
    # raw values live in registers, and a batch of 4096 points is only 64 blocks of 64) --, own tasks per group in [min_tasks, max_own].
    weight = np.array([1.0 + 0.1 * P.nval(o.sets[k].fn) for k, _ in tasks])[order]  # by position
    rows_by_last = [[] for _ in range(n_t)]
    for r, rt in enumerate(row_tasks):
        if len(rt) and not is_global[r]:
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
    n_grp = len(cuts) - 1
    grp_of_pos = np.zeros(n_t, np.int64)
    for g in range(n_grp):
        grp_of_pos[cuts[g]:cuts[g + 1]] = g
    groups = [dict(rows=[], need=set(), own=set(tasks[order[x]] for x in range(cuts[g], cuts[g + 1]))) for g in range(n_grp)]
    const_rows, linear_rows = [], []
    for r, rt in enumerate(row_tasks):
        if is_global[r]:
            continue
        if len(rt) == 0:
            (linear_rows if len(row_z[r]) else const_rows).append(r)
            continue
        g = groups[grp_of_pos[rt[-1]]]
        g["rows"].append(r)
        g["need"].update(int(x) for x in rt)
    # linear rows (mid-point bounds on interpolated states / controls: I_mid U): to the group whose tasks read most of their columns
    if linear_rows:
        cols_of = []
        for g in groups:
            zc = set()
            for x in g["need"]:
                k, p = tasks[order[x]]
                s = o.sets[k]
                lo, hi = s.L.indptr[p * s.fn.n_loc], s.L.indptr[(p + 1) * s.fn.n_loc]
                zc.update(int(c) for c, d in zip(s.L.indices[lo:hi], s.L.data[lo:hi]) if d != 0)
            cols_of.append(zc)
        for r in linear_rows:
            ov = [len(cols_of[gi].intersection(int(c) for c in row_z[r])) for gi in range(len(groups))]
            gi = int(np.argmax(ov))
            if 2 * ov[gi] >= len(row_z[r]):
                groups[gi]["rows"].append(r)
            else:
                is_global[r] = True
    # constants: to the group with the fewest rows so far, in runs (neighbours in the pattern stay neighbours)
    run = max(8, (len(const_rows) + 4 * n_grp - 1) // (4 * n_grp)) if const_rows else 0
    for a in range(0, len(const_rows), max(run, 1)):
        g = min(groups, key=lambda q: len(q["rows"]))
        g["rows"] += const_rows[a:a + run]
    global_rows = [int(r) for r in np.flatnonzero(is_global)]
    # scratch slots: every raw value a global row reads, written by the group that OWNS its task
    sid = {}
    for r in global_rows:
        for x in src[ptr[r]:ptr[r + 1]]:
            if x >= 0 and int(x) not in sid:
                sid[int(x)] = len(sid)
    for x in sid:
        groups[grp_of_pos[pos[P.task_of[x]]]]["need"].add(int(pos[P.task_of[x]]))
    groups = [g for g in groups if g["rows"] or g["need"]]
    first_pos = lambda r: int(row_tasks[r][0]) if len(row_tasks[r]) else n_t
    for g in groups:
        g["tasks"] = [tasks[order[x]] for x in sorted(g["need"])]
        # rows in the order their first task is evaluated (constants last), cut into chunks; a row's array and index inside it
        g["rows"].sort(key=lambda r: (first_pos(r), r))
        # what the group's rows (and the scratch slots it owns) read of each task: values -> the local variables / multipliers those
        # depend on; only their columns of z / lam_g are loaded (time-independent dynamics never read the running sum of the widths)
        used = {kp: set() for kp in g["tasks"]}
        zc, lc = set(), set()
        for r in g["rows"]:
            for x in src[ptr[r]:ptr[r + 1]]:
                if x >= 0:
                    used[tasks[P.task_of[x]]].add(int(P.val_of[x]))
            zc.update(int(c) for c in row_z[r])
        g["scratch"] = {}
        for x, sd in sid.items():
            kp = tasks[P.task_of[x]]
            if kp in g["own"]:
                used[kp].add(int(P.val_of[x]))
                g["scratch"].setdefault(kp, []).append((int(P.val_of[x]), sd))
        g["use"], raw = {}, 0
        for k, p in g["tasks"]:
            s = o.sets[k]
            deps = P.deps(s.fn)
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
        g["ne"] = len(zc) + len(lc)
        nr = len(g["rows"])
        # one chunk: the rows leave through the tile their inputs came in by; several: a region of CHUNK rows behind the inputs
        chunk = int(os.environ.get("MPX_LANES_CHUNK", CHUNK))
        single = nr <= max(g["ne"], 64)
        g["chunk"], g["out_base"] = (max(nr, 1), 0) if single else (chunk, g["ne"])
        g["tile_rows"] = max(g["ne"], nr) if single else g["ne"] + chunk
        if len(g["tasks"]) > max_tasks or raw > max_raw or g["tile_rows"] > max_tile:
            if verbose:
                print(f"assembly_lanes[{kind}]: no plan -- a group of {len(g['tasks'])} tasks, {raw} raw values, {g['ne']} columns, {nr} rows")
            return None
    if verbose:
        print(f"assembly_lanes[{kind}]: {len(groups)} groups, {len(global_rows)} global rows reading {len(sid)} scratch slots, {len(const_rows)} constant rows")
    # hess_l with global rows = time-dependent dynamics (entries of pairs of widths sum over every later task): measured slower than the
    # fused kernel (time_dependent 10x3: 164 against 123 us -- six heavy groups, point Hessians that spill); MPX_LANES_GLOBAL=1 keeps the plan
    if kind == "hes" and global_rows and os.environ.get("MPX_LANES_GLOBAL") != "1":
        return None
    return LanePlan(P, groups, global_rows, sid)


def group_major(o, plan):
    """Reorder the entries of hess_l ('hes') / jac_g ('fgj') group by group, chunk by chunk (the order of a pattern is the
    context's to choose -- assembly.py: pattern --, ``mpx_ccs_perm`` maps any order to CasADi's): the rows a wavefront computes
    become contiguous runs of every evaluation point's array instead of a dozen runs of five.  Global rows go last."""
    P = Pass(o, plan.kind)
    name = "h" if plan.kind == "hes" else "j"
    a_idx = [a for a, (n, _, _) in enumerate(P.arrays) if n == name][0]
    first, count = P.arrays[a_idx][1], P.arrays[a_idx][2]
    in_arr = lambda r: first <= r < first + count
    order = [r for g in plan.groups for r in g["rows"] if in_arr(r)] + [r for r in plan.global_rows if in_arr(r)]
    assert len(order) == count and len(set(order)) == count
    new_of = {r: first + i for i, r in enumerate(order)}
    full = np.array([r for r in range(first)] + order + [r for r in range(first + count, P.n_rows)], np.int64)
    ptr, src, coef = P.ptr, P.src, P.coef
    nt = np.diff(ptr)[full]
    new_ptr = np.concatenate([[0], np.cumsum(nt)]).astype(np.int64)
    take = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in full]) if len(src) else np.zeros(0, np.int64)
    new = (new_ptr, np.ascontiguousarray(src[take]), np.ascontiguousarray(coef[take]))
    loc = np.asarray(order, np.int64) - first
    if plan.kind == "hes":
        o.hess = new
        o.hrow, o.hcol = np.ascontiguousarray(o.hrow[loc]), np.ascontiguousarray(o.hcol[loc])
    else:
        o.fgj = new
        o.jrow, o.jcol = np.ascontiguousarray(o.jrow[loc]), np.ascontiguousarray(o.jcol[loc])
    for g in plan.groups:
        g["rows"] = [new_of.get(r, r) for r in g["rows"]]
    plan.global_rows = [new_of.get(r, r) for r in plan.global_rows]


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


def pass_source(o, thr, plan):
    """-> source text to append to the generated translation unit (``plan``: plan_pass, after group_major)."""
    P = Pass(o, plan.kind)
    kind, ptr, src, coef = plan.kind, P.ptr, P.src, P.coef
    KIND = kind.upper()
    parts = [f"namespace mpxgen {{", f"template <int G> struct LaneGrp{KIND};"]
    # One generated body per group SHAPE, not per group (round 6): equal-degree interior segments of a grid differ in nothing but where
    # their columns of z / lam_g and their rows of the outputs start -- those numbers are the group's PARAMETERS (a constant int table
    # of the code object, read with scalar loads: `par[i]`), everything else (coefficients, tile positions, term order) is the shape's
    # text, and groups whose text is equal share one struct.  Up to round 5 every group was its own struct: 100 x 3 compiled for 49 s
    # into a 459 KB code object, every wavefront of a launch on different instructions.
    shapes, shape_of, par_off, par_all = {}, [], [], []
    for gi, g in enumerate(plan.groups):
        par = []

        def P_(v):  # a group parameter -> the expression that reads it
            par.append(int(v))
            return f"par[{len(par) - 1}]"
        zpos = {c: e for e, c in enumerate(g["zcols"])}
        nz = len(g["zcols"])
        lpos = {c: nz + e for e, c in enumerate(g["lcols"])}
        vname = {kp: f"V{i}" for i, kp in enumerate(g["tasks"])}
        emitted = set()
        task_pos = {kp: i for i, kp in enumerate(g["tasks"])}

        def emit_task(kp):
            k, p = kp
            s = o.sets[k]
            fn, fid, vn = s.fn, o.functions.index(s.fn), vname[kp]
            out = [f"    double {vn}[{max(P.nval(fn), 1)}];", "    {", f"      double loc[{max(fn.n_loc, 1)}], mu[{max(fn.n_out, 1)}];"]
            lv, mv = g["use"][kp]
            for v in range(fn.n_loc):
                if v not in lv:  # (no value the group reads depends on it)
                    out.append(f"      loc[{v}] = 0.0;")
                    continue
                lo, hi = s.L.indptr[p * fn.n_loc + v], s.L.indptr[p * fn.n_loc + v + 1]
                terms = [(float(d), f"T[{zpos[int(c)]} * MPX_LANE_LDW]") for c, d in zip(s.L.indices[lo:hi], s.L.data[lo:hi]) if d != 0]
                out += ["      " + ln for ln in _chain(terms, f"loc[{v}]")]
            cst = ", ".join(_cfloat(float(x)) for x in s.cst[p]) if fn.n_cst else "0.0"
            out.append(f"      const double cst[{max(fn.n_cst, 1)}] = {{{cst}}};")
            if kind == "hes":
                for r in range(fn.n_out):
                    if r not in mv:
                        out.append(f"      mu[{r}] = 0.0;")
                        continue
                    lo, hi = s.G.indptr[p * fn.n_out + r], s.G.indptr[p * fn.n_out + r + 1]
                    terms = [(float(d), f"T[{lpos[int(c)]} * MPX_LANE_LDW]") for c, d in zip(s.G.indices[lo:hi], s.G.data[lo:hi]) if d != 0]
                    if s.fw[p, r] != 0:
                        terms.append((float(s.fw[p, r]), "sg"))
                    out += ["      " + ln for ln in _chain(terms, f"mu[{r}]")]
                out.append(f"      Pt<{fid}>::hes(loc, cst, mu, {vn});")
            else:
                out.append(f"      (void)mu; Pt<{fid}>::jac(loc, cst, {vn}, {vn} + {fn.n_out});")
            out.append("    }")
            for q, sd in g["scratch"].get(kp, []):  # (values global rows read: this group owns the task)
                out.append(f"    io.S[{P_(sd)} * 64] = {vn}[{q}];")
            return out

        def value(x):
            if x == -1:
                return "1.0"
            if x <= -2:
                return f"T[{zpos[-2 - int(x)]} * MPX_LANE_LDW]"
            return f"{vname[P.tasks[P.task_of[x]]]}[{int(P.val_of[x])}]"

        body = []
        rows, ch, ob = g["rows"], g["chunk"], g["out_base"]
        # tasks nobody's LOCAL rows read but whose values go to the scratch array: evaluated up front
        for kp in g["tasks"]:
            if kp in g["scratch"] and not any(P.tasks[P.task_of[x]] == kp for r in rows for x in src[ptr[r]:ptr[r + 1]] if x >= 0):
                body += emit_task(kp)
                emitted.add(kp)
        for c0 in range(0, max(len(rows), 1), ch):
            crow = rows[c0:c0 + ch]
            need = sorted({P.tasks[P.task_of[x]] for r in crow for x in src[ptr[r]:ptr[r + 1]] if x >= 0} - emitted, key=lambda kp: task_pos[kp])
            for kp in need:
                body += emit_task(kp)
                emitted.add(kp)
            body.append("    {")
            body.append(f"      double R[{max(len(crow), 1)}];")
            for j, r in enumerate(crow):
                terms = [(float(coef[e]), value(int(src[e]))) for e in range(ptr[r], ptr[r + 1])]
                if len(terms) <= thr:
                    body += ["      " + ln for ln in _chain(terms, f"R[{j}]")]
                else:  # a long row: lane l of the two-pass kernel sums terms l, l + 64, ... in order, then mpx_wave_total
                    names = [None] * 64
                    body.append("      {")
                    for lane in range(min(64, len(terms))):
                        names[lane] = f"q{lane}"
                        body.append(f"        double q{lane};")
                        body += ["        " + ln for ln in _chain(terms[lane::64], f"q{lane}")]
                    body.append(f"        R[{j}] = {_tree(names, 0, 64)};")
                    body.append("      }")
            # rows -> tile -> their arrays (runs of consecutive rows of one array, in power-of-two pieces)
            body.append("      MPX_LANE_SYNC();" if ob == 0 or c0 > 0 else "")
            body.append(f"      for (int r = 0; r < {len(crow)}; ++r) io.T[({ob} + r) * MPX_LANE_LDW + io.lane] = R[r];".replace("for (", "_Pragma(\"unroll\") for ("))
            body.append("      MPX_LANE_SYNC();")
            j = 0
            while j < len(crow):
                a, i0 = P.array_of(crow[j])
                n = 1
                while j + n < len(crow) and P.array_of(crow[j + n]) == (a, i0 + n):
                    n += 1
                for start, k2, e in _pieces([(i0, n, ob + j)]):
                    body.append(f"      io.template st<{a}, {k2}, {e}>({P_(start)});")
                j += n
            body.append("    }")
        zr, lr = _pieces(_runs(g["zcols"])), _pieces(_runs(g["lcols"]))
        text = [f"  static constexpr int NE = {g['ne']};"]
        text.append("  template <class IO> __device__ static __forceinline__ void load(IO& io, const int* __restrict__ par) {")
        text += [f"    io.template ld<0, {n}, {e}>({P_(a)});" for a, n, e in zr]
        text += [f"    io.template ld<1, {n}, {nz + e}>({P_(a)});" for a, n, e in lr]
        text.append("  }")
        text.append("  template <class IO> __device__ static __forceinline__ void fill(IO& io) {")
        text += [f"    io.template put<{n}, {e}>();" for a, n, e in zr]
        text += [f"    io.template put<{n}, {nz + e}>();" for a, n, e in lr]
        text.append("  }")
        text.append("  template <class IO> __device__ static __forceinline__ void run(IO& io, const double sg, const int* __restrict__ par) {")
        text.append("#pragma clang fp contract(off)")
        text.append("    const double* __restrict__ T = io.T + io.lane; (void)T; (void)sg; (void)par;")
        text += [ln for ln in body if ln]
        text.append("  }")
        text = "\n".join(text)
        if text not in shapes:
            shapes[text] = len(shapes)
            parts.append(f"template <> struct LaneGrp{KIND}<{shapes[text]}> {{")
            parts.append(text)
            parts.append("};")
        shape_of.append(shapes[text])
        par_off.append(len(par_all))
        par_all += par
    # the global rows: a constant table of the code object (row pointers, sources -- scratch slot, -1: 1.0, <= -2: column of z --,
    # coefficients, output array and index); mpx_asml_*_global sums them with lanes <-> evaluation points
    # (every row padded with terms (0.0, the constant 1.0) to a multiple of 8, the long rows of 64: mpx_assembly_lanes.h)
    gptr, gsrc, gcoef, garr, gidx, glong = [0], [], [], [], [], []
    for r in plan.global_rows:
        nt = int(ptr[r + 1] - ptr[r])
        for e in range(ptr[r], ptr[r + 1]):
            x = int(src[e])
            gsrc.append(plan.sid[x] if x >= 0 else x)
            gcoef.append(float(coef[e]))
        pad = (-nt) % (64 if nt > thr else 8) if nt else 8
        gsrc += [-1] * pad
        gcoef += [0.0] * pad
        gptr.append(len(gsrc))
        a, i = P.array_of(r)
        garr.append(a), gidx.append(i), glong.append(1 if nt > thr else 0)
    arr = lambda v, f=str: "{" + ", ".join(f(x) for x in (v if len(v) else [0])) + "}"
    parts.append(f"__device__ const int lane_shape_{kind}[] = {arr(shape_of)};")
    parts.append(f"__device__ const int lane_poff_{kind}[] = {arr(par_off)};")
    parts.append(f"__device__ const int lane_par_{kind}[] = {arr(par_all)};")
    parts.append(f"__device__ const int lane_gptr_{kind}[] = {arr(gptr)};")
    parts.append(f"__device__ const int lane_gsrc_{kind}[] = {arr(gsrc)};")
    parts.append(f"__device__ const double lane_gcoef_{kind}[] = {arr(gcoef, _cfloat)};")
    parts.append(f"__device__ const int lane_garr_{kind}[] = {arr(garr)};")
    parts.append(f"__device__ const int lane_gidx_{kind}[] = {arr(gidx)};")
    parts.append(f"__device__ const int lane_glong_{kind}[] = {arr(glong)};")
    parts.append("}  // namespace mpxgen")
    strides = [o.nnz_hess_] if kind == "hes" else [1, o.n_g_, o.n_z_, o.nnz_jac_]
    parts.append(f"#define MPX_LANE_{KIND}_GROUPS {len(plan.groups)}")
    parts.append(f"#define MPX_LANE_{KIND}_SHAPES {len(shapes)}")
    plan.n_shapes = len(shapes)
    parts.append(f"#define MPX_LANE_{KIND}_TILE_ROWS {plan.tile_rows}")
    parts.append(f"#define MPX_LANE_{KIND}_NGLOBAL {len(plan.global_rows)}")
    parts.append(f"#define MPX_LANE_{KIND}_NSID {len(plan.sid)}")
    parts.append(f"#define MPX_LANE_{KIND}_THR {int(thr)}")
    parts.append(f"#define MPX_LANE_{KIND}_CHECK {o.nnz_hess_ if kind == 'hes' else o.nnz_jac_}")
    parts.append(f"#define MPX_LANE_{KIND}_HASH {pattern_hash(*((o.hrow, o.hcol) if kind == 'hes' else (o.jrow, o.jcol)), o.n_z_, o.n_g_)}")
    parts.append(f"#define MPX_LANE_{KIND}_STRIDES {arr(strides)}")
    parts.append(f"MPX_INSTANTIATE_LANES({kind}, {KIND}, {0 if kind == 'hes' else 1})")
    return "\n".join(parts)


def pattern_hash(row, col, n_z, n_g):
    """31-bit hash of a pattern IN ITS ORDER (+ the sizes): the lane kernels bake the group-major entry order of one transcription
    into their stores, so libmpx compares this number (mpx_assembly.cpp: lane_pattern_hash, the same arithmetic) with the pattern of
    the context a code object is attached to -- an object generated for another order with the same nnz is refused."""
    i = np.arange(len(row), dtype=np.uint64)
    m = np.uint64(0xFFFFFFFF)
    v = ((np.asarray(row, np.uint64) * np.uint64(73856093)) & m) ^ ((np.asarray(col, np.uint64) * np.uint64(19349663)) & m) ^ ((i * np.uint64(83492791)) & m)
    h = (int(v.sum()) + 2654435761 * (int(n_z) & 0xFFFFFFFF) + 40503 * (int(n_g) & 0xFFFFFFFF)) & 0xFFFFFFFF
    return h & 0x7FFFFFFF


def common_source(o):
    """Definitions both passes share (row strides of z and lam_g: compile-time constants of the address arithmetic -- the host always
    passes dense arrays), then the skeleton."""
    return "\n".join(["#ifndef MPX_LANE_LDW", f"#define MPX_LANE_LDW {LDW}", "#endif", f"#define MPX_LANE_ZS {o.n_z_}", f"#define MPX_LANE_LS {o.n_g_}",
                      '#include "mpx_assembly_lanes.h"'])
