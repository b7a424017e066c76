"""Build, load and bind libmpx (the C-ABI library of include/mpx.h) and JIT the per-problem kernels.

There is deliberately no fallback: if the shared library is missing it is built with hipcc, and if
that is impossible every entry point raises.  Nothing in the product path evaluates the NLP on the
CPU.
"""
import ctypes
import hashlib
import os
import shutil
import contextlib
import fcntl
import subprocess
import tempfile
import threading

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB_PATH = os.path.join(PKG, "libmpx.so")
JIT_DIR = os.path.join(PKG, "_jit_cache")
ARCH = "gfx950"

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_int64_p = ctypes.POINTER(ctypes.c_int64)

MPX_F, MPX_G, MPX_GRAD, MPX_JAC, MPX_HESS, MPX_BOUNDARY_ONLY, MPX_JAC_VARIABLE_ONLY, MPX_CCS_ORDER = 1, 2, 4, 8, 16, 32, 64, 128
MPX_WIDTHS_UNCHANGED = 256
MPX_OWNER_RESIDENT = 512
MPX_MID_RESID = 1024
SCHEMES = {"LGR": 0, "LGL": 1, "CGL": 2, "LG": 3}
SCHEME_EQUI = 4


class MpxError(RuntimeError):
    pass


class MpxSchemeError(MpxError, ValueError):
    """A collocation scheme that cannot be transcribed (LG: p nodes where the composite builders need p + 1).  Also a ValueError: the
    reference fails with one at the same point -- ``mpopt(ocp, S, p, "LG")`` constructs, ``create_nlp()`` raises ValueError from
    get_composite_differentiation_matrix (mpopt.py:99 -> 4032-4038), for every grid and both D_MATRIX_METHODs."""


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise MpxError("hipcc not found: libmpx and the collocation kernels cannot be built (no CPU fallback exists)")


def _sources():
    return [os.path.join(CSRC, f) for f in ("mpx_host.cpp", "mpx_layout.cpp", "mpx_shard.cpp", "mpx_equal_area.cpp", "mpx_assembly.cpp", "mpx_colloc.cpp", "mpx_casadi.cpp", "mpx_device.h",
                                            "mpx_internal.h", "mpx_scan.h", "mpx_assembly_kernels.h")] + [os.path.join(INCLUDE, "mpx.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


@contextlib.contextmanager
def _build_lock(path):
    """Exclusive advisory lock for a build step: one process per GPU means N ranks may reach a cold cache at once."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def _flags_stamp():
    """The extra flags the library on disk was built with (diagnostics builds: MPX_LIB_HIPCC_FLAGS); '' for a plain build."""
    try:
        with open(LIB_PATH + ".flags") as f:
            return " ".join(f.read().split())
    except OSError:
        return ""


def build_library(force=False, verbose=False):
    """hipcc -> mpopt_amd/libmpx.so (host runtime + generic kernels, gfx950).  Objects and the library are written under
    private temporary names and moved into place atomically; a file lock serialises concurrent builders.  The flags of
    MPX_LIB_HIPCC_FLAGS are part of the staleness check: a diagnostics build neither poisons later plain runs nor is silently
    skipped."""
    want = " ".join(os.environ.get("MPX_LIB_HIPCC_FLAGS", "").split())  # normalised like the stamp (a whitespace-only value = no flags)
    if not force and not _stale(LIB_PATH, _sources()) and _flags_stamp() == want:
        return LIB_PATH
    with _build_lock(os.path.join(PKG, ".build.lock")):
        if not force and not _stale(LIB_PATH, _sources()) and _flags_stamp() == want:
            return LIB_PATH  # another process built it while we waited
        with tempfile.TemporaryDirectory(dir=PKG, prefix=".build_") as tmp:
            return _build_library_in(tmp, verbose)


def _build_library_in(tmp, verbose):
    cc = hipcc()
    obj = os.path.join(tmp, "mpx_colloc.o")
    hobj = os.path.join(tmp, "mpx_host.o")
    cobj = os.path.join(tmp, "mpx_casadi.o")
    aobj = os.path.join(tmp, "mpx_assembly.o")
    sobj, eobj = os.path.join(tmp, "mpx_shard.o"), os.path.join(tmp, "mpx_equal_area.o")
    lobj = os.path.join(tmp, "mpx_layout.o")
    out = os.path.join(tmp, "libmpx.so")
    extra = os.environ.get("MPX_LIB_HIPCC_FLAGS", "").split()  # diagnostics builds (-DMPX_EA_STAMPS ...)
    cmds = [
        ["g++", "-O2", "-std=c++17", "-fPIC", "-I", INCLUDE, "-c", os.path.join(CSRC, "mpx_colloc.cpp"), "-o", obj],
        ["g++", "-O2", "-std=c++17", "-fPIC", "-I", INCLUDE, "-c", os.path.join(CSRC, "mpx_casadi.cpp"), "-o", cobj],
        [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-c",
         os.path.join(CSRC, "mpx_host.cpp"), "-o", hobj] + extra,
        [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-c",
         os.path.join(CSRC, "mpx_layout.cpp"), "-o", lobj] + extra,
        [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-c",
         os.path.join(CSRC, "mpx_assembly.cpp"), "-o", aobj] + extra,
        [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-c",
         os.path.join(CSRC, "mpx_shard.cpp"), "-o", sobj] + extra,
        [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-c",
         os.path.join(CSRC, "mpx_equal_area.cpp"), "-o", eobj] + extra,
        [cc, f"--offload-arch={ARCH}", "-fPIC", "-shared", hobj, lobj, aobj, sobj, eobj, obj, cobj, "-o", out],
    ]
    for cmd in cmds:
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose:
            print(" ".join(cmd), "\n", r.stdout, r.stderr)
        if r.returncode:
            raise MpxError("building libmpx failed: " + " ".join(cmd) + "\n" + r.stderr[:6000])
    os.replace(out, LIB_PATH)
    stamp = LIB_PATH + ".flags"
    if extra:
        with open(stamp, "w") as f:
            f.write(" ".join(extra))
    elif os.path.exists(stamp):
        os.remove(stamp)
    return LIB_PATH


class mpx_problem(ctypes.Structure):
    _fields_ = [
        ("version", ctypes.c_int32), ("n_phases", ctypes.c_int32), ("nx", ctypes.c_int32), ("nu", ctypes.c_int32),
        ("na", ctypes.c_int32), ("n_segments", ctypes.c_int32), ("poly_orders", c_int32_p), ("scheme", ctypes.c_int32),
        ("tau0", ctypes.c_double), ("tau1", ctypes.c_double), ("n_links", ctypes.c_int32), ("links", c_int32_p),
        ("structure", c_int32_p), ("structure_len", ctypes.c_int64), ("code_object", ctypes.c_void_p),
        ("code_object_size", ctypes.c_size_t), ("device", ctypes.c_int32),
    ]


class mpx_sizes(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in
                ("n_z", "n_p", "n_g", "nnz_jac", "nnz_hess", "n_nodes", "n_tiles", "bytes_fgj", "bytes_hess")]


class mpx_point_set(ctypes.Structure):
    _fields_ = [
        ("fid", ctypes.c_int32), ("n_points", ctypes.c_int32), ("n_loc", ctypes.c_int32), ("n_cst", ctypes.c_int32),
        ("n_out", ctypes.c_int32), ("n_jac", ctypes.c_int32), ("n_hess", ctypes.c_int32),
        ("loc_nterm", c_int32_p), ("loc_idx", c_int32_p), ("loc_coef", c_double_p), ("cst", c_double_p),
        ("mu_nterm", c_int32_p), ("mu_idx", c_int32_p), ("mu_coef", c_double_p),
    ]


class mpx_gather(ctypes.Structure):
    _fields_ = [("n_rows", ctypes.c_int64), ("ptr", c_int64_p), ("src", c_int32_p), ("coef", c_double_p)]


class mpx_assembly(ctypes.Structure):
    _fields_ = [
        ("version", ctypes.c_int32), ("n_z", ctypes.c_int64), ("n_g", ctypes.c_int64), ("nnz_jac", ctypes.c_int64),
        ("nnz_hess", ctypes.c_int64), ("n_sets", ctypes.c_int32), ("sets", ctypes.POINTER(mpx_point_set)),
        ("fgj", mpx_gather), ("hess", mpx_gather),
        ("jac_row", c_int32_p), ("jac_col", c_int32_p), ("hess_row", c_int32_p), ("hess_col", c_int32_p),
        ("code_object", ctypes.c_void_p), ("code_object_size", ctypes.c_size_t), ("device", ctypes.c_int32),
    ]


_lib = None
_lock = threading.Lock()

# every symbol include/mpx.h declares
SYMBOLS = {
    "mpx_colloc_n_nodes": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "mpx_colloc_roots": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, c_double_p]),
    "mpx_colloc_diff_matrix": (ctypes.c_int, [c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p]),
    "mpx_colloc_quad_weights": (ctypes.c_int, [c_double_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, c_double_p]),
    "mpx_colloc_interp_matrix": (ctypes.c_int, [c_double_p, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p]),
    "mpx_create": (ctypes.c_int, [ctypes.POINTER(mpx_problem), ctypes.POINTER(ctypes.c_void_p)]),
    "mpx_create_assembled": (ctypes.c_int, [ctypes.POINTER(mpx_assembly), ctypes.POINTER(ctypes.c_void_p)]),
    "mpx_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mpx_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "mpx_get_sizes": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(mpx_sizes)]),
    "mpx_pattern_jac": (ctypes.c_int, [ctypes.c_void_p, c_int32_p, c_int32_p]),
    "mpx_pattern_hess": (ctypes.c_int, [ctypes.c_void_p, c_int32_p, c_int32_p]),
    "mpx_ccs_perm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int64_p, c_int64_p]),
    "mpx_get_comp_weights": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "mpx_geometry_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "mpx_env_dynamic": (ctypes.c_int, [ctypes.c_int]),
    "mpx_env_knob": (ctypes.c_char_p, [ctypes.c_char_p]),
    "mpx_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "mpx_set_mid_resid_output": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "mpx_eval": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 7),
    "mpx_eval_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 7),
    "mpx_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "mpx_eval_grad_gamma": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 4),
    "mpx_eval_grad_gamma_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 4),
    "mpx_host_alloc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "mpx_host_free": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "mpx_host_register": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "mpx_host_unregister": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "mpx_current_pin_buffers": (ctypes.c_int, [ctypes.c_int]),
    "mpx_current_set_casadi_abi": (ctypes.c_int, [ctypes.c_int]),
    "mpx_current_set_hess_l_output_name": (ctypes.c_int, [ctypes.c_char_p]),
    "mpx_current_keep_jac_constants": (ctypes.c_int, [ctypes.c_int]),
    "mpx_current_jac_stats": (ctypes.c_int, [ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]),
    "mpx_pattern_jac_variable": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8)]),
    "mpx_current_cache_stats": (ctypes.c_int, [ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]),
    "mpx_current_pin_stats": (ctypes.c_int, [ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]),
    "mpx_set_tile_range": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]),
    "mpx_get_tile_jac_range": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_int64_p, c_int64_p]),
    "mpx_get_tile_weights": (ctypes.c_int, [ctypes.c_void_p, c_int64_p]),
    "mpx_get_tile_spans": (ctypes.c_int, [ctypes.c_void_p, c_int32_p, c_int32_p, c_int32_p]),
    "mpx_get_notes": (ctypes.c_char_p, [ctypes.c_void_p]),
    "mpx_get_light_plan": (ctypes.c_int, [ctypes.c_void_p, c_int32_p, c_int64_p, c_int64_p, c_int64_p]),
    "mpx_get_assembled_plan": (ctypes.c_int, [ctypes.c_void_p, c_int32_p, c_int32_p, c_int32_p]),
    "mpx_assembled_attach_kernels": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "mpx_get_partials": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p), c_int64_p]),
    "mpx_shard_setup": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "mpx_shard_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int64_p, c_int64_p, c_int64_p]),
    "mpx_shard_owned": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_int64_p, c_int64_p]),
    "mpx_device_pci_bus_id": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int]),
    "mpx_shard_table": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int64_p]),
    "mpx_shard_pack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "mpx_shard_unpack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "mpx_set_current": (ctypes.c_int, [ctypes.c_void_p]),
    "mpx_resid_plan_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int64_p, c_double_p, ctypes.POINTER(ctypes.c_void_p)]),
    "mpx_resid_plan_create_order": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int64_p, c_double_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "mpx_resid_plan_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mpx_resid_eval": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 7),
    "mpx_resid_eval_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 7),
    "mpx_equal_area_widths_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_double]),
    "mpx_timer_start": (ctypes.c_int, [ctypes.c_void_p]),
    "mpx_timer_stop": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "mpx_profile": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "mpx_profile_read": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_int64_p]),
}


def lib():
    """The loaded library (built on first use)."""
    global _lib
    with _lock:
        if _lib is None:
            try:  # share torch's HIP runtime when torch is around, so device pointers interoperate
                import torch  # noqa: F401
            except Exception:  # pragma: no cover
                pass
            path = build_library()
            L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(L, name)
                fn.restype, fn.argtypes = res, args
            _lib = L
    return _lib


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().mpx_last_error(ctx)
        raise MpxError(f"libmpx error {rc}: {msg.decode() if msg else ''}")


def compile_kernels(source, verbose=False):
    """hipcc --genco of the generated problem source (+ mpx_kernels.h) -> code object bytes.
    Cached on disk under mpopt_amd/_jit_cache keyed by the full text of everything compiled."""
    h = hashlib.sha256()
    h.update(source.encode())
    for dep in ("mpx_kernels.h", "mpx_assembly_kernels.h", "mpx_assembly_fused.h", "mpx_assembly_lanes.h", "mpx_device.h"):
        with open(os.path.join(CSRC, dep), "rb") as f:
            h.update(f.read())
    flags = os.environ.get("MPX_HIPCC_FLAGS", "").split()
    if os.environ.get("MPX_TABLES_STREAM_ABOVE"):  # the host side reads the same variable at context creation (mpx_device.h)
        flags.append(f"-DMPX_TABLES_STREAM_ABOVE={max(12, min(255, int(os.environ['MPX_TABLES_STREAM_ABOVE'])))}")
    h.update(" ".join(flags).encode())
    key = h.hexdigest()[:24]
    os.makedirs(JIT_DIR, exist_ok=True)
    co = os.path.join(JIT_DIR, f"mpx_{key}.hsaco")
    if not os.path.exists(co):
        # private temporary names (several ranks may compile the same key at once), atomic move into the cache
        fd, src_tmp = tempfile.mkstemp(dir=JIT_DIR, prefix=f"mpx_{key}.", suffix=".hip")
        with os.fdopen(fd, "w") as f:
            f.write(source)
        co_tmp = src_tmp[:-4] + ".hsaco.tmp"
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "--genco", "-I", CSRC, "-o", co_tmp, src_tmp]
        cmd[1:1] = flags
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose:
            print(" ".join(cmd), "\n", r.stdout, r.stderr)
        if r.returncode:
            for t in (src_tmp, co_tmp):
                with contextlib.suppress(OSError):
                    os.remove(t)
            raise MpxError("kernel compilation failed: " + " ".join(cmd) + "\n" + r.stderr[:8000])
        os.replace(src_tmp, os.path.join(JIT_DIR, f"mpx_{key}.hip"))  # kept next to the code object for inspection
        os.replace(co_tmp, co)
    with open(co, "rb") as f:
        return f.read(), co


def gpu_available():
    try:
        import torch

        return bool(torch.cuda.is_available())
    except Exception:
        return os.path.exists("/dev/kfd")
