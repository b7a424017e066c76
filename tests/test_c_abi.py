"""The C ABI from C: tests/c_abi/capi_driver.c (plain C, only include/mpx.h) is compiled with gcc, linked against
libmpx.so and driven with a problem description dumped from Python.  Without a GPU it must create a structure-only
context and report sizes / patterns; on the GPU it evaluates a batch through mpx_eval and the values are compared with the
reference goldens."""
import os
import struct
import subprocess

import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import _lib
from helpers import assert_coo_close, build_case, load_golden, rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def build_driver(tmp_path):
    _lib.build_library()
    exe = str(tmp_path / "capi_driver")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "c_abi", "capi_driver.c"),
           "-o", exe, "-L", os.path.dirname(_lib.LIB_PATH), "-lmpx", f"-Wl,-rpath,{os.path.dirname(_lib.LIB_PATH)}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def dump_problem(path, o, code=True):
    ocp = o.ocp
    links = np.asarray(ocp.phase_links, dtype=np.int32).reshape(-1)
    co = o.code_object if (code and o.code_object is not None) else b""
    with open(path, "wb") as f:
        f.write(struct.pack("<8q", ocp.n_phases, ocp.nx, ocp.nu, ocp.na, o.n_segments, _lib.SCHEMES[o.scheme], len(links) // 2, len(o.structure)))
        f.write(struct.pack("<2d", -1.0, 1.0))
        f.write(struct.pack("<q", len(co)))
        f.write(np.ascontiguousarray(o.poly_orders, dtype=np.int32).tobytes())
        f.write(links.tobytes())
        f.write(np.ascontiguousarray(o.structure, dtype=np.int32).tobytes())
        f.write(co)


def read_outputs(path):
    raw = open(path, "rb").read()
    n_z, n_p, n_g, nnz_j, nnz_h, B = struct.unpack_from("<6q", raw, 0)
    off = 48

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a

    out = dict(n_z=n_z, n_p=n_p, n_g=n_g, nnz_j=nnz_j, nnz_h=nnz_h, B=B)
    out["jr"], out["jc"] = take(np.int32, nnz_j), take(np.int32, nnz_j)
    out["hr"], out["hc"] = take(np.int32, nnz_h), take(np.int32, nnz_h)
    out["colind"] = take(np.int64, n_z + 1)
    if B:
        out["f"], out["g"] = take(np.float64, B), take(np.float64, B * n_g).reshape(B, n_g)
        out["grad"] = take(np.float64, B * n_z).reshape(B, n_z)
        out["jv"], out["hv"] = take(np.float64, B * nnz_j).reshape(B, nnz_j), take(np.float64, B * nnz_h).reshape(B, nnz_h)
    return out


@pytest.mark.parametrize("name", ["moon_lander_20x3_LGR", "kitchen_sink_mixed_CGL"])
def test_c_caller_structure_only(name, tmp_path):
    exe = build_driver(tmp_path)
    ocp, mpo, o = build_case(name, with_device=False)
    dump_problem(tmp_path / "problem.bin", o, code=False)
    r = subprocess.run([exe, str(tmp_path / "problem.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = read_outputs(tmp_path / "out.bin")
    assert (out["n_z"], out["n_p"], out["n_g"], out["nnz_j"], out["nnz_h"]) == (o.n_z, o.n_p, o.n_g, o.nnz_jac, o.nnz_hess)
    jr, jc = o.jac_pattern()
    hr, hc = o.hess_pattern()
    assert np.array_equal(out["jr"], jr) and np.array_equal(out["jc"], jc) and np.array_equal(out["hr"], hr) and np.array_equal(out["hc"], hc)
    assert out["colind"][-1] == o.nnz_jac
    G = load_golden(name)
    assert set(zip(out["jr"].tolist(), out["jc"].tolist())) >= set(zip(G["jac_row"].tolist(), G["jac_col"].tolist()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["moon_lander_20x3_LGR", "kitchen_sink_mixed_CGL"])
def test_c_caller_evaluates_on_the_gpu(name, tmp_path):
    exe = build_driver(tmp_path)
    G = load_golden(name)
    ocp, mpo, o = build_case(name, with_device=True)
    dump_problem(tmp_path / "problem.bin", o)
    B = 3
    Z = np.stack([G["z"], G["z0"], 0.5 * (G["z"] + G["z0"])])
    lam = np.stack([G["lam"], G["lam"] * 0.5, -G["lam"]])
    sig = np.array([float(G["sigma"]), 1.0, 0.0])
    with open(tmp_path / "inputs.bin", "wb") as f:
        f.write(struct.pack("<q", B))
        for a in (Z, G["p"], lam, sig):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    r = subprocess.run([exe, str(tmp_path / "problem.bin"), str(tmp_path / "inputs.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = read_outputs(tmp_path / "out.bin")
    assert out["B"] == B
    assert rel_err(out["f"][0], G["f"]) < 1e-10 and rel_err(out["g"][0], G["g"]) < 1e-10 and rel_err(out["grad"][0], G["grad_f"]) < 1e-10
    assert_coo_close(out["jr"], out["jc"], out["jv"][0], G["jac_row"], G["jac_col"], G["jac_val"], 1e-10, "jac_g from C")
    assert_coo_close(out["hr"], out["hc"], out["hv"][0], G["hess_row"], G["hess_col"], G["hess_val"], 1e-10, "hess_l from C")
    ref = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], Z, G["p"], lam_g=lam, sigma=sig)  # same library through ctypes: bitwise
    assert np.array_equal(ref["jac_g"], out["jv"]) and np.array_equal(ref["hess_l"], out["hv"]) and np.array_equal(ref["g"], out["g"])
    o.close()
