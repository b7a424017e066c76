"""Per-kernel resource summary of a problem's code object, without a GPU: VGPRs, AGPRs, SGPRs, LDS, scratch, code bytes (from
`hipcc -S` of the generated source; cross-compiles gfx950).  A/B of kernel edits before they go to a GPU box.
    python tools/isa_summary.py config2|config3|config4|config5|<problem> <S> <deg> <scheme> [substring of kernel names]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import mpopt_amd as M
from mpopt_amd import _lib, mp
import problems


def source_of(args):
    if args[0].startswith("config"):
        builder, S, po, scheme = problems.BENCH_CASES[{"config2": 0, "config3": 1, "config4": 2, "config5": 3}[args[0]]]
        rest = args[1:]
    else:
        builder, S, po, scheme = getattr(problems, args[0]), int(args[1]), int(args[2]), args[3]
        rest = args[4:]
    orders = [po] * S if isinstance(po, int) else list(po)
    o = M.NlpFunctions(builder(mp, M.math), S, orders, scheme, with_device=False)
    return o.source, rest


def summary(source, flt=""):
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "k.hip")
        open(src, "w").write(source)
        cmd = [_lib.hipcc(), f"--offload-arch={_lib.ARCH}", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-I", _lib.CSRC, "-o", os.path.join(d, "k.s"), src]
        extra = os.environ.get("MPX_HIPCC_FLAGS")
        if extra:
            cmd[1:1] = extra.split()
        subprocess.check_call(cmd)
        txt = open(os.path.join(d, "k.s")).read()
    out = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        name, body = m.group(1), m.group(2)
        if flt not in name:
            continue
        g = lambda k: (re.search(rf"\.amdhsa_{k} (\S+)", body) or [None, "?"])[1]
        md = re.search(rf"\.name:\s+{re.escape(name)}\n(.*?)(?=\n  - \.|\Z)", txt, re.S)
        vg = re.search(rf"; NumVgprs: (\d+)", txt[txt.find(name + ":"):]) if (name + ":") in txt else None
        blk = txt[txt.find("\n" + name + ":"):]
        stats = {k: (re.search(rf"; {k}: (\d+)", blk) or [None, "?"])[1] for k in ("NumVgprs", "NumAgprs", "NumSgprs", "ScratchSize", "LDSByteSize", "codeLenInByte", "Occupancy")}
        out.append((name, stats))
    return out


if __name__ == "__main__":
    src, rest = source_of(sys.argv[1:])
    for name, st in summary(src, rest[0] if rest else ""):
        print(f"{name:40s} " + "  ".join(f"{k}={v}" for k, v in st.items()))
