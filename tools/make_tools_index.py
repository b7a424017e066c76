"""Regenerates tools/README.md: one line per script (the first line of its own description), grouped by the round that wrote it."""
import ast
import collections
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
rows = []
for f in sorted(os.listdir(HERE)):
    p = os.path.join(HERE, f)
    if not os.path.isfile(p) or f == "README.md":
        continue
    first = ""
    txt = open(p, errors="replace").read()
    if f.endswith(".py"):
        try:
            first = (ast.get_docstring(ast.parse(txt)) or "").strip().split("\n")[0]
        except SyntaxError:
            pass
    if not first:
        for line in txt.splitlines()[:12]:
            l = line.strip().lstrip('#/*" ').strip()
            if l and not l.startswith("!") and "bin/bash" not in l and l not in ("set -e", "set -u"):
                first = l
                break
    rows.append((f, first[:170]))
groups = collections.OrderedDict()
for f, d in rows:
    m = re.match(r"r(\d)_", f)
    groups.setdefault(f"round {m.group(1)}" if m else "general / rounds 1-2", []).append((f, d))
out = ["# tools/ — measurement, A/B and soak scripts (none is part of the product or of the test suite)\n",
       "One-off scripts kept because `DESIGN.md`, `profiles/*/README.md` and the commit history cite them.  First line of each file's own description; grouped by the round that wrote it.",
       "Scripts that switch libmpx's A/B knobs inside one process set `MPX_ENV_DYNAMIC=1` themselves (include/mpx.h: `mpx_env_dynamic`).  (`python tools/make_tools_index.py` rewrites this file.)\n"]
for g in sorted(groups, key=lambda s: (s != "general / rounds 1-2", s)):
    out += [f"## {g}\n", "| file | what it does |\n|---|---|"] + [f"| `{f}` | {d.replace('|', '/')} |" for f, d in groups[g]] + [""]
open(os.path.join(HERE, "README.md"), "w").write("\n".join(out))
print(len(rows), "files")
