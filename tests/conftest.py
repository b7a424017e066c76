import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# The tests switch the library's A/B knobs (MPX_NO_LIGHT, MPX_BPB, ...) inside one process: have libmpx read them at every call, as it
# did up to round 5, instead of once per process (include/mpx.h: mpx_env_dynamic).  Inherited by the subprocesses the tests start.
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    """Worst PER-ENTRY relative error of every entry class the parity tests compared (helpers.assert_entries), so that the figures
    land in the test log also when the tests pass and output is captured (north_star: 1e-10 relative for FP64 residuals/derivatives)."""
    try:
        from helpers import ENTRY_LOG
    except Exception:
        return
    if not ENTRY_LOG:
        return
    groups = {}
    for what, n, worst, floor in ENTRY_LOG:
        case, _, cls = what.partition(" ")
        g = groups.setdefault(cls or case, [0, 0, -1.0, ""])
        g[0] += 1
        g[1] += n
        if worst > g[2]:
            g[2], g[3] = worst, case
    tr = terminalreporter
    tr.write_sep("-", "per-entry parity: worst |a-b| / max(|b|, class floor) by entry class (tolerance 1e-10)")
    for cls, (cnt, n, worst, case) in sorted(groups.items(), key=lambda kv: -kv[1][2]):
        tr.write_line(f"  {worst:9.2e}  {cls}  ({cnt} comparisons, {n} entries; worst in {case})")
