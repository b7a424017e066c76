"""The scripts under examples/ run end to end on the GPU and print what they promise."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *args], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_moon_lander_example(tmp_path):
    out = run("moon_lander.py", "--plot", str(tmp_path / "moon.png"))
    assert "J = 8.2467" in out and (tmp_path / "moon.png").stat().st_size > 10000


def test_hyper_sensitive_h_adaptive_example():
    out = run("hyper_sensitive_h_adaptive.py")
    assert "refinements" in out and "segment width fractions" in out


def test_two_phase_schwartz_example():
    assert "Solve_Succeeded" in run("two_phase_schwartz.py")


def test_adaptive_widths_example():
    assert "J = 8.2462" in run("adaptive_widths.py")


def test_batched_oracles_example():
    assert "M evaluations/s" in run("batched_oracles.py")
