"""The scripts under examples/ are what a user of the reference reads first: each one runs to completion on the GPU
(own process, like a user would start it) and prints what its docstring promises."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join("examples", script)], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return p.stdout


def test_quickstart_drop_in_surface():
    out = _run("quickstart.py")
    for key in ("apply_M", "saddle", "apply_PC", "M^(1/2) W", "M_RFD", "new X", "solve", "det. step"):
        assert key in out
    assert "nan" not in out.lower()


def test_deterministic_time_steps():
    out = _run("timestep.py")
    assert "nan" not in out.lower() and len(out.splitlines()) >= 3


def test_brownian_steps_converge():
    out = _run("brownian.py")
    steps = re.findall(r"step +(\d+): +(\d+) GMRES iterations \((\S+)\), Lanczos (\d+)", out)
    assert len(steps) == 10
    assert all(int(it) < 60 and float(res) < 1e-6 and int(lz) < 100 for _, it, res, lz in steps)


def test_multi_gpu_script_on_one_rank():
    out = _run("multi_gpu_brownian.py")
    assert len(re.findall(r"step +\d+ on 1 rank", out)) == 10 and "nan" not in out.lower()


def test_diffusion_against_stokes_einstein():
    out = _run("diffusion_check.py")
    ratio = float(re.search(r"ratio ([0-9.]+)", out).group(1))
    assert 0.4 < ratio < 1.0          # hindered by the wall and the neighbours, below the bulk value
