"""GPU: reduced runs of the randomised parity campaigns under tools/ (the full ones -- 500 / 200 / 120 / 80 cases -- are quoted in DESIGN.md):
random decks, wells, patterns and matrices against the CPU oracle through the C ABI.  Each campaign asserts its own bounds and exits
non-zero on a violation."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,ncases,seed", [("fuzz_parity.py", 60, 31), ("fuzz_linsolver.py", 30, 32), ("fuzz_wells.py", 25, 33), ("fuzz_newton.py", 25, 34)])
def test_randomised_campaign(gpu_lib, tool, ncases, seed):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "opm-simulators-legacy_amd"), ROOT, env.get("PYTHONPATH", "")])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(ncases), str(seed)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (tool, r.stdout[-2000:], r.stderr[-3000:])
    last = [l for l in r.stdout.splitlines() if l.startswith("cases")][-1]
    assert "VIOLATIONS" not in r.stdout, last
