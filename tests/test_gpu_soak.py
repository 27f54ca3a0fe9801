"""A slice of the random soaks (tools/soak*.py) inside the suite, so that what the driver runs includes randomised parity against the
oracle and not only fixed cases: GC-RANSAC configurations (samplers, scorings incl. the truncated MSAC, pre-checks, exit, LO knobs),
end-to-end FR() cases incl. ragged batched calls, large NN pairs in four descriptor distributions, the stages around RANSAC, and
non-finite inputs.  Each script asserts equality with the oracle case by case and prints '<name> soak ok'.  Needs an MI355X."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,cases", [("soak_gc.py", 400), ("soak_fr.py", 120), ("soak_nn_big.py", 5), ("soak_misc.py", 60), ("soak_nonfinite.py", 100),
                                          ("soak.py", 60)])
def test_random_soak_slice(script, cases):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(cases)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "soak ok" in r.stdout, r.stdout[-1500:]
