"""Long GPU soaks, kept out of `-m gpu` so that the driver's suite stays well inside its time limit (580 s of 1200): run with
    python -m pytest tests/test_slow_gpu.py -m gpu_slow -q
on a GPU box.  r05 results on MI355X: profiles/r05_slow_suite.log.
  * the eleven differential fuzzers at 8x the case count of the `-m gpu` run (3 min 40 s on the r05 box; their decoder / multimodal cases
    run the CPU oracle on the host)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu_slow
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_differential_fuzz_soak():
    from tests.test_fuzz_gpu import FUZZERS
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_all.py"), "4"], capture_output=True, text=True, timeout=3000)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    results = {l.split()[1]: l for l in r.stdout.splitlines() if l.startswith("RESULT ")}
    for f in FUZZERS:
        assert f in results and " rc=0 " in results[f] and " 0 failures" in results[f], (f, results.get(f), tail)
