"""Long GPU soaks, kept out of `-m gpu` so that the driver's suite stays well inside its time limit (580 s of 1200): run with
    python -m pytest tests/test_slow_gpu.py -m gpu_slow -q
on a GPU box.  r05 results on MI355X: profiles/r05_slow_suite.log.
  * the B = 256 decode regime (gemm_dec_ws_kernel panels) against the CPU oracle - the r01-r03 benchmark batch; `-m gpu` keeps the B = 448
    regime (two row groups per block), which is what bench.py runs;
  * the seven differential fuzzers at 4x the case count of the `-m gpu` run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu_slow
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def crab():
    from crab_amd.build_model import build_crab
    return build_crab("llama", visual=False, audio=False, conditioned=True)


def test_decode_batch_256_regime_vs_cpu_oracle_full_size(crab):
    from tests.test_fullsize_gpu import _decode_regime_vs_cpu_oracle
    _decode_regime_vs_cpu_oracle(crab, 256)


def test_differential_fuzz_soak():
    from tests.test_fuzz_gpu import FUZZERS
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_all.py"), "4"], capture_output=True, text=True, timeout=3000)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    results = {l.split()[1]: l for l in r.stdout.splitlines() if l.startswith("RESULT ")}
    for f in FUZZERS:
        assert f in results and " rc=0 " in results[f] and " 0 failures" in results[f], (f, results.get(f), tail)
