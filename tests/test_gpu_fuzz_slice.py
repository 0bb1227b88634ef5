"""A bounded, seeded slice of the wide parity sweeps (scripts/fuzz_parity.py, fuzz_assoc.py, fuzz_ip.py) inside `pytest -m gpu`: random
shapes, detector configurations, image statistics, association inputs with graded near-duplicates / ties / cutting thresholds, and the
boofcv-ip front end, every case against the CPU oracle through the C ABI -- a few hundred cases, well under a minute.  The full sweeps
(thousands of cases, millions of key points) are run by hand; their totals are recorded in profiles/r03_fuzz_summary.txt."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.mark.parametrize("script,seed,cases", [("fuzz_parity", 20260003, 40), ("fuzz_assoc", 20260003, 120), ("fuzz_ip", 20260003, 60)])
def test_seeded_fuzz_slice(script, seed, cases, capsys):
    from boofcv_amd import api
    api.Context.default()   # fails loudly without a GPU / without libboofhip.so
    mod = __import__(script)
    rc = mod.main(seed, cases)
    out = capsys.readouterr().out
    assert rc == 0 and "MISMATCH" not in out and "EXCEPTION" not in out, out[-2000:]
