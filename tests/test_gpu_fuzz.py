"""-m gpu: a short run of the randomised parity fuzz (tools/fuzz.py): random shapes, key counts, length
distributions (incl. heavy tails and inverted rows), sorted / unsorted probes, strict / weak, direct and
region-partitioned paths, every operator bit-exact against the oracle."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_short(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz.py"), "25", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
