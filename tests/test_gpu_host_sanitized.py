"""-m gpu: the whole host-API suite once more against an ASan + UBSan build of host/bio_ranges_host.cpp (Arrow buffer
arithmetic, nested / dictionary take, the join stream's batch bookkeeping), in a child process with the sanitizer runtimes
preloaded.  Only the host C++ is instrumented -- the device library is the normal build (GPU sanitizers are not used)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_host_api_suite_under_address_and_ub_sanitizers():
    if os.environ.get("BRH_LIB"):
        pytest.skip("already inside the sanitizer child")
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    rts = [subprocess.run([gxx, f"-print-file-name={n}"], capture_output=True, text=True).stdout.strip() for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(r) and os.path.exists(r) for r in rts):
        pytest.skip("sanitizer runtimes not installed")
    pkg = os.path.join(ROOT, "datafusion-bio-functions_amd")
    subprocess.check_call(["make", "-s", "-C", pkg, "asan"])
    env = dict(os.environ, LD_PRELOAD=":".join(rts), BRH_LIB=os.path.join(pkg, "lib", "asan", "libbio_ranges_hip.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:protect_shadow_gap=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_host_api.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert " passed" in p.stdout
