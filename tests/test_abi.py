"""CPU-only: the C-ABI library loads and exports every symbol include/ivx.h declares
(no compute calls without a GPU), and refuses to run without a device."""
import os
import re
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx  # noqa: E402


def _declared():
    src = open(os.path.join(ROOT, "include", "ivx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ivx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    L = pyivx.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libivx_hip.so does not export {n}"
    assert sorted(pyivx.SYMBOLS) == names


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pyivx.IvxError) as ei:
        pyivx.Ctx(0)
    assert ei.value.status == pyivx.ERR_NO_DEVICE


def test_product_path_does_not_touch_oracle():
    """Nothing under the package may import, link or name the checker."""
    pkg = os.path.join(ROOT, "datafusion-bio-functions_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".hip", ".hpp", ".h", ".py", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dp, f)
