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


def test_rust_ffi_declarations_match_the_header():
    # integration/rust/hip_ffi.rs cannot be compiled here (no Rust toolchain): at least every function it
    # declares exists in include/ivx.h with the same number of parameters
    import re

    def sigs(text, pat):
        out = {}
        for m in re.finditer(pat, text, re.S):
            args = re.sub(r"/\*.*?\*/", "", m.group(2), flags=re.S)
            out[m.group(1)] = 0 if args.strip() in ("", "void") else args.count(",") + 1
        return out

    hdr = sigs(open(os.path.join(ROOT, "include", "ivx.h")).read(), r"\b(ivx_\w+)\s*\(([^;{]*?)\)\s*;")
    rust = sigs(open(os.path.join(ROOT, "integration", "rust", "hip_ffi.rs")).read(), r"pub fn (ivx_\w+)\(([^;]*?)\)\s*(?:->\s*[\w\* ]+)?;")
    assert len(rust) >= 19
    assert {k: hdr.get(k) for k in rust} == rust


def test_rust_join_stream_declarations_match_the_header():
    # integration/rust/join_stream_ffi.rs: the brh_* functions it declares exist in include/bio_ranges_host.h with the
    # same number of parameters, and the host library exports them
    import ctypes
    import re

    def sigs(text, pat):
        out = {}
        for m in re.finditer(pat, text, re.S):
            args = re.sub(r"/\*.*?\*/", "", m.group(2), flags=re.S)
            out[m.group(1)] = 0 if args.strip() in ("", "void") else args.count(",") + 1
        return out

    hdr = sigs(open(os.path.join(ROOT, "include", "bio_ranges_host.h")).read(), r"\b(brh_\w+)\s*\(([^;{]*?)\)\s*;")
    rust = sigs(open(os.path.join(ROOT, "integration", "rust", "join_stream_ffi.rs")).read(), r"pub fn (brh_\w+)\(([^;]*?)\)\s*(?:->\s*[\w\* ]+)?;")
    assert len(rust) >= 8
    assert {k: hdr.get(k) for k in rust} == rust
    lib = os.path.join(ROOT, "datafusion-bio-functions_amd", "lib", "libbio_ranges_hip.so")
    if os.path.exists(lib):
        try:
            h = ctypes.CDLL(lib)
        except OSError:
            pytest.skip("host library not loadable here (needs the HIP runtime)")
        for name in rust:
            assert hasattr(h, name), name
