"""CPU-only: the host layer's column checks reproduce the reference's messages
(R/src/array_utils.rs:33-172, :178-295) without touching a GPU."""
import os
import sys

import pyarrow as pa
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import bio_ranges as br  # noqa: E402


def _t(**cols):
    return pa.table(cols)


def test_int32_and_in_range_int64_pass():
    assert br.check_position_column(_t(p=pa.array([1, 2, 3], pa.int32())), "p") is None
    assert br.check_position_column(_t(p=pa.array([1, 2**31 - 1], pa.int64())), "p") is None
    assert br.check_position_column(_t(p=pa.array([5, 2**31 - 1], pa.uint64())), "p") is None


def test_i32_overflow_message():
    # "coordinate value {v} at row {i} overflows i32 (max 2147483647)", array_utils.rs:39-44
    msg = br.check_position_column(_t(p=pa.array([1, 2**31, 7], pa.int64())), "p")
    assert msg == "coordinate value 2147483648 at row 1 overflows i32 (max 2147483647)"
    msg = br.check_position_column(_t(p=pa.array([-2**31 - 1], pa.int64())), "p")
    assert msg == "coordinate value -2147483649 at row 0 overflows i32 (max 2147483647)"
    msg = br.check_position_column(_t(p=pa.array([0, 0, 4294967295], pa.uint32())), "p")
    assert msg == "coordinate value 4294967295 at row 2 overflows i32 (max 2147483647)"


def test_i64_path_accepts_wide_values_and_rejects_u64_overflow():
    assert br.check_position_column(_t(p=pa.array([2**40], pa.int64())), "p", as_i64=True) is None
    msg = br.check_position_column(_t(p=pa.array([2**63], pa.uint64())), "p", as_i64=True)
    assert msg == "coordinate value 9223372036854775808 at row 0 overflows i64 (max 9223372036854775807)"


def test_null_coordinates_rejected():
    t = _t(p=pa.array([1, None, 3], pa.int32()))
    assert br.check_position_column(t, "p") == "coordinate column contains null values; nearest requires non-null coordinates"
    assert br.check_position_column(t, "p", as_i64=True) == "coordinate column contains null values; requires non-null coordinates"


def test_missing_and_unsupported_columns():
    t = _t(contig=pa.array(["a"]), p=pa.array([1.5]))
    assert "start column 'q' not found in batch with columns" in br.check_position_column(t, "q")
    assert "expected Int32, Int64, UInt32, or UInt64" in br.check_position_column(t, "p")


def test_sliced_batches_respect_offsets():
    t = _t(p=pa.array([2**40, 1, 2, 3], pa.int64())).slice(1)
    assert br.check_position_column(t, "p") is None


def test_contig_column_types_and_null_contigs(monkeypatch):
    # ContigArray (array_utils.rs:10-24, :196-229): Utf8 / LargeUtf8 / Utf8View
    for ty in (pa.string(), pa.large_string(), pa.string_view()):
        assert br.check_contig_column(_t(c=pa.array(["chr1", "", "chrX"], ty)), "c") is None
    assert "expected Utf8, LargeUtf8, or Utf8View" in br.check_contig_column(_t(c=pa.array([1, 2])), "c")
    assert "contig column 'q' not found in batch with columns" in br.check_contig_column(_t(c=pa.array(["a"])), "q")
    # NULL contigs pass by default (the reference never reads the validity bitmap: bio_ranges_host.h); a strict session --
    # here, without a GPU, the BIO_STRICT_NULL_CONTIGS switch -- refuses them (sliced batches: the row is counted in the slice)
    t = _t(c=pa.array(["chr1", "chr2", None, "chr3"]))
    assert br.check_contig_column(t, "c") is None
    monkeypatch.setenv("BIO_STRICT_NULL_CONTIGS", "1")
    assert br.check_contig_column(t, "c") == "contig column 'c' contains a NULL at row 2; NULL contigs are not supported"
    assert br.check_contig_column(t.slice(1), "c") == "contig column 'c' contains a NULL at row 1; NULL contigs are not supported"
    assert br.check_contig_column(t.slice(3), "c") is None


def test_host_checks_under_address_and_ub_sanitizers():
    """The host library's column checks (offset arithmetic on sliced Arrow buffers, error formatting) once more against an
    ASan + UBSan build of host/bio_ranges_host.cpp, in a child process with the sanitizer runtimes preloaded (sanitizers
    run on CPU code only; the GPU tiers use the normal build)."""
    import shutil
    import subprocess
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
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert " passed" in p.stdout and "1 skipped" in p.stdout
