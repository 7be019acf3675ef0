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
