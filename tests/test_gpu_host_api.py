"""-m gpu: the table-level operator mirror (Arrow C Data Interface -> C ABI -> HIP) against the
reference's own test tables; these read like R/tests/integration_test.rs."""
import os
import sys

import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import bio_ranges as br  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    s = br.Session(0)
    yield s
    s.close()


def table(rows, pos_type=pa.int64(), contig_type=pa.string()):
    # CSV-registered tables get Int64 positions in the reference
    return pa.table({"contig": pa.array([r[0] for r in rows], contig_type),
                     "pos_start": pa.array([r[1] for r in rows], pos_type),
                     "pos_end": pa.array([r[2] for r in rows], pos_type)})


def rows_of(t, names):
    return sorted(zip(*[t.column(n).to_pylist() for n in names]), key=repr)


def test_count_overlaps_and_coverage_csv(ctx, golden):
    reads, targets = table(golden.tables["ranges_reads"]), table(golden.tables["ranges_targets"])
    out = ctx.count_overlaps(reads, targets)                      # integration_test.rs:610-657
    assert out.schema.names == ["contig", "pos_start", "pos_end", "count"] and out.schema.field("count").type == pa.int64()
    assert out.column("count").to_pylist() == [2, 2, 2, 1, 1, 2, 2, 2, 1, 1, 0]
    out = ctx.coverage(reads, targets)                            # :666-717
    assert out.column("coverage").to_pylist() == [41, 92, 202, 1, 2, 41, 92, 202, 1, 2, 0]


@pytest.mark.parametrize("contig_type,pos_type", [(pa.string(), pa.int32()), (pa.large_string(), pa.int64()),
                                                  (pa.string_view(), pa.uint32()), (pa.string(), pa.uint64())])
def test_count_overlaps_all_accepted_column_types(ctx, golden, contig_type, pos_type):
    reads = table(golden.tables["ranges_reads"], pos_type, contig_type)
    targets = table(golden.tables["ranges_targets"], pos_type, contig_type)
    assert ctx.count_overlaps(reads, targets).column("count").to_pylist() == [2, 2, 2, 1, 1, 2, 2, 2, 1, 1, 0]


def test_count_overlaps_strict_boundary(ctx):
    reads, targets = table([("a", 190, 300)]), table([("a", 100, 190)])          # :1205-1252
    assert ctx.count_overlaps(reads, targets).column("count").to_pylist() == [1]
    assert ctx.count_overlaps(reads, targets, strict=True).column("count").to_pylist() == [0]


def test_coverage_parquet(ctx):
    d = os.path.join(GOLDEN, "data", "ranges")                                   # :726-817
    left, right = pq.read_table(os.path.join(d, "fBrain-DS14718")), pq.read_table(os.path.join(d, "exons"))
    exp = pq.read_table(os.path.join(d, "expected_coverage.parquet"))
    got = ctx.coverage(left, right, strict=True)
    key = ["contig", "pos_start", "pos_end", "coverage"]
    assert rows_of(got, key) == rows_of(exp.cast(pa.schema([("contig", pa.string()), ("pos_start", pa.int32()), ("pos_end", pa.int32()), ("coverage", pa.int64())])), key)


def test_overflowing_coordinate_is_an_error(ctx):
    l = table([("a", 1, 5)]); r = table([("a", 1, 2**31)])
    with pytest.raises(br.BioRangesError, match=r"coordinate value 2147483648 at row 0 overflows i32 \(max 2147483647\)"):
        ctx.count_overlaps(l, r)


def test_nearest_k1_bioframe_rows(ctx, golden):
    reads, targets = table(golden.tables["ranges_reads"]), table(golden.tables["ranges_targets"])
    out = ctx.nearest(reads, targets, 1, True)                                   # :1064-1103
    names = ["right_contig", "right_pos_start", "right_pos_end", "left_contig", "left_pos_start", "left_pos_end", "distance"]
    want = [("chr1", 100, 190, "chr1", 150, 250, 0), ("chr1", 200, 290, "chr1", 150, 250, 0), ("chr1", 400, 600, "chr1", 300, 501, 0),
            ("chr1", 10000, 20000, "chr1", 15000, 15000, 0), ("chr1", 22100, 22100, "chr1", 22000, 22300, 0),
            ("chr2", 100, 190, "chr2", 150, 250, 0), ("chr2", 200, 290, "chr2", 150, 250, 0), ("chr2", 400, 600, "chr2", 300, 500, 0),
            ("chr2", 10000, 20000, "chr2", 15000, 15000, 0), ("chr2", 22100, 22100, "chr2", 22000, 22300, 0),
            ("chr3", 100, 200, "chr3", 234, 300, 34)]
    assert rows_of(out, names) == sorted(want, key=repr)


def test_nearest_k2_no_overlap_and_null_row(ctx):
    l = table([("a", 10, 20), ("a", 30, 40), ("a", 50, 60)]); r = table([("a", 22, 22), ("a", 37, 37), ("b", 1, 1)])
    out = ctx.nearest(l, r, 2, False)                                            # :906-950
    got = rows_of(out, ["left_contig", "left_pos_start", "left_pos_end", "right_contig", "right_pos_start", "right_pos_end", "distance"])
    want = [("a", 10, 20, "a", 22, 22, 2), ("a", 30, 40, "a", 22, 22, 8), ("a", 10, 20, "a", 37, 37, 17), ("a", 50, 60, "a", 37, 37, 13),
            (None, None, None, "b", 1, 1, None)]
    assert got == sorted(want, key=repr)


def test_nearest_without_distance_schema(ctx):
    l = table([("a", 10, 20)]); r = table([("a", 1, 2)])
    out = ctx.nearest(l, r, 1, True, compute_distance=False)                     # :1349-1394
    assert "distance" not in out.schema.names and out.num_rows == 1


def test_sql_join_equi_and_range(ctx, golden):
    reads, targets = table(golden.tables["reads"]), table(golden.tables["targets"])
    out = ctx.sql_range_join(reads, targets)                                     # :61-137
    assert out.num_rows == 16
    case = golden.cases("join")[0]
    want = sorted([tuple(a) + tuple(b) for a, b in case["expect"]], key=repr)
    assert rows_of(out, out.schema.names) == want


def test_sql_join_strict_predicate_and_mixed_types(ctx):
    a = pa.table({"contig": ["a"], "pos_start": pa.array([5], pa.int32()), "pos_end": pa.array([10], pa.int64())})   # interval_join.rs:1724-1730
    b = table([("a", 11, 15), ("a", 10, 15), ("a", 10, 10), ("a", 9, 15), ("a", 5, 15), ("a", 4, 15), ("a", 4, 10), ("a", 6, 8), ("a", 4, 8), ("a", 4, 5), ("a", 5, 5), ("a", 4, 4)])
    assert ctx.sql_range_join(a, b).num_rows == 10                               # :235-310
    assert ctx.sql_range_join(a, b, strict_predicate=True).num_rows == 6         # :314-369


def test_join_nearest_algorithm_multikey_null_rows(ctx):
    a = pa.table({"contig": ["a"], "strand": ["s"], "start": pa.array([5], pa.int32()), "end": pa.array([10], pa.int32())})
    b = pa.table({"contig": ["a", "a", "a", "b"], "strand": ["s", "s", "x", "s"], "start": pa.array([11, 20, 0, 1], pa.int32()),
                  "end": pa.array([13, 21, 1, 2], pa.int32())})
    cols = (["contig", "strand"], "start", "end")
    bi, pi = ctx.interval_join(a, b, cols, cols, strict_predicate=True, nearest_algorithm=True)      # :373-420
    assert pi.to_pylist() == [0, 1, 2, 3] and bi.to_pylist() == [0, 0, None, None]


def test_overlap_udtf_modes(ctx, golden):
    reads, targets = table(golden.tables["ranges_reads"]), table(golden.tables["ranges_targets"])
    full = ctx.overlap(reads, targets)                                           # :1576-1610
    assert full.num_rows == 16 and full.schema.names[:3] == ["left_contig", "left_pos_start", "left_pos_end"]
    assert ctx.overlap(reads, targets, mode="left_all").num_rows == 16           # :1961-2017
    left_only = ctx.overlap(reads, targets, mode="left")                         # :1888-1958 (RIGHT SEMI)
    assert left_only.num_rows == 12 and left_only.schema.names == ["contig", "pos_start", "pos_end"]
    a, b = table([("a", 100, 190)]), table([("a", 190, 300)])
    assert ctx.overlap(a, b).num_rows == 1 and ctx.overlap(a, b, strict=True).num_rows == 0           # :1613-1650


def test_merge_udtf(ctx, golden):
    out = ctx.merge(table(golden.tables["merge_input"]), 0, strict=True)        # :2088-2116
    assert out.schema.names == ["contig", "pos_start", "pos_end", "n_intervals"]
    assert out.schema.field("pos_start").type == pa.int64()
    want = golden.cases("merge")[0]["expect"]
    assert [list(r) for r in zip(*[out.column(n).to_pylist() for n in out.schema.names])] == want
    with pytest.raises(br.BioRangesError, match="min_dist must be >= 0"):
        ctx.merge(table([("a", 1, 2)]), -1)
    assert ctx.merge(table([])).num_rows == 0                                    # :2258-2281


def test_subtract_udtf_with_extra_columns(ctx):
    left = pa.table({"contig": ["a"], "pos_start": [100], "pos_end": [600], "gene": ["BRCA1"]})      # :3667-3702
    right = table([("a", 200, 300), ("a", 400, 500)])
    out = ctx.subtract(left, right)
    assert out.schema.names == ["contig", "pos_start", "pos_end", "gene"]
    assert [list(r) for r in zip(*[out.column(n).to_pylist() for n in out.schema.names])] == \
        [["a", 100, 200, "BRCA1"], ["a", 300, 400, "BRCA1"], ["a", 500, 600, "BRCA1"]]


def test_partition_invariance_tables(ctx, golden):
    case = [c for c in golden.cases("subtract") if c["name"] == "subtract_partitioned_parquet"][0]   # :3758-3890
    out = ctx.subtract(table(case["left"]), table(case["right"]))
    assert [list(r) for r in zip(*[out.column(n).to_pylist() for n in out.schema.names])] == case["expect"]


def _rows(t):
    return [list(r) for r in zip(*[t.column(n).to_pylist() for n in t.schema.names])]


def test_cluster_udtf(ctx, golden):
    for case in golden.cases("cluster"):                                         # :2411-2757, :3792-3802
        out = ctx.cluster(table(golden.rows(case["input"])), case["min_dist"], strict=case["strict"])
        assert out.schema.names == ["contig", "pos_start", "pos_end", "cluster", "cluster_start", "cluster_end"]
        assert all(out.schema.field(n).type == pa.int64() for n in out.schema.names[1:])
        assert _rows(out) == case["expect"], case["name"]
    t = pa.table({"chr": ["a", "a"], "s": pa.array([100, 150], pa.int32()), "e": pa.array([200, 250], pa.int32())})
    out = ctx.cluster(t, 0, cols=("chr", "s", "e"))                              # :2690-2716
    assert out.schema.names[:4] == ["chr", "s", "e", "cluster"] and out.num_rows == 2
    with pytest.raises(br.BioRangesError, match="min_dist must be >= 0"):
        ctx.cluster(table([("a", 1, 2)]), -1)


def test_cluster_udtf_preserves_extra_columns(ctx):
    t = pa.table({"contig": ["a", "a", "a"], "pos_start": pa.array([400, 150, 100], pa.int32()), "pos_end": pa.array([500, 250, 200], pa.int32()),
                  "gene": ["TP53", "BRCA2", "BRCA1"], "score": [0.75, 0.85, 0.95]})      # :3594-3625
    out = ctx.cluster(t)
    assert out.schema.names == ["contig", "pos_start", "pos_end", "gene", "score", "cluster", "cluster_start", "cluster_end"]
    assert out.schema.field("pos_start").type == pa.int32()                      # input fields kept as they are
    assert _rows(out) == [["a", 100, 200, "BRCA1", 0.95, 0, 100, 250], ["a", 150, 250, "BRCA2", 0.85, 0, 100, 250],
                          ["a", 400, 500, "TP53", 0.75, 1, 400, 500]]


def test_complement_udtf(ctx, golden):
    for case in golden.cases("complement"):                                      # :2825-3175, :3803-3822
        view = table(case["view"]) if case["view"] is not None else None
        out = ctx.complement(table(golden.rows(case["input"])), view, strict=case["strict"])
        assert out.schema.names == ["contig", "pos_start", "pos_end"] and out.schema.field("pos_end").type == pa.int64()
        assert _rows(out) == case["expect"], case["name"]
    t = pa.table({"chr": ["a"], "s": [100], "e": [200]})
    v = pa.table({"c": ["a"], "b": [0], "x": [500]})
    out = ctx.complement(t, v, cols=("chr", "s", "e"), view_cols=("c", "b", "x"))   # :3093-3134, view with its own column names
    assert out.schema.names == ["chr", "s", "e"] and _rows(out) == [["a", 0, 100], ["a", 200, 500]]


def test_device_take_matches_arrow_take(ctx):
    import datetime
    import decimal
    import pyarrow.compute as pc
    n = 257
    cols = {
        "i8": pa.array([None if i % 9 == 0 else i % 100 for i in range(n)], pa.int8()),
        "u16": pa.array(range(n), pa.uint16()),
        "f32": pa.array([i / 7 for i in range(n)], pa.float32()),
        "f64": pa.array([None if i % 5 == 0 else i / 3 for i in range(n)], pa.float64()),
        "ts": pa.array([datetime.datetime(2020, 1, 1) + datetime.timedelta(hours=i) for i in range(n)], pa.timestamp("us")),
        "d32": pa.array([datetime.date(2000, 1, 1) + datetime.timedelta(days=i) for i in range(n)], pa.date32()),
        "dec": pa.array([decimal.Decimal(i) / 100 for i in range(n)], pa.decimal128(12, 2)),
        "gene": pa.array([None if i % 11 == 0 else "gene%d" % i * (i % 4) for i in range(n)], pa.string()),
        "big": pa.array(["x" * (i % 50) for i in range(n)], pa.large_string()),
        "bin": pa.array([bytes([i % 256]) * (i % 6) for i in range(n)], pa.binary()),
        "flag": pa.array([None if i % 7 == 0 else (i % 3 == 0) for i in range(n)], pa.bool_()),
        "view": pa.array([None if i % 10 == 0 else ("v%d-" % i) * (i % 7) for i in range(n)], pa.string_view()),
        "bview": pa.array([bytes([65 + i % 26]) * (i % 40) for i in range(n)], pa.binary_view()),
        # dictionary-encoded payload columns (arrow's take gathers the keys, interval_join.rs:1655-1667 takes any type)
        "dict_s": pa.array([None if i % 17 == 0 else "chr%d" % (i % 24) for i in range(n)], pa.string()).dictionary_encode(),
        "dict_i": pa.DictionaryArray.from_arrays(pa.array([None if i % 19 == 0 else i % 5 for i in range(n)], pa.int8()),
                                                 pa.array([10, 20, None, 40, 50], pa.int64())),
        "dict_ls": pa.DictionaryArray.from_arrays(pa.array([i % 3 for i in range(n)], pa.uint16()), pa.array(["+", "-", "."], pa.large_string())),
    }
    t = pa.table(cols).slice(3)                                       # non-zero offsets into every buffer, bitmaps included
    idx = pa.array([None if i % 13 == 0 else (i * 7) % t.num_rows for i in range(1000)], pa.uint32())
    for name in t.schema.names:
        col = t.column(name)
        got = ctx.take(col, idx)
        if "view" in str(col.type):                                  # this pyarrow has no take kernel for views: spell it out
            vals = col.to_pylist()
            want = [None if j is None else vals[j] for j in idx.to_pylist()]
            assert got.type == col.type and got.to_pylist() == want, name
            got.validate(full=True)
            continue
        want = pc.take(col, idx)
        assert got.type == want.type and got.to_pylist() == want.combine_chunks().to_pylist(), name
        if name.startswith("dict"):
            got.validate(full=True)
            assert got.dictionary.to_pylist() == col.chunk(0).dictionary.to_pylist()      # the dictionary rides along unchanged
    with pytest.raises(br.BioRangesError, match="unsupported column type"):
        ctx.take(pa.array([b"abc", b"def"], pa.binary(3)), pa.array([0], pa.uint32()))    # 3-byte elements: no device gather for that width
    with pytest.raises(br.BioRangesError, match="out of bounds"):
        ctx.take(pa.array([1, 2]), pa.array([2], pa.uint32()))


def test_take_nested_columns(ctx):
    """arrow's take handles any type (interval_join.rs:1655-1667): struct, list, large_list, fixed_size_list and map columns
    -- and nestings of them -- go down to the device gathers of their leaves; checked against pyarrow's take"""
    import pyarrow.compute as pc
    n = 300
    def maybe(i, v, m):
        return None if i % m == 0 else v
    cols = {
        "struct": pa.array([maybe(i, {"a": maybe(i, i, 7), "b": maybe(i, "s%d" % i, 5), "c": i % 2 == 0}, 11) for i in range(n)],
                           pa.struct([("a", pa.int64()), ("b", pa.string()), ("c", pa.bool_())])),
        "list_i32": pa.array([maybe(i, [maybe(j, i * 10 + j, 4) for j in range(i % 5)], 9) for i in range(n)], pa.list_(pa.int32())),
        "llist_str": pa.array([maybe(i, ["x" * (j + i % 3) for j in range(i % 4)], 13) for i in range(n)], pa.large_list(pa.string())),
        "fsl_f32": pa.array([maybe(i, [float(i), float(i) + 0.5, -float(i)], 6) for i in range(n)], pa.list_(pa.float32(), 3)),
        "list_struct": pa.array([maybe(i, [{"k": "q%d" % j, "v": maybe(j, j * i, 3)} for j in range(i % 3)], 8) for i in range(n)],
                                pa.list_(pa.struct([("k", pa.string()), ("v", pa.int64())]))),
        "map": pa.array([maybe(i, [("k%d" % j, j + i) for j in range(i % 3)], 10) for i in range(n)], pa.map_(pa.string(), pa.int32())),
        "struct_list": pa.array([{"xs": [i, i + 1][: i % 3], "tag": maybe(i, "t", 4)} for i in range(n)],
                                pa.struct([("xs", pa.list_(pa.int16())), ("tag", pa.string())])),
    }
    t = pa.table(cols).slice(5)                                       # non-zero offsets at every level
    idx = pa.array([None if i % 17 == 0 else (i * 11) % t.num_rows for i in range(700)], pa.uint32())
    for name in t.schema.names:
        col = t.column(name)
        got = ctx.take(col, idx)
        want = pc.take(col, idx).combine_chunks()
        got.validate(full=True)
        assert got.type == want.type and got.to_pylist() == want.to_pylist(), name
    empty = ctx.take(t.column("list_struct"), pa.array([], pa.uint32()))
    assert len(empty) == 0 and empty.type == t.column("list_struct").type
    with pytest.raises(br.BioRangesError, match="out of range"):
        ctx.take(pa.array([[1], [2]]), pa.array([2], pa.uint32()))


def test_sql_range_join_with_device_take(ctx, golden):
    reads, targets = table(golden.tables["reads"]), table(golden.tables["targets"])
    reads = reads.append_column("name", pa.array(["r%d" % i for i in range(reads.num_rows)]))
    a = ctx.sql_range_join(reads, targets, device_take=True)
    b = ctx.sql_range_join(reads, targets, device_take=False)
    assert a.schema.names == b.schema.names and rows_of(a, a.schema.names) == rows_of(b, b.schema.names) and a.num_rows == 16


def test_overlap_udtf_boundary_cases(ctx):
    # R/tests/integration_test.rs:1653-1798: the five two-table cases, row counts as asserted there
    a = table([("a", 100, 200)])
    assert ctx.overlap(a, table([("a", 200, 300)]), strict=True).num_rows == 0      # adjacent, 0-based half-open (:1653-1680)
    assert ctx.overlap(a, table([("a", 200, 300)])).num_rows == 1                   # adjacent, 1-based closed (:1683-1710)
    assert ctx.overlap(a, table([("a", 100, 200)])).num_rows == 1                   # same interval (:1713-1739)
    assert ctx.overlap(a, table([("a", 150, 180)])).num_rows == 1                   # contained (:1742-1768)
    assert ctx.overlap(a, table([("a", 300, 400), ("b", 100, 200)])).num_rows == 0  # no matches, other contig (:1771-1798)


def test_overlap_udtf_left_modes_keep_payload_and_multiplicity(ctx):
    reads = pa.table({"contig": ["chr1", "chr1", "chr1", "chr2"], "pos_start": pa.array([100, 100, 1000, 50], pa.int32()),
                      "pos_end": pa.array([200, 200, 1100, 60], pa.int32()), "name": ["dup", "dup", "miss", "other"]})
    targets = table([("chr1", 90, 150), ("chr1", 120, 180), ("chr2", 55, 56)], pos_type=pa.int32())
    left = ctx.overlap(reads, targets, mode="left")                              # :1888-1958, RightSemi: each matching left row once
    assert left.schema.names == ["contig", "pos_start", "pos_end", "name"]
    assert rows_of(left, left.schema.names) == sorted([("chr1", 100, 200, "dup"), ("chr1", 100, 200, "dup"), ("chr2", 50, 60, "other")], key=repr)
    left_all = ctx.overlap(reads, targets, mode="left_all")                      # :1961-2017, Inner projected on the left: multiplicity kept
    assert rows_of(left_all, left_all.schema.names) == sorted([("chr1", 100, 200, "dup")] * 4 + [("chr2", 50, 60, "other")], key=repr)
    a = pa.table({"chr": ["a", "a"], "s": [100, 100], "e": [200, 201], "label": ["touching", "overlap"]})
    b = pa.table({"chr": ["a"], "s": [200], "e": [300]})
    out = ctx.overlap(a, b, mode="left", cols_left=("chr", "s", "e"), cols_right=("chr", "s", "e"), strict=True)   # :2020-2061
    assert out.schema.names == ["chr", "s", "e", "label"] and _rows(out) == [["a", 100, 201, "overlap"]]


def test_overlap_udtf_custom_columns(ctx):
    # :1801-1832: left_/right_ prefixed output names follow the column arguments
    a = pa.table({"chr": ["a"], "s": [100], "e": [200]})
    b = pa.table({"chr": ["a"], "s": [150], "e": [250]})
    out = ctx.overlap(a, b, cols_left=("chr", "s", "e"), cols_right=("chr", "s", "e"))
    assert out.num_rows == 1 and out.schema.names[0] == "left_chr" and out.schema.names[3] == "right_chr"


def _stream_results(js, batches):
    results = []
    for b in batches:
        results += js.push(b)
    results += js.finish()
    return results


def _stream_pairs(js, batches, results=None, nullable_build=False):
    """push the batches, finish, and return the pairs as (build_idx, global probe row) + the number of results"""
    import numpy as np
    starts = np.cumsum([0] + [b.num_rows for b in batches])
    if results is None:
        results = _stream_results(js, batches)
    bi, pi, seen = [], [], 0
    for r in results:
        assert r["first_batch"] == seen                          # results come in push order, every batch in exactly one group
        off = r["batch_offsets"].to_numpy()
        assert len(off) == r["n_batches"] + 1 and off[0] == 0
        assert (np.diff(off) == [batches[seen + j].num_rows for j in range(r["n_batches"])]).all()
        b = r["build_idx"]
        bi.append(b.fill_null(0xFFFFFFFF).to_numpy() if nullable_build else b.to_numpy())
        pi.append(r["probe_idx"].to_numpy().astype(np.int64) + starts[seen])
        if r["group_done"]:                                      # a bounded-output stream gives several results per group
            seen += r["n_batches"]
    assert seen == len(batches) and (not results or results[-1]["group_done"])
    return (np.concatenate(bi) if bi else np.empty(0, np.uint32)), (np.concatenate(pi) if pi else np.empty(0, np.int64)), len(results)


def test_join_stream_golden_tables_in_small_batches(ctx, golden):
    # the 12 reads x 10 targets join (integration_test.rs:61-84) with the probe side pushed three rows at a time
    reads, targets = table(golden.tables["reads"]), table(golden.tables["targets"])
    want_b, want_p = ctx.interval_join(reads, targets)
    want = sorted(zip(want_b.to_pylist(), want_p.to_pylist()))
    for coalesce in (1, 5, 0):
        js = ctx.join_stream(reads, coalesce_rows=coalesce)
        bi, pi, nres = _stream_pairs(js, [targets.slice(i, 3) for i in range(0, targets.num_rows, 3)])
        js.close()
        assert sorted(zip(bi.tolist(), pi.tolist())) == want and len(want) == 16
        assert nres == {1: 4, 5: 2, 0: 1}[coalesce]


@pytest.mark.parametrize("strict", [False, True])
def test_join_stream_matches_one_shot_join(ctx, strict):
    import numpy as np
    rng = np.random.default_rng(5)
    names = np.array(["chr1", "chr10", "chr2", "chrX", "scaffold_77"])

    def tab(n, keys, mean):
        s = rng.integers(0, 3_000_000, n)
        return pa.table({"contig": pa.array(names[rng.integers(0, keys, n)]), "pos_start": pa.array(s, pa.int64()),
                         "pos_end": pa.array(s + rng.integers(1, 2 * mean, n), pa.int64())})

    build = tab(40_000, 4, 800)                                  # the build side never sees "scaffold_77"
    probe = tab(2_600_000, 5, 150)
    cuts = [0, 0, 10, 8_202, 70_000, 70_000, 1_000_000, 2_400_000, 2_600_000]      # empty batches, a tiny one, big ones
    batches = [probe.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]
    wb, wp = ctx.interval_join(build, probe, strict_predicate=strict)
    want = np.sort((wb.to_numpy().astype(np.uint64) << np.uint64(32)) | wp.to_numpy().astype(np.uint64))
    for coalesce in (50_000, 2_200_000, 0):                      # many small groups; one group on the region-partitioned path; one at finish
        js = ctx.join_stream(build, strict_predicate=strict, coalesce_rows=coalesce)
        bi, pi, nres = _stream_pairs(js, batches)
        js.close()
        got = np.sort((bi.astype(np.uint64) << np.uint64(32)) | pi.astype(np.uint64))
        assert len(got) == len(want) and (got == want).all(), coalesce


def _rand_tables(seed, n_build, n_probe, build_keys=4, probe_keys=5, bmean=800, pmean=150, span=3_000_000):
    import numpy as np
    rng = np.random.default_rng(seed)
    names = np.array(["chr1", "chr10", "chr2", "chrX", "scaffold_77"])

    def tab(n, keys, mean):
        s = rng.integers(0, span, n)
        return pa.table({"contig": pa.array(names[rng.integers(0, keys, n)]), "pos_start": pa.array(s, pa.int64()),
                         "pos_end": pa.array(s + rng.integers(1, 2 * mean, n), pa.int64())})
    return tab(n_build, build_keys, bmean), tab(n_probe, probe_keys, pmean)


@pytest.mark.parametrize("join_type", ["right_semi", "right_anti", "nearest"])
def test_join_stream_semi_anti_nearest_match_the_one_shot_join(ctx, join_type):
    """RightSemi / RightAnti (interval_join.rs:1014-1024, :1433-1463) and Algorithm::CoitreesNearest (:864-870, NULL build
    rows for keys the build side lacks) through the push interface == the one-shot operator on the concatenated probe side."""
    import numpy as np
    build, probe = _rand_tables(11, 30_000, 700_000)
    cuts = [0, 3, 8_195, 8_195, 300_000, 700_000]
    batches = [probe.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]
    if join_type == "nearest":
        wb, wp = ctx.interval_join(build, probe, nearest_algorithm=True)
        want_b, want_p = wb.fill_null(0xFFFFFFFF).to_numpy(), wp.to_numpy().astype(np.int64)
        assert (want_b == 0xFFFFFFFF).any()                          # "scaffold_77" rows have no build side
    else:
        _, wp = ctx.interval_join(build, probe, join_type=br.JOIN_RIGHT_SEMI if join_type == "right_semi" else br.JOIN_RIGHT_ANTI)
        want_p = wp.to_numpy().astype(np.int64)
    for coalesce, budget in ((100_000, 0), (0, 0), (250_000, 70_000)):
        js = ctx.join_stream(build, coalesce_rows=coalesce, join_type=join_type, max_output_rows=budget)
        results = _stream_results(js, batches)
        js.close()
        bi, pi, nres = _stream_pairs(None, batches, results, nullable_build=join_type == "nearest")
        assert (pi == want_p).all()                                  # probe rows ascending, exactly the operator's
        if join_type == "nearest":
            assert (bi == want_b).all()
        else:
            assert all(len(r["build_idx"]) == 0 for r in results)
        if budget:
            assert all(len(r["probe_idx"]) <= budget for r in results) and nres > 3


def test_join_stream_bounded_output_on_a_deep_pile_up(ctx):
    """The low-memory stream (interval_join.rs:1153-1299): a result holds whole probe rows and ends after the row at which
    its running pair count reaches the budget.  One hot region: every probe row there matches thousands of build rows."""
    import numpy as np
    rng = np.random.default_rng(9)
    bs = np.concatenate([rng.integers(1_000_000, 1_001_000, 5000), [0]])
    be = np.concatenate([bs[:-1] + rng.integers(100, 2000, 5000), [200_000_000]])
    build = pa.table({"contig": pa.array(["chr1"] * len(bs)), "pos_start": pa.array(bs, pa.int64()), "pos_end": pa.array(be, pa.int64())})
    ps = np.concatenate([rng.integers(999_000, 1_003_000, 3000), rng.integers(5_000_000, 9_000_000, 40_000)])
    rng.shuffle(ps)
    probe = pa.table({"contig": pa.array(["chr1"] * len(ps)), "pos_start": pa.array(ps, pa.int64()), "pos_end": pa.array(ps + 150, pa.int64())})
    batches = [probe.slice(a, 8192) for a in range(0, probe.num_rows, 8192)]
    wb, wp = ctx.interval_join(build, probe)
    want = np.sort((wb.to_numpy().astype(np.uint64) << np.uint64(32)) | wp.to_numpy().astype(np.uint64))
    rle = np.bincount(wp.to_numpy(), minlength=probe.num_rows)      # matches per probe row
    assert rle.max() > 4000 and len(want) > 4_000_000
    for budget, coalesce in ((100_000, 20_000), (1_000_000, 0), (7, 9_000)):
        js = ctx.join_stream(build, coalesce_rows=coalesce, max_output_rows=budget)
        results = _stream_results(js, batches)
        js.close()
        bi, pi, nres = _stream_pairs(None, batches, results)
        got = np.sort((bi.astype(np.uint64) << np.uint64(32)) | pi.astype(np.uint64))
        assert len(got) == len(want) and (got == want).all(), budget    # the same pair multiset as the unbounded join
        starts = np.cumsum([0] + [b.num_rows for b in batches])
        prev_last = -1
        for r in results:
            p = r["probe_idx"].to_numpy().astype(np.int64) + starts[r["first_batch"]]     # (pair order inside a result is free)
            if len(p) == 0:
                continue
            rows, cnt = np.unique(p, return_counts=True)
            assert (cnt == rle[rows]).all()                             # whole probe rows
            assert rows[0] > prev_last                                  # results walk the probe rows in order
            prev_last = rows[-1]
            assert len(p) < budget + rle[rows[-1]]                     # ends right after the row that reached the budget
            assert r["group_done"] or len(p) >= budget
        assert nres >= len(want) // (budget + rle.max())


def test_join_stream_budget_from_the_environment(ctx, golden, monkeypatch):
    # BIO_MAX_OUTPUT_BATCH_SIZE (interval_join.rs:543-548); the 16-pair golden join in results of >= 5 pairs
    reads, targets = table(golden.tables["reads"]), table(golden.tables["targets"])
    monkeypatch.setenv("BIO_MAX_OUTPUT_BATCH_SIZE", "5")
    js = ctx.join_stream(reads, max_output_rows="env")
    results = _stream_results(js, [targets])
    js.close()
    sizes = [len(r["probe_idx"]) for r in results]
    assert sum(sizes) == 16 and len(sizes) >= 3 and all(5 <= n for n in sizes[:-1])


def test_join_stream_errors(ctx, golden):
    reads, targets = table(golden.tables["reads"]), table(golden.tables["targets"])
    js = ctx.join_stream(reads)
    with pytest.raises(br.BioRangesError, match="contig column 'contig' not found"):
        js.push(targets.rename_columns(["chrom", "pos_start", "pos_end"]))
    with pytest.raises(br.BioRangesError, match=r"overflows i32"):
        js.push(pa.table({"contig": ["chr1"], "pos_start": pa.array([2**31], pa.int64()), "pos_end": pa.array([2**31 + 5], pa.int64())}))
    assert js.finish() == []                                     # nothing was accepted
    js.close()


def test_null_contig_is_refused_by_a_strict_session(golden):
    reads, targets = table(golden.tables["reads"]), table(golden.tables["targets"])
    bad = pa.table({"contig": pa.array(["chr1", None]), "pos_start": pa.array([1, 5], pa.int64()), "pos_end": pa.array([9, 8], pa.int64())})
    s = br.Session(0)
    s.set_strict_null_contigs(True)
    for call in (lambda: s.count_overlaps(bad, targets), lambda: s.count_overlaps(reads, bad), lambda: s.interval_join(reads, bad),
                 lambda: s.merge(bad), lambda: s.nearest(bad, targets)):
        with pytest.raises(br.BioRangesError, match=r"contains a NULL at row 1; NULL contigs are not supported"):
            call()
    s.close()


def test_null_contigs_as_the_reference_treats_them(ctx):
    """Default mode (parity unpinned by the reference's tests, bio_ranges_host.h): the table functions key a NULL slot by
    the bytes its offsets span -- "" for arrays made by the Arrow builders --, the join treats NULL as a key of its own
    that equals only NULL (create_hashes), also in the streamed form."""
    i64 = pa.int64()
    a = pa.table({"contig": pa.array(["chr1", None, "", None]), "pos_start": pa.array([1, 5, 5, 100], i64), "pos_end": pa.array([9, 8, 8, 200], i64)})
    b = pa.table({"contig": pa.array([None, "", "chr1"]), "pos_start": pa.array([6, 6, 6], i64), "pos_end": pa.array([7, 7, 7], i64)})
    # count_overlaps(b rows against a): NULL == "" there -> the NULL and the "" row of b each see a's rows 1 and 2
    got = ctx.count_overlaps(a, b)
    assert got.column("count").to_pylist() == [2, 2, 1]
    # join: NULL matches NULL only (b row 0 <-> a row 1), "" matches "" (b row 1 <-> a row 2), chr1 <-> chr1
    bi, pi = ctx.interval_join(a, b)
    assert sorted(zip(bi.to_pylist(), pi.to_pylist())) == [(0, 2), (1, 0), (2, 1)]
    js = ctx.join_stream(a, coalesce_rows=1)
    res = js.push(b) + js.finish()
    pairs = sorted((x, y) for r in res for x, y in zip(r["build_idx"].to_pylist(), r["probe_idx"].to_pylist()))
    assert pairs == [(0, 2), (1, 0), (2, 1)]
    js.close()


def test_session_metrics_and_memory_limit(golden):
    """BuildProbeJoinMetrics under the reference's names (joins/utils.rs:399-453) and the device-memory reservation
    (interval_join.rs:614-639: over the limit -> ResourcesExhausted, nothing allocated, the session stays usable)."""
    import numpy as np
    s = br.Session(0)
    build, probe = _rand_tables(21, 50_000, 400_000)
    wb, wp = s.interval_join(build, probe)
    m = s.metrics()
    assert m["build_input_batches"] == 1 and m["build_input_rows"] == 50_000 and m["build_mem_used"] > 50_000 * 12
    assert m["input_batches"] == 2 and m["input_rows"] == 2 * 400_000          # the sizing call and the fill call
    assert m["output_batches"] == 1 and m["output_rows"] == len(wb)
    assert m["build_time"] > 0 and m["join_time"] > 0
    s.close()
    s = br.Session(0)
    s.set_memory_limit(1 << 20)                                                # 1 MiB: the build needs more
    with pytest.raises(br.BioRangesError, match=r"Resources exhausted: failed to reserve \d+ bytes"):
        s.interval_join(build, probe)
    s.set_memory_limit(0)
    wb2, wp2 = s.interval_join(build, probe)                                   # the same session, no limit: fine
    key = lambda b, p: np.sort((b.to_numpy().astype(np.uint64) << np.uint64(32)) | p.to_numpy().astype(np.uint64))
    assert (key(wb2, wp2) == key(wb, wp)).all()
    s.close()
