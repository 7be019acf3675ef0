"""Pins the CPU oracle (oracle/ivx_oracle.c) to the reference's own known-answer
tables and fixture files, and cross-checks its fast forms
against brute force and against the reference's vendored superintervals
structure compiled from /root/reference (oracle/_ref, build container only)."""
import os

import numpy as np
import pyarrow.parquet as pq
import pytest

from conftest import GOLDEN, encode_keys, pair_set, synth
from oracle import oracle as orc


def _triples(names, key, s, e, rows):
    return [None if r == orc.NULL_IDX else [names[key[r]], int(s[r]), int(e[r])] for r in rows]


def _canon(rows):
    return sorted(rows, key=lambda r: repr(r))


@pytest.mark.parametrize("idx", range(3))
def test_join_golden(golden, idx):
    case = golden.cases("join")[idx]
    b, p = golden.rows(case["build"]), golden.rows(case["probe"])
    names, ((bk, bs, be), (pk, ps, pe)) = encode_keys(b, p)
    qs, qe = ps.copy(), pe.copy()
    bs2, be2 = bs.copy(), be.copy()
    if case["strict"]:
        # SQL `<`/`>`: both sides' END shrinks by one (intervals.rs:85-115)
        be2 -= 1
        qe -= 1
    for brute in (True, False):
        ob, op = orc.join(bk, bs2, be2, pk, qs, qe, brute=brute)
        got = [[x, y] for x, y in zip(_triples(names, bk, bs, be, ob), _triples(names, pk, ps, pe, op))]
        assert _canon(got) == _canon(case["expect"])


def test_join_range_only(golden):
    case = golden.cases("join_nokey_count")[0]
    b, p = golden.rows(case["build"]), golden.rows(case["probe"])
    _, ((bk, bs, be), (pk, ps, pe)) = encode_keys(b, p)
    ob, _ = orc.join(np.zeros_like(bk), bs, be, np.zeros_like(pk), ps, pe)
    assert len(ob) == case["expect_rows"]


@pytest.mark.parametrize("idx", range(2))
def test_join_nearest_golden(golden, idx):
    # Algorithm::CoitreesNearest inside IntervalJoinExec = nearest_one(.., true),
    # one output row per probe row, NULL build side when the key group is absent
    case = golden.cases("join_nearest")[idx]
    b, p = golden.rows(case["build"]), golden.rows(case["probe"])
    names, ((bk, bs, be), (pk, ps, pe)) = encode_keys(b, p)
    bs2, be2, qs, qe = bs.copy(), be.copy(), ps.copy(), pe.copy()
    if case["strict"]:
        be2 -= 1
        qe -= 1
    ob, op, _ = orc.nearest(bk, bs2, be2, pk, qs, qe, k=1, overlap=True)
    got = [[x, y] for x, y in zip(_triples(names, bk, bs, be, ob), _triples(names, pk, ps, pe, op))]
    assert _canon(got) == _canon(case["expect"])


@pytest.mark.parametrize("op", ["count_overlaps", "coverage"])
def test_count_coverage_golden(golden, op):
    for case in golden.cases(op):
        b, p = golden.rows(case["build"]), golden.rows(case["probe"])
        _, ((bk, bs, be), (pk, ps, pe)) = encode_keys(b, p)
        fn = orc.count_overlaps if op == "count_overlaps" else orc.coverage
        got = fn(bk, bs, be, pk, ps, pe, strict=case["strict"])
        assert got.tolist() == case["expect"], case["name"]


def test_coverage_parquet_golden():
    """R/tests/integration_test.rs:726-817: fBrain-DS14718 -> exons, strict,
    438 694 rows against the polars-bio generated expected_coverage.parquet."""
    d = os.path.join(GOLDEN, "data", "ranges")
    left = pq.read_table(os.path.join(d, "fBrain-DS14718")).to_pandas()
    right = pq.read_table(os.path.join(d, "exons")).to_pandas()
    exp = pq.read_table(os.path.join(d, "expected_coverage.parquet")).to_pandas()
    names = sorted(set(left.contig) | set(right.contig))
    ids = {n: i for i, n in enumerate(names)}
    bk = left.contig.map(ids).to_numpy(np.uint32)
    pk = right.contig.map(ids).to_numpy(np.uint32)
    cov = orc.coverage(bk, left.pos_start.to_numpy(), left.pos_end.to_numpy(),
                       pk, right.pos_start.to_numpy(), right.pos_end.to_numpy(), strict=True)
    right["coverage"] = cov
    a = right.sort_values(["contig", "pos_start", "pos_end", "coverage"]).reset_index(drop=True)
    b = exp.sort_values(["contig", "pos_start", "pos_end", "coverage"]).reset_index(drop=True)
    assert len(a) == len(b) == 438694
    assert (a.contig.values == b.contig.values).all()
    assert (a.pos_start.values == b.pos_start.values).all()
    assert (a.pos_end.values == b.pos_end.values).all()
    assert (a.coverage.values == b.coverage.values).all()
    assert int(cov.sum()) == 12060428 and int((cov != 0).sum()) == 51432


def test_merge_intervals_unit(golden):
    for st in golden.cases("merge_intervals")[0]["sets"]:
        s = np.array([x[0] for x in st["in"]], np.int32)
        e = np.array([x[1] for x in st["in"]], np.int32)
        ms, me = orc.merge_intervals(s, e)
        assert [[int(a), int(b)] for a, b in zip(ms, me)] == st["out"]


def test_nearest_index_unit(golden):
    for st in golden.cases("nearest_unit")[0]["sets"]:
        recs = st["records"]
        nrow = (max(r[2] for r in recs) + 1) if recs else 0
        # row index is the tie-break `position`; rows not named by the test sit in another key
        bk = np.ones(nrow, np.uint32); bs = np.zeros(nrow, np.int32); be = np.zeros(nrow, np.int32)
        for s, e, pos in recs:
            bk[pos], bs[pos], be[pos] = 0, s, e
        ob, _, _ = orc.nearest(bk, bs, be, [0], [st["q"][0]], [st["q"][1]], k=st["k"], overlap=st["overlap"])
        got = [int(x) for x in ob if x != orc.NULL_IDX]
        assert got == st["out"], st


def test_nearest_golden(golden):
    for case in golden.cases("nearest"):
        b, p = golden.rows(case["build"]), golden.rows(case["probe"])
        names, ((bk, bs, be), (pk, ps, pe)) = encode_keys(b, p)
        ob, op, od = orc.nearest(bk, bs, be, pk, ps, pe, k=case["k"], overlap=case["overlap"], strict=case["strict"])
        lt, rt = _triples(names, bk, bs, be, ob), _triples(names, pk, ps, pe, op)
        got = [[x, y, None if x is None else int(d)] for x, y, d in zip(lt, rt, od)]
        assert _canon(got) == _canon(case["expect"]), case["name"]


def test_merge_golden(golden):
    for case in golden.cases("merge"):
        rows = golden.rows(case["input"])
        names, ((k, s, e),) = encode_keys(rows)
        ok, os_, oe, on = orc.merge(k, s, e, min_dist=case["min_dist"], strict=case["strict"])
        got = [[names[a], int(b), int(c), int(d)] for a, b, c, d in zip(ok, os_, oe, on)]
        assert got == case["expect"], case["name"]      # exact order: key asc, start asc


def test_merge_saturating_boundary():
    # cur_end + min_dist saturates instead of wrapping (merge.rs:291; the rule is pinned for
    # cluster, which shares it, at R/tests/integration_test.rs:2638-2687)
    big = np.iinfo(np.int64).max
    ok, os_, oe, on = orc.merge([0, 0], [0, 100], [big - 1, 200], min_dist=big, strict=False)
    assert on.tolist() == [2] and oe.tolist() == [big - 1]


def test_subtract_golden(golden):
    for case in golden.cases("subtract"):
        l, r = golden.rows(case["left"]), golden.rows(case["right"])
        names, ((lk, ls, le), (rk, rs, re)) = encode_keys(l, r)
        ok, os_, oe, orow = orc.subtract(lk, ls, le, rk, rs, re, strict=case["strict"])
        got = [[names[a], int(b), int(c)] for a, b, c in zip(ok, os_, oe)]
        assert got == case["expect"], case["name"]
        for row, a, b in zip(orow, os_, oe):             # every fragment lies inside its left row
            assert ls[row] <= a < b <= le[row]


def test_cluster_golden(golden):
    for case in golden.cases("cluster"):
        rows = golden.rows(case["input"])
        names, ((k, s, e),) = encode_keys(rows)
        c = orc.cluster(k, s, e, min_dist=case["min_dist"], strict=case["strict"], n_keys=len(names))
        got = [[names[a], int(b), int(cc), int(d), int(f), int(g)] for a, b, cc, d, f, g in
               zip(c["key"], c["start"], c["end"], c["cluster"], c["cluster_start"], c["cluster_end"])]
        assert got == case["expect"], case["name"]          # exact order: contig, start, end
        assert c["n_clusters"] == len({r[3] for r in case["expect"]})
        assert int(c["key_clusters"].sum()) == c["n_clusters"]
        for r, row in zip(rows and [rows[i] for i in c["row"]], got):   # out_row points at the input row
            assert r[1:3] == row[1:3]


def test_cluster_key_base_is_coordinator_offset():
    # a partition that holds only contigs 1 and 3 of four gets their global first ids from the
    # coordinator's exclusive scan (cluster.rs:396-417): ids are then identical to the one-partition run
    k = np.array([0, 0, 1, 1, 1, 2, 3, 3], np.uint32)
    s = np.array([0, 50, 0, 5, 100, 7, 0, 1000], np.int64); e = s + 10
    full = orc.cluster(k, s, e, n_keys=4)
    base = np.concatenate([[0], np.cumsum(full["key_clusters"])[:-1]]).astype(np.int64)
    sel = (k == 1) | (k == 3)
    part = orc.cluster(k[sel], s[sel], e[sel], n_keys=4, key_base=base)
    want = full["cluster"][np.isin(full["key"], [1, 3])]
    assert part["cluster"].tolist() == want.tolist()


def test_complement_golden(golden):
    for case in golden.cases("complement"):
        rows, view = golden.rows(case["input"]), case["view"] or []
        names, ((k, s, e), (vk, vs, ve)) = encode_keys(rows, view)
        ok, os_, oe = orc.complement(k, s, e, vk, vs, ve, strict=case["strict"])
        got = [[names[a], int(b), int(c)] for a, b, c in zip(ok, os_, oe)]
        assert got == case["expect"], case["name"]


def test_take_matches_arrow_take():
    # f3: the restatement of `compute::take` against pyarrow's implementation of the same Arrow kernel
    import pyarrow as pa
    import pyarrow.compute as pc
    rng = np.random.default_rng(0)
    vals = rng.integers(-50, 50, 300).astype(np.int32); mask = rng.random(300) < 0.2
    idx = rng.integers(0, 300, 2000).astype(np.uint32); idx[::7] = orc.NULL_IDX
    pidx = pa.array(idx, mask=idx == orc.NULL_IDX)
    out, valid = orc.take_fixed(vals, idx, src_valid=~mask)
    assert pa.array(out, mask=valid == 0).equals(pc.take(pa.array(vals, mask=mask), pidx))
    py = [None if i % 11 == 0 else "s" * (i % 9) + str(i) for i in range(300)]
    col = pa.array(py)
    vb, ob, db = col.buffers()
    off = np.frombuffer(ob, np.int32)[:301]
    sv = np.unpackbits(np.frombuffer(vb, np.uint8), bitorder="little")[:300]
    o, d, v = orc.take_utf8(off, np.frombuffer(db, np.uint8), idx, src_valid=sv)
    got = pa.Array.from_buffers(pa.string(), len(idx), [pa.py_buffer(np.packbits(v, bitorder="little")), pa.py_buffer(o), pa.py_buffer(d)])
    assert got.to_pylist() == pc.take(col, pidx).to_pylist()
    with pytest.raises(IndexError):
        orc.take_fixed(vals, np.array([300], np.uint32))


def test_check_i32():
    # array_utils.rs:33-66: first offending row is reported
    assert orc.check_i32([1, 2, 3]) == -1
    assert orc.check_i32([1, 2**31 - 1, 2**31, 5]) == 2
    assert orc.check_i32([-2**31 - 1]) == 0


@pytest.mark.parametrize("seed", range(6))
def test_tree_join_matches_brute_force(seed):
    nk = [1, 3, 7][seed % 3]
    bk, bs, be = synth(400, 100 + seed, nkeys=nk, mean_len=[50, 1000, 20000][seed % 3], span=100_000)
    pk, ps, pe = synth(900, 200 + seed, nkeys=nk + 1, mean_len=150, span=100_000)
    if seed >= 3:                                        # inverted / degenerate rows on both sides
        be[::17] = bs[::17] - 5
        pe[::13] = ps[::13] - 3
    a = pair_set(*orc.join(bk, bs, be, pk, ps, pe, brute=True))
    ob, op, cnt = orc.join(bk, bs, be, pk, ps, pe, threads=[1, 4][seed % 2], per_row=True)
    assert (pair_set(ob, op) == a).all()
    assert (np.bincount(op, minlength=len(pk)) == cnt).all()
    ex = orc.join_exists(bk, bs, be, pk, ps, pe)
    assert (ex == (cnt > 0)).all()


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("seed", range(4))
def test_oracle_matches_reference_superintervals(seed):
    """The reference's own vendored structure (superintervals.hpp, compiled from
    /root/reference into oracle/_ref) returns the same overlap sets and counts."""
    nk = [1, 4][seed % 2]
    bk, bs, be = synth(5000, 300 + seed, nkeys=nk, mean_len=[1000, 40][seed // 2], span=400_000)
    pk, ps, pe = synth(20000, 400 + seed, nkeys=nk, mean_len=150, span=400_000)
    a = pair_set(*orc.ref_join(bk, bs, be, pk, ps, pe))
    b = pair_set(*orc.join(bk, bs, be, pk, ps, pe))
    assert len(a) == len(b) and (a == b).all()
    assert (orc.ref_count(bk, bs, be, pk, ps, pe) == orc.count_overlaps(bk, bs, be, pk, ps, pe)).all()


def _exons_cluster_case(golden):
    import pyarrow.parquet as pq
    case = golden.cases("cluster_exons")[0]
    t = pq.read_table(os.path.join(GOLDEN, "data", "ranges", "exons")).to_pandas()
    names = sorted(set(t.contig), key=lambda x: x.encode())
    ids = {n: i for i, n in enumerate(names)}
    return case, names, t.contig.map(ids).to_numpy(np.uint32), t.pos_start.to_numpy(np.int64), t.pos_end.to_numpy(np.int64)


def _selected_cluster_rows(case, names, c):
    want = {(r[0], r[1], r[2]) for r in case["select"]}
    rows = [[names[k], int(s), int(e), int(cs), int(ce)] for k, s, e, cs, ce in
            zip(c["key"], c["start"], c["end"], c["cluster_start"], c["cluster_end"]) if (names[k], int(s), int(e)) in want]
    return sorted(rows)


def test_cluster_exons_issue_373(golden):
    # 438 694 exons: the cluster extents the reference pins for five exons (one of them present 11 times)
    case, names, k, s, e = _exons_cluster_case(golden)
    c = orc.cluster(k, s, e, n_keys=len(names))
    assert _selected_cluster_rows(case, names, c) == sorted(case["expect"])


def test_nearest_threaded_equals_single():
    # the threaded k = 1 form the full-size GPU check uses == the plain one
    rng = np.random.default_rng(5)
    nb, npr, nk = 30_000, 50_000, 5
    bk = rng.integers(0, nk, nb).astype(np.uint32); bs = rng.integers(0, 400_000, nb).astype(np.int32)
    be = (bs + rng.integers(0, 900, nb)).astype(np.int32)
    bs[::19] = bs[1::19][: len(bs[::19])]
    pk = rng.integers(0, nk + 1, npr).astype(np.uint32); ps = rng.integers(0, 400_000, npr).astype(np.int32)
    pe = (ps + rng.integers(0, 300, npr)).astype(np.int32)
    for ovl in (True, False):
        for strict in (False, True):
            wb, wp, wd = orc.nearest(bk, bs, be, pk, ps, pe, k=1, overlap=ovl, strict=strict)
            gb, gd = orc.nearest1(bk, bs, be, pk, ps, pe, overlap=ovl, strict=strict, threads=4)
            assert (wp == np.arange(npr)).all() and (gb == wb).all() and (gd == wd).all()
