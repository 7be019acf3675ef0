"""-m gpu: count_overlaps / coverage / nearest / merge / subtract through the C ABI,
bit-exact against the CPU oracle and the reference's golden tables."""
import os
import sys

import numpy as np
import pyarrow.parquet as pq
import pytest

from conftest import GOLDEN, ROOT, encode_keys, synth
from oracle import oracle as orc

sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pyivx.Ctx(0)
    yield c
    c.close()


def _i32(*a):
    return [x.astype(np.int32) for x in a]


def _paths():
    """count_overlaps / coverage probes: rank-grid gathers in input order ("direct"), the region partition with LDS
    slices ("regions"), rank-grid gathers over probe rows routed by coordinate region ("routed")"""
    for path in ("direct", "regions", "routed"):
        os.environ["IVX_ROWVAL_PATH"] = path
        try:
            yield path
        finally:
            del os.environ["IVX_ROWVAL_PATH"]


# ------------------------------------------------------------------ golden tables

@pytest.mark.parametrize("op", ["count_overlaps", "coverage"])
def test_count_coverage_golden(ctx, golden, op):
    for case in golden.cases(op):
        b, p = golden.rows(case["build"]), golden.rows(case["probe"])
        names, ((bk, bs, be), (pk, ps, pe)) = encode_keys(b, p)
        bs, be, ps, pe = _i32(bs, be, ps, pe)
        kind = pyivx.KIND_COUNT if op == "count_overlaps" else pyivx.KIND_COVERAGE
        ix = ctx.build(kind, bk, bs, be, n_keys=len(names))
        fn = ctx.count_overlaps if op == "count_overlaps" else ctx.coverage
        for path in _paths():
            assert fn(ix, pk, ps, pe, strict=case["strict"]).tolist() == case["expect"], (case["name"], path)


def test_coverage_parquet_golden(ctx):
    """R/tests/integration_test.rs:726-817 (438 694 rows, strict) on the HIP path."""
    d = os.path.join(GOLDEN, "data", "ranges")
    left = pq.read_table(os.path.join(d, "fBrain-DS14718")).to_pandas()
    right = pq.read_table(os.path.join(d, "exons")).to_pandas()
    exp = pq.read_table(os.path.join(d, "expected_coverage.parquet")).to_pandas()
    names = sorted(set(left.contig) | set(right.contig))
    ids = {n: i for i, n in enumerate(names)}
    bk = left.contig.map(ids).to_numpy(np.uint32); pk = right.contig.map(ids).to_numpy(np.uint32)
    ix = ctx.build(pyivx.KIND_COVERAGE, bk, left.pos_start.to_numpy(), left.pos_end.to_numpy(), n_keys=len(names))
    for path in _paths():
        cov = ctx.coverage(ix, pk, right.pos_start.to_numpy(), right.pos_end.to_numpy(), strict=True)
        assert int(cov.sum()) == 12060428 and int((cov != 0).sum()) == 51432, path
    right["coverage"] = cov
    a = right.sort_values(["contig", "pos_start", "pos_end", "coverage"]).reset_index(drop=True)
    b = exp.sort_values(["contig", "pos_start", "pos_end", "coverage"]).reset_index(drop=True)
    assert (a.coverage.values == b.coverage.values).all() and (a.pos_start.values == b.pos_start.values).all()
    assert int(cov.sum()) == 12060428 and int((cov != 0).sum()) == 51432
    # and count_overlaps on the same tables against the oracle
    ixc = ctx.build(pyivx.KIND_COUNT, bk, left.pos_start.to_numpy(), left.pos_end.to_numpy(), n_keys=len(names))
    want = orc.count_overlaps(bk, left.pos_start.to_numpy(), left.pos_end.to_numpy(), pk, right.pos_start.to_numpy(), right.pos_end.to_numpy(), strict=True)
    for path in _paths():
        got = ctx.count_overlaps(ixc, pk, right.pos_start.to_numpy(), right.pos_end.to_numpy(), strict=True)
        assert (got == want).all(), path


def _triples(names, key, s, e, rows):
    return [None if r == pyivx.NULL_IDX else [names[key[r]], int(s[r]), int(e[r])] for r in rows]


def test_nearest_golden(ctx, golden):
    for case in golden.cases("nearest"):
        b, p = golden.rows(case["build"]), golden.rows(case["probe"])
        names, ((bk, bs, be), (pk, ps, pe)) = encode_keys(b, p)
        bs, be, ps, pe = _i32(bs, be, ps, pe)
        ix = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=len(names))
        ob, op, od = ctx.nearest(ix, pk, ps, pe, k=case["k"], overlap=case["overlap"], strict=case["strict"])
        lt, rt = _triples(names, bk, bs, be, ob), _triples(names, pk, ps, pe, op)
        got = [[x, y, None if x is None else int(d)] for x, y, d in zip(lt, rt, od)]
        assert sorted(got, key=repr) == sorted(case["expect"], key=repr), case["name"]


def test_join_nearest_golden(ctx, golden):
    for case in golden.cases("join_nearest"):
        b, p = golden.rows(case["build"]), golden.rows(case["probe"])
        names, ((bk, bs, be), (pk, ps, pe)) = encode_keys(b, p)
        bs, be, ps, pe = _i32(bs, be, ps, pe)
        bs2, be2, qs, qe = bs.copy(), be.copy(), ps.copy(), pe.copy()
        if case["strict"]:
            be2 -= 1; qe -= 1
        ix = ctx.build(pyivx.KIND_NEAREST, bk, bs2, be2, n_keys=len(names))
        ob, op, _ = ctx.nearest(ix, pk, qs, qe, k=1, overlap=True, distance=False)
        got = [[x, y] for x, y in zip(_triples(names, bk, bs, be, ob), _triples(names, pk, ps, pe, op))]
        assert sorted(got, key=repr) == sorted(case["expect"], key=repr), case["name"]


def test_nearest_index_unit(ctx, golden):
    for st in golden.cases("nearest_unit")[0]["sets"]:
        recs = st["records"]
        nrow = (max(r[2] for r in recs) + 1) if recs else 0
        bk = np.ones(nrow, np.uint32); bs = np.zeros(nrow, np.int32); be = np.zeros(nrow, np.int32)
        for s, e, pos in recs:
            bk[pos], bs[pos], be[pos] = 0, s, e
        ix = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=2)
        ob, _, _ = ctx.nearest(ix, np.zeros(1, np.uint32), np.array([st["q"][0]], np.int32), np.array([st["q"][1]], np.int32),
                               k=st["k"], overlap=st["overlap"])
        assert [int(x) for x in ob if x != pyivx.NULL_IDX] == st["out"], st


def test_merge_golden(ctx, golden):
    for case in golden.cases("merge"):
        rows = golden.rows(case["input"])
        names, ((k, s, e),) = encode_keys(rows)
        ok, os_, oe, on = ctx.merge(k, s, e, n_keys=max(len(names), 1), min_dist=case["min_dist"], strict=case["strict"])
        got = [[names[a], int(b), int(c), int(d)] for a, b, c, d in zip(ok, os_, oe, on)]
        assert got == case["expect"], case["name"]


def test_subtract_golden(ctx, golden):
    for case in golden.cases("subtract"):
        l, r = golden.rows(case["left"]), golden.rows(case["right"])
        names, ((lk, ls, le), (rk, rs, re)) = encode_keys(l, r)
        ok, os_, oe, orow = ctx.subtract(lk, ls, le, rk, rs, re, n_keys=max(len(names), 1), strict=case["strict"])
        got = [[names[a], int(b), int(c)] for a, b, c in zip(ok, os_, oe)]
        assert got == case["expect"], case["name"]


# ------------------------------------------------------------------ random vs oracle

def _degenerate(seed, bs, be, ps, pe):
    rng = np.random.default_rng(seed)
    be = be.copy(); pe = pe.copy()
    be[::19] = bs[::19] - rng.integers(1, 50, len(bs[::19])).astype(bs.dtype)     # inverted build rows
    pe[::7] = ps[::7] - rng.integers(1, 5, len(ps[::7])).astype(ps.dtype)          # inverted queries
    pe[::5] = ps[::5]                                                               # points
    return be, pe


@pytest.mark.parametrize("seed", range(6))
def test_count_random(ctx, seed):
    nk = [1, 5, 24][seed % 3]
    bk, bs, be = synth(30_000, 10 + seed, nkeys=nk, mean_len=[200, 5000][seed % 2], span=2_000_000)
    pk, ps, pe = synth(120_000, 20 + seed, nkeys=nk + 1, mean_len=150, span=2_000_000)
    if seed >= 3:
        be, pe = _degenerate(seed, bs, be, ps, pe)
    ix = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=nk + 1)
    for strict in (False, True):
        want = orc.count_overlaps(bk, bs, be, pk, ps, pe, strict=strict)
        for path in _paths():                              # (inverted build rows keep the rank grids whatever is asked)
            assert (ctx.count_overlaps(ix, pk, ps, pe, strict=strict) == want).all(), (strict, path)


@pytest.mark.parametrize("chunks", ["2", "5"])
def test_count_coverage_host_columns_in_chunks(ctx, chunks):
    # host-resident columns of a big batch go through the device in chunks (values of chunk c are copied back while chunk
    # c + 1 uploads): same columns as the oracle, with and without a key column
    bk, bs, be = synth(50_000, 301, nkeys=6, mean_len=700, span=4_000_000)
    pk, ps, pe = synth(700_000, 302, nkeys=7, mean_len=150, span=4_000_000)
    os.environ["IVX_HOST_CHUNKS"] = chunks
    try:
        for kind, fn, ofn in ((pyivx.KIND_COUNT, "count_overlaps", orc.count_overlaps), (pyivx.KIND_COVERAGE, "coverage", orc.coverage)):
            ix = ctx.build(kind, bk, bs, be, n_keys=7)
            for strict in (False, True):
                assert (getattr(ctx, fn)(ix, pk, ps, pe, strict=strict) == ofn(bk, bs, be, pk, ps, pe, strict=strict)).all(), (fn, strict)
            ix.free()
            k0 = np.zeros(len(bk), np.uint32)
            ix = ctx.build(kind, None, bs, be, n_keys=1)
            assert (getattr(ctx, fn)(ix, None, ps, pe) == ofn(k0, bs, be, np.zeros(len(pk), np.uint32), ps, pe)).all(), fn
            ix.free()
    finally:
        del os.environ["IVX_HOST_CHUNKS"]


@pytest.mark.parametrize("seed", range(6))
def test_coverage_random(ctx, seed):
    nk = [1, 5, 24][seed % 3]
    mean = [30, 400, 5000][seed % 3]                      # sparse .. heavily merged
    bk, bs, be = synth(40_000, 30 + seed, nkeys=nk, mean_len=mean, span=1_500_000)
    pk, ps, pe = synth(150_000, 40 + seed, nkeys=nk + 1, mean_len=[150, 20000][seed % 2], span=1_500_000)
    if seed >= 3:
        _, pe = _degenerate(seed, bs, be, ps, pe)
    ix = ctx.build(pyivx.KIND_COVERAGE, bk, bs, be, n_keys=nk + 1)
    for strict in (False, True):
        want = orc.coverage(bk, bs, be, pk, ps, pe, strict=strict)
        for path in _paths():
            assert (ctx.coverage(ix, pk, ps, pe, strict=strict) == want).all(), (strict, path)


def test_coverage_extreme_queries(ctx):
    lo, hi = np.iinfo(np.int32).min, np.iinfo(np.int32).max
    bk, bs, be = synth(5000, 77, nkeys=1, mean_len=100, span=1_000_000)
    ps = np.array([lo, lo, 0, -5, hi - 3, 10], np.int32); pe = np.array([hi, 100, hi, 5, hi, hi - 1], np.int32)
    pk = np.zeros(len(ps), np.uint32)
    ix = ctx.build(pyivx.KIND_COVERAGE, bk, bs, be, n_keys=1)
    ixc = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=1)
    for path in _paths():
        for strict in (False, True):
            assert (ctx.coverage(ix, pk, ps, pe, strict=strict) == orc.coverage(bk, bs, be, pk, ps, pe, strict=strict)).all(), path
            assert (ctx.count_overlaps(ixc, pk, ps, pe, strict=strict) == orc.count_overlaps(bk, bs, be, pk, ps, pe, strict=strict)).all(), path


def test_count_coverage_large_batch_default_path(ctx):
    # big enough for the region partition by default: uneven keys, probe keys without build rows, a ragged
    # last chunk, long probe rows (beyond the LDS slice halo) and chromosome-long build rows (upper levels)
    bk, bs, be = synth(250_000, 51, nkeys=30, mean_len=700, span=30_000_000)
    pk, ps, pe = synth(3_300_003, 52, nkeys=33, mean_len=150, span=30_000_000)
    pe[::97] = ps[::97] + 120_000
    bs[:40] = 0; be[:40] = 29_000_000
    pe[5::1001] = ps[5::1001] - 2                          # inverted queries
    for kind, fn, ofn in ((pyivx.KIND_COUNT, ctx.count_overlaps, orc.count_overlaps), (pyivx.KIND_COVERAGE, ctx.coverage, orc.coverage)):
        ix = ctx.build(kind, bk, bs, be, n_keys=33)
        for strict in (False, True):
            assert (fn(ix, pk, ps, pe, strict=strict) == ofn(bk, bs, be, pk, ps, pe, strict=strict)).all(), (kind, strict)
        ix.free()


def test_coverage_rejects_inverted_build(ctx):
    with pytest.raises(pyivx.IvxError) as ei:
        ctx.build(pyivx.KIND_COVERAGE, np.zeros(2, np.uint32), np.array([10, 50], np.int32), np.array([5, 60], np.int32), n_keys=1)
    assert ei.value.status == pyivx.ERR_UNSUPPORTED


@pytest.mark.parametrize("seed", range(8))
def test_nearest_random(ctx, seed):
    nk = [1, 4, 24, 2][seed % 4]
    bk, bs, be = synth(20_000, 50 + seed, nkeys=nk, mean_len=[300, 3000][seed % 2], span=[3_000_000, 60_000][seed // 4])
    pk, ps, pe = synth(60_000, 60 + seed, nkeys=nk + 1, mean_len=100, span=[3_000_000, 60_000][seed // 4])
    if seed % 4 == 3:
        be, pe = _degenerate(seed, bs, be, ps, pe)
        bs[::23] = bs[1::23][: len(bs[::23])]            # duplicate starts -> tie-breaks on end/row
    ix = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=nk + 1)
    for k, ovl, strict in [(1, True, False), (1, False, False), (1, True, True), (3, True, False), (4, False, True), (2, True, True)]:
        wb, wp, wd = orc.nearest(bk, bs, be, pk, ps, pe, k=k, overlap=ovl, strict=strict)
        for path in ("direct", "routed"):                   # k = 1: gathers in input order / probe rows routed by region first
            os.environ["IVX_NEAREST_PATH"] = path
            try:
                gb, gp, gd = ctx.nearest(ix, pk, ps, pe, k=k, overlap=ovl, strict=strict)
                if k == 1:
                    gb2, gp2 = ctx.nearest(ix, pk, ps, pe, k=k, overlap=ovl, strict=strict, distance=False)[:2]
                    assert (gb2 == wb).all() and (gp2 == wp).all(), (k, ovl, strict, path, "no distance")
            finally:
                del os.environ["IVX_NEAREST_PATH"]
            assert len(gb) == len(wb), (k, ovl, strict, path)
            assert (gb == wb).all() and (gp == wp).all() and (gd == wd).all(), (k, ovl, strict, path)
    # probe rows already in (key, start) order: the routed path moves nothing and answers in place
    o = np.lexsort((ps, pk)); pk, ps, pe = pk[o].copy(), ps[o].copy(), pe[o].copy()
    wb, wp, wd = orc.nearest(bk, bs, be, pk, ps, pe, k=1)
    os.environ["IVX_NEAREST_PATH"] = "routed"
    try:
        gb, gp, gd = ctx.nearest(ix, pk, ps, pe, k=1)
    finally:
        del os.environ["IVX_NEAREST_PATH"]
    assert (gb == wb).all() and (gp == wp).all() and (gd == wd).all()


@pytest.mark.parametrize("variant", ["sorted", "one_inversion", "ends_descending"])
def test_nearest_build_rows_already_sorted(ctx, variant):
    # build rows in (key,start,end) order skip the first radix sort of the index build; duplicates keep row order
    bk, bs, be = synth(50_000, 511, nkeys=5, mean_len=400, span=500_000)
    bs[::11] = bs[1::11][: len(bs[::11])]; be[::11] = np.maximum(be[::11], bs[::11])
    bk[::40], bs[::40], be[::40] = bk[1::40][: len(bk[::40])], bs[1::40][: len(bk[::40])], be[1::40][: len(bk[::40])]
    o = np.lexsort((-be, bs, bk)) if variant == "ends_descending" else np.lexsort((be, bs, bk))
    bk, bs, be = bk[o].copy(), bs[o].copy(), be[o].copy()
    if variant == "one_inversion":
        bs[30_000] -= 40_000; be[30_000] -= 40_000
    pk, ps, pe = synth(40_000, 512, nkeys=6, mean_len=100, span=500_000)
    ix = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=6)
    for k, ovl in [(1, True), (1, False), (3, True)]:
        gb, gp, gd = ctx.nearest(ix, pk, ps, pe, k=k, overlap=ovl)
        wb, wp, wd = orc.nearest(bk, bs, be, pk, ps, pe, k=k, overlap=ovl)
        assert len(gb) == len(wb) and (gb == wb).all() and (gp == wp).all() and (gd == wd).all(), (k, ovl)
    ix.free()


def test_nearest_build_long_runs_of_equal_starts(ctx):
    # thousands of build rows share a (key,start): the (key,start)-only sort of the index build hands over to the
    # full sort; tie-breaks on end and row decide most answers
    rng = np.random.default_rng(31)
    nb = 30_000
    bk = rng.integers(0, 3, nb).astype(np.uint32)
    bs = rng.choice(rng.integers(0, 1_000_000, 25), nb).astype(np.int32)
    be = (bs + rng.integers(0, 300, nb)).astype(np.int32)
    pk, ps, pe = synth(20_000, 513, nkeys=4, mean_len=100, span=1_000_000)
    ix = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=4)
    for k, ovl in [(1, True), (1, False), (4, True)]:
        gb, gp, gd = ctx.nearest(ix, pk, ps, pe, k=k, overlap=ovl)
        wb, wp, wd = orc.nearest(bk, bs, be, pk, ps, pe, k=k, overlap=ovl)
        assert len(gb) == len(wb) and (gb == wb).all() and (gp == wp).all() and (gd == wd).all(), (k, ovl)
    ix.free()


@pytest.mark.parametrize("variant", ["many_keys", "i32_extremes", "two_word_sorts"])
def test_nearest_build_sort_forms(ctx, variant):
    # the index build sorts one 64-bit word per row (linearised (key, coordinate) ‖ row) when that fits, else two words
    rng = np.random.default_rng(77)
    nb, nk = 40_000, 5
    if variant == "many_keys":                              # more keys than the LDS copy of the key table takes
        nk = 3000
        bk, bs, be = synth(nb, 601, nkeys=nk, mean_len=300, span=200_000)
        pk, ps, pe = synth(30_000, 602, nkeys=nk, mean_len=100, span=200_000)
    elif variant == "i32_extremes":
        bk = rng.integers(0, nk, nb).astype(np.uint32)
        bs = rng.integers(-2**31, 2**31 - 1, nb, dtype=np.int64)
        be = np.minimum(bs + rng.integers(0, 2**24, nb), 2**31 - 1)
        bs[:4] = [-2**31, 2**31 - 1, -2**31, 0]; be[:4] = [-2**31, 2**31 - 1, 2**31 - 1, 0]
        bs, be = bs.astype(np.int32), be.astype(np.int32)
        pk = rng.integers(0, nk, 30_000).astype(np.uint32)
        ps = rng.integers(-2**31, 2**31 - 1, 30_000, dtype=np.int64)
        pe = np.minimum(ps + rng.integers(0, 2**20, 30_000), 2**31 - 1)
        ps, pe = ps.astype(np.int32), pe.astype(np.int32)
    else:
        bk, bs, be = synth(nb, 603, nkeys=nk, mean_len=300, span=2_000_000)
        bs[::17] = bs[1::17][: len(bs[::17])]; be[::17] = np.maximum(be[::17], bs[::17])
        pk, ps, pe = synth(30_000, 604, nkeys=nk, mean_len=100, span=2_000_000)
        os.environ["IVX_NEAREST_SORT2"] = "1"
    try:
        ix = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=nk)
    finally:
        os.environ.pop("IVX_NEAREST_SORT2", None)
    for k, ovl in [(1, True), (1, False), (3, True)]:
        gb, gp, gd = ctx.nearest(ix, pk, ps, pe, k=k, overlap=ovl)
        wb, wp, wd = orc.nearest(bk, bs, be, pk, ps, pe, k=k, overlap=ovl)
        assert len(gb) == len(wb) and (gb == wb).all() and (gp == wp).all() and (gd == wd).all(), (variant, k, ovl)
    ix.free()


def test_rank_grids_over_clustered_coordinates(ctx):
    # build rows in a few tight clusters far apart (and single rows in between): the sorted rank grids' cell tables have
    # long stretches of empty cells, which a whole wavefront fills (k_grid_bounds); nearest, coverage and count_overlaps
    # all read such grids
    rng = np.random.default_rng(2024)
    parts_k, parts_s = [], []
    for k, centres in enumerate([[1_000, 120_000_000, 240_000_000], [50_000_000], [5, 9_999_999, 10_000_000, 200_000_000]]):
        for c in centres:
            m = int(rng.integers(1, 4000))
            parts_k.append(np.full(m, k, np.uint32)); parts_s.append(c + rng.integers(0, 3000, m))
    bk = np.concatenate(parts_k); bs = np.concatenate(parts_s).astype(np.int32)
    be = (bs + rng.integers(0, 500, len(bs))).astype(np.int32)
    o = rng.permutation(len(bk)); bk, bs, be = bk[o], bs[o], be[o]
    npr = 60_000
    pk = rng.integers(0, 4, npr).astype(np.uint32)
    ps = np.where(rng.random(npr) < 0.5, rng.integers(0, 250_000_000, npr), rng.choice(bs, npr) + rng.integers(-5000, 5000, npr)).astype(np.int32)
    pe = (ps + rng.integers(0, 2000, npr)).astype(np.int32)
    ix = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=4)
    for k, ovl in [(1, True), (1, False), (3, True)]:
        gb, gp, gd = ctx.nearest(ix, pk, ps, pe, k=k, overlap=ovl)
        wb, wp, wd = orc.nearest(bk, bs, be, pk, ps, pe, k=k, overlap=ovl)
        assert len(gb) == len(wb) and (gb == wb).all() and (gp == wp).all() and (gd == wd).all(), (k, ovl)
    ix.free()
    ix = ctx.build(pyivx.KIND_COVERAGE, bk, bs, be, n_keys=4)
    assert (ctx.coverage(ix, pk, ps, pe) == orc.coverage(bk, bs, be, pk, ps, pe)).all()
    ix.free()
    ix = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=4)
    assert (ctx.count_overlaps(ix, pk, ps, pe) == orc.count_overlaps(bk, bs, be, pk, ps, pe)).all()
    ix.free()


def test_nearest_empty_build_and_k0(ctx):
    e = np.empty(0, np.int32)
    ix = ctx.build(pyivx.KIND_NEAREST, np.empty(0, np.uint32), e, e, n_keys=3)
    pk = np.array([0, 2], np.uint32); ps = np.array([5, 9], np.int32); pe = np.array([6, 12], np.int32)
    for k in (0, 1, 3):
        ob, op, od = ctx.nearest(ix, pk, ps, pe, k=k)
        assert (ob == pyivx.NULL_IDX).all() and op.tolist() == [0, 1] and (od == -1).all()


@pytest.mark.parametrize("seed", range(6))
def test_merge_random(ctx, seed):
    nk = [1, 7, 40][seed % 3]
    k, s, e = synth(200_000, 70 + seed, nkeys=nk, mean_len=[20, 300][seed % 2], span=3_000_000, dtype=np.int64)
    if seed >= 3:
        e[::29] = s[::29] - 4                              # inverted rows
        s[::31] = s[1::31][: len(s[::31])]
    md = [0, 0, 5, 1000, 0, 37][seed]
    for strict in (False, True):
        want = orc.merge(k, s, e, min_dist=md, strict=strict)
        for nolin in (False, True):                         # the sort word with and without the linearised (key, start)
            if nolin:
                os.environ["IVX_NO_LIN"] = "1"
            try:
                got = ctx.merge(k, s, e, n_keys=nk, min_dist=md, strict=strict)
            finally:
                os.environ.pop("IVX_NO_LIN", None)
            for g, w in zip(got, want):
                assert len(g) == len(w) and (g == w).all()


def test_sweeps_uneven_contigs(ctx):
    # contigs of very different lengths (and some without rows), as in a genome: the sort word numbers the
    # (contig, start) pairs consecutively, contig by contig
    rng = np.random.default_rng(91)
    spans = np.array([250_000_000, 0, 5_000, 130_000_000, 1, 0, 48_000_000, 16_000, 90_000_000, 0], np.int64)
    nk, n = len(spans), 300_000
    live = np.flatnonzero(spans > 0)
    k = rng.choice(live, n, p=spans[live] / spans[live].sum()).astype(np.uint32)
    s = (rng.random(n) * spans[k]).astype(np.int64) + rng.integers(-1000, 1000, nk)[k]
    e = s + rng.integers(0, 2_000, n)
    s[::13] = s[1::13][: len(s[::13])]; k[::13] = k[1::13][: len(k[::13])]; e[::13] = np.maximum(e[::13], s[::13])
    got = ctx.merge(k, s, e, n_keys=nk)
    want = orc.merge(k, s, e)
    for g, w in zip(got, want):
        assert len(g) == len(w) and (g == w).all()
    _same_cluster(ctx.cluster(k, s, e, n_keys=nk), orc.cluster(k, s, e, n_keys=nk))
    rk = rng.choice(live, n // 5).astype(np.uint32)
    rs = (rng.random(n // 5) * spans[rk]).astype(np.int64); re = rs + rng.integers(0, 5_000, n // 5)
    gs = ctx.subtract(k, s, e, rk, rs, re, n_keys=nk)
    ws = orc.subtract(k, s, e, rk, rs, re)
    for g, w in zip(gs, ws):
        assert len(g) == len(w) and (g == w).all()


@pytest.mark.parametrize("variant", ["sorted", "one_inversion", "all_equal", "sorted_by_start_only"])
def test_sweeps_on_coordinate_sorted_input(ctx, variant):
    """merge / cluster / complement / subtract detect input that already is in (key,start,end) order and skip the
    radix sort (k_copy_sorted); one row out of place, or end-order ties broken the other way, must sort."""
    k, s, e = synth(150_000, 411, nkeys=6, mean_len=400, span=2_000_000, dtype=np.int64)
    e += 1
    s[::9] = s[1::9][: len(s[::9])]; e[::9] = np.maximum(e[::9], s[::9] + 1)      # equal starts, different ends
    k[::50], s[::50], e[::50] = k[1::50][: len(k[::50])], s[1::50][: len(k[::50])], e[1::50][: len(k[::50])]   # fully equal rows
    if variant == "sorted_by_start_only":
        o = np.lexsort((-e, s, k))                          # ends DEscending inside equal (key,start): not sorted
    else:
        o = np.lexsort((e, s, k))
    k, s, e = k[o].copy(), s[o].copy(), e[o].copy()
    if variant == "one_inversion":
        s[70_000], e[70_000] = s[70_000] - 100_000, e[70_000] - 100_000
    if variant == "all_equal":
        k[:] = 2; s[:] = 77; e[:] = 99
    rk, rs, re = synth(40_000, 412, nkeys=7, mean_len=150, span=2_000_000, dtype=np.int64)
    re += 1
    ro = np.lexsort((re, rs, rk)); rk, rs, re = rk[ro].copy(), rs[ro].copy(), re[ro].copy()
    for force in (False, True):                            # IVX_FORCE_SORT: the same answers through the radix sort
        if force:
            os.environ["IVX_FORCE_SORT"] = "1"
        try:
            for strict in (False, True):
                for g, w in zip(ctx.merge(k, s, e, n_keys=7, min_dist=3, strict=strict), orc.merge(k, s, e, min_dist=3, strict=strict)):
                    assert len(g) == len(w) and (g == w).all()
                _same_cluster(ctx.cluster(k, s, e, n_keys=7, min_dist=0, strict=strict), orc.cluster(k, s, e, min_dist=0, strict=strict, n_keys=7))
                for g, w in zip(ctx.complement(k, s, e, rk, rs, re, n_keys=7, strict=strict), orc.complement(k, s, e, rk, rs, re, strict=strict)):
                    assert len(g) == len(w) and (g == w).all()
                for g, w in zip(ctx.subtract(k, s, e, rk, rs, re, n_keys=7, strict=strict), orc.subtract(k, s, e, rk, rs, re, strict=strict)):
                    assert len(g) == len(w) and (g == w).all()
        finally:
            os.environ.pop("IVX_FORCE_SORT", None)


@pytest.mark.parametrize("dups", ["few", "long_runs", "long_runs_in_start_order"])
def test_sweeps_sort_on_key_start_then_fix_runs(ctx, dups):
    """Unsorted sparse input is radix-sorted on (key,start) only and runs of equal (key,start) are ordered by
    (end,row) afterwards; runs longer than 64 rows send the call back to the full-width sort."""
    rng = np.random.default_rng(77)
    n = 150_000
    k = rng.integers(0, 5, n).astype(np.uint32)
    if dups == "few":
        s = rng.integers(0, 3_000_000, n).astype(np.int64)              # a few percent of the rows share a start
        s[::7] = s[1::7][: len(s[::7])]; k[::7] = k[1::7][: len(k[::7])]
    else:
        s = rng.choice(rng.integers(0, 2_000_000_000, 40), n).astype(np.int64)   # 40 distinct starts: runs of thousands
    e = s + rng.integers(1, 5000, n)
    e[::11] = e[1::11][: len(e[::11])]                                  # equal (start,end) too: the row index decides
    if dups == "long_runs_in_start_order":                              # (key, start) order, ends unordered: no radix pass, the repair
        o = np.lexsort((s, k)); k, s, e = k[o].copy(), s[o].copy(), e[o].copy()   # of the runs gives up on their length, then the full sort
    rk, rs, re = synth(30_000, 413, nkeys=6, mean_len=150, span=3_000_000, dtype=np.int64)
    re += 1
    for strict in (False, True):
        for g, w in zip(ctx.merge(k, s, e, n_keys=6, min_dist=2, strict=strict), orc.merge(k, s, e, min_dist=2, strict=strict)):
            assert len(g) == len(w) and (g == w).all()
        _same_cluster(ctx.cluster(k, s, e, n_keys=6, min_dist=0, strict=strict), orc.cluster(k, s, e, min_dist=0, strict=strict, n_keys=6))
        for g, w in zip(ctx.subtract(k, s, e, rk, rs, re, n_keys=6, strict=strict), orc.subtract(k, s, e, rk, rs, re, strict=strict)):
            assert len(g) == len(w) and (g == w).all()


@pytest.mark.parametrize("shape", ["sparse", "dense", "one_key", "many_keys", "empties", "scaffolds"])
def test_merge_one_pass_sweep(ctx, shape):
    """Well-formed rows go through the one-pass sweep over the sort's packed words (ivx_runs.hip k_merge_fused: several
    hundred tiles, look-back windows beyond 64 tiles, runs and keys that straddle tiles); IVX_NO_FUSED_SWEEP=1 is the
    three-pass sweep over unpacked rows.  Both must give the oracle's runs; strict + min_dist 0 with empty rows (whose
    order inside a start matters) must fall back by itself."""
    rng = np.random.default_rng(4242)
    n = 1_500_000
    # (scaffolds: more keys than the linearised sort word's table takes -- the key keeps its own bits in the word)
    nk = {"sparse": 24, "dense": 24, "one_key": 1, "many_keys": 700, "empties": 5, "scaffolds": 5000}[shape]
    span = {"sparse": 200_000_000, "dense": 300_000, "one_key": 50_000_000, "many_keys": 40_000, "empties": 3_000_000, "scaffolds": 150_000}[shape]
    k = rng.integers(0, nk, n).astype(np.uint32)
    if shape == "many_keys":
        k[k % 7 == 3] = 5                                              # keys without rows, one heavy key
    s = rng.integers(0, span, n).astype(np.int64) + rng.integers(-10**6, 10**6, nk)[k]
    e = s + rng.integers(1, 60, n)
    if shape == "empties":
        e[::3] = s[::3]
    s[::17] = s[1::17][: len(s[::17])]; k[::17] = k[1::17][: len(k[::17])]; e[::17] = np.maximum(e[::17], s[::17] + (shape != "empties"))
    for md, strict in ((0, False), (0, True), (9, True), (25, False)):
        want = orc.merge(k, s, e, min_dist=md, strict=strict)
        for env in ({}, {"IVX_NO_FUSED_SWEEP": "1"}, {"IVX_NO_LIN": "1"}, {"IVX_NO_NARROW_RUNS": "1"}):
            os.environ.update(env)
            try:
                got = ctx.merge(k, s, e, n_keys=nk, min_dist=md, strict=strict)
            finally:
                for v in env:
                    os.environ.pop(v, None)
            for g, w in zip(got, want):
                assert len(g) == len(w) and (g == w).all(), (shape, md, strict, env)
    vk, vs, ve = k[:3000].copy(), s[:3000] - 50, e[:3000] + 500
    for g, w in zip(ctx.complement(k, s, e, vk, vs, ve, n_keys=nk), orc.complement(k, s, e, vk, vs, ve)):
        assert len(g) == len(w) and (np.asarray(g).astype(np.int64) == w.astype(np.int64)).all(), shape


def test_merge_i64_extremes(ctx):
    big = np.iinfo(np.int64).max
    k = np.zeros(6, np.uint32)
    s = np.array([0, 100, big - 10, -big, 5, big], np.int64)
    e = np.array([big - 1, 200, big, -big + 3, 5, big], np.int64)
    for md in (0, 7, big):
        for strict in (False, True):
            got = ctx.merge(k, s, e, n_keys=1, min_dist=md, strict=strict)
            want = orc.merge(k, s, e, min_dist=md, strict=strict)
            for g, w in zip(got, want):
                assert len(g) == len(w) and (g == w).all(), (md, strict)
    with pytest.raises(pyivx.IvxError):
        ctx.merge(k, s, e, n_keys=1, min_dist=-1)


@pytest.mark.parametrize("seed", range(6))
def test_subtract_random(ctx, seed):
    nk = [1, 6, 30][seed % 3]
    lk, ls, le = synth(60_000, 80 + seed, nkeys=nk, mean_len=[2000, 200][seed % 2], span=2_000_000, dtype=np.int64)
    rk, rs, re = synth(90_000, 90 + seed, nkeys=nk + 1, mean_len=[100, 1500][seed % 2], span=2_000_000, dtype=np.int64)
    le += 1; re += 1                                       # half-open style rows as in the reference's tests
    if seed >= 3:
        re[::17] = rs[::17] - 3                            # inverted rights
        le[::41] = ls[::41]                                # empty lefts
        ls[::13] = ls[1::13][: len(ls[::13])]; le[::13] = np.maximum(le[::13], ls[::13])
    for strict in (False, True):
        got = ctx.subtract(lk, ls, le, rk, rs, re, n_keys=nk + 1, strict=strict)
        want = orc.subtract(lk, ls, le, rk, rs, re, strict=strict)
        for g, w in zip(got, want):
            assert len(g) == len(w) and (g == w).all()


@pytest.mark.parametrize("shape", ["sparse", "dense", "touching", "many_keys"])
def test_subtract_well_formed_rights_two_searches(ctx, shape):
    """Rights with start <= end take the two-search plan (plan_row<WF>: the gap heads are the merged runs); IVX_SUB_GENERAL=1
    is the five-point plan.  Both against the oracle: empty and duplicate lefts, rights that touch / nest / repeat, lefts
    that reach across many runs, keys without rights."""
    rng = np.random.default_rng(9090)
    nl, nr = 400_000, 150_000
    nk = {"sparse": 8, "dense": 8, "touching": 3, "many_keys": 300}[shape]
    span = {"sparse": 40_000_000, "dense": 300_000, "touching": 200_000, "many_keys": 20_000}[shape]
    lk = rng.integers(0, nk, nl).astype(np.uint32); rk = rng.integers(0, max(nk - 1, 1), nr).astype(np.uint32)
    ls = rng.integers(0, span, nl).astype(np.int64); le = ls + rng.integers(0, 3000, nl)
    rs = rng.integers(0, span, nr).astype(np.int64); re = rs + rng.integers(0, 400, nr)
    if shape == "touching":
        rs = (rs // 50) * 50; re = rs + 50 * rng.integers(0, 4, nr)      # rights on a lattice: touching, nested, equal, empty
        ls = (ls // 25) * 25; le = ls + 25 * rng.integers(0, 9, nl)
    le[::37] = ls[::37]; ls[::11] = ls[1::11][: len(ls[::11])]; le[::11] = np.maximum(le[::11], ls[::11])
    le[::501] = le[::501] + 10 * span                                     # a few lefts across everything
    for strict in (False, True):
        want = orc.subtract(lk, ls, le, rk, rs, re, strict=strict)
        for env in ({}, {"IVX_SUB_GENERAL": "1"}):
            os.environ.update(env)
            try:
                got = ctx.subtract(lk, ls, le, rk, rs, re, n_keys=nk, strict=strict)
            finally:
                for v in env:
                    os.environ.pop(v, None)
            for g, w in zip(got, want):
                assert len(g) == len(w) and (g == w).all(), (shape, strict, env)


@pytest.mark.parametrize("device", [False, True])
def test_subtract_sizing_then_fill_plan(ctx, device):
    """The fill call that follows a sizing call reuses the sizing call's sorted sides; a one-call fill and a
    fill after some other operation touched the context must give the same rows."""
    lk, ls, le = synth(50_000, 181, nkeys=5, mean_len=1500, span=1_000_000, dtype=np.int64)
    rk, rs, re = synth(70_000, 191, nkeys=6, mean_len=300, span=1_000_000, dtype=np.int64)
    le += 1; re += 1
    want = orc.subtract(lk, ls, le, rk, rs, re, strict=False)
    if device:
        import torch
        cols = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda() for a in (lk, ls, le, rk, rs, re)]
    else:
        cols = [lk, ls, le, rk, rs, re]
    def host(a):
        if not device:
            return a
        ctx.synchronize()                                  # device-memory calls are asynchronous on the context's stream
        return a.cpu().numpy()

    def other_op():                                        # reuses the sort / temp scratch of the plan
        ctx.merge(rk, rs, re, n_keys=6)
        ctx.subtract(rk, rs, re, lk, ls, le, n_keys=6)     # another subtract, sizing + fill

    for kw in ({}, {"cap": len(want[0]) + 7}, {"between": other_op}):
        got = ctx.subtract(*cols, n_keys=6, **kw)
        for g, w in zip(got, want):
            g = host(g)
            assert len(g) == len(w) and (g.view(w.dtype) == w).all(), kw.keys()
    # the plan of a sizing call must not serve a fill call with other arguments (strict differs)
    ctx.subtract(*cols, n_keys=6)
    got = ctx.subtract(*cols, n_keys=6, strict=True, cap=2 * len(want[0]) + len(lk) + 7)
    for g, w in zip(got, orc.subtract(lk, ls, le, rk, rs, re, strict=True)):
        g = host(g)
        assert len(g) == len(w) and (g.view(w.dtype) == w).all()


@pytest.mark.parametrize("device", [False, True])
def test_subtract_plan_is_consumed_by_its_fill(ctx, device):
    """sizing(A), fill(A), the caller refills the same buffers with other rows, fill again with a blanket capacity:
    the second fill must subtract the NEW rows (the plan serves one successful fill)."""
    rk, rs, re = synth(70_000, 192, nkeys=6, mean_len=300, span=1_000_000, dtype=np.int64)
    re += 1
    A = list(synth(50_000, 182, nkeys=5, mean_len=1500, span=1_000_000, dtype=np.int64)); A[2] = A[2] + 1
    B = list(synth(50_000, 183, nkeys=5, mean_len=900, span=1_000_000, dtype=np.int64)); B[2] = B[2] + 1
    want_a = orc.subtract(*A, rk, rs, re); want_b = orc.subtract(*B, rk, rs, re)
    assert len(want_a[0]) != len(want_b[0])
    if device:
        import torch
        t = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()
        buf = [t(a) for a in A]; right = [t(a) for a in (rk, rs, re)]
        def refill(cols):
            for d, x in zip(buf, cols):
                d.copy_(t(x))
            torch.cuda.synchronize()
    else:
        buf = [a.copy() for a in A]; right = [rk, rs, re]
        def refill(cols):
            for d, x in zip(buf, cols):
                d[:] = x
    def check(got, want):
        for g, w in zip(got, want):
            if device:
                ctx.synchronize(); g = g.cpu().numpy()
            assert len(g) == len(w) and (g.view(w.dtype) == w).all()
    check(ctx.subtract(*buf, *right, n_keys=6), want_a)        # sizing + planned fill
    refill(B)
    check(ctx.subtract(*buf, *right, n_keys=6, cap=len(want_a[0]) + len(want_b[0])), want_b)


def test_subtract_empty_sides(ctx):
    e64 = np.empty(0, np.int64); ek = np.empty(0, np.uint32)
    lk = np.zeros(3, np.uint32); ls = np.array([1, 5, 9], np.int64); le = np.array([3, 8, 20], np.int64)
    got = ctx.subtract(lk, ls, le, ek, e64, e64, n_keys=1)
    assert got[1].tolist() == [1, 5, 9] and got[2].tolist() == [3, 8, 20] and got[3].tolist() == [0, 1, 2]
    got = ctx.subtract(ek, e64, e64, lk, ls, le, n_keys=1)
    assert len(got[0]) == 0


CLUSTER_COLS = ("key", "start", "end", "row", "cluster", "cluster_start", "cluster_end")


def _same_cluster(got, want):
    for c in CLUSTER_COLS:
        assert len(got[c]) == len(want[c]) and (np.asarray(got[c]).astype(np.int64) == want[c].astype(np.int64)).all(), c
    assert got["n_clusters"] == want["n_clusters"]
    assert (np.asarray(got["key_clusters"]).astype(np.uint64) == want["key_clusters"]).all()


def test_cluster_golden(ctx, golden):
    for case in golden.cases("cluster"):
        rows = golden.rows(case["input"])
        names, ((k, s, e),) = encode_keys(rows)
        c = ctx.cluster(k, s, e, n_keys=max(len(names), 1), min_dist=case["min_dist"], strict=case["strict"])
        got = [[names[a], int(b), int(cc), int(d), int(f), int(g)] for a, b, cc, d, f, g in
               zip(c["key"], c["start"], c["end"], c["cluster"], c["cluster_start"], c["cluster_end"])]
        assert got == case["expect"], case["name"]


@pytest.mark.parametrize("seed", range(6))
def test_cluster_random(ctx, seed):
    nk = [1, 7, 40][seed % 3]
    k, s, e = synth(200_000, 170 + seed, nkeys=nk, mean_len=[20, 300][seed % 2], span=3_000_000, dtype=np.int64)
    if seed >= 3:
        e[::29] = s[::29] - 4                              # inverted rows
        s[::31] = s[1::31][: len(s[::31])]                 # duplicate starts: the row index breaks ties
    md = [0, 0, 5, 1000, 0, 37][seed]
    for strict in (False, True):
        _same_cluster(ctx.cluster(k, s, e, n_keys=nk + 2, min_dist=md, strict=strict),
                      orc.cluster(k, s, e, min_dist=md, strict=strict, n_keys=nk + 2))


@pytest.mark.parametrize("shape", ["sparse", "dense", "many_keys", "sorted", "scaffolds"])
def test_cluster_over_packed_words(ctx, shape):
    """Well-formed rows: cluster() sweeps over the sort's packed words (k_pk_runs<2> + k_pk_cluster_fin); IVX_NO_FUSED_SWEEP=1
    is the scan over unpacked rows.  Several hundred tiles, keys that open inside tiles and wavefronts, equal starts, ids with
    and without a per-key base, the per-key counts alone."""
    rng = np.random.default_rng(777)
    n = 1_200_000
    nk = {"sparse": 24, "dense": 24, "many_keys": 500, "sorted": 9, "scaffolds": 4000}[shape]
    span = {"sparse": 150_000_000, "dense": 200_000, "many_keys": 30_000, "sorted": 5_000_000, "scaffolds": 100_000}[shape]
    k = rng.integers(0, nk, n).astype(np.uint32)
    if shape == "many_keys":
        k[k % 5 == 2] = 7
    s = rng.integers(0, span, n).astype(np.int64) + rng.integers(-10**6, 10**6, nk)[k]
    e = s + rng.integers(1, 80, n)
    s[::19] = s[1::19][: len(s[::19])]; k[::19] = k[1::19][: len(k[::19])]; e[::19] = np.maximum(e[::19], s[::19] + 1)
    if shape == "sorted":
        o = np.lexsort((e, s, k)); k, s, e = k[o].copy(), s[o].copy(), e[o].copy()
    base = (np.arange(nk + 1, dtype=np.int64) * 1_000_003)[: nk + 1]
    for md, strict in ((0, False), (6, True), (40, False)):
        want = orc.cluster(k, s, e, min_dist=md, strict=strict, n_keys=nk + 1)
        for env in ({}, {"IVX_NO_FUSED_SWEEP": "1"}, {"IVX_NO_NARROW_RUNS": "1"}):
            os.environ.update(env)
            try:
                got = ctx.cluster(k, s, e, n_keys=nk + 1, min_dist=md, strict=strict)
                cnt = ctx.cluster(k, s, e, n_keys=nk + 1, min_dist=md, strict=strict, rows=False)
                based = ctx.cluster(k, s, e, n_keys=nk + 1, min_dist=md, strict=strict, key_base=base)
            finally:
                for v in env:
                    os.environ.pop(v, None)
            _same_cluster(got, want)
            assert (np.asarray(cnt["key_clusters"]).astype(np.int64) == np.asarray(got["key_clusters"]).astype(np.int64)).all()
            assert cnt["n_clusters"] == got["n_clusters"]
            kc = np.asarray(got["key_clusters"]).astype(np.int64)
            first = np.concatenate([[0], np.cumsum(kc)[:-1]])
            kk = np.asarray(got["key"]).astype(np.int64)
            assert (np.asarray(based["cluster"]).astype(np.int64) == np.asarray(got["cluster"]).astype(np.int64) - first[kk] + base[kk]).all(), (shape, md, strict, env)


def test_cluster_sharded_ids_match_single_run(ctx):
    # two "partitions" holding disjoint contigs: counts first, exclusive scan over the keys in
    # order (what ClusterIdCoordinator does, cluster.rs:396-417), then ids with key_base
    nk = 9
    k, s, e = synth(50_000, 5, nkeys=nk, mean_len=200, span=2_000_000, dtype=np.int64)
    full = ctx.cluster(k, s, e, n_keys=nk)
    parts = [np.isin(k, [0, 3, 4, 8]), np.isin(k, [1, 2, 5, 6, 7])]
    counts = np.zeros(nk, np.int64)
    for sel in parts:
        counts += np.asarray(ctx.cluster(k[sel], s[sel], e[sel], n_keys=nk, rows=False)["key_clusters"]).astype(np.int64)
    base = np.concatenate([[0], np.cumsum(counts)[:-1]])
    for sel in parts:
        part = ctx.cluster(k[sel], s[sel], e[sel], n_keys=nk, key_base=base)
        want = np.isin(full["key"], np.unique(k[sel]))
        for c in ("key", "start", "end", "cluster", "cluster_start", "cluster_end"):
            assert (np.asarray(part[c]) == np.asarray(full[c])[want]).all(), c
        assert (np.flatnonzero(sel)[part["row"]] == full["row"][want]).all()


def test_cluster_i64_extremes_and_empty(ctx):
    big = np.iinfo(np.int64).max
    k = np.zeros(6, np.uint32)
    s = np.array([0, 100, big - 10, -big, 5, big], np.int64)
    e = np.array([big - 1, 200, big, -big + 3, 5, big], np.int64)
    for md in (0, 7, big):
        for strict in (False, True):
            _same_cluster(ctx.cluster(k, s, e, n_keys=1, min_dist=md, strict=strict), orc.cluster(k, s, e, min_dist=md, strict=strict, n_keys=1))
    z = np.zeros(0, np.int64)
    c = ctx.cluster(np.zeros(0, np.uint32), z, z, n_keys=3)
    assert c["n_clusters"] == 0 and len(c["cluster"]) == 0 and np.asarray(c["key_clusters"]).tolist() == [0, 0, 0]
    with pytest.raises(pyivx.IvxError):
        ctx.cluster(k, s, e, n_keys=1, min_dist=-1)


def test_complement_golden(ctx, golden):
    for case in golden.cases("complement"):
        rows, view = golden.rows(case["input"]), case["view"] or []
        names, ((k, s, e), (vk, vs, ve)) = encode_keys(rows, view)
        ok, os_, oe = ctx.complement(k, s, e, vk, vs, ve, n_keys=max(len(names), 1), strict=case["strict"])
        got = [[names[a], int(b), int(c)] for a, b, c in zip(ok, os_, oe)]
        assert got == case["expect"], case["name"]


@pytest.mark.parametrize("seed", range(8))
def test_complement_random(ctx, seed):
    nk = [1, 6, 30, 12][seed % 4]
    k, s, e = synth(120_000, 270 + seed, nkeys=nk, mean_len=[30, 400][seed % 2], span=4_000_000, dtype=np.int64)
    rng = np.random.default_rng(seed)
    mode = seed % 4
    if mode == 0:                                          # no view at all: implicit [0, i64::MAX) per key
        vk = vs = ve = None
    else:                                                  # overlapping, unmerged views; some keys view-only, some input-only
        nvw = [0, 40, 3000, 50_000][mode]
        vk = rng.integers(0, nk + 3, nvw).astype(np.uint32)
        vs = rng.integers(-1000, 4_000_000, nvw).astype(np.int64)
        ve = vs + rng.integers(-50, [0, 3_000_000, 200_000, 5_000][mode], nvw)
        keep = ~np.isin(k, [1, 4])                         # keys 1 and 4: view rows only
        k, s, e = k[keep], s[keep], e[keep]
    if seed >= 4:
        e[::23] = s[::23] - 7                              # end < start rows: merged ends no longer ascend (serial walk)
    for strict in (False, True):
        got = ctx.complement(k, s, e, vk, vs, ve, n_keys=nk + 3, strict=strict)
        want = orc.complement(k, s, e, vk, vs, ve, strict=strict)
        for g, w in zip(got, want):
            assert len(g) == len(w) and (np.asarray(g).astype(np.int64) == w.astype(np.int64)).all(), (seed, strict)


def test_complement_empty_sides_and_count_only(ctx):
    z64, zk = np.zeros(0, np.int64), np.zeros(0, np.uint32)
    assert len(ctx.complement(zk, z64, z64, n_keys=2)[0]) == 0                       # nothing in, nothing out
    ok, os_, oe = ctx.complement(zk, z64, z64, np.array([1, 0], np.uint32), np.array([5, 0], np.int64), np.array([9, 3], np.int64), n_keys=2)
    assert ok.tolist() == [0, 1] and os_.tolist() == [0, 5] and oe.tolist() == [3, 9]   # views come back whole, key order
    big = np.iinfo(np.int64).max
    ok, os_, oe = ctx.complement(np.array([0], np.uint32), np.array([0], np.int64), np.array([big], np.int64), n_keys=1)
    assert len(ok) == 0                                                                  # the input covers the implicit view


def test_wrong_index_kind(ctx):
    bk, bs, be = synth(100, 1)
    ix = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=1)
    with pytest.raises(pyivx.IvxError) as ei:
        ctx.coverage(ix, bk, bs, be)
    assert ei.value.status == pyivx.ERR_UNSUPPORTED


# ---- f3: compute::take on the device, against pyarrow's take (the same Arrow kernel the reference calls)

@pytest.mark.parametrize("dtype", ["int8", "uint16", "int32", "float32", "int64", "float64"])
def test_take_fixed_matches_arrow(ctx, dtype):
    import pyarrow as pa
    import pyarrow.compute as pc
    rng = np.random.default_rng(3)
    n_src, n = 5000, 40_000
    vals = rng.integers(-100, 100, n_src).astype(dtype)
    mask = rng.random(n_src) < 0.1                                   # source nulls
    col = pa.array(vals, mask=mask)
    idx = rng.integers(0, n_src, n).astype(np.uint32)
    idx[::17] = pyivx.NULL_IDX                                       # null indices (nearest's missing left rows)
    want = pc.take(col, pa.array(idx, mask=idx == pyivx.NULL_IDX))
    bits = np.frombuffer(col.buffers()[0], np.uint8)
    out, valid = ctx.take_fixed(vals, idx, src_valid_bits=bits)
    got = pa.array(out, mask=valid == 0)
    assert got.equals(want)
    out2, _ = ctx.take_fixed(vals, idx[idx != pyivx.NULL_IDX], want_valid=False)
    assert (out2 == vals[idx[idx != pyivx.NULL_IDX]]).all()


def test_take_fixed_wide_records_and_errors(ctx):
    rng = np.random.default_rng(4)
    src = rng.integers(0, 255, (1000, 16)).astype(np.uint8)          # 16-byte records (decimal128 / uuid)
    idx = rng.integers(0, 1000, 7777).astype(np.uint32)
    out, valid = ctx.take_fixed(src, idx)
    assert (out == src[idx]).all() and valid.all()
    src32 = rng.integers(0, 255, (100, 32)).astype(np.uint8)
    out, _ = ctx.take_fixed(src32, idx % 100)
    assert (out == src32[idx % 100]).all()
    with pytest.raises(pyivx.IvxError):                              # out of bounds, as arrow's take
        ctx.take_fixed(src, np.array([5, 1000], np.uint32))
    with pytest.raises(pyivx.IvxError):                              # 3-byte elements are not a supported width
        ctx.take_fixed(np.zeros((4, 3), np.uint8), np.array([0], np.uint32))
    out, valid = ctx.take_fixed(src, np.zeros(0, np.uint32))
    assert len(out) == 0


@pytest.mark.parametrize("large", [False, True])
def test_take_utf8_matches_arrow(ctx, large):
    import pyarrow as pa
    import pyarrow.compute as pc
    rng = np.random.default_rng(5)
    n_src, n = 3000, 50_000
    words = ["", "chr1", "chrUn_KI270742v1", "BRCA1", "ENSG00000139618.19", "é—ü", "x" * 300, "y" * 5000]
    py = [words[i] for i in rng.integers(0, len(words), n_src)]
    for i in range(0, n_src, 13):
        py[i] = None
    col = pa.array(py, pa.large_string() if large else pa.string())
    idx = rng.integers(0, n_src, n).astype(np.uint32)
    idx[::29] = pyivx.NULL_IDX
    want = pc.take(col, pa.array(idx, mask=idx == pyivx.NULL_IDX))
    vb, ob, db = col.buffers()
    off = np.frombuffer(ob, np.int64 if large else np.int32)[: n_src + 1]
    data = np.frombuffer(db, np.uint8)
    out_off, out_data, valid = ctx.take_utf8(off, data, idx, src_valid_bits=np.frombuffer(vb, np.uint8))
    got = pa.Array.from_buffers(col.type, n, [pa.py_buffer(np.packbits(valid, bitorder="little")), pa.py_buffer(out_off), pa.py_buffer(out_data)])
    # arrow keeps the bytes of a null source slot out of the result; so do we only for null INDICES -- compare as values
    assert got.to_pylist() == want.to_pylist()
    e_off, e_data, e_valid = ctx.take_utf8(off, data, np.zeros(0, np.uint32))
    assert e_off.tolist() == [0] and len(e_data) == 0


def test_take_device_buffers(ctx):
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    src = torch.randint(-1000, 1000, (100_000,), generator=g, dtype=torch.int64)
    idx = torch.randint(0, 100_000, (1_000_000,), generator=g, dtype=torch.int32)
    out, valid = ctx.take_fixed(src.to(dev), idx.to(dev))
    assert torch.equal(out.cpu(), src[idx.long()]) and bool(valid.all())


def test_cluster_exons_issue_373(ctx, golden):
    # R/tests/integration_test.rs:4026-4084: 438 694 exons, the pinned cluster extents, with 1 and with 4
    # "target partitions" (contigs dealt round-robin, ids from the coordinator's exclusive scan)
    from test_oracle_golden import _exons_cluster_case, _selected_cluster_rows
    case, names, k, s, e = _exons_cluster_case(golden)
    nk = len(names)
    full = ctx.cluster(k, s, e, n_keys=nk)
    assert _selected_cluster_rows(case, names, full) == sorted(case["expect"])
    want = orc.cluster(k, s, e, n_keys=nk)
    for c in CLUSTER_COLS:
        assert (np.asarray(full[c]).astype(np.int64) == want[c].astype(np.int64)).all(), c
    parts = [np.isin(k, np.arange(p, nk, 4)) for p in range(4)]
    counts = sum(np.asarray(ctx.cluster(k[m], s[m], e[m], n_keys=nk, rows=False)["key_clusters"]).astype(np.int64) for m in parts)
    base = np.concatenate([[0], np.cumsum(counts)[:-1]])
    ids = {}
    for m in parts:
        part = ctx.cluster(k[m], s[m], e[m], n_keys=nk, key_base=base)
        assert _selected_cluster_rows(case, names, part) == sorted(r for r in case["expect"] if names.index(r[0]) in set(np.unique(k[m]).tolist()))
        for cid, cs, ce in zip(part["cluster"], part["cluster_start"], part["cluster_end"]):
            assert ids.setdefault(int(cid), (int(cs), int(ce))) == (int(cs), int(ce))   # :2761-2818: an id never names two extents
    assert len(ids) == full["n_clusters"]


def test_count_coverage_sorted_probe_input(ctx):
    # sorted probe rows take the in-place path of the region pipeline (no scatter, no un-permute); strict mode
    # shrinks the query before routing, so the order check runs on start+1
    bk, bs, be = synth(150_000, 81, nkeys=6, mean_len=500, span=20_000_000)
    pk, ps, pe = synth(900_001, 82, nkeys=6, mean_len=150, span=20_000_000)
    o = np.lexsort((ps, pk)); pk, ps, pe = pk[o].copy(), ps[o].copy(), pe[o].copy()
    for kind, fn, ofn in ((pyivx.KIND_COUNT, ctx.count_overlaps, orc.count_overlaps), (pyivx.KIND_COVERAGE, ctx.coverage, orc.coverage)):
        ix = ctx.build(kind, bk, bs, be, n_keys=6)
        for strict in (False, True):
            want = ofn(bk, bs, be, pk, ps, pe, strict=strict)
            for path in _paths():
                assert (fn(ix, pk, ps, pe, strict=strict) == want).all(), (kind, strict, path)
        ix.free()


def test_count_coverage_and_per_row_with_several_hundred_regions(ctx):
    # a build side that needs more than 255 LDS-sized regions: count_overlaps / coverage / the join's per-row counts and
    # exists go through the 1024-digit partition and the un-permute; unsorted and sorted probe rows, strict and weak
    bk, bs, be = synth(2_600_000, 91, nkeys=3, mean_len=300, span=240_000_000)
    pk, ps, pe = synth(500_000, 92, nkeys=4, mean_len=150, span=240_000_000)
    pe[::83] = ps[::83] + 40_000
    for srt in (False, True):
        if srt:
            o = np.lexsort((ps, pk)); pk, ps, pe = pk[o].copy(), ps[o].copy(), pe[o].copy()
        for kind, fn, ofn in ((pyivx.KIND_COUNT, ctx.count_overlaps, orc.count_overlaps), (pyivx.KIND_COVERAGE, ctx.coverage, orc.coverage)):
            ix = ctx.build(kind, bk, bs, be, n_keys=4)
            for strict in (False, True):
                want = ofn(bk, bs, be, pk, ps, pe, strict=strict)
                for path in _paths():
                    assert (fn(ix, pk, ps, pe, strict=strict) == want).all(), (kind, strict, path, srt)
            ix.free()
        ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=4)
        _, _, want_cnt = orc.join(bk, bs, be, pk, ps, pe, per_row=True, threads=4)
        os.environ["IVX_JOIN_PATH"] = "regions"
        try:
            tot, pr = ctx.overlap_count(ix, pk, ps, pe, per_row=True)
            ex = ctx.exists(ix, pk, ps, pe)
        finally:
            del os.environ["IVX_JOIN_PATH"]
        assert tot == int(want_cnt.sum()) and (pr.astype(np.uint64) == want_cnt).all() and (ex == (want_cnt > 0)).all(), srt
        ix.free()
