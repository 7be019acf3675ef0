"""-m gpu: BASELINE.json configurations at FULL size, bit-exact against the CPU oracle where the oracle
finishes in seconds with 16 host threads (C2, C3: join pair multiset, count_overlaps and coverage columns; C4: nearest), and
through size-independent properties where it does not (C5: merge and subtract on 10^9 rows).

The reference's own scale harness compares row-multiset checksums (R/tests/integration_test.rs:4289-4349) and
its scale fixture compares whole columns (:726-817); here the sorted pair keys / whole columns are compared."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx  # noqa: E402
import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (the checker)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
THREADS = min(16, len(os.sched_getaffinity(0)))


@pytest.fixture(scope="module")
def ctx():
    c = pyivx.Ctx(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def _host(k, s, e):
    return k.cpu().numpy().view(np.uint32), s.cpu().numpy(), e.cpu().numpy()


def _gpu_pair_keys(b, p):
    """sorted build<<32|probe keys of the device pair list (same canonical form as orc.pair_keys)"""
    k = (b.long() << 32) | p.long()                              # both < 2^31 here: no sign trouble
    return torch.sort(k).values.cpu().numpy().view(np.uint64)


def _join_vs_oracle(ctx, bk, bs, be, pk, ps, pe, nk):
    ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=nk)
    total = ctx.overlap_count(ix, pk, ps, pe)
    b, p = ctx.overlap_fill(ix, pk, ps, pe, cap=total)
    torch.cuda.synchronize()
    assert b.numel() == total
    got = _gpu_pair_keys(b, p)
    del b, p
    wb, wp = orc.join_single(*_host(bk, bs, be), *_host(pk, ps, pe), threads=THREADS)
    want = orc.pair_keys(wb, wp)
    assert len(want) == total
    assert np.array_equal(got, want)                              # the pair MULTISET, bit-exact
    ix.free()
    return total


def test_c3_join_100Mx1M_pairs_bit_exact(ctx):
    """BASELINE configs[2] shape, the bench.py headline: every one of the ~37 M pairs equals the oracle's."""
    nb, npr, nk = 1_000_000, 100_000_000, 24
    bk, bs, be = synth.gen_torch(nb, 1000, nk, 0x5EED0004, DEV)
    pk, ps, pe = synth.gen_torch(npr, 150, nk, 0x5EED0005, DEV)
    total = _join_vs_oracle(ctx, bk, bs, be, pk, ps, pe, nk)
    expect = npr * nb * 1149.0 / sum(synth.HG38)
    assert abs(total - expect) < 0.01 * expect


def test_c3_count_coverage_100Mx1M_bit_exact(ctx):
    """BASELINE configs[2]: count_overlaps + coverage columns, 100 M rows each, weak and strict, vs the oracle."""
    nb, npr, nk = 1_000_000, 100_000_000, 24
    bk, bs, be = synth.gen_torch(nb, 1000, nk, 0x5EED0004, DEV)
    pk, ps, pe = synth.gen_torch(npr, 150, nk, 0x5EED0005, DEV)
    hb, hp = _host(bk, bs, be), _host(pk, ps, pe)
    ixc = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=nk)
    ixv = ctx.build(pyivx.KIND_COVERAGE, bk, bs, be, n_keys=nk)
    for strict in (False, True):
        cnt = ctx.count_overlaps(ixc, pk, ps, pe, strict=strict).cpu().numpy()
        assert np.array_equal(cnt, orc.count_overlaps(*hb, *hp, strict=strict, threads=THREADS))
        del cnt
        cov = ctx.coverage(ixv, pk, ps, pe, strict=strict).cpu().numpy()
        assert np.array_equal(cov, orc.coverage(*hb, *hp, strict=strict, threads=THREADS))
        del cov
    ixc.free(); ixv.free()


def test_c4_nearest_50Mx50M_bit_exact(ctx):
    """BASELINE.json configs[3] on one GPU: nearest(), k = 1, 50M x 50M rows over 24 contigs -- the build-row index
    and the distance of EVERY probe row against the oracle (16 host threads), with and without overlapping rows."""
    n = 50_000_000
    bk, bs, be = synth.gen_torch(n, 1000, 24, 0x5EED0006, DEV)
    pk, ps, pe = synth.gen_torch(n, 150, 24, 0x5EED0007, DEV)
    torch.cuda.synchronize()
    ix = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=24)
    hb, hp = _host(bk, bs, be), _host(pk, ps, pe)
    for ovl in (True, False):
        b, p, d = ctx.nearest(ix, pk, ps, pe, k=1, overlap=ovl)
        torch.cuda.synchronize()
        assert b.numel() == n
        gb, gd = b.cpu().numpy().view(np.uint32), d.cpu().numpy()
        assert bool((p.long() == torch.arange(n, device=DEV)).all())
        del b, p, d
        wb, wd = orc.nearest1(*hb, *hp, overlap=ovl, threads=THREADS)
        assert np.array_equal(gb, wb) and np.array_equal(gd, wd), ovl
        del gb, gd, wb, wd
    ix.free()


def test_c2_10Mx100k_single_contig_bit_exact(ctx):
    """BASELINE configs[1]: 10 M probe x 100 k build rows on ONE contig (one key, a few dozen index regions):
    join pairs, rle_right, semi-join flags, count_overlaps and coverage vs the oracle."""
    nb, npr = 100_000, 10_000_000
    bk, bs, be = synth.gen_torch(nb, 1000, 1, 0x5EED0002, DEV)
    pk, ps, pe = synth.gen_torch(npr, 150, 1, 0x5EED0003, DEV)
    hb, hp = _host(bk, bs, be), _host(pk, ps, pe)
    total = _join_vs_oracle(ctx, bk, bs, be, pk, ps, pe, 1)
    expect = npr * nb * 1149.0 / synth.HG38[0]
    assert abs(total - expect) < 0.01 * expect
    # the range-only form of the same join (key == NULL: one key)
    ix = ctx.build(pyivx.KIND_OVERLAP, None, bs, be)
    assert ctx.overlap_count(ix, None, ps, pe) == total
    tot2, per_row = ctx.overlap_count(ix, None, ps, pe, per_row=True)
    wtot, wcnt = orc.join_count(*hb, *hp, threads=THREADS)
    assert tot2 == wtot == total and np.array_equal(per_row.cpu().numpy().view(np.uint32).astype(np.uint64), wcnt)
    assert np.array_equal(ctx.exists(ix, None, ps, pe).cpu().numpy(), (wcnt != 0).astype(np.uint8))
    ix.free()
    ixc = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=1)
    ixv = ctx.build(pyivx.KIND_COVERAGE, bk, bs, be, n_keys=1)
    for strict in (False, True):
        assert np.array_equal(ctx.count_overlaps(ixc, pk, ps, pe, strict=strict).cpu().numpy(),
                              orc.count_overlaps(*hb, *hp, strict=strict, threads=THREADS))
        assert np.array_equal(ctx.coverage(ixv, pk, ps, pe, strict=strict).cpu().numpy(),
                              orc.coverage(*hb, *hp, strict=strict, threads=THREADS))
    ixc.free(); ixv.free()


def _gen64(n, mean, seed):
    k, s, e = synth.gen_torch(n, mean, 24, seed, DEV)
    s64, e64 = s.long(), e.long() + 1                            # half-open i64, as the sweep operators take them
    del s, e
    return k, s64, e64


@pytest.fixture()
def big_ctx():
    """A context of its own for a 10^9-row call: its grow-only scratch (~100 GB after such a call) goes back to the
    device when the test ends instead of staying with the module's context (device memory is oversubscribed into
    host memory on this platform when HBM runs out, and the box caps host memory)."""
    torch.cuda.empty_cache()
    c = pyivx.Ctx(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()
    torch.cuda.empty_cache()


def _chunks(n, step=100_000_000):
    return [(lo, min(n, lo + step)) for lo in range(0, n, step)]


def test_c5_merge_1e9_rows(big_ctx):
    """BASELINE configs[4]: merge() on 10^9 rows, unsorted and coordinate-sorted input (the oracle would need
    minutes: size-independent properties, and the two input orders must give the same runs)."""
    ctx = big_ctx
    n = 1_000_000_000
    k, s64, e64 = _gen64(n, 20, 0x5EED0008)                      # short rows: the merge leaves ~10^6 runs, not 24
    torch.cuda.synchronize()
    ok, os_, oe, on = ctx.merge(k, s64, e64, n_keys=24)
    torch.cuda.synchronize()
    m = ok.numel()
    ok, os_, oe, on = ok.clone(), os_.clone(), oe.clone(), on.clone()      # drop the n-row output buffers
    torch.cuda.empty_cache()
    assert m > 100_000 and int(on.sum()) == n                   # every input row is in exactly one run
    key = ok.long() * (1 << 40) + os_
    assert bool((key[1:] > key[:-1]).all())                      # ordered by (key, start)
    same = ok[1:] == ok[:-1]
    assert bool((os_[1:][same] > oe[:-1][same]).all())           # runs of one key neither touch nor overlap
    assert int((oe - os_).sum()) <= sum(int((e64[a:b] - s64[a:b]).sum()) for a, b in _chunks(n))   # merged length <= summed input length
    # every input row lies inside the run that a binary search over the run starts finds for it
    for a, b in _chunks(n):
        rk = k[a:b].long() * (1 << 40) + s64[a:b]
        r = torch.searchsorted(key, rk, right=True) - 1
        assert bool((r >= 0).all()) and bool((ok[r] == k[a:b]).all()) and bool((e64[a:b] <= oe[r]).all())
        del rk, r
    # idempotence
    ok2, os2, oe2, on2 = ctx.merge(ok, os_, oe, n_keys=24)
    assert ok2.numel() == m and bool((os2 == os_).all()) and bool((oe2 == oe).all()) and bool((on2 == 1).all())
    del ok2, os2, oe2, on2, key, same
    # the same rows coordinate-sorted (the in-place path that skips the radix sort) give the same runs
    o = torch.argsort(k.long() * (1 << 32) + s64)
    k, s64, e64 = k[o].contiguous(), s64[o].contiguous(), e64[o].contiguous()
    del o
    torch.cuda.empty_cache()
    ok3, os3, oe3, on3 = ctx.merge(k, s64, e64, n_keys=24)
    torch.cuda.synchronize()
    assert ok3.numel() == m
    assert bool((ok3 == ok).all()) and bool((os3 == os_).all()) and bool((oe3 == oe).all()) and bool((on3 == on).all())


def test_c5_subtract_1e9_minus_1e8(big_ctx, ctx):
    """BASELINE configs[4]: subtract() of a 10^8-row mask from 10^9 rows."""
    nl, nr = 1_000_000_000, 100_000_000
    lk, ls, le = _gen64(nl, 20, 0x5EED0008)
    rk, rs, re = _gen64(nr, 8, 0x5EED0009)
    torch.cuda.synchronize()
    fk, fs, fe, frow = big_ctx.subtract(lk, ls, le, rk, rs, re, n_keys=24)
    torch.cuda.synchronize()
    big_ctx.close()                                              # its ~100 GB of scratch back to the device now
    torch.cuda.empty_cache()
    nf = fk.numel()
    assert nf > nl // 2
    ixc = ctx.build(pyivx.KIND_COUNT, rk, rs.to(torch.int32), (re - 1).to(torch.int32), n_keys=24)
    prev_row, prev_end = None, None
    for a, b in _chunks(nf):
        k_, s_, e_, r_ = fk[a:b], fs[a:b], fe[a:b], frow[a:b].long()
        assert bool((s_ < e_).all()) and bool((k_ == lk[r_]).all())
        assert bool((s_ >= ls[r_]).all()) and bool((e_ <= le[r_]).all())         # every fragment lies inside its left row
        # the fragments of one left row are consecutive, ascending and disjoint (also across chunk borders)
        same = r_[1:] == r_[:-1]
        assert bool((s_[1:][same] >= e_[:-1][same]).all())
        if prev_row is not None and int(r_[0]) == prev_row:
            assert int(s_[0]) >= prev_end
        prev_row, prev_end = int(r_[-1]), int(e_[-1])
        # no fragment overlaps any mask row: count_overlaps of the closed view [s, e-1] against the mask
        hits = ctx.count_overlaps(ixc, k_, s_.to(torch.int32), (e_ - 1).to(torch.int32))
        assert int(hits.sum()) == 0
        del k_, s_, e_, r_, same, hits
    ixc.free()
    # idempotence on a slice of the fragments: none of them touches the mask, so subtracting it again returns the same
    # rows (as a multiset: fragments of overlapping left rows are not globally ordered by start, the second call's are)
    a, b = 0, 150_000_000
    gk, gs, ge, _ = ctx.subtract(fk[a:b].contiguous(), fs[a:b].contiguous(), fe[a:b].contiguous(), rk, rs, re, n_keys=24)
    pack = lambda k_, s_, e_: torch.sort((k_.long() << 58) | (s_ << 29) | e_).values      # coordinates < 2^29 here
    assert gk.numel() == b - a
    same_rows = bool((pack(gk, gs, ge) == pack(fk[a:b], fs[a:b], fe[a:b])).all())
    assert same_rows
