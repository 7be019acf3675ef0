"""-m gpu: BASELINE.json-sized runs checked through size-independent properties (the oracle cannot
finish these sizes in seconds): pair counts agree between the count / per-row / fill forms, every
emitted pair satisfies the literal predicate, aggregate identities between operators, sortedness
and partition-of-rows invariants of merge / subtract, nearest distances against overlap counts."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx  # noqa: E402
import synth  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ctx():
    c = pyivx.Ctx(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def test_join_100Mx1M_properties(ctx):
    nb, npr, nk = 1_000_000, 100_000_000, 24                   # BASELINE.json configs[2]
    bk, bs, be = synth.gen_torch(nb, 1000, nk, 0x5EED0004, DEV)
    pk, ps, pe = synth.gen_torch(npr, 150, nk, 0x5EED0005, DEV)
    torch.cuda.synchronize()
    ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=nk)
    total = ctx.overlap_count(ix, pk, ps, pe)
    expect = npr * nb * 1149.0 / sum(synth.HG38)
    assert abs(total - expect) < 0.01 * expect                 # uniform-data expectation
    total2, per_row = ctx.overlap_count(ix, pk, ps, pe, per_row=True)
    assert total2 == total and int(per_row.to(torch.int64).sum()) == total
    ob = torch.empty(total, dtype=torch.int32, device=DEV); op = torch.empty_like(ob)
    b, p = ctx.overlap_fill(ix, pk, ps, pe, out=(ob, op))
    torch.cuda.synchronize()
    assert b.numel() == total
    bl, pl = b.long(), p.long()
    assert bool(((bk[bl] == pk[pl]) & (bs[bl] <= pe[pl]) & (be[bl] >= ps[pl])).all())      # literal predicate on every pair
    assert bool((torch.bincount(pl, minlength=npr) == per_row.long()).all())                 # pairs per probe row = rle_right
    h = (bl * 0x9E3779B97F4A7C15 + pl * 0xC2B2AE3D27D4EB4F) & 0x7FFFFFFFFFFFFFFF            # no duplicate pair
    assert int(torch.unique(h).numel()) == total
    ex = ctx.exists(ix, pk, ps, pe)
    assert bool((ex.bool() == (per_row != 0)).all())
    # the same pairs through the gather path on a 4M-row slice (direct vs region-partitioned probes)
    os.environ["IVX_JOIN_PATH"] = "direct"
    try:
        n4 = 4_000_000
        c4 = ctx.overlap_count(ix, pk[:n4], ps[:n4], pe[:n4])
    finally:
        del os.environ["IVX_JOIN_PATH"]
    assert c4 == int(per_row[:n4].long().sum())
    # count_overlaps on the same tables: sum of counts == number of pairs (well-formed intervals)
    ixc = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=nk)
    cnt = ctx.count_overlaps(ixc, pk, ps, pe)
    assert bool((cnt == per_row.long()).all())
    # coverage is at least 1 wherever something overlaps and 0 elsewhere
    ixv = ctx.build(pyivx.KIND_COVERAGE, bk, bs, be, n_keys=nk)
    cov = ctx.coverage(ixv, pk, ps, pe)
    assert bool(((cov > 0) == (cnt > 0)).all()) and int(cov.min()) >= 0


def test_nearest_50Mx50M_properties(ctx):
    n = 50_000_000                                             # BASELINE.json configs[3]
    bk, bs, be = synth.gen_torch(n, 1000, 24, 0x5EED0006, DEV)
    pk, ps, pe = synth.gen_torch(n, 150, 24, 0x5EED0007, DEV)
    torch.cuda.synchronize()
    ix = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=24)
    b, p, d = ctx.nearest(ix, pk, ps, pe, k=1, overlap=True)
    torch.cuda.synchronize()
    assert b.numel() == n and bool((p.long() == torch.arange(n, device=DEV)).all())
    bl = b.long()
    assert bool((b != -1).all())                               # every contig has build rows: no NULLs
    gs, ge = bs[bl].long(), be[bl].long()
    dist = torch.where(pe.long() < gs, gs - pe.long(), torch.where(ge < ps.long(), ps.long() - ge, torch.zeros_like(gs)))
    assert bool((dist == d).all()) and bool((bk[bl] == pk).all())
    # distance 0 exactly where count_overlaps says something overlaps
    ixc = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=24)
    cnt = ctx.count_overlaps(ixc, pk, ps, pe)
    assert bool(((d == 0) == (cnt > 0)).all())
    # include_overlaps = false never returns an overlapping row and is never closer than the k=1 answer
    b2, _, d2 = ctx.nearest(ix, pk[:5_000_000], ps[:5_000_000], pe[:5_000_000], k=1, overlap=False)
    assert bool((d2 > 0).all()) and bool((d2 >= d[:5_000_000]).all())


def test_merge_subtract_200M_properties(ctx):
    n = 200_000_000
    k, s, e = synth.gen_torch(n, 20, 24, 0x5EED0008, DEV)      # short intervals: merge leaves many runs
    s64, e64 = s.long(), e.long() + 1
    del s, e
    torch.cuda.synchronize()
    ok, os_, oe, on = ctx.merge(k, s64, e64, n_keys=24)
    torch.cuda.synchronize()
    m = ok.numel()
    assert int(on.sum()) == n                                  # every input row is in exactly one run
    key = ok.long() * (1 << 40) + os_
    assert bool((key[1:] > key[:-1]).all())                    # ordered by (key, start), no duplicates
    same = ok[1:] == ok[:-1]
    assert bool((os_[1:][same] > oe[:-1][same]).all())         # runs of one key do not touch (weak merge)
    assert int((oe - os_).sum()) <= int((e64 - s64).sum())     # merged length never exceeds the summed input length
    # idempotence: merging the runs again changes nothing
    ok2, os2, oe2, on2 = ctx.merge(ok, os_, oe, n_keys=24)
    assert ok2.numel() == m and bool((os2 == os_).all()) and bool((oe2 == oe).all()) and bool((on2 == 1).all())
    # subtract: left = the runs, right = a 20M-row mask; left minus right minus right again is unchanged,
    # total length left = |left| - |left ∩ merged(right)|
    rk, rs, re = synth.gen_torch(20_000_000, 150, 24, 0x5EED0009, DEV)
    rs64, re64 = rs.long(), re.long() + 1
    fk, fs, fe, frow = ctx.subtract(ok, os_, oe, rk, rs64, re64, n_keys=24)
    torch.cuda.synchronize()
    assert bool((fs < fe).all()) and bool((fs >= os_[frow.long()]).all()) and bool((fe <= oe[frow.long()]).all())
    fkey = fk.long() * (1 << 40) + fs
    assert bool((fkey[1:] > fkey[:-1]).all())                  # fragments ordered like their (disjoint, sorted) left rows
    gk, gs, ge, _ = ctx.subtract(fk, fs, fe, rk, rs64, re64, n_keys=24)
    assert gk.numel() == fk.numel() and bool((gs == fs).all()) and bool((ge == fe).all())
    # no fragment overlaps any right row (count_overlaps on the half-open view: [s, e-1])
    ixc = ctx.build(pyivx.KIND_COUNT, rk, rs, re, n_keys=24)
    hits = ctx.count_overlaps(ixc, fk, fs.to(torch.int32).contiguous(), (fe - 1).to(torch.int32).contiguous())
    assert int(hits.sum()) == 0


def test_cluster_complement_100M_properties(ctx):
    n = 100_000_000
    k, s, e = synth.gen_torch(n, 20, 24, 0x5EED000A, DEV)       # short intervals: tens of millions of clusters
    s64, e64 = s.long(), e.long() + 1
    del s, e
    torch.cuda.synchronize()
    c = ctx.cluster(k, s64, e64, n_keys=24)
    mk, ms, me, mn = ctx.merge(k, s64, e64, n_keys=24)
    torch.cuda.synchronize()
    assert c["n_clusters"] == mk.numel() == int(c["cluster"].max()) + 1          # clusters are exactly merge's runs
    assert bool((c["cluster"][1:] >= c["cluster"][:-1]).all()) and bool(((c["cluster"][1:] - c["cluster"][:-1]) <= 1).all())
    assert bool((c["cluster_start"] <= c["start"]).all()) and bool((c["end"] <= c["cluster_end"]).all())
    ids = c["cluster"].long()
    assert bool((c["cluster_start"] == ms[ids]).all()) and bool((c["cluster_end"] == me[ids]).all())   # each row carries its run's extent
    assert bool((torch.bincount(ids, minlength=mk.numel()) == mn).all())          # and the runs' sizes agree
    rows = c["row"].long()
    assert bool((s64[rows] == c["start"]).all()) and int(torch.bincount(rows, minlength=n).max()) == 1   # a permutation of the input
    assert int(c["key_clusters"].sum()) == c["n_clusters"]
    # complement against one view per contig [0, L): gaps and merged runs tile every view exactly
    lens = torch.tensor(synth.HG38, dtype=torch.int64, device=DEV) + 1000
    vk = torch.arange(24, dtype=torch.int32, device=DEV)
    gk, gs, ge = ctx.complement(k, s64, e64, vk, torch.zeros(24, dtype=torch.int64, device=DEV), lens, n_keys=24)
    torch.cuda.synchronize()
    assert bool((gs < ge).all())
    gaps = torch.zeros(24, dtype=torch.int64, device=DEV).index_add_(0, gk.long(), ge - gs)
    runs = torch.zeros(24, dtype=torch.int64, device=DEV).index_add_(0, mk.long(), me - ms)
    assert bool((gaps + runs == lens).all())
    gkey = gk.long() * (1 << 40) + gs
    assert bool((gkey[1:] > gkey[:-1]).all())                                      # ordered, disjoint
    # no gap touches an input interval
    ixc = ctx.build(pyivx.KIND_COUNT, k, s64.to(torch.int32), (e64 - 1).to(torch.int32), n_keys=24)
    hits = ctx.count_overlaps(ixc, gk, gs.to(torch.int32).contiguous(), (ge - 1).to(torch.int32).contiguous())
    assert int(hits.sum()) == 0


def test_join_100M_sorted_input_same_pairs(ctx):
    # the in-place path for sorted probe rows against the partitioned path on the same rows shuffled
    bk, bs, be = synth.gen_torch(1_000_000, 1000, 24, 0x5EED0004, DEV)
    pk, ps, pe = synth.gen_torch(100_000_000, 150, 24, 0x5EED0005, DEV)
    ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
    total = ctx.overlap_count(ix, pk, ps, pe)
    o = torch.argsort(pk.long() * (1 << 32) + ps.long())
    sk, ss, se = pk[o].contiguous(), ps[o].contiguous(), pe[o].contiguous()
    assert ctx.overlap_count(ix, sk, ss, se) == total
    b, p = ctx.overlap_fill(ix, sk, ss, se, cap=total)
    assert b.numel() == total
    bl, pl = b.long(), p.long()
    assert bool(((bs[bl] <= se[pl]) & (be[bl] >= ss[pl]) & (bk[bl] == sk[pl])).all())       # every pair satisfies the predicate
    per_row = ctx.overlap_count(ix, sk, ss, se, per_row=True)[1]
    assert bool((torch.bincount(pl, minlength=sk.numel()) == per_row.long()).all())         # and each row has its rle_right pairs
