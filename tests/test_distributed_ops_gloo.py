"""CPU, gloo, world 2 and 3: every operator of BASELINE configs C3 / C4 / C5 (and cluster, complement) with contigs
sharded over ranks by LPT (`sharded.ShardedRanges`): the exchanged result must equal the single-process result
column for column -- what the reference pins for its own partitioned forms (R/tests/integration_test.rs:3709-3755,
:3783-3890, :3923-3951, :3987-4020).  Per-rank compute is the CPU oracle behind pyivx.Ctx's method names
(tests/oracle_engine.py); the sharding, row bookkeeping and exchange code is what bench.py runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, synth

sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import shard  # noqa: E402
import sharded  # noqa: E402


def _t(k, s, e):
    return torch.from_numpy(k.astype(np.int32)), torch.from_numpy(s), torch.from_numpy(e)


def _take(cols, mask):
    return tuple(c[mask].contiguous() for c in cols)


def _same(a, b):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    if a.dtype.itemsize == 4:
        a, b = a.view(np.uint32), np.asarray(b).astype(np.uint32, copy=False) if b.dtype.kind == "u" else b.view(np.uint32)
    return a.shape == b.shape and bool((a == b).all())


def _worker(rank, world, port, ret, nk):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as orc
    from oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sr = sharded.ShardedRanges(dist, OracleEngine(), nk)
    ok = {}

    # ---- C3 / C4: count_overlaps + coverage, nearest (int32 closed coordinates)
    bk, bs, be = synth(3000, 21, nkeys=nk, mean_len=900, span=400_000)
    pk, ps, pe = synth(20000, 22, nkeys=nk, mean_len=150, span=400_000)
    rank_of = shard.assign_keys_lpt(np.bincount(bk, minlength=nk) + np.bincount(pk, minlength=nk), world)
    mb, mp_ = rank_of[bk] == rank, rank_of[pk] == rank
    rows_b = torch.from_numpy(np.flatnonzero(mb).astype(np.int32)); rows_p = torch.from_numpy(np.flatnonzero(mp_).astype(np.int32))
    B, P = _take(_t(bk, bs, be), torch.from_numpy(mb)), _take(_t(pk, ps, pe), torch.from_numpy(mp_))
    for strict in (False, True):
        cnt, cov = sr.count_coverage(B, P, rows_p, len(pk), strict=strict)
        ok[f"count{strict}"] = _same(cnt, orc.count_overlaps(bk, bs, be, pk, ps, pe, strict=strict))
        ok[f"coverage{strict}"] = _same(cov, orc.coverage(bk, bs, be, pk, ps, pe, strict=strict))
    for overlap in (True, False):
        nb, nd = sr.nearest1(B, rows_b, P, rows_p, len(pk), overlap=overlap)
        wb, _, wd = orc.nearest(bk, bs, be, pk, ps, pe, k=1, overlap=overlap)
        # a probe row of a contig without build rows has no neighbour on any rank: NULL / -1, as in the whole job
        ok[f"nearest{overlap}"] = _same(nb, wb) and _same(nd, wd)

    # ---- C5 (+ cluster, complement): int64 half-open coordinates
    k, s, e = synth(30000, 23, nkeys=nk, mean_len=40, span=900_000, dtype=np.int64)
    e = e + 1
    rk, rs, re = synth(4000, 24, nkeys=nk, mean_len=25, span=900_000, dtype=np.int64)
    re = re + 1
    rank_of = shard.assign_keys_lpt(np.bincount(k, minlength=nk) + np.bincount(rk, minlength=nk), world)
    ml, mr = rank_of[k] == rank, rank_of[rk] == rank
    rows_l = torch.from_numpy(np.flatnonzero(ml).astype(np.int32))
    L, R = _take(_t(k, s, e), torch.from_numpy(ml)), _take(_t(rk, rs, re), torch.from_numpy(mr))
    for md in (0, 7):
        got = sr.merge(L, min_dist=md)
        ok[f"merge{md}"] = all(_same(g, w) for g, w in zip(got, orc.merge(k, s, e, min_dist=md)))
    got = sr.subtract(L, rows_l, R)
    ok["subtract"] = all(_same(g, w) for g, w in zip(got, orc.subtract(k, s, e, rk, rs, re)))
    got = sr.cluster(L, rows_l, min_dist=3)
    want = orc.cluster(k, s, e, min_dist=3, n_keys=nk)
    ok["cluster"] = all(_same(got[c], want[c]) for c in ("key", "start", "end", "row", "cluster", "cluster_start", "cluster_end"))
    # complement: explicit views on some contigs only, one contig with views and no input rows (it must come last)
    vk = np.array([0, 0, nk - 1], np.uint32) if nk > 1 else np.array([0, 0], np.uint32)
    vs = np.array([100, 500_000, 10], np.int64)[:len(vk)]; ve = np.array([400_000, 1_200_000, 90], np.int64)[:len(vk)]
    keep = k != (nk - 1) if nk > 1 else np.ones(len(k), bool)
    k2, s2, e2 = k[keep], s[keep], e[keep]
    m2 = rank_of[k2] == rank
    mv = rank_of[vk] == rank
    got = sr.complement(_take(_t(k2, s2, e2), torch.from_numpy(m2)), view=_take(_t(vk, vs, ve), torch.from_numpy(mv)))
    ok["complement_views"] = all(_same(g, w) for g, w in zip(got, orc.complement(k2, s2, e2, vk, vs, ve)))
    got = sr.complement(L)
    ok["complement"] = all(_same(g, w) for g, w in zip(got, orc.complement(k, s, e)))

    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if rank == 0:
        ret["ok"] = {r: {n: v for n, v in f.items() if not v} for r, f in enumerate(flags)}
        ret["n"] = len(ok)
    dist.destroy_process_group()


def _run(world, nk, port):
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret, nk), nprocs=world, join=True)
        assert ret.get("n", 0) >= 12
        assert all(not bad for bad in ret["ok"].values()), dict(ret["ok"])


def test_sharded_operators_world2():
    _run(2, 7, 33500 + os.getpid() % 2000)


def test_sharded_operators_world3_with_an_idle_rank():
    # two contigs over three ranks: one rank holds no rows at all and contributes empty pieces to every exchange
    _run(3, 2, 35500 + os.getpid() % 2000)
