"""CPU, world_size 2, gloo: the N>1 path -- keys sharded over ranks by LPT, every rank joins its own
keys with no communication, the per-rank pair buffers are all-gathered (allgatherv); the union must
equal the single-process join.  The per-rank compute here is the CPU oracle (no GPU in this tier);
the sharding / exchange code is the one bench.py runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pair_set, synth

sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import shard  # noqa: E402


def test_lpt_balances_human_contigs():
    import synth as _s
    w = np.array(_s.HG38, dtype=np.int64)
    r = shard.assign_keys_lpt(w, 8)
    load = np.bincount(r, weights=w, minlength=8)
    assert load.max() / load.mean() < 1.12          # 24 contigs over 8 ranks: within ~10 %
    assert sorted(set(r.tolist())) == list(range(8))


def _worker(rank, world, port, ret, nk=7):
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bk, bs, be = synth(4000, 5, nkeys=nk, mean_len=800, span=300_000)
    pk, ps, pe = synth(30000, 6, nkeys=nk, mean_len=150, span=300_000)
    w = np.bincount(bk, minlength=nk) + np.bincount(pk, minlength=nk)
    rank_of = shard.assign_keys_lpt(w, world)
    mb, mp_ = rank_of[bk] == rank, rank_of[pk] == rank
    rows_b, rows_p = np.nonzero(mb)[0], np.nonzero(mp_)[0]
    ob, op = orc.join(bk[mb], bs[mb], be[mb], pk[mp_], ps[mp_], pe[mp_])
    gb = torch.from_numpy(rows_b[ob].astype(np.int64)); gp = torch.from_numpy(rows_p[op].astype(np.int64))   # back to global rows
    (ab, ap), sizes = shard.allgatherv(dist, (gb, gp))
    t = shard.max_over_ranks(dist, 1.0 + rank, torch.device("cpu"))
    if rank == 0:
        wb, wp = orc.join(bk, bs, be, pk, ps, pe)
        ok = (pair_set(ab.numpy().astype(np.uint32), ap.numpy().astype(np.uint32)) == pair_set(wb, wp)).all()
        ret["ok"] = bool(ok) and sum(sizes) == len(wb) and t == float(world)
    dist.destroy_process_group()


def test_sharded_join_allgatherv_world2():
    port = 29500 + os.getpid() % 2000
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret.get("ok") is True


def test_sharded_join_allgatherv_world3_with_an_idle_rank():
    # fewer contigs than ranks: one rank owns no key, joins nothing and contributes an empty slice to the exchange
    port = 31500 + os.getpid() % 2000
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(3, port, ret, 2), nprocs=3, join=True)
        assert ret.get("ok") is True


def _cluster_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nk = 9
    k, s, e = synth(20000, 11, nkeys=nk, mean_len=300, span=2_000_000, dtype=np.int64)
    rank_of = shard.assign_keys_lpt(np.bincount(k, minlength=nk), world)
    mine = rank_of[k] == rank
    counts = orc.cluster(k[mine], s[mine], e[mine], n_keys=nk)["key_clusters"].astype(np.int64)     # count-only call
    base = shard.cluster_key_base(dist, torch.from_numpy(counts))
    part = orc.cluster(k[mine], s[mine], e[mine], n_keys=nk, key_base=base.numpy())
    full = orc.cluster(k, s, e, n_keys=nk)
    sel = np.isin(full["key"], np.flatnonzero(rank_of == rank))
    ok = all((part[c] == full[c][sel]).all() for c in ("key", "start", "end", "cluster", "cluster_start", "cluster_end"))
    flags = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(flags, torch.tensor([int(ok)], dtype=torch.int64))
    if rank == 0:
        ret["ok"] = all(int(f) == 1 for f in flags)
    dist.destroy_process_group()


def test_sharded_cluster_ids_world2():
    # contigs sharded over two ranks; global cluster ids equal the single-process ids
    port = 31500 + os.getpid() % 2000
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_cluster_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret.get("ok") is True
