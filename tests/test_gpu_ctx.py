"""-m gpu: context-level entry points of the C ABI -- ivx_scatter_fixed (the inverse of take: per-row results of a shard
back in input order), the memory reservation (scratch + the live indexes a context built, as the reference's
MemoryReservation holds the build side until the join ends: interval_join.rs:614-639) and ivx_ctx_trim."""
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


def _ctx():
    c = pyivx.Ctx(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    return c


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.int32, np.int64])
def test_scatter_fixed_device_and_host(dtype):
    ctx = _ctx()
    rng = np.random.default_rng(5)
    n_out, n = 1_000_003, 700_001
    idx = rng.permutation(n_out)[:n].astype(np.uint32)
    src = rng.integers(0, 120, n).astype(dtype)
    want = np.full(n_out, 7, dtype)
    want[idx] = src
    out = torch.full((n_out,), 7, dtype=getattr(torch, np.dtype(dtype).name), device=DEV)
    got = ctx.scatter_fixed(torch.from_numpy(src).to(DEV), torch.from_numpy(idx.view(np.int32)).to(DEV), out)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    hout = np.full(n_out, 7, dtype)
    assert np.array_equal(ctx.scatter_fixed(src, idx, hout), want)           # host buffers through the same entry point
    # take after scatter gives the values back (the two are inverses on the named rows)
    back, _ = ctx.take_fixed(got, torch.from_numpy(idx.view(np.int32)).to(DEV), want_valid=False)
    assert np.array_equal(back.cpu().numpy(), src)
    # empty input, and an index outside the output
    assert ctx.scatter_fixed(src[:0], idx[:0], hout) is hout
    bad = idx[:10].copy(); bad[3] = n_out
    with pytest.raises(pyivx.IvxError) as ei:
        ctx.scatter_fixed(src[:10], bad, hout)
    assert ei.value.status == pyivx.ERR_INVALID and "out of bounds" in str(ei.value)
    ctx.close()


def test_memory_limit_counts_live_indexes():
    ctx = _ctx()
    bk, bs, be = synth.gen_torch(2_000_000, 1000, 24, 11, torch.device(DEV))
    ix1 = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
    ix2 = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
    b = ix1.device_bytes
    assert b > 0 and ctx.reserved_bytes() >= 2 * b
    ix2.free()
    r1 = ctx.reserved_bytes()                                     # scratch (now at its steady size for this build) + ix1
    assert r1 >= b
    ctx.set_memory_limit(r1 + b // 2)                             # room for half an index more
    with pytest.raises(pyivx.IvxError) as ei:
        ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
    assert ei.value.status == pyivx.ERR_OOM and "Resources exhausted" in str(ei.value)
    assert ctx.reserved_bytes() == r1                             # the failed build holds nothing
    ix1.free()                                                    # the live index was what stood in the way
    assert ctx.reserved_bytes() == r1 - b
    ix3 = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
    assert ctx.overlap_count(ix3, bk, bs, be) > 0
    ix3.free()
    ctx.close()


def test_trim_gives_scratch_back_to_the_device():
    """A 200 M-row sweep leaves ~20 GB of scratch in its context; after ivx_ctx_trim the device has it back and a second
    context (another DataFusion partition) can run the same call; the first context still works afterwards."""
    dev = torch.device(DEV)
    a = _ctx()
    n = 200_000_000
    k, s, e = synth.gen_torch(n, 20, 24, 0x5EED0008, dev)
    s64, e64 = s.long(), e.long() + 1
    del s, e
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    m1 = a.merge(k, s64, e64, n_keys=24)[0].numel()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    held = a.reserved_bytes()
    assert held > 8 * n                                           # the sort's ping / pong records alone
    assert free0 - torch.cuda.mem_get_info()[0] >= held // 2
    a.trim(1 << 20)
    assert a.reserved_bytes() <= 1 << 20
    a.set_memory_limit(held // 4)                                 # the trimmed context would have to reserve it all again
    def status_of_merge():                                         # (no exception object kept: its traceback would hold the call's output buffers)
        try:
            a.merge(k, s64, e64, n_keys=24)
        except pyivx.IvxError as ex:
            return ex.status
        return pyivx.OK
    assert status_of_merge() == pyivx.ERR_OOM
    a.set_memory_limit(0)
    a.trim(0)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    assert free0 - torch.cuda.mem_get_info()[0] < held // 8       # the device has the bytes back
    b = _ctx()
    assert b.merge(k, s64, e64, n_keys=24)[0].numel() == m1
    b.close()
    assert a.merge(k, s64, e64, n_keys=24)[0].numel() == m1       # grows its scratch again on demand
    a.close()


def test_build_overlap_gives_the_same_pairs():
    """ivx_ctx_set_build_overlap: the build returns before its tail (cell count, scan, scatter, region descriptors) has run,
    the next probe's routing pass runs beside it; pairs, counts and every other use of the index are what they are without
    it -- also when the index is freed at once, probed from a second context, or followed by an unrelated call."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    dev = torch.device(DEV)
    bk, bs, be = synth.gen_torch(300_000, 1000, 24, 21, dev)
    pk, ps, pe = synth.gen_torch(6_000_000, 150, 24, 22, dev)
    hb = (bk.cpu().numpy().view(np.uint32), bs.cpu().numpy(), be.cpu().numpy())
    hp = (pk.cpu().numpy().view(np.uint32), ps.cpu().numpy(), pe.cpu().numpy())
    want = orc.pair_keys(*orc.join_single(*hb, *hp, threads=8))
    keys = lambda b, p: torch.sort((b.long() << 32) | p.long()).values.cpu().numpy().view(np.uint64)
    a, other = _ctx(), _ctx()
    a.set_build_overlap(True)
    for rep in range(3):
        ix = a.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
        if rep == 0:                                             # fill straight away: routing beside the build's tail
            b, p = a.overlap_fill(ix, pk, ps, pe, cap=len(want))
        elif rep == 1:                                           # count (leaves a plan), then the planned fill
            assert a.overlap_count(ix, pk, ps, pe) == len(want)
            b, p = a.overlap_fill(ix, pk, ps, pe, cap=len(want))
        else:                                                    # another context probes the index first; an unrelated call on the builder
            b, p = other.overlap_fill(ix, pk, ps, pe, cap=len(want))
            assert a.merge(bk[:1000], bs[:1000].long(), be[:1000].long() + 1, n_keys=24)[0].numel() > 0
        assert np.array_equal(keys(b, p), want)
        # the small-batch path and the per-row modes wait for the tail themselves
        assert a.overlap_count(ix, pk[:5000], ps[:5000], pe[:5000]) == int((want & np.uint64(0xFFFFFFFF)).astype(np.int64).__lt__(5000).sum())
        a.synchronize(); other.synchronize()
        ix.free()
    ix = a.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
    ix.free()                                                    # freed with its tail possibly still running
    # a build side the lean fill kernel does not take (long rows: several index levels): with the tail overlapped the host does
    # not know that when it launches the fill -- the kernels decide on the device
    lk, ls, le = synth.gen_torch(200_000, 30_000, 24, 23, dev)
    want2 = orc.pair_keys(*orc.join_single(lk.cpu().numpy().view(np.uint32), ls.cpu().numpy(), le.cpu().numpy(), *hp, threads=8))
    ix = a.build(pyivx.KIND_OVERLAP, lk, ls, le, n_keys=24)
    b, p = a.overlap_fill(ix, pk, ps, pe, cap=len(want2))
    assert np.array_equal(keys(b, p), want2)
    a.synchronize()
    ix.free()
    a.close(); other.close()
