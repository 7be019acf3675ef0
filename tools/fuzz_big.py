"""A few large random cases on the DEFAULT paths (batches above the region-pipeline thresholds; build sides below
the 255-region limit, in the 1024-digit range and beyond it; short and long build rows; sparse and match-dense),
join / count_overlaps / coverage against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd")); sys.path.insert(0, ROOT)
import pyivx
from oracle import oracle as orc
ctx = pyivx.Ctx(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for it in range(iters):
    rng = np.random.default_rng(500 + it)
    nk = int(rng.choice([1, 5, 24]))
    span = int(rng.choice([50_000_000, 250_000_000]))
    nb = int(rng.choice([300_000, 1_200_000, 3_000_000, 8_000_000]))
    blen = int(rng.choice([300, 2000, 2000, 40_000]))
    npr = int(rng.choice([2_200_000, 4_100_003]))
    srt = rng.random() < 0.4
    bk = rng.integers(0, nk, nb).astype(np.uint32); bs = rng.integers(0, span, nb).astype(np.int32); be = (bs + rng.integers(0, blen, nb)).astype(np.int32)
    pk = rng.integers(0, nk + 1, npr).astype(np.uint32); ps = rng.integers(0, span, npr).astype(np.int32); pe = (ps + rng.integers(0, 300, npr)).astype(np.int32)
    if srt:
        o = np.lexsort((ps, pk)); pk, ps, pe = pk[o], ps[o], pe[o]
    strict = bool(rng.integers(0, 2))
    tag = f"it={it} nk={nk} span={span} nb={nb} blen={blen} np={npr} sorted={srt} strict={strict}"
    ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=nk)
    if ctx.overlap_count(ix, pk, ps, pe) > 400_000_000:           # keep the oracle's host memory bounded
        ix.free(); print("skip (too many pairs)", tag, flush=True); continue
    wb, wp, wc = orc.join(bk, bs, be, pk, ps, pe, per_row=True, threads=16)
    tot, pr = ctx.overlap_count(ix, pk, ps, pe, per_row=True)
    ob, op = ctx.overlap_fill(ix, pk, ps, pe, cap=len(wb))
    v = lambda b, p: np.sort((b.astype(np.uint64) << np.uint64(32)) | p.astype(np.uint64))
    assert tot == len(wb) and (pr.astype(np.uint64) == wc).all() and (v(ob, op) == v(wb, wp)).all(), "join " + tag
    assert (ctx.exists(ix, pk, ps, pe) == (wc > 0)).all(), "exists " + tag
    ix.free()
    ixc = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=nk)
    assert (ctx.count_overlaps(ixc, pk, ps, pe, strict=strict) == orc.count_overlaps(bk, bs, be, pk, ps, pe, strict=strict)).all(), "count " + tag
    ixc.free()
    ixv = ctx.build(pyivx.KIND_COVERAGE, bk, bs, be, n_keys=nk)
    assert (ctx.coverage(ixv, pk, ps, pe, strict=strict) == orc.coverage(bk, bs, be, pk, ps, pe, strict=strict)).all(), "coverage " + tag
    ixv.free()
    print("ok", tag, flush=True)
print("fuzz_big passed")
