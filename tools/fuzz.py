"""Randomised parity fuzz: random shapes / paths of every operator against the CPU oracle (bit-exact).
usage: python tools/fuzz.py [iterations] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd")); sys.path.insert(0, ROOT)
import pyivx
from oracle import oracle as orc

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = pyivx.Ctx(0)


def rows(rng, n, nkeys, span, mean, dtype=np.int32, inverted=0.0, sort=False, unknown=0):
    k = rng.integers(0, nkeys + unknown, n).astype(np.uint32)
    ln = rng.integers(0, 2 * mean, n) if rng.random() < 0.7 else np.minimum((rng.pareto(1.2, n) * mean).astype(np.int64), 200 * mean)
    s = rng.integers(0, max(span, 1), n)
    e = s + ln
    if inverted:
        m = rng.random(n) < inverted
        e[m] = s[m] - rng.integers(1, 50, m.sum())
    if sort:
        o = np.lexsort((s, k)); k, s, e = k[o], s[o], e[o]
    return k, s.astype(dtype), np.minimum(e, np.iinfo(dtype).max).astype(dtype)


def pairs(b, p):
    return np.sort((b.astype(np.uint64) << np.uint64(32)) | p.astype(np.uint64))


t0 = time.time()
for it in range(iters):
    rng = np.random.default_rng(seed0 * 1000 + it)
    nk = int(rng.choice([1, 2, 7, 24, 300]))
    span = int(rng.choice([10_000, 1_000_000, 250_000_000]))
    nb = int(rng.choice([0, 1, 50, 5_000, 200_000, 1_600_000]))
    npr = int(rng.choice([0, 1, 777, 60_000, 400_000]))
    bmean = int(rng.choice([1, 30, 1000, 20_000]))
    srt = rng.random() < 0.35
    bk, bs, be = rows(rng, nb, nk, span, bmean, inverted=float(rng.choice([0, 0, 0.02])))
    pk, ps, pe = rows(rng, npr, nk, span, int(rng.choice([1, 150, 5000])), inverted=float(rng.choice([0, 0.01])), sort=srt, unknown=int(rng.choice([0, 2])))
    strict = bool(rng.integers(0, 2))
    path = str(rng.choice(["direct", "regions", "regions", "routed"]))
    os.environ["IVX_JOIN_PATH"] = path; os.environ["IVX_ROWVAL_PATH"] = path if (path != "routed" and rng.random() < 0.7) else "routed"
    os.environ["IVX_NEAREST_PATH"] = "routed" if path != "direct" else "direct"
    dense = str(rng.choice(["0", "1", ""]))                       # pair writer of the region path: ring, count-scan-write, by density
    if dense: os.environ["IVX_DENSE"] = dense
    else: os.environ.pop("IVX_DENSE", None)
    # round-2 knobs of the region path: two-pass partition, 12-byte routed rows, occupancy bitmap never / always / by the rule
    part, pack, filt = str(rng.choice(["", "", "two"])), str(rng.choice(["", "", "0"])), str(rng.choice(["", "force", "force", "0"]))
    chunks = str(rng.choice(["", "", "2", "3"]))                  # host-resident fill / count / coverage calls cut into chunks
    fill, rpr = str(rng.choice(["", "", "old"])), str(rng.choice(["", "", "8", "2"]))   # lean / general fill kernel; rows per lane forced (ring overflow routes)
    for name, val in (("IVX_PART", part), ("IVX_PACK", pack), ("IVX_FILTER", filt), ("IVX_HOST_CHUNKS", chunks), ("IVX_FILL", fill), ("IVX_RP_ROWS", rpr)):
        if val: os.environ[name] = val
        else: os.environ.pop(name, None)
    tag = f"it={it} nk={nk} span={span} nb={nb} np={npr} bmean={bmean} sorted={srt} strict={strict} path={path} dense={dense!r} part={part!r} pack={pack!r} filter={filt!r} chunks={chunks!r} fill={fill!r} rows={rpr!r}"
    try:
        # ---- join: count, per-row, exists, fill
        ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=nk)
        if ctx.overlap_count(ix, pk, ps, pe) > 20_000_000:        # keep the oracle's host memory bounded
            ix.free(); print("skip (too many pairs)", tag, flush=True); continue
        wb, wp, wc = orc.join(bk, bs, be, pk, ps, pe, per_row=True, threads=4)
        tot, pr = ctx.overlap_count(ix, pk, ps, pe, per_row=True)
        assert tot == len(wb) and (pr.astype(np.uint64) == wc).all(), "per_row"
        assert ctx.overlap_count(ix, pk, ps, pe) == len(wb), "count"
        assert (ctx.exists(ix, pk, ps, pe) == (wc > 0)).all(), "exists"
        ob, op = ctx.overlap_fill(ix, pk, ps, pe, cap=len(wb))
        assert (pairs(ob, op) == pairs(wb, wp)).all(), "fill"
        ix.free()
        # ---- count_overlaps / coverage (coverage needs well-formed build rows)
        ixc = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=nk)
        assert (ctx.count_overlaps(ixc, pk, ps, pe, strict=strict) == orc.count_overlaps(bk, bs, be, pk, ps, pe, strict=strict)).all(), "count_overlaps"
        ixc.free()
        if not (be < bs).any():
            ixv = ctx.build(pyivx.KIND_COVERAGE, bk, bs, be, n_keys=nk)
            assert (ctx.coverage(ixv, pk, ps, pe, strict=strict) == orc.coverage(bk, bs, be, pk, ps, pe, strict=strict)).all(), "coverage"
            ixv.free()
        # ---- nearest (smaller probe side: the oracle is per-row)
        q = slice(0, min(npr, 20_000))
        kk = int(rng.choice([1, 1, 2, 3]))
        ovl = bool(rng.integers(0, 2))
        ixn = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=nk)
        g = ctx.nearest(ixn, pk[q], ps[q], pe[q], k=kk, overlap=ovl, strict=strict)
        w = orc.nearest(bk, bs, be, pk[q], ps[q], pe[q], k=kk, overlap=ovl, strict=strict)
        for a, b_ in zip(g, w):
            assert len(a) == len(b_) and (np.asarray(a).astype(np.int64) == np.asarray(b_).astype(np.int64)).all(), "nearest"
        ixn.free()
        # ---- sweeps on int64
        k64, s64, e64 = bk, bs.astype(np.int64), be.astype(np.int64)
        md = int(rng.choice([0, 0, 7, 100_000]))
        for a, b_ in zip(ctx.merge(k64, s64, e64, n_keys=nk, min_dist=md, strict=strict), orc.merge(k64, s64, e64, min_dist=md, strict=strict)):
            assert len(a) == len(b_) and (a == b_).all(), "merge"
        c, wcl = ctx.cluster(k64, s64, e64, n_keys=nk, min_dist=md, strict=strict), orc.cluster(k64, s64, e64, min_dist=md, strict=strict, n_keys=nk)
        for name in ("key", "start", "end", "row", "cluster", "cluster_start", "cluster_end"):
            assert (np.asarray(c[name]).astype(np.int64) == wcl[name].astype(np.int64)).all(), "cluster " + name
        rk, rs, re = pk[: npr // 2] % nk, ps[: npr // 2].astype(np.int64), pe[: npr // 2].astype(np.int64)
        for a, b_ in zip(ctx.subtract(k64, s64, e64, rk, rs, re, n_keys=nk, strict=strict), orc.subtract(k64, s64, e64, rk, rs, re, strict=strict)):
            assert len(a) == len(b_) and (a == b_).all(), "subtract"
        view = None if rng.random() < 0.4 else (rk[:2000], rs[:2000], re[:2000])
        got = ctx.complement(k64, s64, e64, *(view or (None, None, None)), n_keys=nk, strict=strict)
        want = orc.complement(k64, s64, e64, *(view or (None, None, None)), strict=strict)
        for a, b_ in zip(got, want):
            assert len(a) == len(b_) and (np.asarray(a).astype(np.int64) == b_.astype(np.int64)).all(), "complement"
    except AssertionError as ex:
        print("MISMATCH", ex, tag, flush=True)
        sys.exit(1)
    if it % 5 == 0:
        print(f"ok {it + 1}/{iters}  {time.time() - t0:.0f}s  last: {tag}", flush=True)
print(f"fuzz passed: {iters} iterations, seed {seed0}, {time.time() - t0:.0f}s")
