"""Device-resident timing of the non-join operators at BASELINE.json scale (1 GPU)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
dev = torch.device("cuda:0")
which = os.environ.get("OPS", "count,coverage,nearest,merge,subtract,cluster,complement,take").split(",")
scale = float(os.environ.get("SCALE", "1"))
ctx = pyivx.Ctx(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)

def timed(fn, reps=3):
    best = 1e9; out = None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out

def report(name, secs, nbytes, extra=""):
    print(f"{name:34s} {secs*1e3:10.3f} ms  {nbytes/secs/1e9:9.1f} GB/s algorithmic  {extra}", flush=True)

if "count" in which or "coverage" in which:
    nb, npr = int(1_000_000 * scale), int(100_000_000 * scale)
    bk, bs, be = synth.gen_torch(nb, 1000, 24, 0x5EED0004, dev)
    pk, ps, pe = synth.gen_torch(npr, 150, 24, 0x5EED0005, dev)
    for kind, name, fn in ((pyivx.KIND_COUNT, "count_overlaps", "count_overlaps"), (pyivx.KIND_COVERAGE, "coverage", "coverage")):
        if name.split("_")[0] not in which and name not in which: continue
        tb, ix = timed(lambda: ctx.build(kind, bk, bs, be, n_keys=24))
        tp, out = timed(lambda: getattr(ctx, fn)(ix, pk, ps, pe))
        km = ctx.last_kernel_ms()
        report(f"{name} build {nb}", tb, 12 * nb)
        report(f"{name} probe {npr}x{nb}", tp, 12 * (npr + nb) + 8 * npr, f"kernel {km:.3f} ms sum={int(out.sum())}")
        ix.free()
if "nearest" in which:
    nb, npr = int(50_000_000 * scale), int(50_000_000 * scale)
    bk, bs, be = synth.gen_torch(nb, 1000, 24, 0x5EED0006, dev)
    pk, ps, pe = synth.gen_torch(npr, 150, 24, 0x5EED0007, dev)
    if os.environ.get("SORTED"):
        o = torch.argsort((bk.to(torch.int64) << 58) | (bs.to(torch.int64) << 29) | be.to(torch.int64))
        bk, bs, be = bk[o].contiguous(), bs[o].contiguous(), be[o].contiguous()
        o = torch.argsort((pk.to(torch.int64) << 58) | (ps.to(torch.int64) << 29) | pe.to(torch.int64))
        pk, ps, pe = pk[o].contiguous(), ps[o].contiguous(), pe[o].contiguous()
        del o
    tb, ix = timed(lambda: ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=24), reps=2)
    report(f"nearest build {nb}", tb, 12 * nb, f"index {ix.device_bytes/1e9:.2f} GB")
    tp, out = timed(lambda: ctx.nearest(ix, pk, ps, pe, k=1))
    report(f"nearest k=1 probe {npr}x{nb}", tp, 12 * (npr + nb) + 16 * npr, f"kernel {ctx.last_kernel_ms():.3f} ms")
    tp, out = timed(lambda: ctx.nearest(ix, pk[:npr // 5], ps[:npr // 5], pe[:npr // 5], k=3), reps=2)
    report(f"nearest k=3 probe {npr//5}x{nb}", tp, 12 * (npr // 5 + nb) + 48 * (npr // 5), f"rows {out[0].numel()}")
    ix.free(); del bk, bs, be, pk, ps, pe
if "merge" in which or "subtract" in which or "cluster" in which or "complement" in which:
    n = int(float(os.environ.get("NMERGE", 200_000_000)) * scale)
    k, s, e = synth.gen_torch(n, 1000, 24, 0x5EED0008, dev)
    s64, e64 = s.to(torch.int64), e.to(torch.int64) + 1
    del s, e
    if os.environ.get("SORTED"):                           # coordinate-sorted input (the sweeps then skip their radix sort)
        o = torch.argsort((k.to(torch.int64) << 58) | (s64 << 29) | e64)
        k, s64, e64 = k[o].contiguous(), s64[o].contiguous(), e64[o].contiguous()
        del o
    if "merge" in which:
        tm, out = timed(lambda: ctx.merge(k, s64, e64, n_keys=24), reps=4)
        m = out[0].numel()
        report(f"merge {n}", tm, 20 * n + 28 * m, f"kernel {ctx.last_kernel_ms():.3f} ms out rows {m}")
    if "subtract" in which:
        nr = n // 10
        rk, rs, re = synth.gen_torch(nr, 150, 24, 0x5EED0009, dev)
        rs64, re64 = rs.to(torch.int64), re.to(torch.int64) + 1
        ts, out = timed(lambda: ctx.subtract(k, s64, e64, rk, rs64, re64, n_keys=24), reps=4)
        m = out[0].numel()
        report(f"subtract {n}-{nr} (count+fill)", ts, 20 * (n + nr) + 20 * m, f"out rows {m}")
    if "subtract" in which:
        del rk, rs, re, rs64, re64
    if "cluster" in which or "complement" in which:
        # dense (everything of a contig chains into one run) and sparse (mean length 20: ~60 % singletons) inputs
        n2 = n // 4
        k2, s2, e2 = synth.gen_torch(n2, 20, 24, 0x5EED000A, dev)
        s2, e2 = s2.to(torch.int64), e2.to(torch.int64) + 1
        for tag, (kk, ss, ee, nn) in (("dense", (k, s64, e64, n)), ("sparse", (k2, s2, e2, n2))):
            if "cluster" in which:
                tc, out = timed(lambda: ctx.cluster(kk, ss, ee, n_keys=24), reps=3)
                report(f"cluster {tag} {nn}", tc, 68 * nn, f"kernel {ctx.last_kernel_ms():.3f} ms clusters {out['n_clusters']}")
                del out
            if "complement" in which:
                tc, out = timed(lambda: ctx.complement(kk, ss, ee, n_keys=24), reps=2)
                m = out[0].numel()
                report(f"complement {tag} {nn} (count+fill)", tc, 20 * nn + 20 * m, f"out rows {m}")
                del out
if "take" in which:
    # payload gather for the headline join's 37M pairs: build-side column (1M rows, random indices),
    # probe-side column (100M rows, ascending-ish indices), a 12-byte string column of the build side
    npairs, nb, npr = int(37_000_000 * scale), int(1_000_000 * scale), int(100_000_000 * scale)
    g = torch.Generator(device=dev).manual_seed(7)
    bi = torch.randint(0, nb, (npairs,), generator=g, device=dev, dtype=torch.int32)
    pi = torch.sort(torch.randint(0, npr, (npairs,), generator=g, device=dev, dtype=torch.int32)).values
    for name, src, idx in (("take i64 build col", torch.arange(nb, device=dev, dtype=torch.int64), bi),
                           ("take i32 probe col", torch.arange(npr, device=dev, dtype=torch.int32), pi),
                           ("take i64 probe col", torch.arange(npr, device=dev, dtype=torch.int64), pi)):
        t, out = timed(lambda: ctx.take_fixed(src, idx, want_valid=False))
        w = src.element_size()
        report(f"{name} {npairs}", t, (4 + 2 * w) * npairs, f"kernel {ctx.last_kernel_ms():.3f} ms")
        del out
    off = torch.arange(0, 12 * (nb + 1), 12, device=dev, dtype=torch.int32)
    data = torch.randint(65, 90, (12 * nb,), generator=g, device=dev, dtype=torch.uint8)
    t, out = timed(lambda: ctx.take_utf8(off, data, bi, want_valid=False))
    report(f"take utf8(12B) build col {npairs} (size+fill)", t, (4 + 4 + 24) * npairs, f"bytes {out[1].numel()}")
