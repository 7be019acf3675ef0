"""count_overlaps / coverage / rle_right on build sides beyond the LDS-slice pipeline (routed gathers): timing."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
dev = torch.device("cuda:0")
ctx = pyivx.Ctx(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
npr = 100_000_000
pk, ps, pe = synth.gen_torch(npr, 150, 24, 0x5EED0005, dev)
for nb in (10_000_000, 50_000_000):
    bk, bs, be = synth.gen_torch(nb, 1000, 24, 0x5EED0004, dev)
    for kind, fn in ((pyivx.KIND_COUNT, "count_overlaps"), (pyivx.KIND_COVERAGE, "coverage")):
        ix = ctx.build(kind, bk, bs, be, n_keys=24)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); out = getattr(ctx, fn)(ix, pk, ps, pe); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{fn} 100M x {nb}: {best*1e3:.3f} ms", flush=True)
        ix.free()
    ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = ctx.overlap_count(ix, pk, ps, pe, per_row=True); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"rle_right 100M x {nb}: {best*1e3:.3f} ms", flush=True)
    ix.free()
