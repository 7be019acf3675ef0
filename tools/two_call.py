"""Device-resident two-call protocol of the headline join: ivx_probe_overlap_count, then ivx_probe_overlap_fill
sized by it (the fill call reuses the rows the count call routed)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
dev = torch.device("cuda:0")
bk, bs, be = synth.gen_torch(int(os.environ.get("NB", 1_000_000)), 1000, 24, 0x5EED0004, dev)
pk, ps, pe = synth.gen_torch(int(os.environ.get("NP", 100_000_000)), 150, 24, 0x5EED0005, dev)
ctx = pyivx.Ctx(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
total = ctx.overlap_count(ix, pk, ps, pe)
ob = torch.empty(total + 16, dtype=torch.int32, device=dev); op = torch.empty_like(ob)
tiny = (np.zeros(4, np.uint32), np.arange(4, dtype=np.int64), np.arange(4, dtype=np.int64) + 1)
for what in ("count + fill", "fill alone", "count alone"):
    best = 1e9
    for _ in range(5):
        ctx.merge(*tiny, n_keys=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if what != "fill alone":
            assert ctx.overlap_count(ix, pk, ps, pe) == total
        if what != "count alone":
            b, p = ctx.overlap_fill(ix, pk, ps, pe, out=(ob, op)); assert b.numel() == total
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{what:14s} {best*1e3:7.3f} ms", flush=True)
