"""Per-call latency of the probe entry points on DataFusion-sized batches (8192 rows), device-resident."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
dev = torch.device("cuda:0")
bk, bs, be = synth.gen_torch(1_000_000, 1000, 24, 0x5EED0004, dev)
ctx = pyivx.Ctx(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
ixc = ctx.build(pyivx.KIND_COUNT, bk, bs, be, n_keys=24)
for n in (8192, 65536, 1 << 20):
    pk, ps, pe = synth.gen_torch(n, 150, 24, 0x5EED0005, dev)
    total = ctx.overlap_count(ix, pk, ps, pe)
    ob = torch.empty(total + 16, dtype=torch.int32, device=dev); op = torch.empty_like(ob)
    out = torch.empty(n, dtype=torch.int64, device=dev)
    for name, fn in (("overlap_count", lambda: ctx.overlap_count(ix, pk, ps, pe)),
                     ("overlap_fill", lambda: ctx.overlap_fill(ix, pk, ps, pe, out=(ob, op))),
                     ("count_overlaps", lambda: ctx.count_overlaps(ixc, pk, ps, pe))):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(f"{name:16s} n={n:8d}: {dt*1e6:8.1f} us per call  {n/dt/1e6:9.1f} M rows/s  (kernel {ctx.last_kernel_ms()*1e3:.1f} us)", flush=True)
