"""direct-gather vs region-partitioned probe as a function of the batch size (device-resident)."""
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
for lg in (17, 18, 19, 20, 21, 22, 23):
    n = 1 << lg
    pk, ps, pe = synth.gen_torch(n, 150, 24, 0x5EED0005, dev)
    total = ctx.overlap_count(ix, pk, ps, pe)
    ob = torch.empty(total + 16, dtype=torch.int32, device=dev); op = torch.empty_like(ob)
    line = f"n=2^{lg}"
    for path in ("direct", "regions"):
        os.environ["IVX_JOIN_PATH"] = path; os.environ["IVX_ROWVAL_PATH"] = path
        for name, fn in (("fill", lambda: ctx.overlap_fill(ix, pk, ps, pe, out=(ob, op))), ("count_ov", lambda: ctx.count_overlaps(ixc, pk, ps, pe))):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
            line += f"  {path}/{name} {dt*1e6:8.1f} us"
    print(line, flush=True)
