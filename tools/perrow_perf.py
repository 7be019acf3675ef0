"""rle_right (per-row counts) and exists on the headline workload: direct gathers vs the region partition."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
dev = torch.device("cuda:0")
bk, bs, be = synth.gen_torch(1_000_000, 1000, 24, 0x5EED0004, dev)
pk, ps, pe = synth.gen_torch(100_000_000, 150, 24, 0x5EED0005, dev)
ctx = pyivx.Ctx(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
for path in ("direct", "regions"):
    os.environ["IVX_JOIN_PATH"] = path
    for name, fn in (("per_row", lambda: ctx.overlap_count(ix, pk, ps, pe, per_row=True)), ("exists", lambda: ctx.exists(ix, pk, ps, pe))):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): out = fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"{path:8s} {name:8s} {dt*1e3:8.3f} ms  kernel {ctx.last_kernel_ms():.3f} ms", flush=True)
