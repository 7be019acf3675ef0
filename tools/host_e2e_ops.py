"""Host-resident count_overlaps / coverage of 100M x 1M rows: wall time of the IVX_MEM_HOST call (chunked or not)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import torch, pyivx, synth
ctx = pyivx.Ctx(0)
bk, bs, be = [x.cpu().numpy() for x in synth.gen_torch(1_000_000, 1000, 24, 0x5EED0004, "cuda:0")]
pk, ps, pe = [x.cpu().numpy() for x in synth.gen_torch(100_000_000, 150, 24, 0x5EED0005, "cuda:0")]
bk, pk = bk.view(np.uint32), pk.view(np.uint32)
for kind, fn in ((pyivx.KIND_COUNT, "count_overlaps"), (pyivx.KIND_COVERAGE, "coverage")):
    ix = ctx.build(kind, bk, bs, be, n_keys=24)
    best = 1e9
    buf = np.zeros(len(pk), np.int64)                          # touched once: fresh pages would cost more than the copy
    for _ in range(3):
        t0 = time.perf_counter(); out = getattr(ctx, fn)(ix, pk, ps, pe, out=buf); best = min(best, time.perf_counter() - t0)
    print(f"{fn} host columns 100M x 1M: {best*1e3:.1f} ms  (chunks={os.environ.get('IVX_HOST_CHUNKS', 'default')}) sum={int(out.sum())}", flush=True)
    ix.free()
