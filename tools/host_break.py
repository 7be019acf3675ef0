"""Wall time of the three host calls of a headline step (build, fill, free), device-resident columns."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
dev = torch.device("cuda:0")
ctx = pyivx.Ctx(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream); ctx.set_build_overlap(True)
bk, bs, be = synth.gen_torch(1_000_000, 1000, 24, 0x5EED0002, dev)
pk, ps, pe = synth.gen_torch(100_000_000, 150, 24, 0x5EED0003, dev)
ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
cap = ctx.overlap_count(ix, pk, ps, pe) + 1024
ob = torch.empty(cap, dtype=torch.int32, device=dev); op = torch.empty(cap, dtype=torch.int32, device=dev)
ix.free()
tb = tf = tx = 0.0
N = 30
for it in range(N + 5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
    t1 = time.perf_counter(); n = ctx.overlap_fill_into(ix, pk, ps, pe, ob, op) if hasattr(ctx, "overlap_fill_into") else ctx.overlap_fill(ix, pk, ps, pe, out=(ob, op))
    t2 = time.perf_counter(); ix.free()
    t3 = time.perf_counter()
    if it >= 5: tb += t1 - t0; tf += t2 - t1; tx += t3 - t2
print(f"build {tb/N*1e6:.1f} us  fill {tf/N*1e6:.1f} us  free {tx/N*1e6:.1f} us  total {(tb+tf+tx)/N*1e6:.1f} us")
