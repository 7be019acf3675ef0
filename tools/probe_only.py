"""Profiling harness: build the C3 index once, then run the probe kernels a few times."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
os.environ.setdefault("IVX_NO_PLAN", "1")      # every fill call does its own routing here
dev = torch.device("cuda:0")
npb = int(os.environ.get("NP", 100_000_000)); nb = int(os.environ.get("NB", 1_000_000)); nk = int(os.environ.get("NK", 24))
mode = os.environ.get("MODE", "count")
reps = int(os.environ.get("REPS", 3))
bk, bs, be = synth.gen_torch(nb, int(os.environ.get("BMEAN", 1000)), nk, 0x5EED0004, dev)
pk, ps, pe = synth.gen_torch(npb, 150, nk, 0x5EED0005, dev)
if os.environ.get("SORTED"):
    o = torch.argsort(pk.to(torch.int64) * (1 << 32) + ps.to(torch.int64)); pk, ps, pe = pk[o].contiguous(), ps[o].contiguous(), pe[o].contiguous()
torch.cuda.synchronize()
ctx = pyivx.Ctx(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=nk)
total = ctx.overlap_count(ix, pk, ps, pe)
ob = torch.empty(total + 16, dtype=torch.int32, device=dev); op = torch.empty_like(ob)
for r in range(reps):
    if mode == "count":
        ctx.overlap_count(ix, pk, ps, pe)
    else:
        ctx.overlap_fill(ix, pk, ps, pe, out=(ob, op))
    print(mode, "sorted" if os.environ.get("SORTED") else "random", npb, nb, nk, "pairs", total, "kernel_ms", round(ctx.last_kernel_ms(), 4), flush=True)
