"""Scratch perf probe for the overlap join (device-resident inputs)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx

def gen(n, mean, nk, seed, dev):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    span = 3_100_000_000 // max(nk, 1) if nk > 1 else 248_956_422
    key = torch.randint(0, nk, (n,), generator=g, device=dev, dtype=torch.int32)
    ln = 1 + torch.randint(0, 2 * mean - 1, (n,), generator=g, device=dev, dtype=torch.int32)
    st = torch.randint(0, span - 2 * mean, (n,), generator=g, device=dev, dtype=torch.int32)
    return key, st, st + ln - 1

dev = torch.device("cuda:0")
ctx = pyivx.Ctx(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for (npb, nb, nk) in [(10_000_000, 100_000, 1), (100_000_000, 1_000_000, 24)]:
    bk, bs, be = gen(nb, 1000, nk, 1, dev)
    pk, ps, pe = gen(npb, 150, nk, 2, dev)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=nk)
        t1 = time.perf_counter()
        total = ctx.overlap_count(ix, pk, ps, pe)
        t2 = time.perf_counter(); kc = ctx.last_kernel_ms()
        ob = torch.empty(total + 16, dtype=torch.int32, device=dev); op = torch.empty_like(ob)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        ctx.overlap_fill(ix, pk, ps, pe, out=(ob, op))
        t4 = time.perf_counter(); kf = ctx.last_kernel_ms()
        print(f"{npb}x{nb} k={nk} pairs={total} build {1e3*(t1-t0):.3f} ms  count {1e3*(t2-t1):.3f} (kernel {kc:.3f})  fill {1e3*(t4-t3):.3f} (kernel {kf:.3f}) idx_bytes={ix.device_bytes}", flush=True)
        ix.free()
