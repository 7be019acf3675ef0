"""Where the PCIe-inclusive join call spends its time: pageable vs pinned inputs, fresh vs touched output buffers."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
npb = int(os.environ.get("NP", 100_000_000)); nb = 1_000_000
bk, bs, be = synth.gen_numpy(nb, 1000, 24, 0x5EED0004)
pk, ps, pe = synth.gen_numpy(npb, 150, 24, 0x5EED0005)
ctx = pyivx.Ctx(0)
ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
total = ctx.overlap_count(ix, pk, ps, pe)
tiny = (np.zeros(4, np.uint32), np.arange(4, dtype=np.int64), np.arange(4, dtype=np.int64) + 1)
def run(what, cols, out):
    best = 1e9
    for _ in range(4):
        ctx.merge(*tiny, n_keys=1)
        t0 = time.perf_counter(); ob, op = ctx.overlap_fill(ix, *cols, out=out() if callable(out) else out); best = min(best, time.perf_counter() - t0)
    print(f"{what:60s} {best*1e3:8.2f} ms  kernel {ctx.last_kernel_ms():.2f} ms", flush=True)
fresh = lambda: (np.empty(total, np.uint32), np.empty(total, np.uint32))
touched = (np.zeros(total, np.uint32), np.zeros(total, np.uint32))
run("pageable in, fresh (untouched) out", (pk, ps, pe), fresh)
run("pageable in, touched out", (pk, ps, pe), touched)
pin = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).pin_memory().numpy()
pkp, psp, pep = pin(pk).view(np.uint32), pin(ps), pin(pe)
pout = (torch.empty(total, dtype=torch.int32).pin_memory().numpy().view(np.uint32), torch.empty(total, dtype=torch.int32).pin_memory().numpy().view(np.uint32))
run("pinned in, touched pageable out", (pkp, psp, pep), touched)
run("pinned in, pinned out", (pkp, psp, pep), pout)
run("pageable in, pinned out", (pk, ps, pe), pout)
