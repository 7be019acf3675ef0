"""k = 1 nearest probe, prefix-max-first overlap lookup on / off, dense and sparse build sides (device-resident)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
dev = torch.device("cuda:0")
ctx = pyivx.Ctx(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
pk, ps, pe = synth.gen_torch(50_000_000, 150, 24, 0x5EED0007, dev)
for nb in (50_000_000, 1_000_000):
    bk, bs, be = synth.gen_torch(nb, 1000, 24, 0x5EED0006, dev)
    ix = ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=24)
    ref = None
    for flag in ("0", "1", "0", "1"):
        os.environ["IVX_NEAREST_PMAX_FIRST"] = flag
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ob, op, od = ctx.nearest(ix, pk, ps, pe, k=1)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        sig = (int(ob.long().sum()), int(od.sum()))
        ref = ref or sig
        print(f"build {nb:>9d} pmax_first={flag}  probe {best * 1e3:7.3f} ms  same={sig == ref}", flush=True)
    ix.free(); del bk, bs, be
