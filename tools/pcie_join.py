"""PCIe-inclusive rate of the headline join: host (numpy) probe columns through IVX_MEM_HOST, pageable numpy buffers (hipHostRegister'ed buffers measured SLOWER on this host: 52 ms vs 40 ms)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
npb = int(os.environ.get("NP", 100_000_000)); nb = 1_000_000
bk, bs, be = synth.gen_numpy(nb, 1000, 24, 0x5EED0004)
pk, ps, pe = synth.gen_numpy(npb, 150, 24, 0x5EED0005)
bk, pk = bk.astype(np.uint32), pk.astype(np.uint32)
ctx = pyivx.Ctx(0)
ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=24)
total = ctx.overlap_count(ix, pk, ps, pe)
for mode in ("pageable",):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ob, op = ctx.overlap_fill(ix, pk, ps, pe, cap=total); best = min(best, time.perf_counter() - t0)
    print(f"{mode:10s} overlap_fill {npb} host rows -> {len(ob)} pairs (pairs copied back): {best*1e3:8.2f} ms  "
          f"{npb/best/1e9:6.2f} G probe rows/s  {len(ob)/best/1e9:6.2f} G pairs/s  kernel {ctx.last_kernel_ms():.2f} ms  "
          f"H2D {12*npb/1e9:.2f} GB + D2H {8*len(ob)/1e9:.2f} GB", flush=True)
