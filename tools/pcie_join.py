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
tiny = (np.zeros(4, np.uint32), np.arange(4, dtype=np.int64), np.arange(4, dtype=np.int64) + 1)


def report(what, best, n_pairs):
    print(f"{what:42s} {npb} host rows -> {n_pairs} pairs: {best*1e3:8.2f} ms  {npb/best/1e9:6.2f} G probe rows/s  "
          f"{n_pairs/best/1e9:6.2f} G pairs/s  H2D {12*npb/1e9:.2f} GB + D2H {8*n_pairs/1e9:.2f} GB", flush=True)


# (a) one fill call with a known capacity: columns in, routing, probe, pairs out
best = 1e9
for _ in range(3):
    ctx.merge(*tiny, n_keys=1)                      # anything else on the context: the next fill call starts from scratch
    t0 = time.perf_counter(); ob, op = ctx.overlap_fill(ix, pk, ps, pe, cap=total); best = min(best, time.perf_counter() - t0)
report("overlap_fill alone (pageable numpy)", best, len(ob))
# (b) the two-call protocol: count (columns in, routing, count), then fill (probe + pairs out; reuses the routed rows)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); ob, op = ctx.overlap_fill(ix, pk, ps, pe); best = min(best, time.perf_counter() - t0)
report("overlap_count + overlap_fill", best, len(ob))
