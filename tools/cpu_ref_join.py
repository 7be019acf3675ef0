"""A reference-authored CPU figure beside the oracle port: the reference's own vendored
superintervals.hpp (compiled from /root/reference into oracle/_ref by oracle/Makefile; the structure behind
Algorithm::SuperIntervals, interval_join.rs:832-845) joins a sample of the headline workload on the build container's
host cores, next to the port (oracle/ivx_oracle.c, orc_join_single_run) on the same sample and thread count.
Build container only: oracle/_ref does not travel to the GPU box.   python tools/cpu_ref_join.py [probe rows] [threads]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import synth
from oracle import oracle as orc

npr = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else len(os.sched_getaffinity(0))
nb = 1_000_000
bk, bs, be = synth.gen_numpy(nb, 1000, 24, 0x5EED0004)
pk, ps, pe = synth.gen_numpy(npr, 150, 24, 0x5EED0005)
orc.lib()
assert orc.ref_available(), "oracle/_ref is missing: make -C oracle ref (needs /root/reference)"

t0 = time.perf_counter(); pb, pp = orc.join_single(bk, bs, be, pk, ps, pe, threads=threads); t_port = time.perf_counter() - t0
want = orc.pair_keys(pb, pp)

def part(i):
    lo, hi = npr * i // threads, npr * (i + 1) // threads
    b, p = orc.ref_join(bk, bs, be, pk[lo:hi], ps[lo:hi], pe[lo:hi])      # (ctypes releases the GIL; every thread builds its own maps, as a partition would)
    return b, p + np.uint32(lo)
t0 = time.perf_counter()
with ThreadPoolExecutor(threads) as ex:
    parts = list(ex.map(part, range(threads)))
t_ref = time.perf_counter() - t0
rb = np.concatenate([x[0] for x in parts]); rp = np.concatenate([x[1] for x in parts])
assert np.array_equal(orc.pair_keys(rb, rp), want), "reference structure and port disagree"
t0 = time.perf_counter(); b1, p1 = orc.ref_join(bk, bs, be, pk[:npr // threads], ps[:npr // threads], pe[:npr // threads]); t_ref1 = time.perf_counter() - t0
print(f"sample: {npr} probe rows x {nb} build rows, 24 contigs, {len(want)} pairs, {threads} threads")
print(f"port   (oracle/ivx_oracle.c orc_join_single_run):      {t_port:7.2f} s  {len(want) / t_port / 1e6:8.2f} M pairs/s  {npr / t_port / 1e6:8.2f} M probe rows/s")
print(f"reference superintervals.hpp (ref_si_join x {threads}):        {t_ref:7.2f} s  {len(want) / t_ref / 1e6:8.2f} M pairs/s  {npr / t_ref / 1e6:8.2f} M probe rows/s")
print(f"reference superintervals.hpp, one thread, {npr // threads} rows: {t_ref1:7.2f} s  {len(b1) / t_ref1 / 1e6:8.2f} M pairs/s")
