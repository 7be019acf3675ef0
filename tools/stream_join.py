"""Probe batches pushed through brh_join_stream (host Arrow batches -> coalesced groups -> GPU): the headline
join's 100M x 1M rows as RecordBatches of BATCH rows.  Reports the host-side push cost and the end-to-end rate."""
import os, sys, time
import numpy as np, pyarrow as pa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import bio_ranges as br, synth
npb = int(os.environ.get("NP", 100_000_000)); nb = 1_000_000
batch = int(os.environ.get("BATCH", 65536)); coalesce = int(os.environ.get("COALESCE", 0))
names = pa.array(["chr%s" % c for c in list(range(1, 23)) + ["X", "Y"]])


def tab(n, mean, seed, sort=False):
    k, s, e = synth.gen_numpy(n, mean, 24, seed)
    if sort:
        o = np.lexsort((s, k)); k, s, e = k[o], s[o], e[o]
    contig = pa.DictionaryArray.from_arrays(pa.array(k.astype(np.int32)), names).cast(pa.string())
    return pa.table({"contig": contig, "pos_start": pa.array(s), "pos_end": pa.array(e)})


build = tab(nb, 1000, 0x5EED0004)
ses = br.Session(0)
for sort in (False, True):
    probe = tab(npb, 150, 0x5EED0005, sort)
    batches = probe.to_batches(max_chunksize=batch)
    js = ses.join_stream(build, coalesce_rows=coalesce)
    t0 = time.perf_counter(); pairs = 0; groups = 0
    for b in batches:
        for r in js.push(b):
            pairs += len(r["build_idx"]); groups += 1
    for r in js.finish():
        pairs += len(r["build_idx"]); groups += 1
    dt = time.perf_counter() - t0
    js.close()
    print(f"{'sorted' if sort else 'random'} probe rows: {len(batches)} batches of {batch} rows, {groups} groups -> {pairs} pairs in {dt*1e3:.1f} ms = "
          f"{npb/dt/1e6:.1f} M probe rows/s, {pairs/dt/1e6:.1f} M pairs/s (one host thread encodes the batches)", flush=True)
    del probe, batches
