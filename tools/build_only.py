"""Index builds of the 1M-row build side, repeated (for kernel traces): KIND=count|coverage|overlap|nearest"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import pyivx, synth
kind = {"count": pyivx.KIND_COUNT, "coverage": pyivx.KIND_COVERAGE, "overlap": pyivx.KIND_OVERLAP, "nearest": pyivx.KIND_NEAREST}[os.environ.get("KIND", "coverage")]
nb = int(os.environ.get("NB", 1_000_000))
ctx = pyivx.Ctx(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
bk, bs, be = synth.gen_torch(nb, 1000, 24, 0x5EED0004, "cuda:0")
for r in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ix = ctx.build(kind, bk, bs, be, n_keys=24); torch.cuda.synchronize()
    print(f"build {os.environ.get('KIND', 'coverage')} {nb}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
    ix.free()
