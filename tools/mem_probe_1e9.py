"""Diagnostic: host / device memory of the 1e9-row subtract scenario, phase by phase (gpurun_out/mem_1e9.log)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))
import torch, pyivx, synth

def mem(tag):
    rss = [l for l in open("/proc/self/status") if l.startswith(("VmRSS", "VmHWM"))]
    cg = ""
    for p in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
        if os.path.exists(p):
            cg = f"cgroup {int(open(p).read()) / 2**30:.1f} GiB"
    free, tot = torch.cuda.mem_get_info()
    mi = {l.split(":")[0]: l.split()[1] for l in open("/proc/meminfo") if l.startswith(("MemAvailable", "Shmem:", "Cached", "Mapped"))}
    print(f"[{time.time() - T0:6.1f}s] {tag}: {' '.join(x.strip() for x in rss)} {cg} dev_used {(tot - free) / 2**30:.1f} GiB "
          f"torch_alloc {torch.cuda.memory_allocated() / 2**30:.1f} reserved {torch.cuda.memory_reserved() / 2**30:.1f} meminfo {mi}", flush=True)

T0 = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ctx = pyivx.Ctx(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
mem("start")
def gen64(n, mean, seed):
    k, s, e = synth.gen_torch(n, mean, 24, seed, "cuda:0")
    s64, e64 = s.long(), e.long() + 1
    return k, s64, e64
lk, ls, le = gen64(n, 20, 0x5EED0008); mem("left generated")
rk, rs, re = gen64(n // 10, 8, 0x5EED0009); mem("right generated")
torch.cuda.synchronize()
ok, os_, oe, on = ctx.merge(lk, ls, le, n_keys=24); torch.cuda.synchronize(); mem(f"merge -> {ok.numel()} runs, {ctx.last_kernel_ms():.1f} ms")
del ok, os_, oe, on; torch.cuda.empty_cache(); mem("merge outputs dropped")
fk, fs, fe, frow = ctx.subtract(lk, ls, le, rk, rs, re, n_keys=24); torch.cuda.synchronize(); mem(f"subtract -> {fk.numel()} fragments, fill {ctx.last_kernel_ms():.1f} ms")
