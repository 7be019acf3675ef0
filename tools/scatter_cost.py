import torch, time
dev='cuda:0'
n=100_000_000
perm=torch.randperm(n, device=dev)
vals=torch.arange(n, device=dev, dtype=torch.int64)
out=torch.empty(n, device=dev, dtype=torch.int64)
for dt,name in ((torch.int64,'i64'),(torch.int32,'i32')):
    v=vals.to(dt); o=out.to(dt)
    torch.cuda.synchronize()
    for r in range(3):
        t0=time.perf_counter(); o[perm]=v; torch.cuda.synchronize(); t1=time.perf_counter()
        print(name,'scatter 100M random', round((t1-t0)*1e3,3),'ms')
    for r in range(2):
        t0=time.perf_counter(); w=v[perm]; torch.cuda.synchronize(); t1=time.perf_counter()
        print(name,'gather 100M random', round((t1-t0)*1e3,3),'ms')
