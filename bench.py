#!/usr/bin/env python3
"""bench.py -- overlap-pairs/s + probe-rows/s of the interval join hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: build the binned overlap index from the build-side
(key,start,end) columns, then stream the probe side through it and write the
(build_idx, probe_idx) pairs -- what IntervalJoinExec does between
collect_left_input and compute::take.

Workload (BASELINE.json metric: IntervalJoinExec 100M x 1M): per GPU 100M probe
rows (mean length 150) against 1M build rows (mean length 1000) over the 24
hg38 contigs, uniform random, unsorted.  With N GPUs every rank processes its
own partition of that size (DataFusion partition <-> GPU, contig groups never
span ranks), no data-path collective: weak scaling.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# HBM bytes per probe call from rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
# MI355X_MICROARCH.md "HBM"), summed over the pipeline's kernels: profiles/r1_n_regions_pipeline_pmc.txt (same as r1_m)
PMC_TRAFFIC_BYTES = {"join_100Mx1M_24contigs": 4.898e9}

WORKLOADS = {
    # name: (probe rows, build rows, contigs, config id in BASELINE.json.configs)
    "join_100Mx1M_24contigs": (100_000_000, 1_000_000, 24, 2),
    "join_10Mx100k_1contig": (10_000_000, 100_000, 1, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="join_100Mx1M_24contigs", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="probe rows timed on the CPU baseline (0 = skip)")
    ap.add_argument("--gather", action="store_true", help="also all-gather the per-rank pair buffers (RCCL all-gatherv) inside the step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo only for rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--probe-rows", type=int, default=0, help="override the probe rows per GPU (rehearsals only; 0 = the workload's size)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns a full-size partition; strong: ONE job, contigs sharded over ranks by LPT")
    args = ap.parse_args()

    import numpy as np
    import torch
    import pyivx
    import shard
    import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    n_probe, n_build, n_contigs, cfg = WORKLOADS[args.workload]
    if args.probe_rows:
        n_probe = args.probe_rows
    if args.scaling == "weak":
        seed = 0x5EED0000 + 2 * cfg + (rank << 8)             # rank salt: every partition is different data
        bk, bs, be = synth.gen_torch(n_build, 1000, n_contigs, seed + 0, dev)
        pk, ps, pe = synth.gen_torch(n_probe, 150, n_contigs, seed + 1, dev)
    else:
        # one fixed job; whole contigs go to ranks (greedy LPT on probe+build rows), no row crosses ranks
        seed = 0x5EED0000 + 2 * cfg
        bk, bs, be = synth.gen_torch(n_build, 1000, n_contigs, seed + 0, dev)
        pk, ps, pe = synth.gen_torch(n_probe, 150, n_contigs, seed + 1, dev)
        if n_contigs >= world:
            w = (torch.bincount(bk, minlength=n_contigs) + torch.bincount(pk, minlength=n_contigs)).cpu().numpy()
            mine = torch.from_numpy(shard.assign_keys_lpt(w, world) == rank).to(dev)
            mb, mp = mine[bk.long()], mine[pk.long()]
            bk, bs, be = bk[mb].contiguous(), bs[mb].contiguous(), be[mb].contiguous()
            pk, ps, pe = pk[mp].contiguous(), ps[mp].contiguous(), pe[mp].contiguous()
        else:
            # fewer contigs than ranks (the single-contig workload): build side replicated, probe rows split
            # evenly, still no exchange (SURVEY 8e fallback)
            lo, hi = n_probe * rank // world, n_probe * (rank + 1) // world
            pk, ps, pe = pk[lo:hi].contiguous(), ps[lo:hi].contiguous(), pe[lo:hi].contiguous()
        n_build, n_probe = int(bk.numel()), int(pk.numel())
    torch.cuda.synchronize()

    ctx = pyivx.Ctx(local_rank)                                # raises if the HIP library / gfx950 is missing
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    # sizing pass (untimed): how many pairs this partition yields -> output capacity
    ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=n_contigs)
    pairs = ctx.overlap_count(ix, pk, ps, pe)
    ix.free()
    cap = pairs + 1024
    ob = torch.empty(cap, dtype=torch.int32, device=dev)
    op = torch.empty(cap, dtype=torch.int32, device=dev)
    expect = n_probe * n_build * 1149.0 / sum(synth.HG38[:n_contigs])    # uniform-data expectation
    if n_contigs == 24 and args.scaling == "weak" and not args.probe_rows and abs(pairs - expect) > 0.01 * expect:
        raise SystemExit(f"pair count {pairs} is not within 1% of the uniform expectation {expect:.0f}")

    probe_ms = []
    build_ms = []

    def step():
        ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=n_contigs)
        build_ms.append(ctx.last_kernel_ms())
        b, p = ctx.overlap_fill(ix, pk, ps, pe, out=(ob, op))
        probe_ms.append(ctx.last_kernel_ms())
        assert b.numel() == pairs
        ix.free()
        if args.gather and dist is not None:
            shard.allgatherv(dist, (ob[:pairs], op[:pairs]))

    for _ in range(args.warmup):
        step()
    probe_ms.clear(); build_ms.clear()
    import gc
    gc.collect(); gc.disable()                                  # no collector pauses inside the 10-odd milliseconds being timed
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    tot_pairs, tot_rows = pairs, n_probe
    if dist is not None:
        elapsed = shard.max_over_ranks(dist, elapsed, dev)
        c = torch.tensor([pairs, n_probe], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        tot_pairs, tot_rows = int(c[0]), int(c[1])

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        kern_ms = float(np.mean(probe_ms))
        alg_bytes = 12 * n_probe + 12 * n_build + 8 * pairs          # SURVEY.md 8(d), per launch of the probe kernel
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "overlap-pairs/sec + probe-rows/sec, IntervalJoinExec 100Mx1M",
            "value": tot_pairs * args.steps / elapsed,
            "unit": "overlap-pairs/s",
            "probe_rows_per_s": tot_rows * args.steps / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": args.workload, "probe_rows_per_gpu": n_probe, "build_rows_per_gpu": n_build,
                       "contigs": n_contigs, "pairs_per_gpu": pairs, "parallelism": f"partition-per-gpu x{world}",
                       "gather": bool(args.gather)},
            "roofline": {"bound": "hbm", "kernel": "overlap probe pipeline: k_part_hist + k_part_scatter + k_probe_regions<fill>", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": PMC_TRAFFIC_BYTES.get(args.workload) if args.scaling == "weak" and not args.probe_rows else None,
                         "kernel_ms": kern_ms, "algorithmic_bytes": alg_bytes,
                         "build_ms": float(np.mean(build_ms))},
        }
        tr = out["roofline"]["traffic"]
        # real HBM traffic of the pipeline (PMC) over its measured time: the bandwidth utilisation rocprof sees
        out["roofline"]["traffic_frac"] = (tr / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if tr else None
        if world == 1 and args.cpu_sample > 0:
            from oracle import oracle as orc                     # checker used as the timed CPU baseline only
            ns = min(args.cpu_sample, n_probe)
            hb = (bk.cpu().numpy().view(np.uint32), bs.cpu().numpy(), be.cpu().numpy())
            hp = (pk[:ns].cpu().numpy().view(np.uint32), ps[:ns].cpu().numpy(), pe[:ns].cpu().numpy())
            cores = min(16, len(os.sched_getaffinity(0)))      # the 1-GPU box's CPU share
            orc.lib()
            c0 = time.perf_counter()
            cb, cp = orc.join(*hb, *hp, threads=cores)
            cpu_s = time.perf_counter() - c0
            # the sample is also a parity check of the timed GPU result
            sel = op[:pairs] < ns
            gpu_pairs_in_sample = int(sel.sum())
            assert gpu_pairs_in_sample == len(cb), (gpu_pairs_in_sample, len(cb))
            out["cpu_baseline"] = {"value": len(cb) / cpu_s, "unit": "overlap-pairs/s", "cores": cores, "kind": "port",
                                   "probe_rows_per_s": ns / cpu_s,
                                   "sample": f"first {ns} probe rows x all {n_build} build rows, index build + probe + pair "
                                             f"materialisation, {cpu_s:.2f} s (oracle/ivx_oracle.c orc_join_tree, OpenMP)"}
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
