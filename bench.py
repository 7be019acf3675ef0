#!/usr/bin/env python3
"""bench.py -- overlap-pairs/s + probe-rows/s of the interval join hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: build the binned overlap index from the build-side
(key,start,end) columns, then stream the probe side through it and write the
(build_idx, probe_idx) pairs -- what IntervalJoinExec does between
collect_left_input and compute::take.

Workload (BASELINE.json metric: IntervalJoinExec 100M x 1M): per GPU 100M probe
rows (mean length 150) against 1M build rows (mean length 1000) over the 24
hg38 contigs, uniform random, unsorted.

Multi-GPU (one process per GPU, launched by torch.distributed.run):
  --scaling weak   (default) every rank owns a full-size partition of its own
                   (DataFusion partition <-> GPU; contig groups never span ranks),
                   no data-path collective: per-GPU work is fixed as N grows.
  --scaling strong ONE 100M x 1M job (BASELINE config 3's shape): whole contigs go
                   to ranks by greedy LPT, every rank joins its contigs, and with
                   --gather (default in this mode) the per-rank pair buffers are
                   all-gathered (RCCL all-gatherv, exact sizes) inside the step.
In a weak run with N > 1 the strong-scaling job is timed as well, after the
official loop, and reported under "strong" (a labelled extra, not `value`).

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "datafusion-bio-functions_amd"))

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# HBM bytes per step measured with rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
# MI355X_MICROARCH.md "HBM"), summed over the step's kernels; valid only for the kernels named next to it, so the
# figure is dropped from the line when the pipeline's kernels change (PIPELINE below is what the library runs today)
PIPELINE = "k_part_onepass + k_probe_regions<fill, paged>"
PMC_TRAFFIC = {"join_100Mx1M_24contigs": {"bytes": None, "source": None, "pipeline": PIPELINE}}
_pmc = os.path.join(ROOT, "profiles", "r2_bench_traffic.json")
if os.path.exists(_pmc):
    with open(_pmc) as _f:
        PMC_TRAFFIC.update(json.load(_f))

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
    ap.add_argument("--gather", action="store_true", help="all-gather the per-rank pair buffers (RCCL all-gatherv) inside the step (default with --scaling strong)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo only for rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--probe-rows", type=int, default=0, help="override the probe rows per GPU (rehearsals only; 0 = the workload's size)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-extras", action="store_true", help="skip the two-call, end-to-end and strong-scaling extras")
    ap.add_argument("--check-union", action="store_true", help="strong + gather: rank 0 checks the gathered pair set against a single-rank join (rehearsals)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import pyivx
    import shard
    import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
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

    n_probe_w, n_build_w, n_contigs, cfg = WORKLOADS[args.workload]
    if args.probe_rows:
        n_probe_w = args.probe_rows

    ctx = pyivx.Ctx(local_rank)                                # raises if the HIP library / gfx950 is missing
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def make_job(scaling):
        """-> this rank's build / probe columns, and (strong) the global row numbers of its rows"""
        if scaling == "weak":
            seed = 0x5EED0000 + 2 * cfg + (rank << 8)             # rank salt: every partition is different data
            b = synth.gen_torch(n_build_w, 1000, n_contigs, seed + 0, dev)
            p = synth.gen_torch(n_probe_w, 150, n_contigs, seed + 1, dev)
            return b, p, None, None
        # one fixed job; whole contigs go to ranks (greedy LPT on probe+build rows), no row crosses ranks
        seed = 0x5EED0000 + 2 * cfg
        bk, bs, be = synth.gen_torch(n_build_w, 1000, n_contigs, seed + 0, dev)
        pk, ps, pe = synth.gen_torch(n_probe_w, 150, n_contigs, seed + 1, dev)
        if n_contigs >= world:
            w = (torch.bincount(bk, minlength=n_contigs) + torch.bincount(pk, minlength=n_contigs)).cpu().numpy()
            mine = torch.from_numpy(shard.assign_keys_lpt(w, world) == rank).to(dev)
            mb, mp = mine[bk.long()], mine[pk.long()]
            rows_b, rows_p = torch.nonzero(mb).squeeze(1).to(torch.int32), torch.nonzero(mp).squeeze(1).to(torch.int32)
            b = (bk[mb].contiguous(), bs[mb].contiguous(), be[mb].contiguous())
            p = (pk[mp].contiguous(), ps[mp].contiguous(), pe[mp].contiguous())
        else:
            # fewer contigs than ranks (the single-contig workload): build side replicated, probe rows split
            # evenly, still no exchange (SURVEY 8e fallback)
            lo, hi = n_probe_w * rank // world, n_probe_w * (rank + 1) // world
            rows_b = torch.arange(n_build_w, dtype=torch.int32, device=dev)
            rows_p = torch.arange(lo, hi, dtype=torch.int32, device=dev)
            b = (bk, bs, be)
            p = (pk[lo:hi].contiguous(), ps[lo:hi].contiguous(), pe[lo:hi].contiguous())
        return b, p, rows_b, rows_p

    def run_job(scaling, gather, steps, warmup):
        """-> dict(value, rows_per_s, ms_per_step, pairs, n_probe, n_build, probe_ms, build_ms) of K timed steps"""
        (bk, bs, be), (pk, ps, pe), rows_b, rows_p = make_job(scaling)
        n_build, n_probe = int(bk.numel()), int(pk.numel())
        torch.cuda.synchronize()
        # sizing pass (untimed): how many pairs this partition yields -> output capacity
        ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=n_contigs)
        pairs = ctx.overlap_count(ix, pk, ps, pe)
        ix.free()
        cap = pairs + 1024
        ob = torch.empty(cap, dtype=torch.int32, device=dev)
        op = torch.empty(cap, dtype=torch.int32, device=dev)
        if scaling == "weak" and n_contigs == 24 and not args.probe_rows:
            expect = n_probe * n_build * 1149.0 / sum(synth.HG38[:n_contigs])    # uniform-data expectation
            if abs(pairs - expect) > 0.01 * expect:
                raise SystemExit(f"pair count {pairs} is not within 1% of the uniform expectation {expect:.0f}")
        probe_ms, build_ms = [], []
        gathered = [None]

        def step():
            ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=n_contigs)
            build_ms.append(ctx.last_kernel_ms())
            b, p = ctx.overlap_fill(ix, pk, ps, pe, out=(ob, op))
            probe_ms.append(ctx.last_kernel_ms())
            assert b.numel() == pairs
            ix.free()
            if gather and dist is not None:
                if rows_b is not None:                          # strong: pairs in the job's global row numbers
                    gathered[0] = shard.allgatherv(dist, (rows_b[ob[:pairs].long()], rows_p[op[:pairs].long()]))
                else:
                    gathered[0] = shard.allgatherv(dist, (ob[:pairs], op[:pairs]))

        for _ in range(warmup):
            step()
        probe_ms.clear(); build_ms.clear()
        gc.collect(); gc.disable()                              # no collector pauses inside the 10-odd milliseconds being timed
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        gc.enable()
        tot_pairs, tot_rows = pairs, n_probe
        if dist is not None:
            elapsed = shard.max_over_ranks(dist, elapsed, dev)
            c = torch.tensor([pairs, n_probe], dtype=torch.int64, device=shard.comm_device(dist, dev))
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            tot_pairs, tot_rows = int(c[0]), int(c[1])
        return dict(value=tot_pairs * steps / elapsed, rows_per_s=tot_rows * steps / elapsed, ms_per_step=1e3 * elapsed / steps,
                    pairs=pairs, tot_pairs=tot_pairs, n_probe=n_probe, n_build=n_build, probe_ms=float(np.mean(probe_ms)),
                    build_ms=float(np.mean(build_ms)), cols=((bk, bs, be), (pk, ps, pe)), out=(ob, op), gathered=gathered[0])

    gather = (args.gather or args.scaling == "strong") and not args.no_gather
    r = run_job(args.scaling, gather, args.steps, args.warmup)
    (bk, bs, be), (pk, ps, pe) = r["cols"]
    ob, op = r["out"]
    pairs, n_probe, n_build = r["pairs"], r["n_probe"], r["n_build"]

    out = None
    if rank == 0:
        alg_bytes = 12 * n_probe + 12 * n_build + 8 * pairs          # every input read once, every pair written once
        step_ms = r["build_ms"] + r["probe_ms"]                      # device time of build + probe (HIP events on the launch stream)
        achieved = alg_bytes / (step_ms * 1e-3) / 1e9
        tr = PMC_TRAFFIC.get(args.workload, {})
        tr_ok = bool(tr.get("bytes")) and tr.get("pipeline") == PIPELINE and args.scaling == "weak" and not args.probe_rows
        out = {
            "metric": "overlap-pairs/sec + probe-rows/sec, IntervalJoinExec 100Mx1M",
            "value": r["value"], "unit": "overlap-pairs/s", "probe_rows_per_s": r["rows_per_s"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": args.workload, "probe_rows_per_gpu": n_probe, "build_rows_per_gpu": n_build,
                       "contigs": n_contigs, "pairs_per_gpu": pairs, "parallelism": f"partition-per-gpu x{world}" if args.scaling == "weak" else f"contigs sharded by LPT over {world} ranks",
                       "gather": bool(gather and world > 1)},
            # the dominant cost is the probe pipeline; the fraction is taken over build + probe device time, the metric's
            # definition; probe_only_* are the same figures without the index build
            "roofline": {"bound": "hbm", "kernel": f"index build (7 kernels) + {PIPELINE}", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": tr.get("bytes") if tr_ok else None, "traffic_source": tr.get("source") if tr_ok else None,
                         "kernel_ms": step_ms, "algorithmic_bytes": alg_bytes,
                         "build_ms": r["build_ms"], "probe_ms": r["probe_ms"],
                         "probe_only_achieved": alg_bytes / (r["probe_ms"] * 1e-3) / 1e9,
                         "probe_only_frac": alg_bytes / (r["probe_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        }
        if tr_ok:   # real HBM traffic of the step (PMC) over its measured time: the bandwidth utilisation rocprof sees
            out["roofline"]["traffic_frac"] = tr["bytes"] / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS

    # ---- extras (not `value`): what a caller that does not know the pair count pays, and the PCIe-inclusive call
    if rank == 0 and world == 1 and not args.no_extras:
      try:
        tiny = (np.zeros(4, np.uint32), np.arange(4, dtype=np.int64), np.arange(4, dtype=np.int64) + 1)
        ix = ctx.build(pyivx.KIND_OVERLAP, bk, bs, be, n_keys=n_contigs)
        best = 1e9
        for _ in range(5):      # count (sizes the buffers) + fill (reuses the count call's routing), device-resident
            ctx.merge(*tiny, n_keys=1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tot = ctx.overlap_count(ix, pk, ps, pe)
            b, p = ctx.overlap_fill(ix, pk, ps, pe, out=(ob, op))
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            assert tot == pairs and b.numel() == pairs
        out["two_call_ms"] = 1e3 * best
        # host-resident columns in (pageable, as Arrow buffers are), pairs out into host buffers the caller already owns
        hp = (pk.cpu().numpy().view(np.uint32), ps.cpu().numpy(), pe.cpu().numpy())
        hob, hop = np.zeros(pairs + 1024, np.uint32), np.zeros(pairs + 1024, np.uint32)
        best = 1e9
        for _ in range(3):
            ctx.merge(*tiny, n_keys=1)
            t0 = time.perf_counter()
            b, p = ctx.overlap_fill(ix, *hp, out=(hob, hop))
            best = min(best, time.perf_counter() - t0)
            assert len(b) == pairs
        out["e2e"] = {"ms": 1e3 * best, "pairs_per_s": pairs / best, "probe_rows_per_s": n_probe / best,
                      "h2d_bytes": 12 * n_probe, "d2h_bytes": 8 * pairs, "link_gbps": (12 * n_probe + 8 * pairs) / best / 1e9,
                      "what": "ivx_probe_overlap_fill with IVX_MEM_HOST: pageable host columns in, pairs out to host buffers (index already built); the library cuts the batch into 4 chunks and copies the pairs of one back while the next uploads"}
        ix.free()
      except Exception as ex:                                    # noqa: BLE001 -- extras never cost the official line
        out["extras_error"] = f"{type(ex).__name__}: {ex}"

    # ---- CPU baseline beside it + parity of the timed GPU result against the oracle's pair set
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import oracle as orc                     # checker, used as the timed CPU baseline only
        ns = min(args.cpu_sample, n_probe)
        hb = (bk.cpu().numpy().view(np.uint32), bs.cpu().numpy(), be.cpu().numpy())
        hp = (pk[:ns].cpu().numpy().view(np.uint32), ps[:ns].cpu().numpy(), pe[:ns].cpu().numpy())
        cores = min(16, len(os.sched_getaffinity(0)))      # the 1-GPU box's CPU share
        orc.lib()
        c0 = time.perf_counter()
        cb, cp = orc.join_single(*hb, *hp, threads=cores)
        cpu_s = time.perf_counter() - c0
        # parity of the timed GPU result: the pair MULTISET over the sampled rows equals the oracle's, bit for bit
        sel = op[:pairs] < ns
        gk = torch.sort((ob[:pairs][sel].long() << 32) | op[:pairs][sel].long()).values.cpu().numpy().view(np.uint64)
        if not np.array_equal(gk, orc.pair_keys(cb, cp)):
            raise SystemExit("GPU pair set differs from the oracle's")
        out["parity"] = f"pair multiset of the first {ns} probe rows == oracle ({len(cb)} pairs, sorted build<<32|probe keys compared)"
        out["cpu_baseline"] = {"value": len(cb) / cpu_s, "unit": "overlap-pairs/s", "cores": cores, "kind": "port",
                               "probe_rows_per_s": ns / cpu_s,
                               "sample": f"first {ns} probe rows x all {n_build} build rows, index build + single-walk probe with per-thread "
                                         f"pair buffers + concatenation, {cpu_s:.2f} s (oracle/ivx_oracle.c orc_join_single_run, OpenMP)"}

    # ---- rehearsal check: the gathered union of a strong-scaling job equals the single-rank join
    if args.check_union and args.scaling == "strong" and gather and world > 1:
        (gb, gp), sizes = r["gathered"]
        if rank == 0:
            ab, ap_ = synth.gen_torch(n_build_w, 1000, n_contigs, 0x5EED0000 + 2 * cfg, dev), synth.gen_torch(n_probe_w, 150, n_contigs, 0x5EED0000 + 2 * cfg + 1, dev)
            ix = ctx.build(pyivx.KIND_OVERLAP, *ab, n_keys=n_contigs)
            wb, wp = ctx.overlap_fill(ix, *ap_)
            want = torch.sort((wb.long() << 32) | wp.long()).values
            got = torch.sort((gb.long() << 32) | gp.long()).values
            ok = want.numel() == got.numel() == sum(sizes) and bool((want == got).all())
            out["union_check"] = "gathered pair set == single-rank join" if ok else "MISMATCH"
            ix.free()
            if not ok:
                print(json.dumps(out), flush=True)
                raise SystemExit("gathered pair set differs from the single-rank join")

    # ---- labelled extra of a multi-GPU weak run: the ONE-job (strong-scaling) form of the same workload with the all-gatherv.
    #      Never at the price of the official line: any failure here is recorded, not raised (every rank takes the same path:
    #      the flag is agreed on with an all-reduce before the collective part starts and after it ends)
    if world > 1 and args.scaling == "weak" and not args.no_extras:
        del r, ob, op, bk, bs, be, pk, ps, pe
        torch.cuda.empty_cache()
        err = None
        try:
            s = run_job("strong", not args.no_gather, args.steps, args.warmup)
        except Exception as ex:                                   # noqa: BLE001 -- reported in the JSON line
            err = f"{type(ex).__name__}: {ex}"
        if rank == 0:
            if err is None:
                out["strong"] = {"value": s["value"], "unit": "overlap-pairs/s", "ms_per_step": s["ms_per_step"], "probe_rows_per_s": s["rows_per_s"],
                                 "pairs_total": s["tot_pairs"], "probe_rows_rank0": s["n_probe"], "gather": not args.no_gather,
                                 "what": f"one {n_probe_w} x {n_build_w} job, {n_contigs} contigs sharded by LPT over {world} ranks, all-gatherv of the pair buffers in the step"}
            else:
                out["strong"] = {"error": err}

    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
