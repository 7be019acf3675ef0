#!/usr/bin/env python3
"""bench.py -- overlap-pairs/s + probe-rows/s of the interval join hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: build the binned overlap index from the build-side
(key,start,end) columns, then stream the probe side through it and write the
(build_idx, probe_idx) pairs -- what IntervalJoinExec does between
collect_left_input and compute::take.

Workload (BASELINE.json metric: IntervalJoinExec 100M x 1M): per GPU 100M probe
rows (mean length 150) against 1M build rows (mean length 1000) over the 24
hg38 contigs, uniform random, unsorted.  The other BASELINE configs have workloads
of their own (--workload): count_overlaps + coverage 100M x 1M (C3), nearest
50M x 50M (C4), merge + subtract on 200 M or 10^9 intervals (C5); their steps are
the same build + probe / sort + sweep calls the operators make.

Multi-GPU (one process per GPU, launched by torch.distributed.run):
  --scaling weak   (default) every rank owns a full-size partition of its own
                   (DataFusion partition <-> GPU; contig groups never span ranks),
                   no data-path collective: per-GPU work is fixed as N grows.
  --scaling strong ONE job (the BASELINE configs' shape): whole contigs go
                   to ranks by greedy LPT (every rank generates only its own rows),
                   every rank runs the operator on its contigs, and with
                   --gather (default in this mode) the per-rank results are
                   exchanged (RCCL all-gatherv, exact sizes) inside the step:
                   pair buffers for the join, (row, value) lists scattered back to
                   input order for count / coverage / nearest, per-contig pieces
                   spliced in key order for merge / subtract (sharded.py).
In a weak run of the join with N > 1 the strong-scaling job is timed as well, after
the official loop, and reported under "strong" (a labelled extra, not `value`).
At N = 1 the default run also times every other operator after the official line
and reports them under "ops" (labelled extras; --no-ops skips them).

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
PIPELINE = "k_part_onepass + k_fill_fast + k_fill_rest"
PMC_TRAFFIC = {"join_100Mx1M_24contigs": {"bytes": None, "source": None, "pipeline": PIPELINE}}
for _name in ("r2_bench_traffic.json", "r3_bench_traffic.json"):
    _pmc = os.path.join(ROOT, "profiles", _name)
    if os.path.exists(_pmc):
        with open(_pmc) as _f:
            PMC_TRAFFIC.update(json.load(_f))

WORKLOADS = {
    # name: kind, rows of the two sides, mean lengths, contigs, index in BASELINE.json.configs
    "join_100Mx1M_24contigs": dict(kind="join", probe=100_000_000, build=1_000_000, contigs=24, cfg=2),
    "join_10Mx100k_1contig": dict(kind="join", probe=10_000_000, build=100_000, contigs=1, cfg=1),
    "count_coverage_100Mx1M": dict(kind="count_coverage", probe=100_000_000, build=1_000_000, contigs=24, cfg=2),
    "nearest_50Mx50M": dict(kind="nearest", probe=50_000_000, build=50_000_000, contigs=24, cfg=3),
    "merge_subtract_200M": dict(kind="merge_subtract", probe=200_000_000, build=20_000_000, contigs=24, cfg=4),
    "merge_subtract_1B": dict(kind="merge_subtract", probe=1_000_000_000, build=100_000_000, contigs=24, cfg=4),
}
# algorithmic bytes per call (SURVEY.md 8d / DESIGN.md section 3): every input read once, every output written once
ALG = {
    "join": lambda n_p, n_b, n_out: 12 * n_p + 12 * n_b + 8 * n_out,
    "count_overlaps": lambda n_p, n_b, n_out=0: 12 * (n_p + n_b) + 8 * n_p,
    "coverage": lambda n_p, n_b, n_out=0: 12 * (n_p + n_b) + 8 * n_p,
    "nearest": lambda n_p, n_b, n_out=0: 12 * (n_p + n_b) + 16 * n_p,
    "merge": lambda n_in, _b, n_out: 20 * n_in + 28 * n_out,
    "subtract": lambda n_l, n_r, n_out: 20 * (n_l + n_r) + 20 * n_out,
    "cluster": lambda n_in, _b=0, n_out=0: 20 * n_in + 48 * n_in,
    "complement": lambda n_in, n_v, n_out: 20 * (n_in + n_v) + 20 * n_out,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="join_100Mx1M_24contigs", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="probe rows timed on the CPU baseline (0 = skip)")
    ap.add_argument("--gather", action="store_true", help="exchange the per-rank results (RCCL all-gatherv) inside the step (default with --scaling strong)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo only for rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses cuda:0")
    ap.add_argument("--probe-rows", type=int, default=0, help="override the probe / left rows of the job (rehearsals only; the build / right side scales along for the non-join workloads)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-extras", action="store_true", help="skip the two-call, end-to-end and strong-scaling extras")
    ap.add_argument("--no-ops", action="store_true", help="skip the per-operator block (count / coverage / nearest / merge / subtract / cluster / complement)")
    ap.add_argument("--ops-only", action="store_true", help="print the per-operator block alone (no join line)")
    ap.add_argument("--ops", default="count,coverage,nearest,merge,subtract,cluster,complement,big", help="operators of the block")
    ap.add_argument("--no-build-overlap", action="store_true", help="build the index synchronously (default: ivx_ctx_set_build_overlap -- the build's tail runs beside the probe's routing pass; the build columns stay alive for the whole step, as the contract asks)")
    ap.add_argument("--check-union", action="store_true", help="strong + gather: rank 0 checks the exchanged result against a single-rank run of the whole job (rehearsals)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import pyivx
    import shard
    import sharded
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

    W = WORKLOADS[args.workload]
    kind, n_probe_w, n_build_w, n_contigs, cfg = W["kind"], W["probe"], W["build"], W["contigs"], W["cfg"]
    if args.probe_rows:
        if kind != "join":
            n_build_w = max(1, n_build_w * args.probe_rows // n_probe_w)
        n_probe_w = args.probe_rows

    ctx = pyivx.Ctx(local_rank)                                # raises if the HIP library / gfx950 is missing
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_build_overlap(not args.no_build_overlap)
    gather = (args.gather or args.scaling == "strong") and not args.no_gather
    cores = min(16, len(os.sched_getaffinity(0)))              # the 1-GPU box's CPU share

    def lpt_keep(counts):
        """this rank's contigs of ONE job: whole contigs dealt by greedy LPT on the rows they hold"""
        return torch.from_numpy(shard.assign_keys_lpt(counts.cpu().numpy(), world) == rank).to(dev)

    def timed_loop(step, steps, warmup):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize; -> seconds (max over ranks)"""
        for _ in range(warmup):
            step()
        gc.collect(); gc.disable()                              # no collector pauses inside the milliseconds being timed
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
        if dist is not None:
            elapsed = shard.max_over_ranks(dist, elapsed, dev)
        return elapsed

    def sum_over_ranks(*vals):
        if dist is None:
            return [int(v) for v in vals]
        c = torch.tensor(list(vals), dtype=torch.int64, device=shard.comm_device(dist, dev))
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        return [int(x) for x in c.tolist()]

    # =================================================================== the join (the official line)
    def make_join_job(scaling):
        """-> this rank's build / probe columns, and (strong) the job's row numbers of its rows"""
        if scaling == "weak":
            seed = 0x5EED0000 + 2 * cfg + (rank << 8)             # rank salt: every partition is different data
            b = synth.gen_torch(n_build_w, 1000, n_contigs, seed + 0, dev)
            p = synth.gen_torch(n_probe_w, 150, n_contigs, seed + 1, dev)
            return b, p, None, None
        # one fixed job; whole contigs go to ranks (greedy LPT on probe+build rows), no row crosses ranks; a rank
        # generates its own rows only (counter-based generator: the contig draw alone gives the per-contig weights)
        seed = 0x5EED0000 + 2 * cfg
        if n_contigs >= world:
            keep = lpt_keep(synth.key_counts_torch(n_build_w, n_contigs, seed, dev) + synth.key_counts_torch(n_probe_w, n_contigs, seed + 1, dev))
            *b, rows_b = synth.gen_torch_sharded(n_build_w, 1000, n_contigs, seed + 0, dev, keep)
            *p, rows_p = synth.gen_torch_sharded(n_probe_w, 150, n_contigs, seed + 1, dev, keep)
        else:
            # fewer contigs than ranks (the single-contig workload): build side replicated, probe rows split
            # evenly, still no exchange (SURVEY 8e fallback)
            lo, hi = n_probe_w * rank // world, n_probe_w * (rank + 1) // world
            *b, rows_b = synth.gen_torch_sharded(n_build_w, 1000, n_contigs, seed + 0, dev)
            *p, rows_p = synth.gen_torch_sharded(n_probe_w, 150, n_contigs, seed + 1, dev, lo=lo, hi=hi)
        return tuple(b), tuple(p), rows_b, rows_p

    def run_join(scaling, gather, steps, warmup):
        """-> dict(value, rows_per_s, ms_per_step, pairs, n_probe, n_build, probe_ms, build_ms) of K timed steps"""
        (bk, bs, be), (pk, ps, pe), rows_b, rows_p = make_join_job(scaling)
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
                if rows_b is not None:                          # strong: pairs in the job's row numbers (one gather kernel per column)
                    gb, gp = ctx.take_fixed(rows_b, ob[:pairs], want_valid=False)[0], ctx.take_fixed(rows_p, op[:pairs], want_valid=False)[0]
                    gathered[0] = shard.allgatherv(dist, (gb, gp))
                else:
                    gathered[0] = shard.allgatherv(dist, (ob[:pairs], op[:pairs]))

        for _ in range(warmup):
            step()
        probe_ms.clear(); build_ms.clear()
        elapsed = timed_loop(step, steps, 0)
        tot_pairs, tot_rows = sum_over_ranks(pairs, n_probe)
        return dict(value=tot_pairs * steps / elapsed, rows_per_s=tot_rows * steps / elapsed, ms_per_step=1e3 * elapsed / steps,
                    pairs=pairs, tot_pairs=tot_pairs, n_probe=n_probe, n_build=n_build, probe_ms=float(np.mean(probe_ms)),
                    build_ms=float(np.mean(build_ms)), cols=((bk, bs, be), (pk, ps, pe)), out=(ob, op), gathered=gathered[0])

    # =================================================================== the other configs as workloads of their own
    def half_open64(k, s, e):
        """the sweep operators take int64 half-open rows"""
        return k, s.to(torch.int64), e.to(torch.int64) + 1

    def make_two_sided(scaling, mean_b, mean_p, seed0):
        """-> (build cols, rows_b, probe cols, rows_p, n_probe of the whole job)"""
        if scaling == "weak":
            seed = seed0 + (rank << 8)
            return synth.gen_torch(n_build_w, mean_b, n_contigs, seed, dev), None, synth.gen_torch(n_probe_w, mean_p, n_contigs, seed + 1, dev), None, n_probe_w
        keep = lpt_keep(synth.key_counts_torch(n_build_w, n_contigs, seed0, dev) + synth.key_counts_torch(n_probe_w, n_contigs, seed0 + 1, dev))
        *b, rows_b = synth.gen_torch_sharded(n_build_w, mean_b, n_contigs, seed0, dev, keep)
        *p, rows_p = synth.gen_torch_sharded(n_probe_w, mean_p, n_contigs, seed0 + 1, dev, keep)
        return tuple(b), rows_b, tuple(p), rows_p, n_probe_w

    def whole_two_sided(mean_b, mean_p, seed0):
        return synth.gen_torch(n_build_w, mean_b, n_contigs, seed0, dev), synth.gen_torch(n_probe_w, mean_p, n_contigs, seed0 + 1, dev)

    def run_other(scaling, gather, steps, warmup):
        sr = sharded.ShardedRanges(dist, ctx, n_contigs) if dist is not None else None
        ex = gather and dist is not None
        res = {}
        if kind == "count_coverage":
            seed0 = 0x5EED0000 + 2 * cfg + 0x10
            B, rows_b, P, rows_p, n_tot = make_two_sided(scaling, 1000, 150, seed0)
            n_b, n_p = int(B[0].numel()), int(P[0].numel())
            if rows_p is None and ex:
                rows_p = torch.arange(n_p, dtype=torch.int32, device=dev)       # weak + gather: the rank's own numbering

            def step():
                if ex:
                    res["out"] = sr.count_coverage(B, P, rows_p, n_tot if scaling == "strong" else n_p * world, gather=scaling == "strong")
                    if scaling == "weak":
                        res["out"] = shard.allgatherv(dist, res["out"])[0]
                else:
                    outs = []
                    for kd, fn in ((pyivx.KIND_COUNT, ctx.count_overlaps), (pyivx.KIND_COVERAGE, ctx.coverage)):
                        ix = ctx.build(kd, *B, n_keys=n_contigs)
                        outs.append(fn(ix, *P))
                        ctx.synchronize()
                        ix.free()
                    res["out"] = tuple(outs)
            alg = ALG["count_overlaps"](n_p, n_b) + ALG["coverage"](n_p, n_b)
            meta = dict(metric="probe-rows/sec, count_overlaps + coverage 100Mx1M", unit="probe-rows/s", units=n_p, n_a=n_p, n_b=n_b,
                        kernel="count index build + probe, coverage index build + probe")

            def check():
                wb, wp = whole_two_sided(1000, 150, seed0)
                want = []
                for kd, fn in ((pyivx.KIND_COUNT, ctx.count_overlaps), (pyivx.KIND_COVERAGE, ctx.coverage)):
                    ix = ctx.build(kd, *wb, n_keys=n_contigs)
                    want.append(fn(ix, *wp)); ctx.synchronize(); ix.free()
                return all(bool((g == w).all()) for g, w in zip(res["out"], want))
        elif kind == "nearest":
            seed0 = 0x5EED0000 + 2 * cfg + 0x10
            B, rows_b, P, rows_p, n_tot = make_two_sided(scaling, 1000, 150, seed0)
            n_b, n_p = int(B[0].numel()), int(P[0].numel())

            def step():
                if ex and scaling == "strong":
                    res["out"] = sr.nearest1(B, rows_b, P, rows_p, n_tot)
                else:
                    ix = ctx.build(pyivx.KIND_NEAREST, *B, n_keys=n_contigs)
                    ob, _, od = ctx.nearest(ix, *P, k=1)
                    ctx.synchronize()
                    ix.free()
                    res["out"] = shard.allgatherv(dist, (ob, od))[0] if ex else (ob, od)
            alg = ALG["nearest"](n_p, n_b)
            meta = dict(metric="probe-rows/sec, nearest k=1 50Mx50M", unit="probe-rows/s", units=n_p, n_a=n_p, n_b=n_b,
                        kernel="nearest index build + k=1 probe")

            def check():
                wb, wp = whole_two_sided(1000, 150, seed0)
                ix = ctx.build(pyivx.KIND_NEAREST, *wb, n_keys=n_contigs)
                ob, _, od = ctx.nearest(ix, *wp, k=1); ctx.synchronize(); ix.free()
                return bool((res["out"][0] == ob).all()) and bool((res["out"][1] == od).all())
        else:   # merge + subtract
            seed0 = 0x5EED0008
            if scaling == "weak":
                L = half_open64(*synth.gen_torch(n_probe_w, 20, n_contigs, seed0 + (rank << 8), dev))
                R = half_open64(*synth.gen_torch(n_build_w, 8, n_contigs, seed0 + 1 + (rank << 8), dev))
                rows_l = None
            else:
                keep = lpt_keep(synth.key_counts_torch(n_probe_w, n_contigs, seed0, dev) + synth.key_counts_torch(n_build_w, n_contigs, seed0 + 1, dev))
                *l, rows_l = synth.gen_torch_sharded(n_probe_w, 20, n_contigs, seed0, dev, keep)
                *r, _ = synth.gen_torch_sharded(n_build_w, 8, n_contigs, seed0 + 1, dev, keep)
                L, R = half_open64(*l), half_open64(*r)
                del l, r
            n_l, n_r = int(L[0].numel()), int(R[0].numel())
            sizes = {}

            def step():
                if ex and scaling == "strong":
                    res["merge"] = sr.merge(L)
                    res["subtract"] = sr.subtract(L, rows_l, R)
                else:
                    res["merge"] = ctx.merge(*L, n_keys=n_contigs)
                    res["subtract"] = ctx.subtract(*L, *R, n_keys=n_contigs)
                    if ex:
                        res["merge"] = shard.allgatherv(dist, res["merge"])[0]
                        res["subtract"] = shard.allgatherv(dist, res["subtract"])[0]
                sizes["m"], sizes["s"] = int(res["merge"][0].numel()), int(res["subtract"][0].numel())
            step()                                               # sizes of the outputs (untimed)
            if ex:
                sizes["m"], sizes["s"] = sizes["m"] // world, sizes["s"] // world      # (exchanged results hold every rank's rows)
            alg = ALG["merge"](n_l, 0, sizes["m"]) + ALG["subtract"](n_l, n_r, sizes["s"])
            meta = dict(metric="input-intervals/sec, merge + subtract", unit="intervals/s", units=n_l, n_a=n_l, n_b=n_r,
                        kernel="radix sort + merge scans; two sorts + subtract count / fill")

            def check():
                wl = half_open64(*synth.gen_torch(n_probe_w, 20, n_contigs, seed0, dev))
                wr = half_open64(*synth.gen_torch(n_build_w, 8, n_contigs, seed0 + 1, dev))
                wm = ctx.merge(*wl, n_keys=n_contigs)
                ok = all(bool((g == w).all()) for g, w in zip(res["merge"], wm))
                ws = ctx.subtract(*wl, *wr, n_keys=n_contigs)
                return ok and all(g.numel() == w.numel() and bool((g == w).all()) for g, w in zip(res["subtract"], ws))

        ctx.acc_ms = 0.0
        for _ in range(warmup):
            step()
        ctx.acc_ms = 0.0
        elapsed = timed_loop(step, steps, 0)
        kernel_ms = ctx.acc_ms / steps
        ctx.acc_ms = None
        (tot_units,) = sum_over_ranks(meta["units"])
        out = None
        if rank == 0:
            achieved = alg / (kernel_ms * 1e-3) / 1e9
            out = {"metric": meta["metric"], "value": tot_units * steps / elapsed, "unit": meta["unit"], "n_gpus": world, "steps": steps,
                   "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
                   "dtype": "int64" if kind == "merge_subtract" else "int32", "data": "synthetic",
                   "config": {"workload": args.workload, "rows_a_rank0": meta["n_a"], "rows_b_rank0": meta["n_b"], "contigs": n_contigs,
                              "parallelism": f"partition-per-gpu x{world}" if scaling == "weak" else f"contigs sharded by LPT over {world} ranks",
                              "gather": bool(ex)},
                   "roofline": {"bound": "hbm", "kernel": meta["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel_ms": kernel_ms, "algorithmic_bytes": alg}}
        if args.check_union and scaling == "strong" and ex:
            ok = check() if rank == 0 else True
            if rank == 0:
                out["union_check"] = "exchanged result == single-rank run of the whole job" if ok else "MISMATCH"
                if not ok:
                    print(json.dumps(out), flush=True)
                    raise SystemExit("sharded result differs from the single-rank run")
        return out

    # =================================================================== per-operator block (N = 1 extras)
    def ops_block():
        """Every other operator of the path at BASELINE size, device-resident, timed by THIS run: wall ms per call (best of
        the repetitions, synchronize on both sides), device ms of its kernels (HIP events on the launch stream), the
        SURVEY 8(d) algorithmic bytes and the roofline fraction they give, and the oracle timed on a stated sample."""
        from oracle import oracle as orc                     # checker, used as the timed CPU baseline only
        orc.lib()
        which = set(args.ops.split(","))
        ops = {}

        def timed(fn, reps=3):
            best, kms, out = 1e9, 0.0, None
            for _ in range(reps):
                out = None
                torch.cuda.synchronize(); ctx.acc_ms = 0.0
                t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if dt < best:
                    best, kms = dt, ctx.acc_ms
            ctx.acc_ms = None
            return best, kms, out

        def entry(name, secs, kms, alg, rows, extra=None, cpu=None):
            e = {"ms": 1e3 * secs, "kernel_ms": kms, "algorithmic_bytes": alg, "achieved_gbps": alg / secs / 1e9,
                 "frac": alg / secs / 1e9 / HBM_PEAK_GBPS, "rows_per_s": rows / secs}
            if extra:
                e.update(extra)
            if cpu:
                e["cpu"] = cpu
            ops[name] = e

        def cpu_fig(fn, rows, sample, threads, keep=None):
            t0 = time.perf_counter(); res = fn(); dt = time.perf_counter() - t0
            if keep is not None:
                keep.append(res)
            return {"rows_per_s": rows / dt, "seconds": dt, "cores": threads, "kind": "port", "sample": sample}

        def host32(k, s, e, n=None):
            return k[:n].cpu().numpy().view(np.uint32), s[:n].cpu().numpy(), e[:n].cpu().numpy()

        if which & {"count", "coverage"}:
            nb, npr = 1_000_000, 100_000_000
            bk, bs, be = synth.gen_torch(nb, 1000, 24, 0x5EED0004, dev)
            pk, ps, pe = synth.gen_torch(npr, 150, 24, 0x5EED0005, dev)
            ns = 10_000_000
            hb, hp = host32(bk, bs, be), host32(pk, ps, pe, ns)
            for kd, name, fn, ofn in ((pyivx.KIND_COUNT, "count", ctx.count_overlaps, orc.count_overlaps), (pyivx.KIND_COVERAGE, "coverage", ctx.coverage, orc.coverage)):
                if name not in which:
                    continue
                tb, kb, ix = timed(lambda: ctx.build(kd, bk, bs, be, n_keys=24))
                tp, kp, out = timed(lambda: fn(ix, pk, ps, pe))
                want = []
                cpu = cpu_fig(lambda: ofn(*hb, *hp, threads=cores), ns, f"first {ns} probe rows x all {nb} build rows (index build + probe)", cores, want)
                if not np.array_equal(out[:ns].cpu().numpy(), want[0]):
                    raise SystemExit(f"{name}: GPU column differs from the oracle's on the sampled rows")
                full = {"count": "count_overlaps_100Mx1M", "coverage": "coverage_100Mx1M"}[name]
                entry(full, tb + tp, kb + kp, ALG["count_overlaps"](npr, nb), npr, {"build_ms": 1e3 * tb, "probe_ms": 1e3 * tp, "probe_kernel_ms": kp,
                      "parity": f"first {ns} rows == oracle"}, cpu)
                ix.free(); del out
            del bk, bs, be, pk, ps, pe
        if "nearest" in which:
            nb = npr = 50_000_000
            bk, bs, be = synth.gen_torch(nb, 1000, 24, 0x5EED0006, dev)
            pk, ps, pe = synth.gen_torch(npr, 150, 24, 0x5EED0007, dev)
            tb, kb, ix = timed(lambda: ctx.build(pyivx.KIND_NEAREST, bk, bs, be, n_keys=24), reps=2)
            tp, kp, out = timed(lambda: ctx.nearest(ix, pk, ps, pe, k=1))
            ns = 20_000_000                                      # the CPU sample is a smaller (sparser) job of the same shape
            hb, hp = host32(bk, bs, be, ns), host32(pk, ps, pe, ns)
            cpu = cpu_fig(lambda: orc.nearest1(*hb, *hp, threads=cores), ns, f"first {ns} build rows x first {ns} probe rows (index build + k=1 search)", cores)
            entry("nearest_50Mx50M", tb + tp, kb + kp, ALG["nearest"](npr, nb), npr, {"build_ms": 1e3 * tb, "probe_ms": 1e3 * tp, "build_kernel_ms": kb, "probe_kernel_ms": kp}, cpu)
            ix.free(); del out, bk, bs, be, pk, ps, pe
        if which & {"merge", "subtract", "cluster", "complement"}:
            n = 200_000_000
            # dense: mean length 1000 (every contig chains into ONE run: the degenerate case); sparse: mean length 20
            # (~60 % singletons, >= 10^6 output runs: the 28 * N_out term is exercised)
            for tag, mean, seed in (("dense", 1000, 0x5EED0008), ("sparse", 20, 0x5EED000A)):
                k, s64, e64 = half_open64(*synth.gen_torch(n, mean, 24, seed, dev))
                ns = 10_000_000
                hk, hs, he = k[:ns].cpu().numpy().view(np.uint32), s64[:ns].cpu().numpy(), e64[:ns].cpu().numpy()
                if "merge" in which:
                    tm, km, out = timed(lambda: ctx.merge(k, s64, e64, n_keys=24), reps=4)
                    m = int(out[0].numel()); del out
                    cpu = cpu_fig(lambda: orc.merge(hk, hs, he), ns, f"first {ns} rows (sort + sweep)", 1)
                    entry(f"merge_200M_{tag}", tm, km, ALG["merge"](n, 0, m), n, {"out_rows": m}, cpu)
                    if tag == "sparse":
                        # the same rows coordinate-sorted (the usual state of BED / VCF-derived tables): no sort pass at all
                        o = torch.argsort((k.to(torch.int64) << 40) | s64)
                        k2, s2, e2 = k[o].contiguous(), s64[o].contiguous(), e64[o].contiguous()
                        del o
                        tm, km, out = timed(lambda: ctx.merge(k2, s2, e2, n_keys=24), reps=4)
                        m2 = int(out[0].numel()); del out, k2, s2, e2
                        entry("merge_200M_sparse_sorted_input", tm, km, ALG["merge"](n, 0, m2), n, {"out_rows": m2, "what": "the sparse rows in (contig, start) order"})
                if "subtract" in which and tag == "sparse":
                    nr = n // 10
                    rk, rs64, re64 = half_open64(*synth.gen_torch(nr, 8, 24, 0x5EED0009, dev))
                    ts, ks, out = timed(lambda: ctx.subtract(k, s64, e64, rk, rs64, re64, n_keys=24), reps=3)
                    m = int(out[0].numel()); del out
                    nsl, nsr = 5_000_000, 500_000
                    hr = rk[:nsr].cpu().numpy().view(np.uint32), rs64[:nsr].cpu().numpy(), re64[:nsr].cpu().numpy()
                    cpu = cpu_fig(lambda: orc.subtract(hk[:nsl], hs[:nsl], he[:nsl], *hr), nsl, f"first {nsl} left rows - first {nsr} right rows (two sorts + sweep)", 1)
                    entry("subtract_200M_20M", ts, ks, ALG["subtract"](n, nr, m), n, {"out_rows": m, "what": "sizing call + fill call"}, cpu)
                    del rk, rs64, re64
                if "subtract" in which and tag == "dense":
                    nr = n // 10
                    rk, rs64, re64 = half_open64(*synth.gen_torch(nr, 150, 24, 0x5EED0009, dev))
                    ts, ks, out = timed(lambda: ctx.subtract(k, s64, e64, rk, rs64, re64, n_keys=24), reps=3)
                    m = int(out[0].numel()); del out
                    entry("subtract_200M_20M_dense", ts, ks, ALG["subtract"](n, nr, m), n, {"out_rows": m, "what": "sizing call + fill call; mean lengths 1000 / 150 (round-2 workload)"})
                    del rk, rs64, re64
                if "cluster" in which:
                    tc, kc, out = timed(lambda: ctx.cluster(k, s64, e64, n_keys=24), reps=2)
                    ncl = out["n_clusters"]; del out
                    nsc = 4_000_000
                    cpu = cpu_fig(lambda: orc.cluster(hk[:nsc], hs[:nsc], he[:nsc], n_keys=24), nsc, f"first {nsc} rows (sort + sweep)", 1)
                    entry(f"cluster_200M_{tag}", tc, kc, ALG["cluster"](n), n, {"clusters": ncl}, cpu)
                if "complement" in which:
                    tc, kc, out = timed(lambda: ctx.complement(k, s64, e64, n_keys=24), reps=2)
                    m = int(out[0].numel()); del out
                    nsc = 4_000_000
                    cpu = cpu_fig(lambda: orc.complement(hk[:nsc], hs[:nsc], he[:nsc]), nsc, f"first {nsc} rows (sort + merge + gaps)", 1)
                    entry(f"complement_200M_{tag}", tc, kc, ALG["complement"](n, 0, m), n, {"out_rows": m}, cpu)
                del k, s64, e64
                torch.cuda.empty_cache()
        if "big" in which and which & {"merge", "subtract"}:
            # BASELINE config C5 at its full single-GPU size: 10^9 intervals (and a 10^8-row mask); a context of its own,
            # whose ~100 GB of scratch goes back to the device afterwards
            big = pyivx.Ctx(local_rank); big.set_stream(torch.cuda.current_stream().cuda_stream)
            try:
                n = 1_000_000_000
                k, s64, e64 = half_open64(*synth.gen_torch(n, 20, 24, 0x5EED0008, dev))
                def btimed(fn, reps):
                    best, kms, out = 1e9, 0.0, None
                    for _ in range(reps):
                        out = None
                        torch.cuda.synchronize(); big.acc_ms = 0.0
                        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                        if dt < best:
                            best, kms = dt, big.acc_ms
                    big.acc_ms = None
                    return best, kms, out
                if "merge" in which:
                    tm, km, out = btimed(lambda: big.merge(k, s64, e64, n_keys=24), 3)
                    m = int(out[0].numel()); del out
                    torch.cuda.empty_cache()
                    entry("merge_1B", tm, km, ALG["merge"](n, 0, m), n, {"out_rows": m})
                if "subtract" in which:
                    nr = 100_000_000
                    rk, rs64, re64 = half_open64(*synth.gen_torch(nr, 8, 24, 0x5EED0009, dev))
                    ts, ks, out = btimed(lambda: big.subtract(k, s64, e64, rk, rs64, re64, n_keys=24), 2)
                    m = int(out[0].numel()); del out
                    entry("subtract_1B_100M", ts, ks, ALG["subtract"](n, nr, m), n, {"out_rows": m, "what": "sizing call + fill call"})
                    del rk, rs64, re64
                del k, s64, e64
            finally:
                big.close()
                torch.cuda.empty_cache()
        return ops

    # =================================================================== run
    if kind != "join":
        out = run_other(args.scaling, gather, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps(out), flush=True)
        ctx.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    if args.ops_only:
        print(json.dumps({"ops": ops_block()}), flush=True)
        ctx.close()
        return

    r = run_join(args.scaling, gather, args.steps, args.warmup)
    (bk, bs, be), (pk, ps, pe) = r["cols"]
    ob, op = r["out"]
    pairs, n_probe, n_build = r["pairs"], r["n_probe"], r["n_build"]

    out = None
    if rank == 0:
        alg_bytes = ALG["join"](n_probe, n_build, pairs)             # every input read once, every pair written once
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
                       "gather": bool(gather and world > 1), "build_overlap": not args.no_build_overlap},
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
        del hp, hob, hop
      except Exception as ex:                                    # noqa: BLE001 -- extras never cost the official line
        out["extras_error"] = f"{type(ex).__name__}: {ex}"

    # ---- CPU baseline beside it + parity of the timed GPU result against the oracle's pair set
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import oracle as orc                     # checker, used as the timed CPU baseline only
        ns = min(args.cpu_sample, n_probe)
        hb = (bk.cpu().numpy().view(np.uint32), bs.cpu().numpy(), be.cpu().numpy())
        hp = (pk[:ns].cpu().numpy().view(np.uint32), ps[:ns].cpu().numpy(), pe[:ns].cpu().numpy())
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
        del hb, hp, cb, cp, gk, sel

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
            s = run_join("strong", not args.no_gather, args.steps, args.warmup)
        except Exception as ex:                                   # noqa: BLE001 -- reported in the JSON line
            err = f"{type(ex).__name__}: {ex}"
        if rank == 0:
            if err is None:
                out["strong"] = {"value": s["value"], "unit": "overlap-pairs/s", "ms_per_step": s["ms_per_step"], "probe_rows_per_s": s["rows_per_s"],
                                 "pairs_total": s["tot_pairs"], "probe_rows_rank0": s["n_probe"], "gather": not args.no_gather,
                                 "what": f"one {n_probe_w} x {n_build_w} job, {n_contigs} contigs sharded by LPT over {world} ranks, all-gatherv of the pair buffers in the step"}
            else:
                out["strong"] = {"error": err}

    # ---- every other operator, timed by this run (N = 1; after everything the official line needs, never at its cost)
    if rank == 0 and world == 1 and not args.no_ops and not args.no_extras and not args.probe_rows and args.workload == "join_100Mx1M_24contigs":
        r = None
        del ob, op, bk, bs, be, pk, ps, pe
        gc.collect(); torch.cuda.empty_cache()
        try:
            out["ops"] = ops_block()
        except Exception as ex:                                   # noqa: BLE001
            out["ops_error"] = f"{type(ex).__name__}: {ex}"

    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
