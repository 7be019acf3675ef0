"""Multi-GPU plumbing for the interval path.

Every operator is independent per equi-key (contig), so keys are the unit of
sharding: `assign_keys_lpt` gives each rank whole keys (greedy longest
processing time on probe+build rows), each rank builds and probes its keys with
no communication, and the only exchange is the optional all-gather of the
variable-length per-rank match buffers (`allgatherv`) for a single consumer.
Works with any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs,
"gloo" on CPU for tests.
"""
import numpy as np


def assign_keys_lpt(weights, n_ranks):
    """weights[k] = rows of key k (probe + build).  Returns rank_of_key (int array):
    heaviest key first onto the currently lightest rank."""
    weights = np.asarray(weights, dtype=np.int64)
    rank_of = np.zeros(len(weights), dtype=np.int64)
    load = np.zeros(n_ranks, dtype=np.int64)
    for k in np.argsort(-weights, kind="stable"):
        r = int(np.argmin(load))
        rank_of[k] = r
        load[r] += weights[k]
    return rank_of


def comm_device(dist, device, group=None):
    """Where tensors must live to go through the process group: the GPU for "nccl" (RCCL), host memory for "gloo"
    (CPU tests, and rehearsals of several ranks on one GPU, where RCCL refuses two ranks per device)."""
    import torch
    return torch.device("cpu") if dist.get_backend(group) == "gloo" else device


def allgatherv(dist, tensors, group=None):
    """All-gather a tuple of equally long 1-D tensors whose length differs per rank (the (build_idx, probe_idx)
    pair buffers).  Returns (the concatenations over ranks in rank order, the per-rank sizes).

    One small all-gather of the sizes, then every rank's slice travels ONCE to each peer, straight into its place
    (exact offset, exact length) of the preallocated result: a single group of point-to-point sends / receives
    (RCCL groups them into one all-gatherv-shaped exchange over xGMI; no padding to the longest rank, no staging
    copies, no concatenation afterwards)."""
    import torch
    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    home = tensors[0].device
    dev = comm_device(dist, home, group)
    tensors = [t.to(dev) for t in tensors]
    n = torch.tensor([tensors[0].numel()], dtype=torch.int64, device=dev)
    all_n = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_n, n, group=group)
    sizes = [int(x) for x in all_n.tolist()]
    offs = [0]
    for x in sizes:
        offs.append(offs[-1] + x)
    outs, ops = [], []
    for t in tensors:
        t = t.contiguous()
        out = torch.empty(offs[-1], dtype=t.dtype, device=dev)
        out[offs[me]:offs[me + 1]] = t
        for r in range(world):
            if r == me:
                continue
            peer = dist.get_global_rank(group, r) if group is not None else r
            if sizes[me]:
                ops.append(dist.P2POp(dist.isend, t, peer, group))
            if sizes[r]:
                ops.append(dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], peer, group))
        outs.append(out)
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return tuple(o.to(home) for o in outs), sizes


def max_over_ranks(dist, seconds, device):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=comm_device(dist, device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def cluster_key_base(dist, key_clusters, group=None):
    """cluster(): the one exchange step with real data dependence (cluster.rs:322-420,
    ClusterIdCoordinator).  Every rank passes the clusters it counted per key (int64
    tensor [n_keys], zero for keys it does not hold; ivx_cluster's key_clusters from a
    count-only call).  One all-reduce(SUM) makes the global per-key counts; the exclusive
    scan over keys in id order (= contig names in byte order) is the id of each key's
    first cluster, identical on all ranks -- the key_base argument of ivx_cluster."""
    import torch
    tot = key_clusters.clone().to(torch.int64).to(comm_device(dist, key_clusters.device, group))
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    tot = tot.to(key_clusters.device)
    return torch.cumsum(tot, 0) - tot
