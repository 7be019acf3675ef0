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


def allgatherv(dist, tensors, group=None):
    """All-gather a tuple of equally long 1-D tensors whose length differs per rank
    (the (build_idx, probe_idx) pair buffers).  Returns the concatenation over ranks
    in rank order.  One size all-gather + one padded all-gather per tensor."""
    import torch
    world = dist.get_world_size(group)
    n = torch.tensor([tensors[0].numel()], dtype=torch.int64, device=tensors[0].device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s) for s in sizes]
    mx = max(max(sizes), 1)
    outs = []
    for t in tensors:
        pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
        pad[: t.numel()] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        outs.append(torch.cat([p[:s] for p, s in zip(parts, sizes)]))
    return tuple(outs), sizes


def max_over_ranks(dist, seconds, device):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
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
    tot = key_clusters.clone().to(torch.int64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    return torch.cumsum(tot, 0) - tot
