"""Every range operator over contigs sharded across ranks (BASELINE configs C3, C4, C5).

The reference partitions these operators itself: count_overlaps / coverage / nearest fan the probe batches out
round-robin over ONE shared index and return them in stream order (count_overlaps.rs:143-153), merge / subtract /
cluster / complement hash-partition on contig (merge.rs:148-153, subtract.rs:207-218); its tests pin that the
partitioned forms return what the single-partition form returns (R/tests/integration_test.rs:3709-3755, :3783-3890,
:3923-3951, :3987-4020).  Here a contig (equi-key) is the unit: `shard.assign_keys_lpt` deals whole contigs to ranks,
a rank holds the rows of its contigs together with their row numbers in the whole job, runs the operator on them with
no communication, and ONE exchange (exact-size all-gatherv, shard.allgatherv) makes the whole job's result:

  per-row operators (count_overlaps, coverage, nearest k = 1): the (row, value...) lists of all ranks are gathered and
      scattered into columns in INPUT order (ivx_scatter_fixed);
  ordered operators (merge, subtract, cluster, complement): every rank's output is already ordered by key, so the
      gathered pieces are spliced key by key (`splice_by_key`), which is the single-rank output exactly; row ids
      (subtract's left row, cluster's input row) are translated to the job's row numbers first (ivx_take_fixed);
  cluster ids additionally need every rank's clusters per contig BEFORE the rows are written
      (shard.cluster_key_base, the reference's ClusterIdCoordinator).

`engine` is a pyivx.Ctx (device tensors) -- or, in the CPU tests, an object with the same methods on CPU tensors; this
module does no interval arithmetic of its own.  Fewer contigs than ranks: the surplus ranks hold no rows and contribute
empty pieces (per-row operators may instead be given a replicated build side and a slice of the probe rows: the
scatter is the same).
"""
import shard

NULL_IDX32 = -1          # IVX_NULL_IDX (0xFFFFFFFF) as it reads in an int32 tensor


class ShardedRanges:
    def __init__(self, dist, engine, n_keys, group=None):
        self.dist, self.eng, self.n_keys, self.group = dist, engine, int(n_keys), group

    # ---------------------------------------------------------------- exchange helpers
    def rows_to_input_order(self, rows, cols, n_total, fills):
        """rows: the job's row numbers of this rank's rows (int32); cols: value columns aligned with rows.
        -> one column per value column, n_total long, in input order (on every rank)."""
        import torch
        (g_rows, *g_cols), _ = shard.allgatherv(self.dist, (rows, *cols), self.group)
        outs = []
        for c, fill in zip(g_cols, fills):
            out = torch.full((n_total,), fill, dtype=c.dtype, device=c.device)
            outs.append(self.eng.scatter_fixed(c, g_rows, out))
        return tuple(outs)

    def splice_by_key(self, okey, cols, late=None):
        """okey: key column of this rank's output (rows grouped by ascending key; with `late` -- bool [n_keys], true =
        the key belongs to the second group -- first the early keys ascending, then the late ones); cols: the other
        output columns.  -> (key, *cols) of the whole job in the single-rank order, on every rank."""
        import torch
        dist, nk = self.dist, self.n_keys
        world = dist.get_world_size(self.group)
        dev = okey.device
        cdev = shard.comm_device(dist, dev, self.group)
        cnt = torch.bincount(okey.long(), minlength=nk)[:nk].to(torch.int64) if okey.numel() else torch.zeros(nk, dtype=torch.int64, device=dev)
        lt = torch.zeros(nk, dtype=torch.int64, device=dev) if late is None else late.to(device=dev, dtype=torch.int64)
        mine = torch.stack([cnt, lt * (cnt > 0)]).to(cdev).contiguous()
        table = torch.empty((world, 2, nk), dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(table.view(-1), mine.view(-1), group=self.group)
        table = table.cpu()
        cnts, lates = table[:, 0, :].tolist(), table[:, 1, :].tolist()
        (g_key, *g_cols), sizes = shard.allgatherv(dist, (okey, *cols), self.group)
        base = [0]
        for s in sizes:
            base.append(base[-1] + s)
        # where each (rank, key) piece sits in the gathered buffers: a rank's own rows are in (group, key) order
        pos = [[0] * nk for _ in range(world)]
        for r in range(world):
            p = base[r]
            for g in (0, 1):
                for k in range(nk):
                    if cnts[r][k] and lates[r][k] == g:
                        pos[r][k] = p
                        p += cnts[r][k]
        pieces = []
        for g in (0, 1):
            for k in range(nk):
                for r in range(world):
                    if cnts[r][k] and lates[r][k] == g:
                        lo, hi = pos[r][k], pos[r][k] + cnts[r][k]
                        if pieces and pieces[-1][1] == lo:
                            pieces[-1][1] = hi                  # consecutive in the gathered buffer too
                        else:
                            pieces.append([lo, hi])
        def cat(col):
            if len(pieces) == 1 and pieces[0] == [0, int(col.shape[0])]:
                return col
            if not pieces:
                return col[:0]
            return torch.cat([col[lo:hi] for lo, hi in pieces])
        return tuple(cat(c) for c in (g_key, *g_cols))

    def to_job_rows(self, rows, idx, nullable=False):
        """local row ids -> the job's row numbers (rows: int32 [n_local]); IVX_NULL_IDX stays IVX_NULL_IDX"""
        import torch
        if idx.numel() == 0:
            return idx
        out, valid = self.eng.take_fixed(rows, idx, want_valid=nullable)
        if nullable:
            out = torch.where(valid.bool(), out, torch.full_like(out, NULL_IDX32))
        return out

    # ---------------------------------------------------------------- per-row operators
    def count_coverage(self, build, probe, rows_p, n_total, strict=False, gather=True, coverage=True, count=True):
        """build / probe: this rank's (key, start, end) columns; -> (count, coverage) int64 columns of the whole job in
        input order (gather) or this rank's columns (no gather)."""
        import pyivx
        eng, outs = self.eng, []
        for want, kind, fn in ((count, pyivx.KIND_COUNT, "count_overlaps"), (coverage, pyivx.KIND_COVERAGE, "coverage")):
            if not want:
                continue
            ix = eng.build(kind, *build, n_keys=self.n_keys)
            outs.append(getattr(eng, fn)(ix, *probe, strict=strict))
            eng.synchronize()
            ix.free()
        if not gather:
            return tuple(outs)
        return self.rows_to_input_order(rows_p, outs, n_total, [0] * len(outs))

    def nearest1(self, build, rows_b, probe, rows_p, n_total, strict=False, overlap=True, gather=True):
        """k = 1: -> (build row in the job's numbering or IVX_NULL_IDX, distance or -1) per probe row, input order"""
        import pyivx
        eng = self.eng
        ix = eng.build(pyivx.KIND_NEAREST, *build, n_keys=self.n_keys)
        ob, _, od = eng.nearest(ix, *probe, k=1, overlap=overlap, strict=strict)
        eng.synchronize()
        ix.free()
        ob = self.to_job_rows(rows_b, ob, nullable=True)
        if not gather:
            return ob, od
        return self.rows_to_input_order(rows_p, (ob, od), n_total, (NULL_IDX32, -1))

    # ---------------------------------------------------------------- ordered operators
    def merge(self, rows, min_dist=0, strict=False, gather=True):
        out = self.eng.merge(*rows, n_keys=self.n_keys, min_dist=min_dist, strict=strict)
        return self.splice_by_key(out[0], out[1:]) if gather else out

    def subtract(self, left, rows_l, right, strict=False, gather=True):
        k, s, e, row = self.eng.subtract(*left, *right, n_keys=self.n_keys, strict=strict)
        row = self.to_job_rows(rows_l, row)
        return self.splice_by_key(k, (s, e, row)) if gather else (k, s, e, row)

    def complement(self, rows, view=None, strict=False, gather=True):
        import torch
        v = view if view is not None else (None, None, None)
        k, s, e = self.eng.complement(*rows, *v, n_keys=self.n_keys, strict=strict)
        if not gather:
            return k, s, e
        # keys that only have view rows come after the keys with input rows (complement.rs:394-465)
        has_in = torch.bincount(rows[0].long(), minlength=self.n_keys)[:self.n_keys] > 0 if rows[0].numel() else torch.zeros(self.n_keys, dtype=torch.bool, device=k.device)
        return self.splice_by_key(k, (s, e), late=~has_in)

    def cluster(self, rows, rows_g, min_dist=0, strict=False, gather=True):
        """-> dict(key, start, end, row, cluster, cluster_start, cluster_end) of the whole job, ids global"""
        import torch
        eng = self.eng
        counts = eng.cluster(*rows, n_keys=self.n_keys, min_dist=min_dist, strict=strict, rows=False)["key_clusters"]
        base = shard.cluster_key_base(self.dist, counts.to(torch.int64), self.group)
        out = eng.cluster(*rows, n_keys=self.n_keys, min_dist=min_dist, strict=strict, key_base=base)
        out["row"] = self.to_job_rows(rows_g, out["row"])
        names = ("start", "end", "row", "cluster", "cluster_start", "cluster_end")
        if gather:
            got = self.splice_by_key(out["key"], tuple(out[c] for c in names))
            out = dict(zip(("key",) + names, got))
        return out
