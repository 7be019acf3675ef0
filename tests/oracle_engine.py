"""CPU tier only: the oracle behind the method names of pyivx.Ctx, on torch CPU tensors, so that the sharding / exchange
code of `sharded.py` (which does no interval arithmetic itself) runs under gloo without a GPU.  Test infrastructure:
the product path never sees this class."""
import numpy as np
import torch

from oracle import oracle as orc

KIND_OVERLAP, KIND_COUNT, KIND_COVERAGE, KIND_NEAREST = 0, 1, 2, 3


def _n(t, dt=None):
    if t is None:
        return None
    a = t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return a if dt is None else a.astype(dt, copy=False)


def _k(t):
    return _n(t).astype(np.int64).astype(np.uint32)


def _t32(a):
    """uint32 numpy -> int32 tensor with the same bits (as the device tensors hold them)"""
    return torch.from_numpy(np.ascontiguousarray(a, np.uint32).view(np.int32).copy())


class _Index:
    def __init__(self, kind, cols):
        self.kind, self.cols = kind, cols

    def free(self):
        self.cols = None


class OracleEngine:
    def synchronize(self):
        pass

    def build(self, kind, key, start, end, n_keys=None):
        return _Index(kind, (_k(key), _n(start, np.int32), _n(end, np.int32)))

    def count_overlaps(self, ix, key, start, end, strict=False):
        return torch.from_numpy(orc.count_overlaps(*ix.cols, _k(key), _n(start, np.int32), _n(end, np.int32), strict=strict))

    def coverage(self, ix, key, start, end, strict=False):
        return torch.from_numpy(orc.coverage(*ix.cols, _k(key), _n(start, np.int32), _n(end, np.int32), strict=strict))

    def nearest(self, ix, key, start, end, k=1, overlap=True, strict=False):
        ob, op, od = orc.nearest(*ix.cols, _k(key), _n(start, np.int32), _n(end, np.int32), k=k, overlap=overlap, strict=strict)
        return _t32(ob), _t32(op), torch.from_numpy(od)

    def merge(self, key, start, end, n_keys=None, min_dist=0, strict=False):
        k, s, e, n = orc.merge(_k(key), _n(start, np.int64), _n(end, np.int64), min_dist=min_dist, strict=strict)
        return _t32(k), torch.from_numpy(s), torch.from_numpy(e), torch.from_numpy(n)

    def subtract(self, lkey, ls, le, rkey, rs, re, n_keys=None, strict=False):
        k, s, e, row = orc.subtract(_k(lkey), _n(ls, np.int64), _n(le, np.int64), _k(rkey), _n(rs, np.int64), _n(re, np.int64), strict=strict)
        return _t32(k), torch.from_numpy(s), torch.from_numpy(e), _t32(row)

    def complement(self, key, start, end, vkey=None, vstart=None, vend=None, n_keys=None, strict=False):
        v = (None, None, None) if vstart is None else (_k(vkey), _n(vstart, np.int64), _n(vend, np.int64))
        k, s, e = orc.complement(_k(key), _n(start, np.int64), _n(end, np.int64), *v, strict=strict)
        return _t32(k), torch.from_numpy(s), torch.from_numpy(e)

    def cluster(self, key, start, end, n_keys=None, min_dist=0, strict=False, key_base=None, rows=True):
        out = orc.cluster(_k(key), _n(start, np.int64), _n(end, np.int64), min_dist=min_dist, strict=strict, n_keys=n_keys,
                          key_base=None if key_base is None else _n(key_base, np.int64))
        res = {"key_clusters": torch.from_numpy(out["key_clusters"].astype(np.int64)), "n_clusters": out["n_clusters"]}
        if rows:
            for c in ("key", "row"):
                res[c] = _t32(out[c])
            for c in ("start", "end", "cluster", "cluster_start", "cluster_end"):
                res[c] = torch.from_numpy(out[c])
        return res

    def take_fixed(self, src, idx, want_valid=True):
        out, valid = orc.take_fixed(_n(src), _n(idx).view(np.uint32))
        return torch.from_numpy(out), (torch.from_numpy(valid) if want_valid else None)

    def scatter_fixed(self, src, idx, out):
        o = out.numpy()
        o[_n(idx).view(np.uint32).astype(np.int64)] = _n(src)
        return out
