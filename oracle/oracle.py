"""ctypes binding of the CPU oracle (oracle/ivx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libivx_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_superintervals.so")

NULL_IDX = 0xFFFFFFFF

_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the C restatement (and, where /root/reference exists, the
    reference-built superintervals checker)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "ivx_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "all"])
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REF)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_join_brute.restype = C.c_uint64
        _lib.orc_join_tree.restype = C.c_uint64
        _lib.orc_nearest.restype = C.c_uint64
        _lib.orc_nearest1_mt.restype = C.c_uint64
        _lib.orc_merge.restype = C.c_uint64
        _lib.orc_subtract.restype = C.c_uint64
        _lib.orc_merge_intervals_i32.restype = C.c_uint64
        _lib.orc_check_i32.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _k(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _side32(key, s, e):
    key, s, e = _k(key), _c32(s), _c32(e)
    assert len(key) == len(s) == len(e)
    return key, s, e, C.c_uint64(len(key))


def join(bkey, bs, be, pkey, ps, pe, brute=False, threads=1, per_row=False):
    """-> (build_rows u32[], probe_rows u32[]) [, per_row u64[]]"""
    L = lib()
    bk, bs_, be_, nb = _side32(bkey, bs, be)
    pk, ps_, pe_, npr = _side32(pkey, ps, pe)
    if brute:
        n = L.orc_join_brute(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, None, None, C.c_uint64(0))
        ob = np.empty(n, np.uint32); op = np.empty(n, np.uint32)
        L.orc_join_brute(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, _p(ob), _p(op), C.c_uint64(n))
        return ob, op
    cnt = np.zeros(len(pk), np.uint64)
    n = L.orc_join_tree(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, None, None, C.c_uint64(0),
                        _p(cnt), C.c_int(threads))
    ob = np.empty(n, np.uint32); op = np.empty(n, np.uint32)
    L.orc_join_tree(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, _p(ob), _p(op), C.c_uint64(n),
                    _p(cnt), C.c_int(threads))
    return (ob, op, cnt) if per_row else (ob, op)


def join_single(bkey, bs, be, pkey, ps, pe, threads=1):
    """The join in one walk per probe row with per-thread buffers (the reference's probe-loop shape);
    -> (build_rows u32[], probe_rows u32[]) grouped by probe row in probe order."""
    L = lib()
    L.orc_join_single_run.restype = C.c_void_p
    L.orc_join_single_total.restype = C.c_uint64
    bk, bs_, be_, nb = _side32(bkey, bs, be)
    pk, ps_, pe_, npr = _side32(pkey, ps, pe)
    h = C.c_void_p(L.orc_join_single_run(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, C.c_int(threads)))
    try:
        n = L.orc_join_single_total(h)
        ob = np.empty(n, np.uint32); op = np.empty(n, np.uint32)
        L.orc_join_single_copy(h, _p(ob), _p(op))
    finally:
        L.orc_join_single_free(h)
    return ob, op


def pair_keys(b, p):
    """(build_row, probe_row) pairs as sorted uint64 keys build<<32|probe: the order-free form two pair
    multisets are compared in (the reference pins only the row multiset, never the match order)."""
    k = (np.asarray(b).astype(np.uint64) << np.uint64(32)) | np.asarray(p).astype(np.uint64)
    k.sort()
    return k


def join_count(bkey, bs, be, pkey, ps, pe, threads=1):
    """Total pairs + per-row counts, nothing materialised."""
    L = lib()
    bk, bs_, be_, nb = _side32(bkey, bs, be)
    pk, ps_, pe_, npr = _side32(pkey, ps, pe)
    cnt = np.zeros(len(pk), np.uint64)
    n = L.orc_join_tree(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, None, None, C.c_uint64(0),
                        _p(cnt), C.c_int(threads))
    return int(n), cnt


def join_exists(bkey, bs, be, pkey, ps, pe):
    L = lib()
    bk, bs_, be_, nb = _side32(bkey, bs, be)
    pk, ps_, pe_, npr = _side32(pkey, ps, pe)
    out = np.zeros(len(pk), np.uint8)
    L.orc_join_exists(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, _p(out))
    return out


def count_overlaps(bkey, bs, be, pkey, ps, pe, strict=False, threads=1):
    L = lib()
    bk, bs_, be_, nb = _side32(bkey, bs, be)
    pk, ps_, pe_, npr = _side32(pkey, ps, pe)
    out = np.zeros(len(pk), np.int64)
    L.orc_count_overlaps_mt(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, C.c_int(int(strict)), _p(out), C.c_int(threads))
    return out


def coverage(bkey, bs, be, pkey, ps, pe, strict=False, threads=1):
    L = lib()
    bk, bs_, be_, nb = _side32(bkey, bs, be)
    pk, ps_, pe_, npr = _side32(pkey, ps, pe)
    out = np.zeros(len(pk), np.int64)
    L.orc_coverage_mt(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, C.c_int(int(strict)), _p(out), C.c_int(threads))
    return out


def merge_intervals(s, e):
    s = _c32(s).copy(); e = _c32(e).copy()
    m = lib().orc_merge_intervals_i32(_p(s), _p(e), C.c_uint64(len(s)))
    return s[:m], e[:m]


def nearest(bkey, bs, be, pkey, ps, pe, k=1, overlap=True, strict=False):
    """-> (build_rows u32 (NULL_IDX = null), probe_rows u32, distance i64 (-1 = null))"""
    L = lib()
    bk, bs_, be_, nb = _side32(bkey, bs, be)
    pk, ps_, pe_, npr = _side32(pkey, ps, pe)
    cap = len(pk) * max(int(k), 1)
    ob = np.empty(cap, np.uint32); op = np.empty(cap, np.uint32); od = np.empty(cap, np.int64)
    n = L.orc_nearest(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, C.c_int(int(strict)),
                      C.c_uint32(int(k)), C.c_int(int(overlap)), _p(ob), _p(op), _p(od), C.c_uint64(cap))
    return ob[:n], op[:n], od[:n]


def nearest1(bkey, bs, be, pkey, ps, pe, overlap=True, strict=False, threads=16):
    """k = 1 over `threads` host threads -> (build_rows u32 per probe row (NULL_IDX = null), distance i64 (-1 = null))"""
    L = lib()
    bk, bs_, be_, nb = _side32(bkey, bs, be)
    pk, ps_, pe_, npr = _side32(pkey, ps, pe)
    ob = np.empty(len(pk), np.uint32); od = np.empty(len(pk), np.int64)
    L.orc_nearest1_mt(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, C.c_int(int(strict)),
                      C.c_int(int(overlap)), _p(ob), _p(od), C.c_int(int(threads)))
    return ob, od


def merge(key, s, e, min_dist=0, strict=False):
    """-> (key u32, start i64, end i64, n_intervals i64), keys ascending."""
    L = lib()
    key, s, e = _k(key), _c64(s), _c64(e)
    n = len(key)
    ok = np.empty(n, np.uint32); os_ = np.empty(n, np.int64); oe = np.empty(n, np.int64); on = np.empty(n, np.int64)
    m = L.orc_merge(_p(key), _p(s), _p(e), C.c_uint64(n), C.c_int64(int(min_dist)), C.c_int(int(strict)),
                    _p(ok), _p(os_), _p(oe), _p(on), C.c_uint64(n))
    return ok[:m], os_[:m], oe[:m], on[:m]


def subtract(lkey, ls, le, rkey, rs, re, strict=False):
    """-> (key u32, start i64, end i64, left_row u32)"""
    L = lib()
    lkey, ls, le = _k(lkey), _c64(ls), _c64(le)
    rkey, rs, re = _k(rkey), _c64(rs), _c64(re)
    args = (_p(lkey), _p(ls), _p(le), C.c_uint64(len(lkey)), _p(rkey), _p(rs), _p(re), C.c_uint64(len(rkey)),
            C.c_int(int(strict)))
    m = L.orc_subtract(*args, None, None, None, None, C.c_uint64(0))
    ok = np.empty(m, np.uint32); os_ = np.empty(m, np.int64); oe = np.empty(m, np.int64); orow = np.empty(m, np.uint32)
    L.orc_subtract(*args, _p(ok), _p(os_), _p(oe), _p(orow), C.c_uint64(m))
    return ok, os_, oe, orow


def cluster(key, s, e, min_dist=0, strict=False, n_keys=None, key_base=None):
    """-> dict(key u32, start i64, end i64, row u32, cluster i64, cluster_start i64, cluster_end i64,
               key_clusters u64[n_keys], n_clusters) -- rows sorted by (key, start, end, row)."""
    L = lib()
    key, s, e = _k(key), _c64(s), _c64(e)
    n = len(key)
    nk = int(n_keys if n_keys is not None else (int(key.max()) + 1 if n else 0))
    ok = np.empty(n, np.uint32); os_ = np.empty(n, np.int64); oe = np.empty(n, np.int64); orow = np.empty(n, np.uint32)
    oc = np.empty(n, np.int64); ocs = np.empty(n, np.int64); oce = np.empty(n, np.int64)
    kc = np.zeros(max(nk, 1), np.uint64)
    kb = None if key_base is None else _c64(key_base)
    L.orc_cluster.restype = C.c_uint64
    tot = L.orc_cluster(_p(key), _p(s), _p(e), C.c_uint64(n), C.c_uint32(nk), C.c_int64(int(min_dist)), C.c_int(int(strict)),
                        _p(kb) if kb is not None else None, _p(ok), _p(os_), _p(oe), _p(orow), _p(oc), _p(ocs), _p(oce), _p(kc))
    return dict(key=ok, start=os_, end=oe, row=orow, cluster=oc, cluster_start=ocs, cluster_end=oce,
                key_clusters=kc[:nk], n_clusters=int(tot))


def complement(key, s, e, vkey=None, vs=None, ve=None, strict=False):
    """-> (key u32, start i64, end i64); no view rows = implicit [0, i64::MAX) per key."""
    L = lib()
    key, s, e = _k(key), _c64(s), _c64(e)
    if vkey is None:
        vkey, vs, ve = np.empty(0, np.uint32), np.empty(0, np.int64), np.empty(0, np.int64)
    vkey, vs, ve = _k(vkey), _c64(vs), _c64(ve)
    L.orc_complement.restype = C.c_uint64
    args = (_p(key), _p(s), _p(e), C.c_uint64(len(key)), _p(vkey), _p(vs), _p(ve), C.c_uint64(len(vkey)), C.c_int(int(strict)))
    m = L.orc_complement(*args, None, None, None, C.c_uint64(0))
    ok = np.empty(m, np.uint32); os_ = np.empty(m, np.int64); oe = np.empty(m, np.int64)
    L.orc_complement(*args, _p(ok), _p(os_), _p(oe), C.c_uint64(m))
    return ok, os_, oe


def take_fixed(src, idx, src_valid=None):
    """f3: arrow `compute::take` on a fixed-width column (interval_join.rs:1655-1667, nearest.rs:469-482):
    out[i] = src[idx[i]]; a NULL_IDX index or a null source slot gives a null output slot (valid 0),
    a NULL_IDX slot holds zero bytes.  -> (out, valid u8)"""
    src = np.asarray(src); idx = np.asarray(idx, np.uint32)
    null = idx == NULL_IDX
    if (idx[~null] >= len(src)).any():
        raise IndexError("take: index out of bounds")
    safe = np.where(null, 0, idx).astype(np.int64)
    out = src[safe].copy() if len(src) else np.zeros((len(idx),) + src.shape[1:], src.dtype)
    out[null] = 0
    valid = ~null
    if src_valid is not None:
        valid = np.asarray(src_valid, bool)[safe] & ~null
    return out, valid.astype(np.uint8)


def take_utf8(offsets, data, idx, src_valid=None):
    """same for Utf8/LargeUtf8: -> (out_offsets, out_data u8, valid u8); a NULL_IDX slot is an empty string."""
    offsets = np.asarray(offsets); data = np.asarray(data, np.uint8); idx = np.asarray(idx, np.uint32)
    null = idx == NULL_IDX
    safe = np.where(null, 0, idx).astype(np.int64)
    lens = np.where(null, 0, offsets[safe + 1] - offsets[safe]) if len(idx) else np.zeros(0, offsets.dtype)
    out_off = np.concatenate([[0], np.cumsum(lens)]).astype(offsets.dtype)
    parts = [data[offsets[j]:offsets[j + 1]] for j, nl in zip(safe, null) if not nl]
    out_data = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    valid = ~null
    if src_valid is not None:
        valid = np.asarray(src_valid, bool)[safe] & ~null
    return out_off, out_data, valid.astype(np.uint8)


def check_i32(v):
    v = _c64(v)
    return int(lib().orc_check_i32(_p(v), C.c_uint64(len(v))))


# ---- the reference's own vendored structure, compiled from /root/reference ----

_ref = None


def ref_available():
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF)
        _ref.ref_si_join.restype = C.c_uint64
    return _ref


def ref_join(bkey, bs, be, pkey, ps, pe):
    R = ref()
    bk, bs_, be_, nb = _side32(bkey, bs, be)
    pk, ps_, pe_, npr = _side32(pkey, ps, pe)
    n = R.ref_si_join(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, None, None, C.c_uint64(0))
    ob = np.empty(n, np.uint32); op = np.empty(n, np.uint32)
    R.ref_si_join(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, _p(ob), _p(op), C.c_uint64(n))
    return ob, op


def ref_count(bkey, bs, be, pkey, ps, pe):
    R = ref()
    bk, bs_, be_, nb = _side32(bkey, bs, be)
    pk, ps_, pe_, npr = _side32(pkey, ps, pe)
    out = np.zeros(len(pk), np.int64)
    R.ref_si_count(_p(bk), _p(bs_), _p(be_), nb, _p(pk), _p(ps_), _p(pe_), npr, _p(out))
    return out
