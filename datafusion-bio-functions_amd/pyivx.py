"""ctypes binding of libivx_hip.so (include/ivx.h) for tests and bench.py.

Plumbing only: numpy arrays go through the IVX_MEM_HOST entry points, torch CUDA
tensors through IVX_MEM_DEVICE with their data_ptr().  There is no fallback of
any kind: if the HIP library is missing or no gfx950 device is usable this
module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libivx_hip.so")

MEM_HOST, MEM_DEVICE = 0, 1
KIND_OVERLAP, KIND_COUNT, KIND_COVERAGE, KIND_NEAREST = 0, 1, 2, 3
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_CAPACITY, ERR_UNSUPPORTED = range(7)
NULL_IDX = 0xFFFFFFFF

# every symbol include/ivx.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "ivx_ctx_create", "ivx_ctx_free", "ivx_last_error", "ivx_ctx_set_stream", "ivx_ctx_use_own_stream",
    "ivx_ctx_synchronize",
    "ivx_ctx_last_kernel_ms", "ivx_version", "ivx_index_build", "ivx_index_free", "ivx_index_rows",
    "ivx_index_device_bytes", "ivx_probe_overlap_count", "ivx_probe_overlap_fill", "ivx_probe_exists",
    "ivx_probe_count", "ivx_probe_coverage", "ivx_probe_nearest", "ivx_merge", "ivx_subtract",
    "ivx_cluster", "ivx_complement", "ivx_take_fixed", "ivx_take_utf8", "ivx_take_bits", "ivx_take_view",
    "ivx_ctx_metrics", "ivx_ctx_reset_metrics", "ivx_ctx_set_memory_limit", "ivx_ctx_trim", "ivx_scatter_fixed",
    "ivx_ctx_reserved_bytes", "ivx_ctx_set_build_overlap",
]


class Metrics(C.Structure):
    _fields_ = [("build_time", C.c_double), ("join_time", C.c_double), ("build_input_batches", C.c_uint64),
                ("build_input_rows", C.c_uint64), ("build_mem_used", C.c_uint64), ("input_batches", C.c_uint64),
                ("input_rows", C.c_uint64), ("output_batches", C.c_uint64), ("output_rows", C.c_uint64)]


class IvxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"ivx status {status}: {msg}")
        self.status = status


_lib = None


def _preload_hip_runtime():
    """PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process do not share device state, so when torch is installed make its
    copy the one that is resident before libivx_hip.so binds to libamdhip64.so.7."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IvxError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() (no CPU fallback exists)")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.ivx_last_error.restype = C.c_char_p
        L.ivx_version.restype = C.c_char_p
        L.ivx_ctx_last_kernel_ms.restype = C.c_double
        L.ivx_index_rows.restype = C.c_uint64
        L.ivx_index_device_bytes.restype = C.c_uint64
        L.ivx_ctx_reserved_bytes.restype = C.c_uint64
        _lib = L
    return _lib


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        return C.c_void_p(x.data_ptr())
    return x.ctypes.data_as(C.c_void_p)


def _mem_of(*arrs):
    kinds = {_is_torch(a) for a in arrs if a is not None}
    if len(kinds) > 1:
        raise ValueError("mix of host and device buffers in one call")
    return MEM_DEVICE if kinds == {True} else MEM_HOST


def _np(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def _cols(key, s, e, dt):
    """normalise one (key,start,end) side; returns (key,start,end,n,mem)"""
    if _is_torch(s):
        assert s.is_contiguous() and e.is_contiguous() and (key is None or key.is_contiguous())
        return key, s, e, int(s.numel()), MEM_DEVICE
    key, s, e = _np(key, np.uint32), _np(s, dt), _np(e, dt)
    return key, s, e, len(s), MEM_HOST


class Index:
    def __init__(self, ctx, handle, kind):
        self.ctx, self.h, self.kind = ctx, handle, kind

    @property
    def rows(self):
        return lib().ivx_index_rows(self.h)

    @property
    def device_bytes(self):
        return lib().ivx_index_device_bytes(self.h)

    def free(self):
        if self.h:
            lib().ivx_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Ctx:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        st = lib().ivx_ctx_create(C.c_int(device), C.byref(self.h))
        if st != OK:
            self.h = None
            raise IvxError(st, "ivx_ctx_create failed (no usable gfx950 device; there is no CPU fallback)")

    def close(self):
        if self.h:
            lib().ivx_ctx_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    acc_ms = None       # set to 0.0 to sum the device time (hipEvents) of every compute call made from here on

    def _chk0(self, st):
        if st != OK:
            raise IvxError(st, lib().ivx_last_error(self.h).decode())

    def _chk(self, st):
        """status check of a COMPUTE call (one that brackets its kernels with the context's events)"""
        self._chk0(st)
        if self.acc_ms is not None:
            self.acc_ms += self.last_kernel_ms()

    def set_stream(self, stream_ptr):
        """use this hipStream_t verbatim (0/None = HIP's default stream)"""
        self._chk0(lib().ivx_ctx_set_stream(self.h, C.c_void_p(stream_ptr or 0)))

    def use_own_stream(self):
        self._chk0(lib().ivx_ctx_use_own_stream(self.h))

    def synchronize(self):
        self._chk0(lib().ivx_ctx_synchronize(self.h))

    def last_kernel_ms(self):
        return lib().ivx_ctx_last_kernel_ms(self.h)

    def metrics(self):
        """BuildProbeJoinMetrics under the reference's names (joins/utils.rs:399-453)."""
        m = Metrics()
        self._chk0(lib().ivx_ctx_metrics(self.h, C.byref(m)))
        return {f: getattr(m, f) for f, _ in Metrics._fields_}

    def reset_metrics(self):
        lib().ivx_ctx_reset_metrics(self.h)

    def set_memory_limit(self, nbytes):
        self._chk0(lib().ivx_ctx_set_memory_limit(self.h, C.c_uint64(int(nbytes))))

    def set_build_overlap(self, on=True):
        """an overlap index build returns once the routing tables are final; its tail runs beside the next probe's routing
        pass (include/ivx.h: the build columns must then outlive the index's first probe call)"""
        self._chk0(lib().ivx_ctx_set_build_overlap(self.h, C.c_int(int(on))))

    def reserved_bytes(self):
        return lib().ivx_ctx_reserved_bytes(self.h)

    def trim(self, keep_bytes=0):
        """scratch (and the pool of recycled index buffers) back to the device"""
        self._chk0(lib().ivx_ctx_trim(self.h, C.c_uint64(int(keep_bytes))))

    # ---- index ----
    def build(self, kind, key, start, end, n_keys=None):
        key, s, e, n, mem = _cols(key, start, end, np.int32)
        if n_keys is None:
            n_keys = 1 if key is None else (int(key.max()) + 1 if n else 1)
        h = C.c_void_p()
        self._chk(lib().ivx_index_build(self.h, C.c_int(kind), C.c_int(mem), _ptr(key), _ptr(s), _ptr(e),
                                         C.c_uint64(n), C.c_uint32(n_keys), C.byref(h)))
        return Index(self, h, kind)

    # ---- a3 ----
    def overlap_count(self, ix, key, start, end, per_row=False):
        key, s, e, n, mem = _cols(key, start, end, np.int32)
        total = C.c_uint64(0)
        pr = None
        if per_row:
            if mem == MEM_DEVICE:
                import torch
                pr = torch.empty(n, dtype=torch.int32, device=s.device)
            else:
                pr = np.empty(n, np.uint32)
        self._chk(lib().ivx_probe_overlap_count(self.h, ix.h, C.c_int(mem), _ptr(key), _ptr(s), _ptr(e),
                                                 C.c_uint64(n), _ptr(pr), C.byref(total)))
        return (total.value, pr) if per_row else total.value

    def overlap_fill(self, ix, key, start, end, cap=None, out=None):
        """-> (build_idx, probe_idx) of length `written`.  Host arrays: cap defaults to a count pass."""
        key, s, e, n, mem = _cols(key, start, end, np.int32)
        if out is not None:
            ob, op = out
            cap = int(ob.numel()) if _is_torch(ob) else len(ob)
        else:
            if cap is None:
                cap = self.overlap_count(ix, key, s, e)
            if mem == MEM_DEVICE:
                import torch
                ob = torch.empty(max(cap, 1), dtype=torch.int32, device=s.device)
                op = torch.empty(max(cap, 1), dtype=torch.int32, device=s.device)
            else:
                ob = np.empty(max(cap, 1), np.uint32); op = np.empty(max(cap, 1), np.uint32)
        written = C.c_uint64(0)
        st = lib().ivx_probe_overlap_fill(self.h, ix.h, C.c_int(mem), _ptr(key), _ptr(s), _ptr(e), C.c_uint64(n),
                                          _ptr(ob), _ptr(op), C.c_uint64(cap), C.byref(written))
        if st == ERR_CAPACITY:
            raise IvxError(st, f"need {written.value} pairs, cap {cap}")
        self._chk(st)
        w = written.value
        return ob[:w], op[:w]

    def exists(self, ix, key, start, end):
        key, s, e, n, mem = _cols(key, start, end, np.int32)
        if mem == MEM_DEVICE:
            import torch
            out = torch.empty(n, dtype=torch.uint8, device=s.device)
        else:
            out = np.empty(n, np.uint8)
        self._chk(lib().ivx_probe_exists(self.h, ix.h, C.c_int(mem), _ptr(key), _ptr(s), _ptr(e), C.c_uint64(n), _ptr(out)))
        return out

    # ---- a4 / a5 ----
    def _per_row_i64(self, fn, ix, key, start, end, strict, out=None):
        key, s, e, n, mem = _cols(key, start, end, np.int32)
        if out is not None:
            assert len(out) >= n                                    # (a caller's own int64 buffer, e.g. one it has touched before)
        elif mem == MEM_DEVICE:
            import torch
            out = torch.empty(n, dtype=torch.int64, device=s.device)
        else:
            out = np.empty(n, np.int64)
        self._chk(fn(self.h, ix.h, C.c_int(mem), _ptr(key), _ptr(s), _ptr(e), C.c_uint64(n), C.c_int(int(strict)), _ptr(out)))
        return out

    def count_overlaps(self, ix, key, start, end, strict=False, out=None):
        return self._per_row_i64(lib().ivx_probe_count, ix, key, start, end, strict, out)

    def coverage(self, ix, key, start, end, strict=False, out=None):
        return self._per_row_i64(lib().ivx_probe_coverage, ix, key, start, end, strict, out)

    # ---- a6 ----
    def nearest(self, ix, key, start, end, k=1, overlap=True, strict=False, distance=True):
        key, s, e, n, mem = _cols(key, start, end, np.int32)
        cap = n * max(int(k), 1)
        if mem == MEM_DEVICE:
            import torch
            ob = torch.empty(max(cap, 1), dtype=torch.int32, device=s.device)
            op = torch.empty(max(cap, 1), dtype=torch.int32, device=s.device)
            od = torch.empty(max(cap, 1), dtype=torch.int64, device=s.device) if distance else None
        else:
            ob = np.empty(max(cap, 1), np.uint32); op = np.empty(max(cap, 1), np.uint32)
            od = np.empty(max(cap, 1), np.int64) if distance else None
        rows = C.c_uint64(0)
        self._chk(lib().ivx_probe_nearest(self.h, ix.h, C.c_int(mem), _ptr(key), _ptr(s), _ptr(e), C.c_uint64(n),
                                           C.c_int(int(strict)), C.c_uint32(int(k)), C.c_int(int(overlap)),
                                           _ptr(ob), _ptr(op), _ptr(od), C.c_uint64(cap), C.byref(rows)))
        r = rows.value
        return ob[:r], op[:r], (od[:r] if distance else None)

    # ---- a7..a9 ----
    def merge(self, key, start, end, n_keys=None, min_dist=0, strict=False):
        key, s, e, n, mem = _cols(key, start, end, np.int64)
        if n_keys is None:
            n_keys = 1 if key is None else (int(key.max()) + 1 if n else 1)
        cap = max(n, 1)
        if mem == MEM_DEVICE:
            import torch
            dev = s.device
            ok = torch.empty(cap, dtype=torch.int32, device=dev); os_ = torch.empty(cap, dtype=torch.int64, device=dev)
            oe = torch.empty(cap, dtype=torch.int64, device=dev); on = torch.empty(cap, dtype=torch.int64, device=dev)
        else:
            ok = np.empty(cap, np.uint32); os_ = np.empty(cap, np.int64); oe = np.empty(cap, np.int64); on = np.empty(cap, np.int64)
        m = C.c_uint64(0)
        self._chk(lib().ivx_merge(self.h, C.c_int(mem), _ptr(key), _ptr(s), _ptr(e), C.c_uint64(n), C.c_uint32(n_keys),
                                   C.c_int64(int(min_dist)), C.c_int(int(strict)), _ptr(ok), _ptr(os_), _ptr(oe), _ptr(on),
                                   C.c_uint64(cap), C.byref(m)))
        m = m.value
        return ok[:m], os_[:m], oe[:m], on[:m]

    def subtract(self, lkey, ls, le, rkey, rs, re, n_keys=None, strict=False, cap=None, between=None):
        """cap=None: sizing call, then the fill call (which reuses the sizing call's sorted sides).  cap=N: one call into
        buffers of N rows.  between: callable run between the two calls (tests)."""
        lkey, ls, le, nl, mem = _cols(lkey, ls, le, np.int64)
        rkey, rs, re, nr, mem2 = _cols(rkey, rs, re, np.int64)
        assert mem == mem2
        if n_keys is None:
            mk = 0
            for k_, n_ in ((lkey, nl), (rkey, nr)):
                if k_ is not None and n_:
                    mk = max(mk, int(k_.max()))
            n_keys = mk + 1
        args = (C.c_int(mem), _ptr(lkey), _ptr(ls), _ptr(le), C.c_uint64(nl), _ptr(rkey), _ptr(rs), _ptr(re),
                C.c_uint64(nr), C.c_uint32(n_keys), C.c_int(int(strict)))
        def bufs(cap):
            if mem == MEM_DEVICE:
                import torch
                dev = ls.device
                return (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int64, device=dev),
                        torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.int32, device=dev))
            return np.empty(cap, np.uint32), np.empty(cap, np.int64), np.empty(cap, np.int64), np.empty(cap, np.uint32)
        # sizing call (counting half only), then the fill call: the fragment count has no useful a-priori bound
        if cap is None:
            m = C.c_uint64(0)
            self._chk(lib().ivx_subtract(self.h, *args, None, None, None, None, C.c_uint64(0), C.byref(m)))
            cap = max(m.value, 1)
            if between is not None:
                between()
        ok, os_, oe, orow = bufs(cap)
        m2 = C.c_uint64(0)
        self._chk(lib().ivx_subtract(self.h, *args, _ptr(ok), _ptr(os_), _ptr(oe), _ptr(orow), C.c_uint64(cap), C.byref(m2)))
        m2 = m2.value
        return ok[:m2], os_[:m2], oe[:m2], orow[:m2]

    # ---- f1, f2 ----
    def cluster(self, key, start, end, n_keys=None, min_dist=0, strict=False, key_base=None, rows=True):
        """-> dict(key, start, end, row, cluster, cluster_start, cluster_end [rows sorted by (key,start,end,row)],
                   key_clusters[n_keys], n_clusters); rows=False only counts the clusters per key."""
        key, s, e, n, mem = _cols(key, start, end, np.int64)
        if n_keys is None:
            n_keys = 1 if key is None else (int(key.max()) + 1 if n else 1)
        cap = max(n, 1)
        if mem == MEM_DEVICE:
            import torch
            dev = s.device
            mk = lambda dt, m=cap: torch.empty(m, dtype=dt, device=dev)
            i32, i64 = torch.int32, torch.int64
            if key_base is not None:
                key_base = torch.as_tensor(key_base, dtype=torch.int64, device=dev).contiguous()
        else:
            mk = lambda dt, m=cap: np.empty(m, dt)
            i32, i64 = np.uint32, np.int64
            if key_base is not None:
                key_base = np.ascontiguousarray(key_base, np.int64)
        out = {}
        if rows:
            out = dict(key=mk(i32), start=mk(i64), end=mk(i64), row=mk(i32), cluster=mk(i64), cluster_start=mk(i64), cluster_end=mk(i64))
        kc = mk(i64, max(n_keys, 1))
        m = C.c_uint64(0)
        g = lambda name: _ptr(out[name]) if rows else None
        self._chk(lib().ivx_cluster(self.h, C.c_int(mem), _ptr(key), _ptr(s), _ptr(e), C.c_uint64(n), C.c_uint32(n_keys),
                                     C.c_int64(int(min_dist)), C.c_int(int(strict)), _ptr(key_base) if key_base is not None else None,
                                     g("key"), g("start"), g("end"), g("row"), g("cluster"), g("cluster_start"), g("cluster_end"),
                                     _ptr(kc), C.byref(m)))
        out = {k_: v[:n] for k_, v in out.items()}
        out["key_clusters"] = kc[:n_keys]
        out["n_clusters"] = m.value
        return out

    def complement(self, key, start, end, vkey=None, vstart=None, vend=None, n_keys=None, strict=False):
        """-> (key, start, end); without view rows every key gets the implicit view [0, i64::MAX)."""
        key, s, e, n, mem = _cols(key, start, end, np.int64)
        if vstart is None:
            nv, vk, vs, ve = 0, None, None, None
        else:
            vk, vs, ve, nv, mem2 = _cols(vkey, vstart, vend, np.int64)
            assert mem == mem2
        if n_keys is None:
            mx = 0
            for k_, n_ in ((key, n), (vk, nv)):
                if k_ is not None and n_:
                    mx = max(mx, int(k_.max()))
            n_keys = mx + 1
        args = (C.c_int(mem), _ptr(key), _ptr(s), _ptr(e), C.c_uint64(n), _ptr(vk), _ptr(vs), _ptr(ve), C.c_uint64(nv),
                C.c_uint32(n_keys), C.c_int(int(strict)))
        def bufs(cap):
            if mem == MEM_DEVICE:
                import torch
                dev = s.device
                return torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.int64, device=dev)
            return np.empty(cap, np.uint32), np.empty(cap, np.int64), np.empty(cap, np.int64)
        cap = max(n + 2 * nv + 2 * n_keys + 16, 1)          # enough unless views overlap each other heavily; else one retry
        m2 = C.c_uint64(0)
        ok, os_, oe = bufs(cap)
        st = lib().ivx_complement(self.h, *args, _ptr(ok), _ptr(os_), _ptr(oe), C.c_uint64(cap), C.byref(m2))
        if st == ERR_CAPACITY:
            cap = m2.value
            ok, os_, oe = bufs(cap)
            st = lib().ivx_complement(self.h, *args, _ptr(ok), _ptr(os_), _ptr(oe), C.c_uint64(cap), C.byref(m2))
        self._chk(st)
        m2 = m2.value
        return ok[:m2], os_[:m2], oe[:m2]

    # ---- f3: compute::take of payload columns ----
    def take_fixed(self, src, idx, src_valid_bits=None, want_valid=True):
        """src: 1-D array (numpy or torch) of a fixed-width type (1, 2, 4, 8, 16 or 32 bytes per element; wider
        records as a 2-D uint8 array [n_src, width]); idx: uint32 rows, NULL_IDX = null.  -> (out, valid u8 or None)"""
        dev = _is_torch(src)
        if _is_torch(idx) != dev:
            raise ValueError("mix of host and device buffers in one call")
        if dev:
            import torch
            assert src.is_contiguous() and idx.is_contiguous()
            n_src = src.shape[0]
            width = src.element_size() * (src.shape[1] if src.dim() == 2 else 1)
            n = int(idx.numel())
            out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
            valid = torch.empty(max(n, 1), dtype=torch.uint8, device=src.device) if want_valid else None
            svb = src_valid_bits
        else:
            src = np.ascontiguousarray(src); idx = np.ascontiguousarray(idx, np.uint32)
            n_src = src.shape[0]
            width = src.dtype.itemsize * (src.shape[1] if src.ndim == 2 else 1)
            n = len(idx)
            out = np.empty((n,) + src.shape[1:], src.dtype)
            valid = np.empty(max(n, 1), np.uint8) if want_valid else None
            svb = None if src_valid_bits is None else np.ascontiguousarray(src_valid_bits, np.uint8)
        self._chk(lib().ivx_take_fixed(self.h, C.c_int(MEM_DEVICE if dev else MEM_HOST), _ptr(src), C.c_uint32(width), C.c_uint64(n_src),
                                        _ptr(svb), _ptr(idx), C.c_uint64(n), _ptr(out), _ptr(valid)))
        return out, (valid[:n] if valid is not None else None)

    def scatter_fixed(self, src, idx, out):
        """out[idx[i]] = src[i] (idx: uint32 rows of `out`); -> out"""
        dev = _is_torch(src)
        if _is_torch(idx) != dev or _is_torch(out) != dev:
            raise ValueError("mix of host and device buffers in one call")
        if dev:
            assert src.is_contiguous() and idx.is_contiguous() and out.is_contiguous() and src.dtype == out.dtype
            width, n, n_out = src.element_size(), int(idx.numel()), int(out.numel())
        else:
            src = np.ascontiguousarray(src, out.dtype); idx = np.ascontiguousarray(idx, np.uint32)
            assert out.flags.c_contiguous
            width, n, n_out = out.dtype.itemsize, len(idx), len(out)
        assert (int(src.numel()) if dev else len(src)) == n
        self._chk(lib().ivx_scatter_fixed(self.h, C.c_int(MEM_DEVICE if dev else MEM_HOST), _ptr(src), C.c_uint32(width), _ptr(idx),
                                           C.c_uint64(n), _ptr(out), C.c_uint64(n_out)))
        return out

    def take_utf8(self, offsets, data, idx, src_valid_bits=None, want_valid=True):
        """offsets: int32 (Utf8/Binary) or int64 (LargeUtf8/LargeBinary) [n_src+1]; data: uint8 bytes.
        -> (out_offsets, out_data u8, valid u8 or None)"""
        dev = _is_torch(offsets)
        if dev:
            import torch
            large = offsets.dtype == torch.int64
            n_src, nbytes, n = int(offsets.numel()) - 1, int(data.numel()), int(idx.numel())
            mk = lambda m, dt: torch.empty(max(m, 1), dtype=dt, device=offsets.device)
            odt, u8 = offsets.dtype, torch.uint8
            svb = src_valid_bits
        else:
            offsets = np.ascontiguousarray(offsets); data = np.ascontiguousarray(data, np.uint8); idx = np.ascontiguousarray(idx, np.uint32)
            large = offsets.dtype == np.int64
            assert large or offsets.dtype == np.int32
            n_src, nbytes, n = len(offsets) - 1, len(data), len(idx)
            mk = lambda m, dt: np.empty(max(m, 1), dt)
            odt, u8 = offsets.dtype, np.uint8
            svb = None if src_valid_bits is None else np.ascontiguousarray(src_valid_bits, np.uint8)
        mem = C.c_int(MEM_DEVICE if dev else MEM_HOST)
        out_off = mk(n + 1, odt)
        valid = mk(n, u8) if want_valid else None
        need = C.c_uint64(0)
        head = (self.h, mem, C.c_int(int(large)), _ptr(offsets), _ptr(data), C.c_uint64(n_src), C.c_uint64(nbytes), _ptr(svb), _ptr(idx), C.c_uint64(n))
        # optimistic size from the mean source string length; the call reports the exact need if it is short
        cap = int(n * (nbytes / max(n_src, 1)) * 1.5) + 64
        out_data = mk(cap, u8)
        st = lib().ivx_take_utf8(*head, _ptr(out_off), _ptr(out_data), C.c_uint64(cap), C.byref(need), _ptr(valid))
        if st == ERR_CAPACITY:
            cap = max(need.value, 1)
            out_data = mk(cap, u8)
            st = lib().ivx_take_utf8(*head, _ptr(out_off), _ptr(out_data), C.c_uint64(cap), C.byref(need), _ptr(valid))
        self._chk(st)
        return out_off[: n + 1], out_data[: need.value], (valid[:n] if valid is not None else None)
