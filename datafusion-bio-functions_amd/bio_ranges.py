"""Table-level mirror of the reference's range functions, driven through libbio_ranges_hip.so
(Arrow C Data Interface) -- what a DataFusion session would expose as

    count_overlaps('l','r'), coverage('l','r'), nearest('l','r',k,overlap,distance),
    overlap('l','r'[,mode]), merge('t'[,min_dist]), subtract('l','r'), cluster('t'[,min_dist]), complement('t'[,'view']),
    and the SQL range join handled by IntervalJoinExec

Argument meaning and output schemas follow R/src/table_function.rs and the providers
(`left_`/`right_` prefixes: nearest.rs:57-78, overlap.rs:106-126; appended `count`/`coverage`
column: count_overlaps.rs:60-66; Int64 start/end + n_intervals: merge.rs:43-48).  pyarrow only
plays DataFusion's part (holding the tables, `take` of payload columns); every interval
computation happens in the HIP kernels.  There is no CPU fallback.
"""
import ctypes as C
import os

import pyarrow as pa
import pyarrow.compute as pc

import pyivx

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libbio_ranges_hip.so")

DEFAULT_COLS = ("contig", "pos_start", "pos_end")
WEAK, STRICT = 0, 1
JOIN_INNER, JOIN_RIGHT_SEMI, JOIN_RIGHT_ANTI = 0, 1, 2


class _ArrowSchema(C.Structure):
    pass


class _ArrowArray(C.Structure):
    pass


_ArrowSchema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
                         ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(_ArrowSchema))),
                         ("dictionary", C.POINTER(_ArrowSchema)), ("release", C.c_void_p), ("private_data", C.c_void_p)]
_ArrowArray._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                        ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)),
                        ("children", C.POINTER(C.POINTER(_ArrowArray))), ("dictionary", C.POINTER(_ArrowArray)),
                        ("release", C.c_void_p), ("private_data", C.c_void_p)]


class _Batch(C.Structure):
    _fields_ = [("array", C.POINTER(_ArrowArray)), ("schema", C.POINTER(_ArrowSchema))]


class _Columns(C.Structure):
    _fields_ = [("keys", C.POINTER(C.c_char_p)), ("n_keys", C.c_int), ("start", C.c_char_p), ("end", C.c_char_p)]


class BioRangesError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        pyivx.lib()                                   # loads libivx_hip.so (and the HIP runtime) first
        _lib = C.CDLL(os.environ.get("BRH_LIB") or _LIB_PATH)     # BRH_LIB: another build of the host library (the sanitizer build)
        _lib.brh_last_error.restype = C.c_char_p
    return _lib


class _Exported:
    """A pyarrow Table exported as one struct array (kept alive for the call)."""

    def __init__(self, table):
        self.batch = table.combine_chunks().to_batches()[0] if table.num_rows else pa.RecordBatch.from_pylist([], schema=table.schema)
        self.arr, self.sch = _ArrowArray(), _ArrowSchema()
        self.batch._export_to_c(C.addressof(self.arr), C.addressof(self.sch))
        self.c = _Batch(C.pointer(self.arr), C.pointer(self.sch))

    def close(self):
        for obj in (self.arr, self.sch):
            if obj.release:
                C.CFUNCTYPE(None, C.c_void_p)(obj.release)(C.addressof(obj))


def _cols(cols):
    keys = cols[0] if isinstance(cols[0], (list, tuple)) else [cols[0]]
    arr = (C.c_char_p * len(keys))(*[k.encode() for k in keys])
    c = _Columns(arr, len(keys), cols[1].encode(), cols[2].encode())
    c._keep = arr
    return c


def _out():
    return _ArrowArray(), _ArrowSchema()


def _import(a, s):
    return pa.Array._import_from_c(C.addressof(a), C.addressof(s))


class Session:
    """create_bio_session() counterpart: owns the GPU context."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib().brh_session_create(C.c_int(device), C.byref(self.h))
        if rc != 0:
            raise BioRangesError(f"no usable gfx950 device (status {rc}); there is no CPU fallback")

    def close(self):
        if self.h:
            lib().brh_session_free(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise BioRangesError(lib().brh_last_error(self.h).decode())

    def metrics(self):
        """BuildProbeJoinMetrics of the session (joins/utils.rs:399-453): build_time, join_time [ms], build_input_batches,
        build_input_rows, build_mem_used, input_batches, input_rows, output_batches, output_rows."""
        m = pyivx.Metrics()
        self._chk(lib().brh_session_metrics(self.h, C.byref(m)))
        return {f: getattr(m, f) for f, _ in pyivx.Metrics._fields_}

    def set_strict_null_contigs(self, on=True):
        """refuse batches with NULL contigs instead of treating them as the reference does (bio_ranges_host.h)"""
        self._chk(lib().brh_session_set_strict_null_contigs(self.h, C.c_int(int(on))))

    def set_memory_limit(self, nbytes):
        """device bytes the session may reserve (MemoryReservation, interval_join.rs:614-639); 0 = no limit"""
        self._chk(lib().brh_session_set_memory_limit(self.h, C.c_uint64(int(nbytes))))

    # ---- count_overlaps / coverage: RangeTableFunction (table_function.rs:521-560)
    def _count(self, left, right, cols_left, cols_right, strict, coverage):
        L, R = _Exported(left), _Exported(right)
        a, s = _out()
        try:
            self._chk(lib().brh_count_overlaps(self.h, L.c, _cols(cols_left), R.c, _cols(cols_right),
                                               C.c_int(STRICT if strict else WEAK), C.c_int(int(coverage)), C.byref(a), C.byref(s)))
        finally:
            L.close(); R.close()
        col = _import(a, s)
        return right.append_column("coverage" if coverage else "count", col)

    def count_overlaps(self, left, right, cols_left=DEFAULT_COLS, cols_right=DEFAULT_COLS, strict=False):
        return self._count(left, right, cols_left, cols_right, strict, False)

    def coverage(self, left, right, cols_left=DEFAULT_COLS, cols_right=DEFAULT_COLS, strict=False):
        return self._count(left, right, cols_left, cols_right, strict, True)

    # ---- nearest: NearestTableFunction (table_function.rs:286-367)
    def nearest(self, left, right, k=1, overlap=True, compute_distance=True, cols_left=DEFAULT_COLS, cols_right=DEFAULT_COLS,
                strict=False):
        L, R = _Exported(left), _Exported(right)
        (la, ls), (ra, rs), (da, ds) = _out(), _out(), _out()
        try:
            self._chk(lib().brh_nearest(self.h, L.c, _cols(cols_left), R.c, _cols(cols_right), C.c_int(STRICT if strict else WEAK),
                                        C.c_uint32(int(k)), C.c_int(int(overlap)), C.c_int(int(compute_distance)),
                                        C.byref(la), C.byref(ls), C.byref(ra), C.byref(rs), C.byref(da), C.byref(ds)))
        finally:
            L.close(); R.close()
        li, ri = _import(la, ls), _import(ra, rs)
        out = {}
        for name in left.schema.names:                       # nearest.rs:57-78: left_*, right_*, distance
            out["left_" + name] = pc.take(left.column(name), li)
        for name in right.schema.names:
            out["right_" + name] = pc.take(right.column(name), ri)
        if compute_distance:
            out["distance"] = _import(da, ds)
        return pa.table(out)

    # ---- IntervalJoinExec (the SQL range join), `build` = the SQL join's left table
    def interval_join(self, build, probe, cols_build=DEFAULT_COLS, cols_probe=DEFAULT_COLS, join_type=JOIN_INNER,
                      strict_predicate=False, nearest_algorithm=False):
        B, P = _Exported(build), _Exported(probe)
        (ba, bs), (pa_, ps) = _out(), _out()
        try:
            self._chk(lib().brh_interval_join(self.h, B.c, _cols(cols_build), P.c, _cols(cols_probe), C.c_int(join_type),
                                              C.c_int(int(strict_predicate)), C.c_int(int(nearest_algorithm)),
                                              C.byref(ba), C.byref(bs), C.byref(pa_), C.byref(ps)))
        finally:
            B.close(); P.close()
        return _import(ba, bs), _import(pa_, ps)

    def sql_range_join(self, build, probe, cols_build=DEFAULT_COLS, cols_probe=DEFAULT_COLS, strict_predicate=False,
                       nearest_algorithm=False, device_take=False):
        """SELECT * FROM build JOIN probe ON key = key AND range predicate  (build columns, then probe columns)."""
        bi, pi = self.interval_join(build, probe, cols_build, cols_probe, JOIN_INNER, strict_predicate, nearest_algorithm)
        if device_take:                                      # payload gather on the GPU as well
            cols = list(self.take_table(build, bi).values()) + list(self.take_table(probe, pi).values())
        else:
            cols = [pc.take(build.column(n), bi) for n in build.schema.names] + [pc.take(probe.column(n), pi) for n in probe.schema.names]
        names = [f"l.{n}" for n in build.schema.names] + [f"r.{n}" for n in probe.schema.names]
        return pa.table(cols, names=names)

    def join_stream(self, build, cols_build=DEFAULT_COLS, cols_probe=DEFAULT_COLS, strict_predicate=False, coalesce_rows=0,
                    join_type="inner", max_output_rows=0):
        """IntervalJoinStream as a push interface: index `build` once, then push probe batches; see JoinStream.
        join_type: "inner" | "right_semi" | "right_anti" | "nearest" (Algorithm::CoitreesNearest);
        max_output_rows: 0 = one result per group, N = the low-memory stream's output budget, "env" = the reference's
        default (BIO_MAX_OUTPUT_BATCH_SIZE or 100000)."""
        return JoinStream(self, build, cols_build, cols_probe, strict_predicate, coalesce_rows, join_type, max_output_rows)

    # ---- overlap UDTF (overlap.rs:154-226): FROM right AS b, left AS a  => the user's RIGHT table is the build side
    def overlap(self, left, right, mode="join", cols_left=DEFAULT_COLS, cols_right=DEFAULT_COLS, strict=False):
        if mode == "left":                                   # LeftDistinct: RIGHT SEMI JOIN, left rows that have a match
            _, pi = self.interval_join(right, left, cols_right, cols_left, JOIN_RIGHT_SEMI, strict)
            return left.take(pi)
        bi, pi = self.interval_join(right, left, cols_right, cols_left, JOIN_INNER, strict)
        if mode == "left_all":                               # Inner projecting left only (overlap multiplicity kept)
            return left.take(pi)
        out = {"left_" + n: pc.take(left.column(n), pi) for n in left.schema.names}
        out.update({"right_" + n: pc.take(right.column(n), bi) for n in right.schema.names})
        return pa.table(out)

    # ---- merge (table_function.rs:441-464)
    def merge(self, table, min_dist=0, cols=DEFAULT_COLS, strict=False):
        if min_dist < 0:
            raise BioRangesError(f"merge() min_dist must be >= 0, got {min_dist}")          # table_function.rs:237
        T = _Exported(table)
        outs = [_out() for _ in range(4)]
        try:
            self._chk(lib().brh_merge(self.h, T.c, _cols(cols), C.c_int64(int(min_dist)), C.c_int(STRICT if strict else WEAK),
                                      *[C.byref(x) for pair in outs for x in pair]))
        finally:
            T.close()
        key = cols[0] if isinstance(cols[0], str) else cols[0][0]
        return pa.table([_import(*o) for o in outs], names=[key, cols[1], cols[2], "n_intervals"])

    # ---- subtract (table_function.rs:574-612); extra left columns are carried along (subtract.rs:501-527)
    def subtract(self, left, right, cols_left=DEFAULT_COLS, cols_right=DEFAULT_COLS, strict=False):
        L, R = _Exported(left), _Exported(right)
        outs = [_out() for _ in range(4)]
        try:
            self._chk(lib().brh_subtract(self.h, L.c, _cols(cols_left), R.c, _cols(cols_right), C.c_int(STRICT if strict else WEAK),
                                         *[C.byref(x) for pair in outs for x in pair]))
        finally:
            L.close(); R.close()
        contig, start, end, row = [_import(*o) for o in outs]
        key = cols_left[0] if isinstance(cols_left[0], str) else cols_left[0][0]
        cols = []
        for n in left.schema.names:
            cols.append(contig if n == key else start if n == cols_left[1] else end if n == cols_left[2] else pc.take(left.column(n), row))
        return pa.table(cols, names=left.schema.names)


    # ---- f3: compute::take of payload columns on the device (interval_join.rs:1655-1667, nearest.rs:469-482)
    def take(self, column, idx):
        """column: pyarrow Array / ChunkedArray of any flat, dictionary-encoded or nested (struct / list / map) type;
        idx: UInt32 array (nulls -> null rows).  Raises for layouts the gather does not cover (unions, run-end encoding)."""
        if isinstance(column, pa.ChunkedArray):
            column = column.combine_chunks() if column.num_chunks != 1 else column.chunk(0)
        if isinstance(idx, pa.ChunkedArray):
            idx = idx.combine_chunks()
        ca, cs, ia, is_ = _ArrowArray(), _ArrowSchema(), _ArrowArray(), _ArrowSchema()
        column._export_to_c(C.addressof(ca), C.addressof(cs))
        idx._export_to_c(C.addressof(ia), C.addressof(is_))
        a, sc = _out()
        try:
            self._chk(lib().brh_take(self.h, C.byref(ca), C.byref(cs), C.byref(ia), C.byref(is_), C.byref(a), C.byref(sc)))
        finally:
            for obj in (ca, cs, ia, is_):
                if obj.release:
                    C.CFUNCTYPE(None, C.c_void_p)(obj.release)(C.addressof(obj))
        return _import(a, sc)

    def take_table(self, table, idx, prefix=""):
        return {prefix + n: self.take(table.column(n), idx) for n in table.schema.names}

    # ---- cluster (table_function.rs:574-612; ClusterProvider cluster.rs:29-82)
    def cluster(self, table, min_dist=0, cols=DEFAULT_COLS, strict=False):
        if min_dist < 0:
            raise BioRangesError(f"cluster() min_dist must be >= 0, got {min_dist}")        # table_function.rs:237
        T = _Exported(table)
        outs = [_out() for _ in range(7)]
        try:
            self._chk(lib().brh_cluster(self.h, T.c, _cols(cols), C.c_int64(int(min_dist)), C.c_int(STRICT if strict else WEAK),
                                        *[C.byref(x) for pair in outs for x in pair]))
        finally:
            T.close()
        contig, start, end, row, cl, cs, ce = [_import(*o) for o in outs]
        key = cols[0] if isinstance(cols[0], str) else cols[0][0]
        if table.num_columns > 3:                            # extra columns: every input field kept as it is (cluster.rs:50-53)
            out_cols = [pc.take(table.column(n), row) for n in table.schema.names]
            names = list(table.schema.names)
        else:
            out_cols, names = [contig, start, end], [key, cols[1], cols[2]]
        return pa.table(out_cols + [cl, cs, ce], names=names + ["cluster", "cluster_start", "cluster_end"])

    # ---- complement (table_function.rs:614-786; ComplementProvider complement.rs:28-75)
    def complement(self, table, view=None, cols=DEFAULT_COLS, view_cols=None, strict=False):
        T = _Exported(table)
        V = _Exported(view) if view is not None else None
        outs = [_out() for _ in range(3)]
        try:
            vb = V.c if V is not None else _Batch(None, None)
            self._chk(lib().brh_complement(self.h, T.c, _cols(cols), vb, _cols(view_cols or cols), C.c_int(STRICT if strict else WEAK),
                                           *[C.byref(x) for pair in outs for x in pair]))
        finally:
            T.close()
            if V is not None:
                V.close()
        key = cols[0] if isinstance(cols[0], str) else cols[0][0]
        return pa.table([_import(*o) for o in outs], names=[key, cols[1], cols[2]])


class JoinStream:
    """brh_join_stream: the build side indexed once, probe RecordBatches pushed one by one and coalesced into
    groups before they go to the GPU.  push()/finish() return the results that became ready, each a dict
    first_batch, n_batches, group_done, build_idx, probe_idx (rows counted over the group's concatenated batches),
    batch_offsets."""

    JOIN_TYPES = {"inner": 0, "right_semi": 1, "right_anti": 2, "nearest": 3}

    def __init__(self, session, build, cols_build, cols_probe, strict_predicate, coalesce_rows, join_type="inner", max_output_rows=0):
        self.session = session
        budget = (1 << 64) - 1 if max_output_rows == "env" else int(max_output_rows)
        self.h = C.c_void_p()
        B = _Exported(build)
        try:
            session._chk(lib().brh_join_stream_open(session.h, B.c, _cols(cols_build), _cols(cols_probe), C.c_int(int(strict_predicate)),
                                                    C.c_uint64(int(coalesce_rows)), C.c_int(self.JOIN_TYPES[join_type]), C.c_uint64(budget),
                                                    C.byref(self.h)))
        finally:
            B.close()

    def _drain(self, n):
        out = []
        for _ in range(n):
            first, nb, done = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
            (ba, bs), (pa_, ps), (oa, os_) = _out(), _out(), _out()
            self.session._chk(lib().brh_join_stream_next(self.h, C.byref(first), C.byref(nb), C.byref(done), C.byref(ba), C.byref(bs),
                                                         C.byref(pa_), C.byref(ps), C.byref(oa), C.byref(os_)))
            out.append({"first_batch": first.value, "n_batches": nb.value, "group_done": bool(done.value), "build_idx": _import(ba, bs),
                        "probe_idx": _import(pa_, ps), "batch_offsets": _import(oa, os_)})
        return out

    def push(self, batch):
        P = _Exported(batch if isinstance(batch, pa.Table) else pa.Table.from_batches([batch]))
        n = C.c_int(0)
        try:
            self.session._chk(lib().brh_join_stream_push(self.h, P.c, C.byref(n)))
        finally:
            P.close()
        return self._drain(n.value)

    def finish(self):
        n = C.c_int(0)
        self.session._chk(lib().brh_join_stream_finish(self.h, C.byref(n)))
        return self._drain(n.value)

    def close(self):
        if self.h:
            lib().brh_join_stream_close(self.h)
            self.h = C.c_void_p()


def check_position_column(table, column, as_i64=False):
    """PosArray::resolve / resolve_i64 checks alone (no GPU): returns None or the error text."""
    T = _Exported(table)
    buf = C.create_string_buffer(512)
    try:
        rc = lib().brh_check_position_column(None, T.c, column.encode(), C.c_int(int(as_i64)), buf, C.c_int(512))
    finally:
        T.close()
    return None if rc == 0 else buf.value.decode()


def check_contig_column(table, column):
    """ContigArray checks alone (no GPU): Utf8 / LargeUtf8 / Utf8View (NULL contigs only fail under
    BIO_STRICT_NULL_CONTIGS=1 / a strict session); returns None or the error text."""
    T = _Exported(table)
    buf = C.create_string_buffer(512)
    try:
        rc = lib().brh_check_contig_column(None, T.c, column.encode(), buf, C.c_int(512))
    finally:
        T.close()
    return None if rc == 0 else buf.value.decode()
