/*
 * ivx.h -- C ABI of the MI355X-native interval build-and-probe library
 *          (libivx_hip.so, hand-written HIP for gfx950).
 *
 * This is the drop-in boundary for ONE path of
 * biodatageeks/datafusion-bio-functions: the interval index build + probe
 * behind `datafusion-bio-function-ranges` (IntervalJoinExec and the
 * overlap / count_overlaps / coverage / nearest / merge / subtract table
 * functions).  The reference has no FFI of its own (no extern "C" anywhere);
 * each entry point below names the reference Rust function whose inner loop
 * it replaces -- the call a maintainer would swap for an `unsafe extern "C"`
 * call in the stream implementation (see INTEGRATION.md for the Rust stub).
 * R/ = datafusion/bio-function-ranges/ in the reference tree.
 *
 * Conventions
 *  - Plain pointers and sizes only.  No Arrow, torch or C++ types.
 *  - `mem` says where EVERY data pointer of that call lives:
 *      IVX_MEM_HOST   host memory (Arrow buffers); the library copies them to
 *                     device scratch with hipMemcpyAsync on the ctx stream;
 *      IVX_MEM_DEVICE device memory of the ctx's GPU (e.g. buffers the caller
 *                     already keeps in HBM); no copies, kernels run on the
 *                     ctx stream.
 *    Scalar out-parameters (`uint64_t *total`, `*n_out`, ...) are always host
 *    pointers; writing them synchronises the ctx stream.
 *  - Keys never cross the boundary as strings: the host maps the equi-key
 *    column(s) (contig, or contig+strand, ...) to dense ids 0..n_keys-1.  The
 *    reference groups rows by a 64-bit hash of the key columns and never
 *    compares key values (interval_join.rs:857, :922-928); dense ids differ
 *    from that only under a hash collision (documented divergence).
 *    `key == NULL` means "one key" (range-only join: interval_join.rs on
 *    [(lit 1, lit 1)], bio_physical_planner.rs:125-146).
 *  - Join/count/coverage/nearest coordinates are int32, closed [start,end]
 *    (array_utils.rs:66-135 `resolve()`); merge/subtract are int64
 *    (`resolve_i64()`, :137-172).  Range checks and null checks stay in the
 *    host layer, with the reference's wording.
 *  - "build" is the reference's left / collected side, "probe" the streamed
 *    right side.  Row indices are uint32 (interval_join.rs:759-760, :1616).
 *  - `strict` for count/coverage/nearest shrinks the QUERY (qs+1, qe-1) as the
 *    UDTFs do (interval_tree.rs:185-188, :253-256; nearest.rs:341-344).  The
 *    SQL join path instead rewrites `<`/`>` into `end-1` on both sides
 *    (intervals.rs:85-115); callers of ivx_probe_overlap_* pass already
 *    adjusted columns, exactly as IntervalJoinExec evaluates them.
 *  - Every function returns an ivx_status; ivx_last_error(ctx) holds the text
 *    (maps to DataFusionError::Execution).
 *  - An ivx_index is immutable after build and may be probed concurrently
 *    from several host threads, each with its OWN ivx_ctx on the same device
 *    (the reference shares Arc<JoinLeftData> across partitions,
 *    interval_join.rs:466-480).
 *  - There is NO CPU fallback: without a usable gfx950 device
 *    ivx_ctx_create fails with IVX_ERR_NO_DEVICE.
 */
#ifndef IVX_H
#define IVX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ivx_ctx ivx_ctx;
typedef struct ivx_index ivx_index;
typedef int32_t ivx_status;

enum {
    IVX_OK = 0,
    IVX_ERR_INVALID = 1,      /* bad argument (null pointer, key id >= n_keys, min_dist < 0 ...) */
    IVX_ERR_NO_DEVICE = 2,    /* no HIP device / not gfx950 */
    IVX_ERR_HIP = 3,          /* a HIP runtime call failed; text in ivx_last_error */
    IVX_ERR_OOM = 4,          /* device or pinned allocation failed (ResourcesExhausted) */
    IVX_ERR_CAPACITY = 5,     /* caller's output buffers too small; the needed size is returned */
    IVX_ERR_UNSUPPORTED = 6   /* e.g. wrong index kind for this probe */
};

enum { IVX_MEM_HOST = 0, IVX_MEM_DEVICE = 1 };

/* which reference structure the index replaces */
enum {
    IVX_KIND_OVERLAP = 0,   /* COITree per key            interval_join.rs:745-763 (a1+a2)        */
    IVX_KIND_COUNT = 1,     /* CountOverlapIndex          interval_tree.rs:20-39, :113-143        */
    IVX_KIND_COVERAGE = 2,  /* merged COITree             interval_tree.rs:52-111 (coverage=true) */
    IVX_KIND_NEAREST = 3    /* NearestIntervalIndex       nearest_index.rs:44-71                  */
};

#define IVX_NULL_IDX 0xFFFFFFFFu   /* NULL build row (interval_join.rs:1233 u32::MAX marker) */

/* ---- context ----------------------------------------------------------- */
ivx_status ivx_ctx_create(int device_ordinal, ivx_ctx **out);
void       ivx_ctx_free(ivx_ctx *ctx);
const char *ivx_last_error(const ivx_ctx *ctx);
/* run on a caller-owned hipStream_t, used verbatim (e.g. torch's current
 * stream; NULL is HIP's default stream, which orders against the caller's
 * other default-stream work).  A fresh ctx runs on its own non-blocking
 * stream; ivx_ctx_use_own_stream switches back to it. */
ivx_status ivx_ctx_set_stream(ivx_ctx *ctx, void *hip_stream);
ivx_status ivx_ctx_use_own_stream(ivx_ctx *ctx);
ivx_status ivx_ctx_synchronize(ivx_ctx *ctx);
/* Build overlap (off by default; IVX_BUILD_OVERLAP=1 turns it on for new contexts).  With it on, ivx_index_build of an
 * IVX_KIND_OVERLAP index from IVX_MEM_DEVICE columns returns as soon as the per-key tables and the region layout are final --
 * all that the routing pass of a big probe batch reads -- and lets the rest of the build (cell count, scan, scatter, region
 * descriptors) run on a second stream of the context.  The next big ivx_probe_overlap_count / _fill call routes its rows
 * beside that tail and only its probe kernel waits for it; every other use of the index (any context) and every other call
 * on this context is ordered behind the tail first, so results never change.  What changes is the caller's side of the
 * IVX_MEM_DEVICE contract: the build columns must stay unchanged until the index's first probe call has returned, or
 * ivx_ctx_synchronize (which waits for the tail too) -- as IntervalJoinExec keeps the build batches alive in JoinLeftData
 * for the whole join (interval_join.rs:466-480).  ivx_index_free waits for the tail by itself. */
ivx_status ivx_ctx_set_build_overlap(ivx_ctx *ctx, int on);
/* device time (ms, hipEvent) of the kernels of the last call on this ctx */
double     ivx_ctx_last_kernel_ms(const ivx_ctx *ctx);
const char *ivx_version(void);

/* BuildProbeJoinMetrics (joins/utils.rs:399-453) under the reference's names, accumulated over the calls made on this
 * context since creation / the last reset: build_* by ivx_index_build, the rest by the probe and sweep calls (one
 * call = one input batch and, if it returns rows, one output batch).  Times are host wall time of the calls in ms
 * (what the reference's timers bracket); build_mem_used = device bytes of the indexes built. */
typedef struct ivx_metrics {
    double build_time, join_time;
    uint64_t build_input_batches, build_input_rows, build_mem_used;
    uint64_t input_batches, input_rows, output_batches, output_rows;
} ivx_metrics;
ivx_status ivx_ctx_metrics(const ivx_ctx *ctx, ivx_metrics *out);
void       ivx_ctx_reset_metrics(ivx_ctx *ctx);
/* Device memory this context may hold at once (0 = no limit): its scratch buffers, the index it is building and every
 * index it has built that is still alive -- an index stays reserved against the limit of the context that built it until
 * ivx_index_free, as the reference holds the build side's MemoryReservation until the join stream ends
 * (interval_join.rs:614-639; try_grow fails with ResourcesExhausted).  A call that would go over the limit returns
 * IVX_ERR_OOM and allocates nothing.  Not counted: the caller's own buffers, and freed indexes' buffers waiting in the
 * library's recycling pool (ivx_ctx_trim returns those to the device). */
ivx_status ivx_ctx_set_memory_limit(ivx_ctx *ctx, uint64_t bytes);
/* What counts against that limit right now (MemoryReservation::size): scratch + the live indexes this context built. */
uint64_t   ivx_ctx_reserved_bytes(const ivx_ctx *ctx);
/* Give scratch back to the device: waits for the context's stream, then frees scratch buffers, largest first, until
 * at most keep_bytes remain (0 = all), and empties the device's pool of recycled index buffers.  Scratch is grow-only
 * otherwise (a 10^9-row sweep leaves ~100 GB behind); the reference returns its reservation when the stream ends
 * (interval_join.rs:614-639) -- call this where the Rust side drops a stream.  Drops the state a sizing / count call
 * left for its fill call. */
ivx_status ivx_ctx_trim(ivx_ctx *ctx, uint64_t keep_bytes);

/* ---- index build: replaces collect_left_input's update_hashmap +
 *      IntervalJoinAlgorithm::new (interval_join.rs:584-668, :745-847, :903-931),
 *      build_count_index_from_batches / build_coitree_from_batches
 *      (interval_tree.rs:75-143) and build_nearest_indexes (nearest.rs:498-547) */
ivx_status ivx_index_build(ivx_ctx *ctx, int kind, int mem,
                           const uint32_t *key /* nullable */, const int32_t *start, const int32_t *end,
                           uint64_t n, uint32_t n_keys, ivx_index **out);
/* The caller must have synchronised (ivx_ctx_synchronize) every ctx that probed the index with
 * IVX_MEM_DEVICE buffers: those calls may return with kernels still in flight, and the index's
 * device buffers go back to a pool that later builds draw from. */
void       ivx_index_free(ivx_index *ix);
uint64_t   ivx_index_rows(const ivx_index *ix);
uint64_t   ivx_index_device_bytes(const ivx_index *ix);

/* ---- a3: IntervalJoinAlgorithm::get + probe loop
 *      (interval_join.rs:849-900, :1614-1653).  Index kind OVERLAP. ---------
 * count: total pairs and, if per_row != NULL, the reference's rle_right
 *        (matches per probe row, uint32). */
ivx_status ivx_probe_overlap_count(ivx_ctx *ctx, const ivx_index *ix, int mem,
                                   const uint32_t *key, const int32_t *start, const int32_t *end, uint64_t n,
                                   uint32_t *per_row /* nullable */, uint64_t *total);
/* fill: the (build_idx, probe_idx) pairs = the reference's (left_indexes,
 *       index_right) before compute::take.  Pair ORDER is unspecified (the
 *       reference pins only the row multiset).  If more than cap pairs exist
 *       nothing useful is written, *written = pairs needed and the call
 *       returns IVX_ERR_CAPACITY.  cap also serves as the hint for the expected
 *       pairs per row (cap / n): size it from ivx_probe_overlap_count, not with
 *       a blanket maximum, or large batches run with needlessly small rounds.
 *       A total-only count call (per_row = NULL) of a large batch leaves the
 *       probe rows routed to the index regions (and, for IVX_MEM_HOST, copied to
 *       the device) in the context: the fill call for the same index, column
 *       pointers, n and stream that is the next call on that context skips that
 *       work (and takes the counted total, not cap, as the density hint) --
 *       the three columns must not change between the two calls.  The state
 *       serves ONE successful fill call (it survives an IVX_ERR_CAPACITY
 *       retry); any other call drops it and a fill call does everything itself.
 *       An IVX_MEM_HOST fill of a big batch (16 M rows or more) that has no such
 *       state to use runs in chunks: the pairs of one chunk are copied to
 *       build_idx / probe_idx by a helper thread while the next chunk's columns
 *       are uploaded (both directions of the link at once).  On IVX_ERR_CAPACITY
 *       the buffers may then hold the pairs of the first chunks. */
ivx_status ivx_probe_overlap_fill(ivx_ctx *ctx, const ivx_index *ix, int mem,
                                  const uint32_t *key, const int32_t *start, const int32_t *end, uint64_t n,
                                  uint32_t *build_idx, uint32_t *probe_idx, uint64_t cap, uint64_t *written);
/* a3': RightSemi / RightAnti (interval_join.rs:1014-1024, :1433-1447):
 *      exists[i] = 1 iff probe row i has a match. */
ivx_status ivx_probe_exists(ivx_ctx *ctx, const ivx_index *ix, int mem,
                            const uint32_t *key, const int32_t *start, const int32_t *end, uint64_t n,
                            uint8_t *exists);

/* ---- a4: CountOverlapIndex::query_count in get_count_stream
 *      (interval_tree.rs:41-49, :249-267).  Index kind COUNT. */
ivx_status ivx_probe_count(ivx_ctx *ctx, const ivx_index *ix, int mem,
                           const uint32_t *key, const int32_t *start, const int32_t *end, uint64_t n,
                           int strict, int64_t *out);
/* ---- a5: get_coverage in get_stream (interval_tree.rs:145-152, :181-208).
 *      Index kind COVERAGE. */
ivx_status ivx_probe_coverage(ivx_ctx *ctx, const ivx_index *ix, int mem,
                              const uint32_t *key, const int32_t *start, const int32_t *end, uint64_t n,
                              int strict, int64_t *out);

/* ---- a6: NearestIntervalIndex::nearest_one / nearest_k in get_nearest_stream
 *      (nearest_index.rs:91-235, nearest.rs:330-456) and Algorithm::CoitreesNearest
 *      (interval_join.rs:864-870).  Index kind NEAREST.
 *      Every probe row yields max(1, found) output rows, in probe order:
 *      build_idx (IVX_NULL_IDX = NULL left columns), probe_idx, and if
 *      distance != NULL candidate_distance on the RAW probe coordinates
 *      (-1 = NULL).  cap >= n*max(k,1) always suffices. */
ivx_status ivx_probe_nearest(ivx_ctx *ctx, const ivx_index *ix, int mem,
                             const uint32_t *key, const int32_t *start, const int32_t *end, uint64_t n,
                             int strict, uint32_t k, int include_overlaps,
                             uint32_t *build_idx, uint32_t *probe_idx, int64_t *distance /* nullable */,
                             uint64_t cap, uint64_t *rows);

/* ---- a7+a8: StreamCollector sort + MergeStream sweep
 *      (grouped_stream.rs:50-113, merge.rs:282-350).
 *      Output rows ordered by (key id, start); give key ids in the byte order
 *      of the contig names to reproduce the reference's group order.
 *      cap >= n always suffices. */
ivx_status ivx_merge(ivx_ctx *ctx, int mem,
                     const uint32_t *key /* nullable */, const int64_t *start, const int64_t *end, uint64_t n,
                     uint32_t n_keys, int64_t min_dist, int strict,
                     uint32_t *out_key, int64_t *out_start, int64_t *out_end, int64_t *out_n,
                     uint64_t cap, uint64_t *n_out);

/* ---- a7+a9: SubtractStream / SubtractStreamExtra sweep
 *      (subtract.rs:390-462, :575-655).  out_row (nullable) = the left input row
 *      of each fragment (for the extra-columns `take`).  Passing cap = 0 with
 *      NULL outputs only counts (the sizing call).  The sizing call leaves its
 *      sorted sides and per-row output offsets in the context: a fill call with
 *      the same input pointers, sizes, n_keys, strict and stream that is the next
 *      sort/sweep call on that context runs the output pass only -- the six input
 *      columns must not change between the two calls (the size would be stale
 *      anyway).  The state serves ONE successful fill call (it survives an
 *      IVX_ERR_CAPACITY retry); any other call on the context drops it; a fill call
 *      then (or one made without a sizing call) does all the work itself. */
ivx_status ivx_subtract(ivx_ctx *ctx, int mem,
                        const uint32_t *lkey, const int64_t *lstart, const int64_t *lend, uint64_t nl,
                        const uint32_t *rkey, const int64_t *rstart, const int64_t *rend, uint64_t nr,
                        uint32_t n_keys, int strict,
                        uint32_t *out_key, int64_t *out_start, int64_t *out_end, uint32_t *out_row,
                        uint64_t cap, uint64_t *n_out);

/* ---- f1: ClusterStream / ClusterStreamExtra (cluster.rs:443-477, :598-661, :813-884) and the
 *      ClusterIdCoordinator offsets (cluster.rs:380-420).  Output has exactly n rows, sorted by
 *      (key, start, end, input row): the sorted row itself (out_key/out_start/out_end, and
 *      out_row = its input row for the extra-columns `take`), the id of its cluster and the
 *      cluster's extent.  Any output may be NULL.  Ids count clusters from 0 in output order
 *      (keys ascending = contig names in byte order); a caller that holds only some of the
 *      contigs (one DataFusion partition, one GPU of a sharded job) passes key_base[n_keys] =
 *      the global id of each key's first cluster, computed from everybody's key_clusters
 *      (out, [n_keys], clusters per key) -- call once with the row outputs NULL to get them.
 *      *n_clusters = clusters in this call. */
ivx_status ivx_cluster(ivx_ctx *ctx, int mem,
                       const uint32_t *key /* nullable */, const int64_t *start, const int64_t *end, uint64_t n,
                       uint32_t n_keys, int64_t min_dist, int strict, const int64_t *key_base /* nullable */,
                       uint32_t *out_key, int64_t *out_start, int64_t *out_end, uint32_t *out_row,
                       int64_t *out_cluster, int64_t *out_cluster_start, int64_t *out_cluster_end,
                       uint64_t *key_clusters /* nullable */, uint64_t *n_clusters);

/* ---- f2: ComplementStream (complement.rs:297-357, :394-465).  Input rows are merged per key
 *      (strict: start < cur_end, else <=), gaps are emitted against the key's view intervals
 *      (vkey/vstart/vend, nv rows; sorted per key, not merged); a key with input rows and no
 *      view row gets the implicit view [0, INT64_MAX).  Output order as the reference: keys
 *      with input rows ascending, then keys that only have view rows (their views, whole).
 *      Passing cap = 0 with NULL outputs only counts. */
ivx_status ivx_complement(ivx_ctx *ctx, int mem,
                          const uint32_t *key /* nullable */, const int64_t *start, const int64_t *end, uint64_t n,
                          const uint32_t *vkey /* nullable */, const int64_t *vstart, const int64_t *vend, uint64_t nv,
                          uint32_t n_keys, int strict,
                          uint32_t *out_key, int64_t *out_start, int64_t *out_end,
                          uint64_t cap, uint64_t *n_out);

/* ---- f3: `compute::take` of payload columns with the index arrays the probes return
 *      (interval_join.rs:1655-1667, nearest.rs:469-482).  idx[i] == IVX_NULL_IDX or a null source
 *      slot (src_valid_bits: Arrow validity bitmap of the source, bit offset 0, nullable) gives
 *      out_valid[i] = 0 (one byte per output row, nullable) and zero bytes / an empty string;
 *      any other idx[i] >= n_src is IVX_ERR_INVALID.
 *      fixed: width = bytes per element, one of 1, 2, 4, 8, 16, 32. */
ivx_status ivx_take_fixed(ivx_ctx *ctx, int mem, const void *src, uint32_t width, uint64_t n_src,
                          const uint8_t *src_valid_bits, const uint32_t *idx, uint64_t n,
                          void *out, uint8_t *out_valid);

/*      The inverse: out[idx[i]] = src[i] for i < n (idx[i] >= n_out is IVX_ERR_INVALID; rows no index names keep their
 *      contents; equal indices: one of the values).  What puts a per-row result column (count_overlaps, coverage,
 *      nearest) computed on a SHARD of the probe rows back in the order of the whole input -- the reference's partitioned
 *      forms return their batches in stream order (count_overlaps.rs:143-153, R/tests/integration_test.rs:3783-3890);
 *      with contigs sharded over GPUs the rows come back as (row, value) lists instead. */
ivx_status ivx_scatter_fixed(ivx_ctx *ctx, int mem, const void *src, uint32_t width, const uint32_t *idx, uint64_t n,
                             void *out, uint64_t n_out);

/*      Boolean columns (bit-packed, LSB first): out_bits gets (n+7)/8 bytes. */
ivx_status ivx_take_bits(ivx_ctx *ctx, int mem, const uint8_t *src_bits, uint64_t n_src, const uint8_t *src_valid_bits,
                         const uint32_t *idx, uint64_t n, uint8_t *out_bits, uint8_t *out_valid);

/*      Utf8 / Binary (large = 0, int32 offsets) and LargeUtf8 / LargeBinary (large = 1, int64).
 *      out_offsets[n+1] is written whenever given; *data_bytes always returns the bytes needed.
 *      out_data = NULL sizes only; data_cap < *data_bytes is IVX_ERR_CAPACITY. */
ivx_status ivx_take_utf8(ivx_ctx *ctx, int mem, int large, const void *offsets, const uint8_t *data, uint64_t n_src,
                         uint64_t src_data_bytes, const uint8_t *src_valid_bits, const uint32_t *idx, uint64_t n,
                         void *out_offsets, uint8_t *out_data, uint64_t data_cap, uint64_t *data_bytes,
                         uint8_t *out_valid);

/*      Utf8View / BinaryView: views = [n_src] 16-byte views, data_bufs[n_bufs] the variadic data buffers
 *      with their sizes.  The output is self-contained: out_views[n] plus ONE data buffer holding the
 *      gathered strings longer than 12 bytes (views rewritten to buffer 0).  out_data = NULL sizes only
 *      (*data_bytes); more than 2^31-1 gathered bytes is IVX_ERR_UNSUPPORTED. */
ivx_status ivx_take_view(ivx_ctx *ctx, int mem, const void *views, const uint8_t *const *data_bufs, const uint64_t *data_buf_bytes,
                         uint32_t n_bufs, uint64_t n_src, const uint8_t *src_valid_bits, const uint32_t *idx, uint64_t n,
                         void *out_views, uint8_t *out_data, uint64_t data_cap, uint64_t *data_bytes, uint8_t *out_valid);

#ifdef __cplusplus
}
#endif
#endif
