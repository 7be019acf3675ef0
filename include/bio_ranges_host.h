/*
 * bio_ranges_host.h -- host-side mirror of the reference's operator interface over the
 * Arrow C Data Interface (libbio_ranges_hip.so, plain C++; no Arrow library needed).
 *
 * The reference's operators take Arrow RecordBatches and column-name triples
 * (contig, start, end); this layer does what their host code does before and after
 * the hot loop, then calls the HIP kernels through include/ivx.h:
 *   - contig column access: Utf8 / LargeUtf8 / Utf8View   (R/src/array_utils.rs:10-24, :196-229)
 *   - position columns: Int32 / Int64 / UInt32 / UInt64, null check and checked
 *     narrowing with the reference's error text            (array_utils.rs:33-172, :231-295)
 *   - key dictionary: contig (or several key columns) -> dense uint32 ids, assigned in
 *     byte order of the names so merge/subtract emit groups like
 *     StreamCollector::take_groups                          (R/src/grouped_stream.rs:105-113)
 *   - output columns as Arrow arrays (count/coverage Int64; UInt32 index arrays with
 *     validity for nearest; Utf8 + Int64 columns for merge/subtract).
 * Gathering payload columns with the returned index arrays (`compute::take`) stays with
 * the caller, exactly as in the reference (interval_join.rs:1655-1667, nearest.rs:469-482).
 *
 * Every function returns 0 on success; brh_last_error() gives the message
 * (DataFusionError::Execution text).  Input batches are borrowed, outputs are
 * released by the caller through the usual ArrowArray.release callback.
 */
#ifndef BIO_RANGES_HOST_H
#define BIO_RANGES_HOST_H
#include <stdint.h>
#include "ivx.h"   /* ivx_metrics */
#ifdef __cplusplus
extern "C" {
#endif

#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char *format; const char *name; const char *metadata; int64_t flags; int64_t n_children;
    struct ArrowSchema **children; struct ArrowSchema *dictionary;
    void (*release)(struct ArrowSchema *); void *private_data;
};
struct ArrowArray {
    int64_t length; int64_t null_count; int64_t offset; int64_t n_buffers; int64_t n_children;
    const void **buffers; struct ArrowArray **children; struct ArrowArray *dictionary;
    void (*release)(struct ArrowArray *); void *private_data;
};
#endif

typedef struct brh_session brh_session;

/* one table = one struct-typed ArrowArray + its ArrowSchema (RecordBatch export) */
typedef struct { const struct ArrowArray *array; const struct ArrowSchema *schema; } brh_batch;

/* column names: keys[0..n_keys) are the equi-key columns (the contig; more for multi-column
 * joins such as contig+strand, R/tests/integration_test.rs:393-397), then start and end */
typedef struct { const char *const *keys; int n_keys; const char *start; const char *end; } brh_columns;

int  brh_session_create(int device_ordinal, brh_session **out);      /* fails without a gfx950 device */
void brh_session_free(brh_session *s);
const char *brh_last_error(const brh_session *s);

/* FilterOp (R/src/filter_op.rs:4-10) */
enum { BRH_WEAK = 0, BRH_STRICT = 1 };
/* JoinType subset with real semantics in IntervalJoinExec (interval_join.rs:1014-1024) */
enum { BRH_JOIN_INNER = 0, BRH_JOIN_RIGHT_SEMI = 1, BRH_JOIN_RIGHT_ANTI = 2,
       BRH_JOIN_NEAREST = 3 /* join stream only: Inner with Algorithm::CoitreesNearest (interval_join.rs:864-870) */ };
#define BRH_MAX_OUTPUT_ENV UINT64_MAX   /* join stream: take the output budget from BIO_MAX_OUTPUT_BATCH_SIZE (default 100000) */

/* count_overlaps('left','right') / coverage(...): CountOverlapsProvider (R/src/count_overlaps.rs:107-169),
 * get_count_stream / get_stream (interval_tree.rs:155-280).  `left` is indexed, `right` streamed;
 * out = the Int64 column ("count"/"coverage") to append to `right`. */
int brh_count_overlaps(brh_session *s, brh_batch left, brh_columns lcols, brh_batch right, brh_columns rcols,
                       int filter_op, int coverage, struct ArrowArray *out, struct ArrowSchema *out_schema);

/* nearest('left','right',k,overlap,compute_distance): NearestProvider / get_nearest_stream
 * (R/src/nearest.rs:127-166, :266-496).  Outputs: left_idx UInt32 (nullable), right_idx UInt32,
 * distance Int64 (nullable; NULL pointers when compute_distance == 0). */
int brh_nearest(brh_session *s, brh_batch left, brh_columns lcols, brh_batch right, brh_columns rcols,
                int filter_op, uint32_t k, int include_overlaps, int compute_distance,
                struct ArrowArray *left_idx, struct ArrowSchema *left_idx_schema,
                struct ArrowArray *right_idx, struct ArrowSchema *right_idx_schema,
                struct ArrowArray *distance, struct ArrowSchema *distance_schema);

/* IntervalJoinExec (interval_join.rs:442-550, :1418-1677): `build` is the SQL join's left side.
 * strict_predicate = the planner's `<`/`>` rewrite: both sides' END minus one (intervals.rs:85-115).
 * nearest_algorithm != 0 = Algorithm::CoitreesNearest (one row per probe row, NULL build index when
 * nothing is found).  Inner: (build_idx, probe_idx); RightSemi/RightAnti: probe_idx only
 * (build_idx output left released/empty). */
int brh_interval_join(brh_session *s, brh_batch build, brh_columns bcols, brh_batch probe, brh_columns pcols,
                      int join_type, int strict_predicate, int nearest_algorithm,
                      struct ArrowArray *build_idx, struct ArrowSchema *build_idx_schema,
                      struct ArrowArray *probe_idx, struct ArrowSchema *probe_idx_schema);

/* merge('table'[,min_dist]...): MergeProvider/MergeStream (R/src/merge.rs:84-112, :263-350).
 * Outputs contig Utf8, start Int64, end Int64, n_intervals Int64. */
int brh_merge(brh_session *s, brh_batch table, brh_columns cols, int64_t min_dist, int filter_op,
              struct ArrowArray *contig, struct ArrowSchema *contig_schema,
              struct ArrowArray *start, struct ArrowSchema *start_schema,
              struct ArrowArray *end, struct ArrowSchema *end_schema,
              struct ArrowArray *n_intervals, struct ArrowSchema *n_schema);

/* subtract('left','right'): SubtractStream / SubtractStreamExtra (R/src/subtract.rs:354-462, :529-662).
 * Outputs contig Utf8, start Int64, end Int64 and left_row UInt32 (the row of `left` each fragment
 * came from, for the extra-columns take). */
int brh_subtract(brh_session *s, brh_batch left, brh_columns lcols, brh_batch right, brh_columns rcols,
                 int filter_op,
                 struct ArrowArray *contig, struct ArrowSchema *contig_schema,
                 struct ArrowArray *start, struct ArrowSchema *start_schema,
                 struct ArrowArray *end, struct ArrowSchema *end_schema,
                 struct ArrowArray *left_row, struct ArrowSchema *left_row_schema);

/* cluster('table'[,min_dist]...): ClusterProvider / ClusterStream / ClusterStreamExtra
 * (R/src/cluster.rs:29-82, :517-760, :762-977).  One output row per input row, sorted by
 * (contig, start, end, input row): contig Utf8, start Int64, end Int64, row UInt32 (the input row,
 * for the extra-columns take, cluster.rs:787-806), cluster Int64, cluster_start Int64, cluster_end Int64. */
int brh_cluster(brh_session *s, brh_batch table, brh_columns cols, int64_t min_dist, int filter_op,
                struct ArrowArray *contig, struct ArrowSchema *contig_schema,
                struct ArrowArray *start, struct ArrowSchema *start_schema,
                struct ArrowArray *end, struct ArrowSchema *end_schema,
                struct ArrowArray *row, struct ArrowSchema *row_schema,
                struct ArrowArray *cluster, struct ArrowSchema *cluster_schema,
                struct ArrowArray *cluster_start, struct ArrowSchema *cluster_start_schema,
                struct ArrowArray *cluster_end, struct ArrowSchema *cluster_end_schema);

/* complement('table'[,'view_table']...): ComplementProvider / ComplementStream
 * (R/src/complement.rs:28-75, :236-478).  view.array == NULL means no view table: every contig gets
 * [0, i64::MAX).  Outputs contig Utf8, start Int64, end Int64 named after the INPUT table's columns. */
int brh_complement(brh_session *s, brh_batch table, brh_columns cols, brh_batch view, brh_columns view_cols, int filter_op,
                   struct ArrowArray *contig, struct ArrowSchema *contig_schema,
                   struct ArrowArray *start, struct ArrowSchema *start_schema,
                   struct ArrowArray *end, struct ArrowSchema *end_schema);

/* compute::take of ONE payload column with an index array a join / nearest call returned
 * (interval_join.rs:1655-1667, nearest.rs:469-482), on the device.  column: any fixed-width primitive
 * (ints, floats, date/time/timestamp/duration, decimal128/256, fixed-size binary of 1/2/4/8/16/32 bytes),
 * Boolean, Utf8 / LargeUtf8 / Binary / LargeBinary, or Utf8View / BinaryView (the output is compacted
 * into one data buffer), or a dictionary-encoded column over any of the flat value types (the keys are gathered, the
 * output carries its own copy of the dictionary); idx: UInt32, nulls allowed (-> null output slots).
 * Nested columns -- Struct, List / LargeList / FixedSizeList, Map, in any nesting over the types above -- are taken level
 * by level (a struct's row selection on every child, a list's rows as ranges of child elements); their leaves go through
 * the same device gathers, offsets and validity of the nesting levels are host work.  Union and run-end-encoded layouts
 * return an error.  The output has the column's type and is nullable. */
int brh_take(brh_session *s, const struct ArrowArray *column, const struct ArrowSchema *column_schema,
             const struct ArrowArray *idx, const struct ArrowSchema *idx_schema,
             struct ArrowArray *out, struct ArrowSchema *out_schema);

/* IntervalJoinStream as a push interface (interval_join.rs:934-1140, :1418-1677).  open indexes the build side once
 * (collect_left_input, :584-700); push takes one probe RecordBatch (FetchProbeBatch / ProcessProbeBatch).  Probe
 * batches are coalesced into groups of at least `coalesce_rows` rows (0 = 4 Mi) before they go to the GPU -- the job
 * of CoalesceBatchesExec in a DataFusion plan: a device call per 8192-row batch would be all launch latency.
 * join_type:
 *   BRH_JOIN_INNER       (build_idx, probe_idx) pairs (:1614-1653)
 *   BRH_JOIN_RIGHT_SEMI / _ANTI   probe_idx only: the probe rows with / without a match, ascending (:1014-1024, :1433-1463)
 *   BRH_JOIN_NEAREST     Algorithm::CoitreesNearest: one row per probe row, build_idx NULL where the key has no build row
 *                        (:864-870, :1226-1238, :1628-1635)
 * max_output_rows: 0 = every group gives ONE result (the regular mode).  > 0 = the reference's low-memory stream
 *   (:1153-1299): a result holds whole probe rows and ends after the row at which its running output-row count
 *   reaches the budget (:1199-1216, :1256-1273), so it stays below budget + the matches of its last row;
 *   BRH_MAX_OUTPUT_ENV = the reference's default, BIO_MAX_OUTPUT_BATCH_SIZE or 100000 (:543-548).
 * next: probe_idx counts over the group's concatenated rows; batch_offsets Int64 [n_batches + 1] = the first row of
 * each of the group's batches, so the caller can concatenate its buffered batches (or split the pairs) and `take`
 * the payload columns as the reference does (:1655-1667); *group_done = 1 with the group's last result (the
 * caller may drop the group's buffered batches).  *n_ready = results waiting in the queue.  Probe rows whose key
 * the build side does not have never match.  finish flushes the last, partial group. */
typedef struct brh_join_stream brh_join_stream;
int  brh_join_stream_open(brh_session *s, brh_batch build, brh_columns bcols, brh_columns pcols, int strict_predicate,
                          uint64_t coalesce_rows, int join_type, uint64_t max_output_rows, brh_join_stream **out);
int  brh_join_stream_push(brh_join_stream *js, brh_batch probe, int *n_ready);
int  brh_join_stream_finish(brh_join_stream *js, int *n_ready);
int  brh_join_stream_next(brh_join_stream *js, uint64_t *first_batch, uint64_t *n_batches, int *group_done,
                          struct ArrowArray *build_idx, struct ArrowSchema *build_idx_schema,
                          struct ArrowArray *probe_idx, struct ArrowSchema *probe_idx_schema,
                          struct ArrowArray *batch_offsets, struct ArrowSchema *batch_offsets_schema);
void brh_join_stream_close(brh_join_stream *js);

/* the checks alone (no GPU): resolve a position column like PosArray::resolve / resolve_i64 would;
 * 0 = fine, else the error text is set.  Used by the CPU-only tests. */
int brh_check_position_column(brh_session *s_or_null, brh_batch table, const char *column, int as_i64,
                              char *errbuf, int errbuf_len);
/* ... and a contig column like ContigArray (array_utils.rs:10-24, :178-229): Utf8 / LargeUtf8 / Utf8View.  NULL contigs
 * pass unless the session (or, without one, BIO_STRICT_NULL_CONTIGS=1) is strict about them, see below. */
int brh_check_contig_column(brh_session *s_or_null, brh_batch table, const char *column, char *errbuf, int errbuf_len);
/* NULL contigs.  Default (on = 0), as the reference, which never reads a contig column's validity bitmap: the table
 * functions key a NULL slot by the bytes its offsets span (`value(i)`: "" for builder-made arrays), the join treats NULL as a
 * key of its own that matches only NULL (create_hashes; hashes are compared, never values: interval_join.rs:857, :922-928).
 * None of the reference's tests pins this (parity unpinned).  on = 1: a batch with a NULL contig is refused with an error
 * instead (the default of new sessions when BIO_STRICT_NULL_CONTIGS=1 is set). */
int brh_session_set_strict_null_contigs(brh_session *s, int on);

/* BuildProbeJoinMetrics of the session's context under the reference's names (joins/utils.rs:399-453), and the device
 * memory the session may reserve (MemoryReservation::try_grow, interval_join.rs:614-639): ivx.h ivx_ctx_metrics /
 * ivx_ctx_set_memory_limit. */
int brh_session_metrics(brh_session *s, ivx_metrics *out);
int brh_session_set_memory_limit(brh_session *s, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif
