/*
 * ivx_oracle.h -- CPU restatement of the reference's interval algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the timed CPU baseline.
 *
 * Reference = biodatageeks/datafusion-bio-functions, crate
 * datafusion/bio-function-ranges (abbreviated R/ below).  Each function cites
 * the reference file:line whose arithmetic it restates.
 *
 * Parity pin: checked against the reference's own golden tables and fixture
 * files (tests/golden/, see tests/test_oracle_golden.py) and, in the build
 * container, against the reference's vendored superintervals.hpp compiled
 * into oracle/_ref (see oracle/Makefile).
 *
 * Conventions shared with the product C ABI (include/ivx.h):
 *   - keys are dense uint32 ids (the host maps contig strings to ids; the
 *     reference groups by a 64-bit hash of the key columns,
 *     R/src/physical_planner/joins/interval_join.rs:922-928);
 *   - join/count/coverage/nearest coordinates are int32, closed intervals;
 *   - merge/subtract coordinates are int64;
 *   - "build" = the reference's left / collected side, "probe" = the streamed
 *     right side.
 */
#ifndef IVX_ORACLE_H
#define IVX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NULL_IDX 0xFFFFFFFFu

/* ---- a3: overlap join (interval_join.rs:849-900, :1614-1653) ------------
 * Emits every (build_row, probe_row) with key equal and
 * build.start <= probe.end && build.end >= probe.start.
 * Pairs are grouped by probe row in probe order (the reference's RLE
 * expansion); order inside one probe row is unspecified by the reference.
 * Returns the total number of pairs; writes at most cap of them. */
uint64_t orc_join_brute(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                        const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                        uint32_t *out_build, uint32_t *out_probe, uint64_t cap);

/* Same result set through a per-key sorted, max-end-augmented implicit tree
 * (the algorithm class of coitrees 0.4.0 COITree::new/query as used at
 * interval_join.rs:751-763, :857-862).  threads>1 splits the probe side
 * across OpenMP threads sharing one read-only index (DataFusion's
 * target_partitions probing an Arc'd build side, interval_join.rs:466-480).
 * per_row (nullable) receives the match count of every probe row. */
uint64_t orc_join_tree(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                       const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                       uint32_t *out_build, uint32_t *out_probe, uint64_t cap,
                       uint64_t *per_row, int threads);

/* The same pair set in a single walk per probe row with per-thread growable buffers (the shape of the
 * reference's probe loop, interval_join.rs:1614-1653).  run -> total -> copy -> free. */
void *orc_join_single_run(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                          const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np, int threads);
uint64_t orc_join_single_total(const void *h);
void orc_join_single_copy(const void *h, uint32_t *out_build, uint32_t *out_probe);
void orc_join_single_free(void *h);

/* a3': RightSemi / RightAnti existence (interval_join.rs:1014-1024, :1433-1447)
 * exists[i] = 1 iff probe row i has at least one match. */
void orc_join_exists(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                     const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                     uint8_t *exists);

/* ---- a4: count_overlaps (interval_tree.rs:20-50, :249-267) --------------
 * Per key: starts and ends sorted independently;
 * count = #{starts <= qe} - #{ends < qs}, 0 when qe < qs or key unknown.
 * strict != 0 shrinks the query first: qs += 1, qe -= 1 (wrapping). */
void orc_count_overlaps(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                        const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                        int strict, int64_t *out);

void orc_count_overlaps_mt(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                           const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                           int strict, int64_t *out, int threads);

/* ---- a5: coverage (interval_tree.rs:52-73 merge_intervals, :145-152
 * get_coverage, :181-208 stream loop) ------------------------------------ */
void orc_coverage(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                  const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                  int strict, int64_t *out);

void orc_coverage_mt(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                     const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                     int strict, int64_t *out, int threads);

/* merge_intervals alone (interval_tree.rs:52-73) on one key, in place on
 * (s,e) of length n; returns the merged length. */
uint64_t orc_merge_intervals_i32(int32_t *s, int32_t *e, uint64_t n);

/* ---- a6: nearest (nearest_index.rs:44-266, nearest.rs:330-456) ----------
 * For every probe row emits max(1, found) output rows:
 *   out_build[r]  build row or ORC_NULL_IDX (validity 0),
 *   out_probe[r]  probe row,
 *   out_dist[r]   candidate_distance on the RAW probe coordinates
 *                 (nearest.rs:367-374), -1 where null.  out_dist nullable.
 * Returns rows written (needs cap >= np*max(k,1)). */
uint64_t orc_nearest(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                     const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                     int strict, uint32_t k, int include_overlaps,
                     uint32_t *out_build, uint32_t *out_probe, int64_t *out_dist, uint64_t cap);
/* k = 1, one output row per probe row at the probe row's index, `threads` host threads (full-size checks) */
uint64_t orc_nearest1_mt(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                         const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                         int strict, int include_overlaps, uint32_t *out_build, int64_t *out_dist, int threads);

/* ---- a7+a8: merge (grouped_stream.rs:50-113, merge.rs:282-350) ----------
 * Groups by key (ascending key id = the host's byte-lexicographic contig
 * order), sorts (start,end), sweeps.  Returns rows; writes at most cap. */
uint64_t orc_merge(const uint32_t *key, const int64_t *s, const int64_t *e, uint64_t n,
                   int64_t min_dist, int strict,
                   uint32_t *out_key, int64_t *out_s, int64_t *out_e, int64_t *out_n, uint64_t cap);

/* ---- a7+a9: subtract (subtract.rs:390-462, :575-655) ---------------------
 * out_row (nullable) receives the left input row of each fragment (the
 * extra-columns variant sorts by (start,end,row)). */
uint64_t orc_subtract(const uint32_t *lkey, const int64_t *ls, const int64_t *le, uint64_t nl,
                      const uint32_t *rkey, const int64_t *rs, const int64_t *re, uint64_t nr,
                      int strict,
                      uint32_t *out_key, int64_t *out_s, int64_t *out_e, uint32_t *out_row,
                      uint64_t cap);

/* ---- f1: cluster (cluster.rs:443-477 count, :598-661 emit, :380-420 global ids)
 * Rows come out sorted by (key, start, end, row); row r of the output gets the
 * id of its cluster (clusters numbered in output order: keys ascending = contig
 * names in byte order, cluster.rs:396-417), and the cluster's extent.
 * key_base (nullable, [nkeys]) overrides the id of each key's first cluster --
 * what ClusterIdCoordinator hands a partition that holds only some contigs.
 * key_clusters (nullable, [nkeys]) receives the clusters per key.  Returns the
 * cluster count. */
uint64_t orc_cluster(const uint32_t *key, const int64_t *s, const int64_t *e, uint64_t n, uint32_t nkeys,
                     int64_t min_dist, int strict, const int64_t *key_base,
                     uint32_t *out_key, int64_t *out_s, int64_t *out_e, uint32_t *out_row,
                     int64_t *out_cluster, int64_t *out_cs, int64_t *out_ce, uint64_t *key_clusters);

/* ---- f2: complement (complement.rs:297-357 merge + gaps, :394-465 driver) ----
 * Input rows merged per key (strict: s < cur_end, else s <= cur_end; no
 * min_dist), gaps emitted against the key's view intervals (sorted by
 * (start,end), NOT merged); a key with input rows and no view row gets the
 * implicit view [0, INT64_MAX).  Keys with input rows come first (ascending),
 * then keys that only have view rows (ascending), whose views are emitted
 * whole.  Returns rows; writes at most cap. */
uint64_t orc_complement(const uint32_t *key, const int64_t *s, const int64_t *e, uint64_t n,
                        const uint32_t *vkey, const int64_t *vs, const int64_t *ve, uint64_t nv,
                        int strict, uint32_t *out_key, int64_t *out_s, int64_t *out_e, uint64_t cap);

/* array_utils.rs:33-66, :90-104: checked i64 -> i32; returns -1 when all fit,
 * else the first offending row (the reference reports value and row). */
int64_t orc_check_i32(const int64_t *v, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
