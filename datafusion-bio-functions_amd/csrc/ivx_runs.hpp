// ivx_runs.hpp -- the interval-merge sweep as parallel scans (merge, complement, cluster).  Two forms: the general one below
// over sorted columns (any input), and the sweep over the sort's packed words for well-formed rows (ivx_merge_runs_packed /
// ivx_cluster_rows_packed; ivx_runs.hip says why that one may ignore the order of equal starts and needs only a maximum).
//
// The reference walks the sorted rows of a contig with one running state
// (cur_start, cur_end, cur_count): merge.rs:286-311 and merge_intervals,
// interval_tree.rs:60-70.  `cur_end` after a row is a function of `cur_end`
// before it:
//        f_(s,e)(x) = merges(x) ? max(x, e) : e ,  merges(x) <=> x >= T(s)
// which always has the shape  x < M ? c : x  with c <= M (M = max(T,e), c = e);
// the first row of a contig is the constant function e.  That family is closed
// under composition:
//        (M2,c2) o (M1,c1) = M2 > M1 ? (M2, c2) : (M1, c1 < M2 ? c2 : c1)
// so an inclusive scan with this operator yields cur_end after every row
// without any serial walk.  Run heads, run ids and run lengths follow from a
// second (sum,max) scan.  Exact for ANY input (unsorted ends, end < start,
// saturating cur_end + min_dist), not just well-formed intervals.
#pragma once
#include "ivx_internal.hpp"

// How the sweeps' sort packs a row into ONE 64-bit word (ivx_sweep.hip sort64): the end in the low bits_e bits (minus
// min_e), above it either key ‖ (start - min_s) -- bits_s bits of start -- or, lin, the (key, start) pair's number in key-major
// order: base[key] + (start - kmin[key]).
struct Pack64 { i64 min_s, min_e; u32 bits_s, bits_e; const u64 *base; const long long *kmin; u32 lin, nkeys;
                u32 small, pad; };         // small: every start and end lies within +-2^61 (differences cannot overflow)

// Rows sorted by (key, start, end) as the merge sweep reads them: three columns -- wide (u32 key, i64 start, i64 end: 20 bytes
// per row in every pass of the sweep) or, s32 != nullptr, narrow (start and end as 32-bit offsets from min_s / min_e: 12
// bytes), which the sort's unpack pass writes whenever the coordinates' ranges fit, i.e. for any genome.
// (Reading the sort's packed 8-byte words directly was built and measured too: merge of 200 M rows 12.4 -> 13.0 ms.  The
// unpack pass got cheaper, but every pass of the sweep then decodes (key, start) from a linearised position -- a key lookup
// and variable 64-bit shifts per row -- and the striped run-head pass doubled its time.)
struct SortedRows {
    const u32 *ks; const i64 *ss, *es;
    const u32 *s32, *e32; i64 min_s, min_e;
    const u8 *k8;               // narrow rows of at most 256 keys: the key column as bytes (ks' storage)
#ifdef __HIPCC__
    // row i: its key, start, end and whether it is the first row of its key
    __device__ __forceinline__ void get(u64 i, u32 &k, i64 &s, i64 &e, bool &first) const
    {
        if (k8 != nullptr) { k = k8[i]; first = i == 0 || k8[i - 1] != k; }
        else { k = ks[i]; first = i == 0 || ks[i - 1] != k; }
        if (s32 != nullptr) { s = (i64)((u64)min_s + s32[i]); e = (i64)((u64)min_e + e32[i]); }
        else { s = ss[i]; e = es[i]; }
    }
#endif
};

struct ivx_runs_out {
    u32 *key; i64 *start; i64 *end; i64 *count;     // device buffers, capacity n (any may be nullptr)
};

// ks/ss/es: rows sorted by (key, start, end) on the device.  Writes the runs in
// order and returns their number in *m (host; synchronises the stream).
// Uses scratch WS_T5..WS_T7 and WS_SCAN*.
ivx_status ivx_merge_runs(ivx_ctx *ctx, const u32 *ks, const i64 *ss, const i64 *es, u64 n,
                          i64 min_dist, int strict, const ivx_runs_out &out, u64 *m);
// the same over either form of sorted rows
ivx_status ivx_merge_runs_rows(ivx_ctx *ctx, const SortedRows &rows, u64 n, i64 min_dist, int strict, const ivx_runs_out &out, u64 *m);

// The same runs in ONE pass over the sort's packed words (sorted on their (key, start) bits; ivx_runs.hip "one-pass merge
// sweep"), for inputs that allow it: ivx_merge_packed_ok -- malformed = some row has end < start, has_empty = some row has
// end == start.  Uses scratch WS_SCAN0.
bool ivx_merge_packed_ok(const Pack64 &p, u64 n, u32 nkeys, i64 min_dist, int strict, bool malformed, bool has_empty);
ivx_status ivx_merge_runs_packed(ivx_ctx *ctx, const u64 *w, u64 n, const Pack64 &p, i64 min_dist, int strict, const ivx_runs_out &out, u64 *m);

// cluster(): the same sweep, but every sorted row gets the id and the extent of its run
// (cluster.rs:598-661).  Ids count runs in order from 0, or from key_base[key] for the first run
// of each key when key_base is given ([nkeys], device; ClusterIdCoordinator offsets,
// cluster.rs:396-417).  key_clusters (nullable, [nkeys], device) receives the runs per key.
// Uses scratch WS_T5..WS_T9.
struct ivx_cluster_out { i64 *cluster; i64 *start; i64 *end; u64 *key_clusters; };
ivx_status ivx_cluster_rows(ivx_ctx *ctx, const u32 *ks, const i64 *ss, const i64 *es, u64 n, u32 nkeys,
                            i64 min_dist, int strict, const i64 *key_base, const ivx_cluster_out &out, u64 *m);
// ... and over the sort's packed words when ivx_merge_packed_ok allows (w: sorted on the (key, start) bits at least; ks: the
// unpacked key column, for the per-key id base): tile maxima, run heads, one pass that leaves every row's run number and run
// start, one that turns the numbers into ids and run ends -- instead of three passes over 20-byte rows, a 16-byte state per
// row and two per-row passes.
ivx_status ivx_cluster_rows_packed(ivx_ctx *ctx, const u64 *w, const Pack64 &p, const u32 *ks, u64 n, u32 nkeys,
                                   i64 min_dist, int strict, const i64 *key_base, const ivx_cluster_out &out, u64 *m);
