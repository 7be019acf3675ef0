// ivx_grid.hpp -- "rank grid": O(1) rank queries over an int32 column grouped by key.
//
// The reference answers count_overlaps / coverage / nearest with binary
// searches (`partition_point`) over per-contig sorted Vec<i32>
// (interval_tree.rs:46-47, nearest_index.rs:145-146, :194-195, :227-232).  A
// binary search is ~20 dependent random HBM/L2 reads per query on a GPU; the
// grid replaces it with two: a direct-address cell table (cells 2^sh wide per
// key, CSR offsets) and the cell's few values.
//   rank_le(k, x) = koff[k] + #{v of key k : v <= x}
//                 = binstart[cell(k,x)] + #{v in that cell : v <= x}
#pragma once
#include "ivx_device.hpp"

// position (in the grouped order, across keys) of the first value of key k that is > x.
// VS = distance between consecutive values in 32-bit words (values that live inside records)
template <int VS = 1>
__device__ __forceinline__ u32 grid_rank_le(const RankGridView &g, u32 sh, u32 k, i32 x)
{
    const u32 cnt = g.kcnt[k];
    const u32 off = g.koff[k];
    if (cnt == 0) return off;
    const i64 d = (i64)x - (i64)g.origin[k];
    if (d < 0) return off;
    if (d > (i64)g.span[k]) return off + cnt;
    const u32 c = g.kbase[k] + (u32)(d >> sh);
    const u32 a = g.binstart[c], b = g.binstart[c + 1];
    u32 r = a;
    for (u32 j = a; j < b; j++) r += g.val[(u64)j * VS] <= x ? 1u : 0u;
    return r;
}

// #{v < x}: partition_point(|v| v < x)
template <int VS = 1>
__device__ __forceinline__ u32 grid_rank_lt(const RankGridView &g, u32 sh, u32 k, i32 x)
{
    if (x == INT32_MIN) return g.koff[k];
    return grid_rank_le<VS>(g, sh, k, x - 1);
}

// host side (ivx_grid.hip)
ivx_status ivx_keystats(ivx_ctx *ctx, const u32 *key, const i32 *v, u64 n, u32 nkeys,
                        i32 *kmin, i32 *kmax, u32 *kcnt, u32 *errflag);
// Build a rank grid over (key[i], v[i]), i < n.  Index memory is owned by ix.
// kcnt_hint: per-key row counts are recomputed; key may be NULL (single key).
// sorted = the rows come grouped by ascending key with non-decreasing v inside a key AND v is index memory
// that outlives the grid: no atomics, no copy of the values.  vstride (sorted only) = words between
// consecutive values; probe such a grid with grid_rank_le<vstride>.
ivx_status ivx_grid_build(ivx_ctx *ctx, ivx_index *ix, const u32 *key, const i32 *v, u64 n, u32 nkeys, RankGridView *out, bool sorted = false,
                          u32 vstride = 1);
ivx_status ivx_keystats(ivx_ctx *ctx, const u32 *key, const i32 *v, u64 n, u32 nkeys,
                        i32 *kmin, i32 *kmax, u32 *kcnt, u32 *errflag, u32 vstride);
// ... and the histogram of the length classes of [v, vend] (33 counters, zeroed by the caller; vstride must be 1)
// zero (nullable): short word ranges the initialising kernel clears on the way (the caller's flags / headers)
struct ivx_zero_ranges { u32 *p[3]; u32 n[3]; };
ivx_status ivx_keystats_len(ivx_ctx *ctx, const u32 *key, const i32 *v, u64 n, u32 nkeys,
                            i32 *kmin, i32 *kmax, u32 *kcnt, u32 *errflag, u32 vstride, const i32 *vend, u32 *lenhist,
                            const ivx_zero_ranges *zero = nullptr);
