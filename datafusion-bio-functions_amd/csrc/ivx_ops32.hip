// ivx_ops32.hip -- count_overlaps, coverage and nearest on int32 coordinates:
// index builds (device radix sort + scans + rank grids) and the probe kernels.
//
//   a4 CountOverlapIndex::new/query_count   interval_tree.rs:20-50, get_count_stream :249-267
//   a5 merge_intervals + get_coverage       interval_tree.rs:52-73, :145-152, get_stream :181-208
//   a6 NearestIntervalIndex                 nearest_index.rs:44-266, get_nearest_stream nearest.rs:330-456
//
// All kernels are integer/index work bound by random L2/HBM access latency; each
// probe row costs a handful of dependent gathers instead of the reference's
// O(log n) binary searches.
#include <cstdlib>
#include <cstring>
#include "ivx_grid.hpp"
#include "ivx_runs.hpp"
#include "ivx_scan.hpp"
#include "ivx_sort.hpp"

namespace {

constexpr int OT = 256;
constexpr u32 SIGN = 0x80000000u;

__device__ __forceinline__ i32 wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
__device__ __forceinline__ i32 wsub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }

ivx_status check_keyflag(ivx_ctx *ctx)
{
    u32 *errflag = (u32 *)(ctx->d_scalars + 8);
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 8, errflag, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*(u32 *)(ctx->h_scalars + 8)) return ctx->fail(IVX_ERR_INVALID, "build key id >= n_keys");
    return IVX_OK;
}

// ======================================================================= a4

__device__ __forceinline__ i64 count_one(const RankGridView &gs, const RankGridView &ge, u32 shs, u32 she, u32 k, i32 qs, i32 qe)
{
    if (k < gs.nkeys && gs.kcnt[k] != 0 && !(qe < qs)) {                 // :42-44, unknown contig -> 0 (:265)
        const u32 started = grid_rank_le(gs, shs, k, qe);                // starts.partition_point(|v| v <= end)
        const u32 ended_before = grid_rank_lt(ge, she, k, qs);           // ends.partition_point(|v| v < start)
        return (i64)started - (i64)ended_before;
    }
    return 0;
}

// gate (nullable): run only if *gate == 0 (the routed path found the probe rows in region order and moved nothing)
__global__ __launch_bounds__(OT) void k_probe_count(RankGridView gs, RankGridView ge, const u32 *__restrict__ pkey,
                                                    const i32 *__restrict__ ps, const i32 *__restrict__ pe, u64 n,
                                                    int strict, i64 *__restrict__ out, const u32 *gate)
{
    if (gate && *gate != 0) return;
    const u32 shs = gs.hdr[0], she = ge.hdr[0];
    for (u64 i = (u64)blockIdx.x * OT + threadIdx.x; i < n; i += (u64)gridDim.x * OT) {
        const u32 k = pkey ? pkey[i] : 0u;
        i32 qs = ps[i], qe = pe[i];
        if (strict) { qs = wadd(qs, 1); qe = wsub(qe, 1); }              // interval_tree.rs:253-256
        out[i] = count_one(gs, ge, shs, she, k, qs, qe);
    }
}

__global__ __launch_bounds__(OT) void k_any_inverted(const i32 *__restrict__ s, const i32 *__restrict__ e, u64 n, u32 *flag)
{
    const u64 i = (u64)blockIdx.x * OT + threadIdx.x;
    if (i < n && e[i] < s[i]) *flag = 1;
}

// large unsorted batches go through the region partition (ivx_join_regions.hip) unless told otherwise
bool rowval_regions_wanted(const ivx_index *ix, u64 n)
{
    if (!(ix->flags & IVX_IXF_REGION_ROWVAL) || ix->jv_nreg == 0 || ix->jv_nreg > IVX_MAXREG_WIDE) return false;   // one partition pass only
    const char *f = getenv("IVX_ROWVAL_PATH");                          // tests: "direct" | "regions"
    if (f && !strcmp(f, "direct")) return false;
    if (f && !strcmp(f, "regions")) return true;
    return n >= (1u << 21);                                             // measured crossover (tools/crossover.py)
}

// ======================================================================= a5

// pack rows for the stable sort by (key, first): w0 = first(biased)<<32 | end bits, w1 = key<<32 | row
__global__ __launch_bounds__(OT) void k_pack_first(const u32 *__restrict__ key, const i32 *__restrict__ s, const i32 *__restrict__ e,
                                                   u64 n, u32 nkeys, u64 *w0, u64 *w1, u32 *flags)
{
    const u64 i = (u64)blockIdx.x * OT + threadIdx.x;
    if (i >= n) return;
    const u32 k = key ? key[i] : 0u;
    if (k >= nkeys) flags[0] = 1;
    if (e[i] < s[i]) flags[1] = 1;
    w0[i] = ((u64)((u32)s[i] ^ SIGN) << 32) | (u32)e[i];
    w1[i] = ((u64)k << 32) | (u32)i;
}

__global__ __launch_bounds__(OT) void k_unpack_first(const u64 *__restrict__ w0, const u64 *__restrict__ w1, u64 n,
                                                     u32 *ks, i64 *ss, i64 *es)
{
    const u64 i = (u64)blockIdx.x * OT + threadIdx.x;
    if (i >= n) return;
    ks[i] = (u32)(w1[i] >> 32);
    ss[i] = (i64)(i32)((u32)(w0[i] >> 32) ^ SIGN);
    es[i] = (i64)(i32)(u32)w0[i];
}

// merged nodes (i64 from the run sweep) -> i32 columns + per-node weight max(1, last-first) (i32 wrapping, :148)
__global__ __launch_bounds__(OT) void k_nodes(const i64 *__restrict__ rs, const i64 *__restrict__ re, u64 m,
                                              i32 *nfirst, i32 *nlast, u64 *w)
{
    const u64 j = (u64)blockIdx.x * OT + threadIdx.x;
    if (j > m) return;
    if (j == m) { w[j] = 0; return; }
    const i32 f = (i32)rs[j], l = (i32)re[j];
    nfirst[j] = f; nlast[j] = l;
    const i32 d = wsub(l, f);
    w[j] = (u64)(i64)(d > 1 ? d : 1);
}

__device__ __forceinline__ i32 cov_term(i32 qs, i32 qe, i32 first, i32 last)
{   // interval_tree.rs:148  max(1, min(end + 1, node.last) - max(start - 1, node.first))
    const i32 a = wadd(qe, 1), b = wsub(qs, 1);
    const i32 hi = a < last ? a : last;
    const i32 lo = b > first ? b : first;
    const i32 d = wsub(hi, lo);
    return d > 1 ? d : 1;
}

__device__ __forceinline__ i64 coverage_one(const CoverageView &cv, u32 shf, u32 shl, u32 k, i32 qs, i32 qe)
{
        i32 cov = 0;
        if (k < cv.first.nkeys && cv.first.kcnt[k] != 0) {
            // merged nodes are disjoint and ascending: those with first <= qe are a prefix [..hi),
            // those with last >= qs a suffix [lo..): the overlapping ones are [lo,hi)
            const u32 hi = grid_rank_le(cv.first, shf, k, qe);
            const u32 lo = grid_rank_lt(cv.last, shl, k, qs);
            if (lo < hi) {
                if (qs == INT32_MIN || qe == INT32_MAX || hi - lo <= 2) {
                    for (u32 j = lo; j < hi; j++) cov = wadd(cov, cov_term(qs, qe, cv.nfirst[j], cv.nlast[j]));
                } else {
                    // interior nodes lie inside [qs,qe]: their term is max(1, last-first) = pw[j+1]-pw[j]
                    const i64 inner = cv.pw[hi - 1] - cv.pw[lo + 1];
                    cov = wadd(cov_term(qs, qe, cv.nfirst[lo], cv.nlast[lo]), cov_term(qs, qe, cv.nfirst[hi - 1], cv.nlast[hi - 1]));
                    cov = (i32)((u32)cov + (u32)(u64)inner);
                }
            }
        }
        return (i64)cov;                                                 // :207 `count as i64`
}

__global__ __launch_bounds__(OT) void k_probe_coverage(CoverageView cv, const u32 *__restrict__ pkey,
                                                       const i32 *__restrict__ ps, const i32 *__restrict__ pe, u64 n,
                                                       int strict, i64 *__restrict__ out, const u32 *gate)
{
    if (gate && *gate != 0) return;
    const u32 shf = cv.first.hdr[0], shl = cv.last.hdr[0];
    for (u64 i = (u64)blockIdx.x * OT + threadIdx.x; i < n; i += (u64)gridDim.x * OT) {
        const u32 k = pkey ? pkey[i] : 0u;
        i32 qs = ps[i], qe = pe[i];
        if (strict) { qs = wadd(qs, 1); qe = wsub(qe, 1); }              // :185-188
        out[i] = coverage_one(cv, shf, shl, k, qs, qe);
    }
}

// count_overlaps / coverage over probe rows routed by coordinate region (ivx_route_rows; build sides too big for the
// LDS-slice pipeline): the rank-grid gathers of a wavefront then fall into one stretch of the grids.  XCD x sweeps the
// x-th eighth of the routed rows (gridDim.x is a multiple of 8); values at the routed positions.
template <bool COVERAGE>
__global__ __launch_bounds__(OT) void k_rowval_routed(RankGridView gs, RankGridView ge, CoverageView cv, const u32 *__restrict__ rkey, u32 nreg,
                                                      const u64 *__restrict__ pse, const u32 *__restrict__ offs, u32 nblk,
                                                      i64 *__restrict__ vd, const u32 *unsorted)
{
    __shared__ u32 s_rfirst[IVX_MAXREG_WIDE + 2];
    if (*unsorted == 0) return;
    for (u32 t = threadIdx.x; t <= nreg; t += OT) s_rfirst[t] = offs[(u64)t * nblk];
    __syncthreads();
    const u64 total = s_rfirst[nreg];
    const u32 sa = COVERAGE ? cv.first.hdr[0] : gs.hdr[0], sb = COVERAGE ? cv.last.hdr[0] : ge.hdr[0];
    const u32 xcd = blockIdx.x & 7u, nb = gridDim.x >> 3, bi = blockIdx.x >> 3;
    const u64 seg_lo = total * xcd / 8, seg_hi = total * (xcd + 1) / 8;
    for (u64 i = seg_lo + (u64)bi * OT + threadIdx.x; i < seg_hi; i += (u64)nb * OT) {
        u32 a = 0, b = nreg;                                            // last region whose first row is <= i
        while (a < b) { const u32 m = (a + b + 1) >> 1; if (s_rfirst[m] <= i) a = m; else b = m - 1; }
        const u32 k = rkey[a];
        const u64 w = pse[i];
        const i32 qs = (i32)(u32)w, qe = (i32)(u32)(w >> 32);           // strict-adjusted by the routing pass
        vd[i] = COVERAGE ? coverage_one(cv, sa, sb, k, qs, qe) : count_one(gs, ge, sa, sb, k, qs, qe);
    }
}

// ======================================================================= a6

// sort words of a build row: w0 = (start,end) in unsigned order, start major; w1 = (key, row)
__global__ __launch_bounds__(OT) void k_pack_se(const u32 *__restrict__ key, const i32 *__restrict__ s, const i32 *__restrict__ e,
                                                u64 n, u32 nkeys, u64 *w0, u64 *w1, u32 *flags)
{
    const u64 i = (u64)blockIdx.x * OT + threadIdx.x;
    if (i >= n) return;
    const u32 k = key ? key[i] : 0u;
    if (k >= nkeys) flags[0] = 1;
    const u64 sb = (u32)s[i] ^ SIGN, eb = (u32)e[i] ^ SIGN;
    w0[i] = (sb << 32) | eb;
    w1[i] = ((u64)k << 32) | (u32)i;
    if (i) {                                                            // below the row before it? (flags[1]: input not sorted)
        const u32 pk = key ? key[i - 1] : 0u;
        const u64 pw = (((u64)((u32)s[i - 1] ^ SIGN)) << 32) | ((u32)e[i - 1] ^ SIGN);
        if (k != pk ? k < pk : w0[i] < pw) flags[1] = 1;
    }
}

// sorted words -> records {major, minor, row} (+ the key column).  y0/y1 (nullable): the same rows re-packed
// with the halves of w0 swapped, i.e. ready for the end-major sort
__global__ __launch_bounds__(OT) void k_unpack_rec(const u64 *__restrict__ w0, const u64 *__restrict__ w1, u64 n,
                                                   u32 *ks, ivx_nrec *rec, u64 *y0, u64 *y1)
{
    const u64 i = (u64)blockIdx.x * OT + threadIdx.x;
    if (i >= n) return;
    const u64 a = w0[i], b = w1[i];
    ivx_nrec r;
    r.a = (i32)((u32)(a >> 32) ^ SIGN); r.b = (i32)((u32)a ^ SIGN); r.row = (u32)b; r.pmax = 0;
    rec[i] = r;
    if (ks) ks[i] = (u32)(b >> 32);
    if (y0) { y0[i] = (a << 32) | (a >> 32); y1[i] = b; }
}

struct SegMax { i32 v; u32 head; };
struct SegMaxOp {
    using T = SegMax;
    __host__ __device__ static T identity() { T t; t.v = INT32_MIN; t.head = 0; return t; }
    __device__ static T combine(const T &a, const T &b)
    {
        T r; r.head = a.head | b.head; r.v = b.head ? b.v : (a.v > b.v ? a.v : b.v); return r;
    }
    __device__ static T shfl_up(const T &x, int d)
    {
        T r; r.v = __shfl_up(x.v, d, IVX_WAVE); r.head = __shfl_up(x.head, d, IVX_WAVE); return r;
    }
};

// the scan's elements come straight from the records and the key column, its results go into the records
struct SegMaxIn {
    const u32 *ks; const ivx_nrec *rs;
    __device__ SegMax operator()(u64 i) const
    {
        SegMax t; t.v = rs[i].b; t.head = (i == 0 || ks[i] != ks[i - 1]) ? 1u : 0u; return t;
    }
};
struct SegMaxOut {
    ivx_nrec *rs;
    __device__ void operator()(u64 i, const SegMax &v) const { rs[i].pmax = v.v; }
};

struct Cand { i32 s, e; u32 row; };

__device__ __forceinline__ i64 cand_dist(i32 qs, i32 qe, i32 s, i32 e)
{   // nearest_index.rs:252-260
    if (qe < s) return (i64)s - (i64)qe;
    if (e < qs) return (i64)qs - (i64)e;
    return 0;
}
__device__ __forceinline__ int cmp_meta(const Cand &a, const Cand &b)
{   // :245-250
    if (a.s != b.s) return a.s < b.s ? -1 : 1;
    if (a.e != b.e) return a.e < b.e ? -1 : 1;
    return a.row < b.row ? -1 : (a.row > b.row ? 1 : 0);
}
__device__ __forceinline__ int cmp_cand(i32 qs, i32 qe, const Cand &a, const Cand &b)
{   // :262-266
    const i64 ad = cand_dist(qs, qe, a.s, a.e), bd = cand_dist(qs, qe, b.s, b.e);
    if (ad != bd) return ad < bd ? -1 : 1;
    return cmp_meta(a, b);
}

struct NearestCtx {
    NearestView nv; u32 sh_s, sh_e, sh_p;
    // one 16-byte load per candidate
    __device__ __forceinline__ Cand by_start(u32 j) const { const ivx_nrec r = nv.rs[j]; Cand c; c.s = r.a; c.e = r.b; c.row = r.row; return c; }
    __device__ __forceinline__ Cand by_end(u32 j) const { const ivx_nrec r = nv.re[j]; Cand c; c.s = r.b; c.e = r.a; c.row = r.row; return c; }
};

// nearest_index.rs:222-234; positions are absolute (key offset included)
// (issuing the two rank lookups side by side and loading the two records the decision needs together -- three dependent
//  round trips instead of five or six -- was measured and dropped: routed probe 3.66 -> 4.5 ms, sorted 1.5 -> 1.6 ms per 50M
//  rows; the probe is bound by the sectors it misses, not by the length of its dependency chain)
// pmax_first: the same answer from ONE rank lookup when the row does overlap something.  idx = the key's first position whose
// running max of ends reaches qs; the row there has end >= qs itself (it is where the maximum first got there), every
// overlapping row sits at or behind it, and rows are in start order: so it is the first overlap iff its start <= qe, i.e. iff
// idx < pend -- the reference's test prefix_max_end[pend - 1] >= qs says exactly that -- and if its start is beyond qe nothing
// overlaps.  The by_start rank (pend) is then only looked up for rows WITHOUT an overlap, which still need it: two or three
// index sectors per overlapping row instead of five or six (the routed probe is bound by the sectors it misses).
__device__ __forceinline__ bool first_overlap_pmax(const NearestCtx &x, u32 k, u32 off, u32 cnt, i32 qs, i32 qe, Cand *out)
{
    if (qe < qs) return false;
    const u32 idx = grid_rank_lt<4>(x.nv.pmax, x.sh_p, k, qs);       // prefix_max_end.partition_point(< start), over the whole key
    if (idx >= off + cnt) return false;
    const Cand c = x.by_start(idx);
    if (c.s > qe) return false;
    *out = c;
    return true;
}

__device__ __forceinline__ bool first_overlap(const NearestCtx &x, u32 k, u32 off, i32 qs, i32 qe, u32 *plen_abs, Cand *out)
{
    const u32 pend = grid_rank_le<4>(x.nv.by_start, x.sh_s, k, qe);  // by_start.partition_point(first <= end)
    *plen_abs = pend;
    if (qe < qs) return false;
    if (pend == off || x.nv.rs[pend - 1].pmax < qs) return false;
    u32 idx = grid_rank_lt<4>(x.nv.pmax, x.sh_p, k, qs);              // prefix_max_end[..plen].partition_point(< start)
    if (idx > pend) idx = pend;
    *out = x.by_start(idx);
    return true;
}

// the one nearest build row of key k for the (already strict-adjusted) query [qs,qe]
__device__ __forceinline__ bool nearest_one(const NearestCtx &x, u32 k, i32 qs, i32 qe, int include_overlaps, Cand *best, bool pmax_first = false)
{
    const NearestView &nv = x.nv;
    bool found = false;
    if (k < nv.by_start.nkeys && nv.by_start.kcnt[k] != 0) {
        const u32 off = nv.by_start.koff[k], cnt = nv.by_start.kcnt[k];
        u32 pend = 0;
        if (include_overlaps && pmax_first) {
            found = first_overlap_pmax(x, k, off, cnt, qs, qe, best);
            if (!found) pend = grid_rank_le<4>(nv.by_start, x.sh_s, k, qe);
        }
        else if (include_overlaps) found = first_overlap(x, k, off, qs, qe, &pend, best);
        else pend = grid_rank_le<4>(nv.by_start, x.sh_s, k, qe);
        if (!found) {                                                    // nearest_non_overlap_one :192-220
            const u32 li = grid_rank_lt<4>(nv.by_end, x.sh_e, k, qs);    // by_end.partition_point(last < start)
            const bool hl = li > off, hr = pend < off + cnt;
            if (hl && hr) {
                const Cand l = x.by_end(li - 1), r = x.by_start(pend);
                *best = cmp_cand(qs, qe, l, r) <= 0 ? l : r; found = true;
            } else if (hl) { *best = x.by_end(li - 1); found = true; }
            else if (hr) { *best = x.by_start(pend); found = true; }
        }
    }
    return found;
}

// gate (nullable): run only if *gate == 0 (the routed path found the probe rows in region order and moved nothing)
__global__ __launch_bounds__(OT) void k_probe_nearest1(NearestView nv, const u32 *__restrict__ pkey, const i32 *__restrict__ ps,
                                                       const i32 *__restrict__ pe, u64 n, int strict, int include_overlaps,
                                                       u32 *__restrict__ ob, u32 *__restrict__ op, i64 *__restrict__ od, const u32 *gate, int pmax_first = 1)
{
    if (gate && *gate != 0) return;
    NearestCtx x; x.nv = nv; x.sh_s = nv.by_start.hdr[0]; x.sh_e = nv.by_end.hdr[0]; x.sh_p = nv.pmax.hdr[0];
    for (u64 i = (u64)blockIdx.x * OT + threadIdx.x; i < n; i += (u64)gridDim.x * OT) {
        const u32 k = pkey ? pkey[i] : 0u;
        const i32 rs = ps[i], re = pe[i];
        i32 qs = rs, qe = re;
        if (strict) { qs = wadd(qs, 1); qe = wsub(qe, 1); }              // nearest.rs:341-344
        Cand best{};
        bool found;
        const u32 k0 = (u32)__builtin_amdgcn_readfirstlane((int)k);       // sorted probe rows: one key per wavefront, its per-key tables (fifteen lookups per row) go through the scalar unit: 2.0 -> 1.5 ms per 50M rows
        if (__ballot(k != k0) == 0) found = nearest_one(x, k0, qs, qe, include_overlaps, &best, pmax_first != 0);
        else found = nearest_one(x, k, qs, qe, include_overlaps, &best, pmax_first != 0);
        ob[i] = found ? best.row : IVX_NULL_IDX;
        op[i] = (u32)i;
        if (od) od[i] = found ? cand_dist(rs, re, best.s, best.e) : -1;  // raw coordinates, nearest.rs:367-374
    }
}

// The same over probe rows ROUTED by coordinate region (ivx_route_rows): rows of one region sit together, so the
// gathers of a wavefront fall into a few MB of the index instead of all of it.  pse = (qs,qe) already strict-adjusted;
// the key of a row follows from its region (rkey); results at the routed position.
constexpr int NR_T = 256;
__global__ __launch_bounds__(NR_T) void k_nearest_routed(NearestView nv, const u32 *__restrict__ rkey, u32 nreg, const u64 *__restrict__ pse,
                                                         const u32 *__restrict__ offs, u32 nblk, u32 adj, int include_overlaps,
                                                         u32 *__restrict__ vb, i64 *__restrict__ vd, const u32 *unsorted, int pmax_first)
{
    __shared__ u32 s_rfirst[IVX_MAXREG_WIDE + 2];
    if (*unsorted == 0) return;
    for (u32 t = threadIdx.x; t <= nreg; t += NR_T) s_rfirst[t] = offs[(u64)t * nblk];
    __syncthreads();
    const u64 total = s_rfirst[nreg];
    NearestCtx x; x.nv = nv; x.sh_s = nv.by_start.hdr[0]; x.sh_e = nv.by_end.hdr[0]; x.sh_p = nv.pmax.hdr[0];
    // workgroups are dealt to the 8 XCDs round robin: XCD x sweeps the x-th eighth of the routed rows, so that every
    // L2 sees one stretch of the index and the index is read from HBM once, not once per XCD (gridDim.x is a multiple of 8)
    const u32 xcd = blockIdx.x & 7u, nb = gridDim.x >> 3, bi = blockIdx.x >> 3;
    const u64 seg_lo = total * xcd / 8, seg_hi = total * (xcd + 1) / 8;
    for (u64 i = seg_lo + (u64)bi * NR_T + threadIdx.x; i < seg_hi; i += (u64)nb * NR_T) {
        u32 a = 0, b = nreg;                                            // last region whose first row is <= i (empty regions share a start: take the last)
        while (a < b) { const u32 m = (a + b + 1) >> 1; if (s_rfirst[m] <= i) a = m; else b = m - 1; }
        const u32 k = rkey[a];
        const u64 w = pse[i];
        const i32 qs = (i32)(u32)w, qe = (i32)(u32)(w >> 32);
        Cand best{};
        // (reading the per-key tables through the scalar unit when the wavefront shares its region, as k_probe_nearest1
        // does, is slower here -- 2.6 -> 2.85 ms per 50M rows: the lanes' reads of one address are a single request already,
        // and this kernel waits on the scattered index sectors of its unsorted rows, not on issuing loads)
        const bool found = nearest_one(x, k, qs, qe, include_overlaps, &best, pmax_first != 0);
        vb[i] = found ? best.row : IVX_NULL_IDX;
        if (vd) vd[i] = found ? cand_dist(wsub(qs, (i32)adj), wadd(qe, (i32)adj), best.s, best.e) : -1;
    }
}

// k > 1: up to k candidates per probe row into tmp[i*k ..], rows-per-probe into cnt[i]
__global__ __launch_bounds__(OT) void k_probe_nearestk(NearestView nv, const u32 *__restrict__ pkey, const i32 *__restrict__ ps,
                                                       const i32 *__restrict__ pe, u64 n, int strict, int include_overlaps, u32 kk,
                                                       u32 *__restrict__ tmp, i64 *__restrict__ tmpd, u32 *__restrict__ cnt)
{
    NearestCtx x; x.nv = nv; x.sh_s = nv.by_start.hdr[0]; x.sh_e = nv.by_end.hdr[0]; x.sh_p = nv.pmax.hdr[0];
    for (u64 i = (u64)blockIdx.x * OT + threadIdx.x; i < n; i += (u64)gridDim.x * OT) {
        const u32 k = pkey ? pkey[i] : 0u;
        const i32 rs = ps[i], re = pe[i];
        i32 qs = rs, qe = re;
        if (strict) { qs = wadd(qs, 1); qe = wsub(qe, 1); }
        u32 *mine = tmp + i * (u64)kk;                                   // the build rows taken so far (the emit pass only copies)
        i64 *mined = tmpd ? tmpd + i * (u64)kk : nullptr;                // and their distances on the raw coordinates (nearest.rs:367-374)
        auto take = [&](u32 at, const Cand &c) { mine[at] = c.row; if (mined) mined[at] = cand_dist(rs, re, c.s, c.e); };
        u32 found = 0;
        if (k < nv.by_start.nkeys && nv.by_start.kcnt[k] != 0) {
            const u32 off = nv.by_start.koff[k], cnt_k = nv.by_start.kcnt[k];
            const u32 pend = grid_rank_le<4>(nv.by_start, x.sh_s, k, qe);
            // seen-set of nearest_k (:122, :134, :186): rows already taken
            auto seen = [&](u32 row) { for (u32 q = 0; q < found; q++) if (mine[q] == row) return true; return false; };
            if (include_overlaps && pend > off && nv.rs[pend - 1].pmax >= qs) {
                // all overlaps in (start,end,row) order = by_start order filtered by end >= qs (:125-137)
                u32 j = grid_rank_lt<4>(nv.pmax, x.sh_p, k, qs);
                for (; j < pend && found < kk; j++) {
                    const ivx_nrec r = nv.rs[j];
                    if (r.b >= qs && !seen(r.row)) { Cand c; c.s = r.a; c.e = r.b; c.row = r.row; take(found++, c); }
                }
            }
            if (found < kk) {                                            // alternate the two cursors (:144-189)
                u32 li = grid_rank_lt<4>(nv.by_end, x.sh_e, k, qs), ri = pend;
                while (found < kk) {
                    const bool hl = li > off, hr = ri < off + cnt_k;
                    if (!hl && !hr) break;
                    bool take_left;
                    if (hl && hr) take_left = cmp_cand(qs, qe, x.by_end(li - 1), x.by_start(ri)) <= 0;
                    else take_left = hl;
                    Cand c;
                    if (take_left) { li--; c = x.by_end(li); }
                    else { c = x.by_start(ri); ri++; }
                    if (!include_overlaps && cand_dist(qs, qe, c.s, c.e) == 0) continue;
                    if (!seen(c.row)) take(found++, c);
                }
            }
        }
        cnt[i] = found;
    }
}

__global__ __launch_bounds__(OT) void k_rows_of(const u32 *__restrict__ cnt, u64 n, u64 *rows)
{
    const u64 i = (u64)blockIdx.x * OT + threadIdx.x;
    if (i < n) rows[i] = cnt[i] ? cnt[i] : 1u;
    if (i == n) rows[i] = 0;
}

__global__ __launch_bounds__(OT) void k_nearest_emit(u64 n, u32 kk, const u32 *__restrict__ tmp, const i64 *__restrict__ tmpd,
                                                     const u32 *__restrict__ cnt, const u64 *__restrict__ offs,
                                                     u64 cap, u32 *ob, u32 *op, i64 *od)
{
    for (u64 i = (u64)blockIdx.x * OT + threadIdx.x; i < n; i += (u64)gridDim.x * OT) {
        const u32 c = cnt[i];
        u64 at = offs[i];
        if (c == 0) {                                                    // nearest.rs:424-430
            if (at < cap) { ob[at] = IVX_NULL_IDX; op[at] = (u32)i; if (od) od[at] = -1; }
            continue;
        }
        for (u32 q = 0; q < c; q++, at++) {
            if (at >= cap) break;
            ob[at] = tmp[i * (u64)kk + q];
            op[at] = (u32)i;
            if (od) od[at] = tmpd[i * (u64)kk + q];
        }
    }
}

u32 grid1(u64 n) { return (u32)((n + OT - 1) / OT); }

// after a stable sort on (key,start) alone: order the runs of equal (key,start) by end (rows of equal end keep their
// row order); a run longer than NFIX_MAXRUN raises *toolong and the caller sorts on all three fields instead
constexpr u32 NFIX_MAXRUN = 64;
__global__ __launch_bounds__(OT) void k_fix_runs_se(u64 *__restrict__ w0, u64 *__restrict__ w1, u64 n, u32 *toolong)
{
    const u64 i = (u64)blockIdx.x * OT + threadIdx.x;
    if (i >= n) return;
    auto same = [&](u64 a, u64 b) { return (w0[a] >> 32) == (w0[b] >> 32) && (w1[a] >> 32) == (w1[b] >> 32); };
    if (i && same(i - 1, i)) return;
    if (i + 1 >= n || !same(i, i + 1)) return;
    u32 len = 2;
    while (i + len < n && len <= NFIX_MAXRUN && same(i, i + len)) len++;
    if (len > NFIX_MAXRUN) { *toolong = 1; return; }
    for (u32 a = 1; a < len; a++) {
        const u64 x0 = w0[i + a], x1 = w1[i + a];
        u32 b = a;
        while (b > 0 && w0[i + b - 1] > x0) { w0[i + b] = w0[i + b - 1]; w1[i + b] = w1[i + b - 1]; b--; }
        w0[i + b] = x0; w1[i + b] = x1;
    }
}

// ---- the same two orders through ONE 64-bit sort word per row (the usual case: genomic coordinates)
// Per key the coordinates of both columns span [origin, origin + span]; `lin = base[key] + (v - origin[key])` with
// base = running sum of (span + 1) numbers every (key, coordinate) in key-major order.  If lin's bits and a row
// number's bits fit 64 together, `lin(start) ‖ row` sorts as 8-byte records over ceil(linbits / 8) digit passes
// (4 for a human genome, where the packed (key, start) of the two-word form takes 5), the end is fetched by row
// afterwards, and the end-major order is the same sort of `lin(end) ‖ position in start order`.
constexpr u32 NLIN_KEYS_LDS = 2048;

__global__ void k_init_minmax(i32 *kmin, i32 *kmax, u32 nkeys)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nkeys) { kmin[i] = INT32_MAX; kmax[i] = INT32_MIN; }
}

__global__ __launch_bounds__(OT) void k_nstats(const u32 *__restrict__ key, const i32 *__restrict__ s, const i32 *__restrict__ e, u64 n,
                                               u32 nkeys, i32 *kmin, i32 *kmax, u32 *flags)
{
    extern __shared__ i32 sh[];
    const bool priv = nkeys <= NLIN_KEYS_LDS;
    i32 *smin = sh, *smax = sh + nkeys;
    if (priv) {
        for (u32 k = threadIdx.x; k < nkeys; k += OT) { smin[k] = INT32_MAX; smax[k] = INT32_MIN; }
        __syncthreads();
    }
    // only "has rows" matters of a key's count; a bound is touched by an atomic only when this row moves it (a plain
    // read comes first: after the first rows of a key almost none do)
    bool bad = false, unsorted = false;
    constexpr int U = 4;                                                // rows per thread in flight
    for (u64 i0 = (u64)blockIdx.x * (OT * U) + threadIdx.x; i0 < n; i0 += (u64)gridDim.x * (OT * U)) {
        u32 k[U], pk[U]; i32 a[U], b[U], pa[U], pb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 i = i0 + (u64)u * OT;
            const bool in = i < n;
            k[u] = in ? (key ? key[i] : 0u) : 0u; a[u] = in ? s[i] : 0; b[u] = in ? e[i] : 0;
            const bool hp = in && i > 0;
            pk[u] = hp ? (key ? key[i - 1] : 0u) : 0u; pa[u] = hp ? s[i - 1] : INT32_MIN; pb[u] = hp ? e[i - 1] : INT32_MIN;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (i0 + (u64)u * OT >= n) continue;
            if (k[u] >= nkeys) { bad = true; continue; }
            // below the row before it? (flags[1]: input not sorted)
            if (k[u] != pk[u] ? k[u] < pk[u] : (a[u] != pa[u] ? a[u] < pa[u] : b[u] < pb[u])) unsorted = true;
            const i32 lo = a[u] < b[u] ? a[u] : b[u], hi = a[u] < b[u] ? b[u] : a[u];
            i32 *pmin = priv ? &smin[k[u]] : &kmin[k[u]], *pmax = priv ? &smax[k[u]] : &kmax[k[u]];
            if (lo < *(volatile i32 *)pmin) atomicMin(pmin, lo);
            if (hi > *(volatile i32 *)pmax) atomicMax(pmax, hi);
        }
    }
    if (bad) flags[0] = 1;
    if (unsorted) flags[1] = 1;
    if (priv) {
        __syncthreads();
        for (u32 k = threadIdx.x; k < nkeys; k += OT)
            if (smin[k] <= smax[k]) { atomicMin(&kmin[k], smin[k]); atomicMax(&kmax[k], smax[k]); }
    }
}

// one workgroup: origin, base (nkeys + 1 entries), and hdr = {linbits, rowbits, fits}
__global__ __launch_bounds__(1024) void k_nlin_layout(const i32 *kmin, const i32 *kmax, u32 nkeys, u64 n,
                                                      i32 *origin, u64 *base, u32 *hdr)
{
    __shared__ u64 red[1024 / IVX_WAVE + 1];
    const u32 t = threadIdx.x;
    u64 run = 0;
    for (u32 k0 = 0; k0 < nkeys; k0 += 1024) {
        const u32 k = k0 + t;
        const bool has = k < nkeys && kmin[k] <= kmax[k];              // (a key without rows keeps its sentinels)
        const u64 w = has ? (u64)((i64)kmax[k] - (i64)kmin[k]) + 1 : 0;
        u64 tot;
        const u64 ex = block_excl_scan<u64, 1024>(w, red, &tot);
        if (k < nkeys) { base[k] = run + ex; origin[k] = has ? kmin[k] : 0; }
        run += tot;
    }
    if (t == 0) {
        base[nkeys] = run;
        const u32 linbits = run > 1 ? 64u - (u32)__clzll(run - 1) : 1u;
        const u32 rowbits = n > 1 ? 64u - (u32)__clzll(n - 1) : 1u;
        hdr[0] = linbits; hdr[1] = rowbits; hdr[2] = linbits + rowbits <= 64 ? 1u : 0u;
    }
}

__global__ __launch_bounds__(OT) void k_pack_lin(const u32 *__restrict__ key, const i32 *__restrict__ v, const i32 *__restrict__ e, u64 n,
                                                 const i32 *__restrict__ origin, const u64 *__restrict__ base, u32 rowbits, u64 *w, i32 *pay)
{
    const u64 i = (u64)blockIdx.x * OT + threadIdx.x;
    if (i >= n) return;
    const u32 k = key ? key[i] : 0u;
    w[i] = ((base[k] + (u64)((i64)v[i] - (i64)origin[k])) << rowbits) | i;
    pay[i] = e[i];                                                      // the end rides along as the record's payload
}

// key of a linearised coordinate: last k with base[k] <= lin (keys without rows share their successor's base and are
// skipped by taking the LAST such k among those with rows: base[k + 1] > lin)
__device__ __forceinline__ u32 lin_key(const u64 *base, u32 nkeys, u64 lin)
{
    u32 a = 0, b = nkeys;                                              // first k with base[k + 1] > lin
    while (a < b) { const u32 m = (a + b) >> 1; if (base[m + 1] > lin) b = m; else a = m + 1; }
    return a;
}

// sorted start words -> records in (key, start, end, row) order + the key column + the end-major sort words.
// Rows of equal (key, start) arrive in row order; each finds its place among them by (end, row).
constexpr u32 NLIN_MAXRUN = 64;
__global__ __launch_bounds__(OT) void k_unpack_lin(const u64 *__restrict__ w, u64 n, u32 nkeys, const i32 *__restrict__ origin,
                                                   const u64 *__restrict__ base_g, const i32 *__restrict__ e, u32 rowbits,
                                                   u32 *ks, ivx_nrec *rs, u64 *y, u32 *toolong)
{
    __shared__ u64 s_base[NLIN_KEYS_LDS + 1];
    const bool lds = nkeys <= NLIN_KEYS_LDS;
    if (lds) { for (u32 k = threadIdx.x; k <= nkeys; k += OT) s_base[k] = base_g[k]; __syncthreads(); }
    const u64 i = (u64)blockIdx.x * OT + threadIdx.x;
    if (i >= n) return;
    const u64 rowmask = (1ull << rowbits) - 1;                          // (rowbits <= 32)
    const u64 x = w[i], lin = x >> rowbits;
    const u32 row = (u32)(x & rowmask);
    const i32 end = e[i];                                               // (payloads, in sorted order)
    u64 h = i, t = i + 1;                                               // the run of equal lin around i
    while (h > 0 && i - h < NLIN_MAXRUN && (w[h - 1] >> rowbits) == lin) h--;
    while (t < n && t - i < NLIN_MAXRUN && (w[t] >> rowbits) == lin) t++;
    u64 pos = i;
    if (t - h > 1) {
        if (t - h > NLIN_MAXRUN) { *toolong = 1; return; }
        u32 below = 0;
        for (u64 j = h; j < t; j++) {
            if (j == i) continue;
            const i32 ej = e[j];
            below += (ej < end || (ej == end && j < i)) ? 1u : 0u;
        }
        pos = h + below;
    }
    const u32 k = lds ? lin_key(s_base, nkeys, lin) : lin_key(base_g, nkeys, lin);
    const u64 kb = lds ? s_base[k] : base_g[k];
    const i32 org = origin[k];
    ivx_nrec r;
    r.a = (i32)((i64)org + (i64)(lin - kb)); r.b = end; r.row = row; r.pmax = 0;
    rs[pos] = r;
    ks[pos] = k;
    y[pos] = ((kb + (u64)((i64)end - (i64)org)) << rowbits) | pos;
}

// sorted end words -> records in (key, end, start, row) order: the payload is the row's place in the start order
__global__ __launch_bounds__(OT) void k_unpack_lin_end(const u64 *__restrict__ y, u64 n, u32 rowbits, const ivx_nrec *__restrict__ rs, ivx_nrec *re)
{
    const u64 i = (u64)blockIdx.x * OT + threadIdx.x;
    if (i >= n) return;
    const ivx_nrec r = rs[(u32)(y[i] & ((1ull << rowbits) - 1))];
    ivx_nrec o; o.a = r.b; o.b = r.a; o.row = r.row; o.pmax = 0;
    re[i] = o;
}

// IVX_OK with *done = false: the input does not fit the one-word form (coordinate spans, or a long run of equal
// (key, start)) -- the caller takes the two-word sorts
ivx_status nearest_sorted_records_lin(ivx_ctx *ctx, const u32 *key, const i32 *s, const i32 *e, u64 n, u32 nkeys,
                                      u32 *ks, ivx_nrec *rs, ivx_nrec *re, bool *done)
{
    *done = false;
    if (n < 4096 || n > 0xFFFFFFFFull || getenv("IVX_NEAREST_SORT2")) return IVX_OK;
    hipStream_t st = ctx->stream;
    i32 *kmin, *kmax, *origin; u64 *base;
    IVX_TRY(ctx->get_scratch(WS_GRID0, nkeys * sizeof(i32), (void **)&kmin));
    IVX_TRY(ctx->get_scratch(WS_GRID1, nkeys * sizeof(i32), (void **)&kmax));
    IVX_TRY(ctx->get_scratch(WS_T3, nkeys * sizeof(i32), (void **)&origin));
    IVX_TRY(ctx->get_scratch(WS_T4, ((size_t)nkeys + 1) * sizeof(u64), (void **)&base));
    u32 *flags = (u32 *)(ctx->d_scalars + 8);                           // [0] bad key, [1] unsorted / run too long
    u32 *hdr = (u32 *)(ctx->d_scalars + 10);
    hipLaunchKernelGGL(k_init_minmax, dim3((nkeys + OT - 1) / OT), dim3(OT), 0, st, kmin, kmax, nkeys);
    const size_t shm = nkeys <= NLIN_KEYS_LDS ? (size_t)nkeys * 8 : 0;
    hipLaunchKernelGGL(k_nstats, dim3(ivx_stream_grid(n, OT * 16, 4096)), dim3(OT), shm, st, key, s, e, n, nkeys, kmin, kmax, flags);
    hipLaunchKernelGGL(k_nlin_layout, dim3(1), dim3(1024), 0, st, (const i32 *)kmin, (const i32 *)kmax, nkeys, n, origin, base, hdr);
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 8, ctx->d_scalars + 8, 4 * sizeof(u64), hipMemcpyDeviceToHost, st));
    IVX_HIP(ctx, hipStreamSynchronize(st));
    const u32 *hf = (const u32 *)(ctx->h_scalars + 8), *hh = (const u32 *)(ctx->h_scalars + 10);
    if (hf[0]) return ctx->fail(IVX_ERR_INVALID, "build key id >= n_keys");
    const bool presorted = hf[1] == 0 && !getenv("IVX_FORCE_SORT");
    const u32 linbits = hh[0], rowbits = hh[1];
    if (!hh[2]) return IVX_OK;
    u64 *a[1], *b[1]; u32 *pay[2];
    IVX_TRY(ctx->get_scratch(WS_SA0, n * sizeof(u64), (void **)&a[0]));
    IVX_TRY(ctx->get_scratch(WS_SB0, n * sizeof(u64), (void **)&b[0]));
    IVX_TRY(ctx->get_scratch(WS_SA1, n * sizeof(u32), (void **)&pay[0]));
    IVX_TRY(ctx->get_scratch(WS_SB1, n * sizeof(u32), (void **)&pay[1]));
    const ivx_sort_field f[1] = {{0, (int)rowbits, (int)(rowbits + linbits)}};
    int in_b = 0;
    hipLaunchKernelGGL(k_pack_lin, dim3(grid1(n)), dim3(OT), 0, st, key, s, e, n, (const i32 *)origin, (const u64 *)base, rowbits, a[0], (i32 *)pay[0]);
    if (!presorted) IVX_TRY(ivx_radix_sort(ctx, 1, a, b, n, f, 1, &in_b, true, pay));
    u64 *w = in_b ? b[0] : a[0], *y = in_b ? a[0] : b[0];
    IVX_HIP(ctx, hipMemsetAsync(flags + 1, 0, sizeof(u32), st));
    hipLaunchKernelGGL(k_unpack_lin, dim3(grid1(n)), dim3(OT), 0, st, (const u64 *)w, n, nkeys, (const i32 *)origin, (const u64 *)base,
                       (const i32 *)pay[in_b], rowbits, ks, rs, y, flags + 1);
    a[0] = y; b[0] = w;
    IVX_TRY(ivx_radix_sort(ctx, 1, a, b, n, f, 1, &in_b, true));
    hipLaunchKernelGGL(k_unpack_lin_end, dim3(grid1(n)), dim3(OT), 0, st, (const u64 *)(in_b ? b[0] : a[0]), n, rowbits, (const ivx_nrec *)rs, re);
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 8, ctx->d_scalars + 8, sizeof(u64), hipMemcpyDeviceToHost, st));
    IVX_HIP(ctx, hipStreamSynchronize(st));
    if (hf[1]) return IVX_OK;                                           // a run of equal (key, start) beyond NLIN_MAXRUN rows
    IVX_HIP(ctx, hipGetLastError());
    *done = true;
    return IVX_OK;
}

// The two orders of the nearest index as records: rs by (key,start,end,row) (nearest_index.rs:50-55), re by
// (key,end,start,row) (:77-82), ks = the key column (the same in both orders).  The second order is a STABLE
// sort of the first by (key,end) alone -- ties keep their (start,row) order -- which is 5 digit passes
// instead of 9 for human-genome coordinates.
ivx_status nearest_sorted_records(ivx_ctx *ctx, const u32 *key, const i32 *s, const i32 *e, u64 n, u32 nkeys,
                                  u32 *ks, ivx_nrec *rs, ivx_nrec *re)
{
    if (n == 0) return IVX_OK;
    {
        bool done = false;
        IVX_TRY(nearest_sorted_records_lin(ctx, key, s, e, n, nkeys, ks, rs, re, &done));
        if (done) return IVX_OK;
        IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(u64), ctx->stream));
    }
    u64 *a[2], *b[2];
    IVX_TRY(ctx->get_scratch(WS_SA0, n * sizeof(u64), (void **)&a[0]));
    IVX_TRY(ctx->get_scratch(WS_SA1, n * sizeof(u64), (void **)&a[1]));
    IVX_TRY(ctx->get_scratch(WS_SB0, n * sizeof(u64), (void **)&b[0]));
    IVX_TRY(ctx->get_scratch(WS_SB1, n * sizeof(u64), (void **)&b[1]));
    u32 *flags = (u32 *)(ctx->d_scalars + 8);
    hipLaunchKernelGGL(k_pack_se, dim3(grid1(n)), dim3(OT), 0, ctx->stream, key, s, e, n, nkeys, a[0], a[1], flags);
    // build rows that already come in (key,start,end) order need no first sort
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 8, flags, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (((const u32 *)(ctx->h_scalars + 8))[0]) return ctx->fail(IVX_ERR_INVALID, "build key id >= n_keys");
    const bool presorted = ((const u32 *)(ctx->h_scalars + 8))[1] == 0 && !getenv("IVX_FORCE_SORT");
    int in_b = 0;
    if (!presorted) {
        // (key,start) first -- equal (key,start) rows are rare -- then the short runs by end in place; all three
        // fields only if a run turns out long
        bool done = false;
        if (n >= 4096 && !getenv("IVX_FORCE_SORT")) {
            const ivx_sort_field f2[2] = {{0, 32, 64}, {1, 32, 64}};
            IVX_TRY(ivx_radix_sort(ctx, 2, a, b, n, f2, 2, &in_b));
            u64 *const *o = in_b ? b : a;
            IVX_HIP(ctx, hipMemsetAsync(flags + 1, 0, sizeof(u32), ctx->stream));
            hipLaunchKernelGGL(k_fix_runs_se, dim3(grid1(n)), dim3(OT), 0, ctx->stream, o[0], o[1], n, flags + 1);
            IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 8, flags, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
            IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            done = ((const u32 *)(ctx->h_scalars + 8))[1] == 0;
            if (!done) {
                IVX_HIP(ctx, hipMemsetAsync(flags + 1, 0, sizeof(u32), ctx->stream));
                hipLaunchKernelGGL(k_pack_se, dim3(grid1(n)), dim3(OT), 0, ctx->stream, key, s, e, n, nkeys, a[0], a[1], flags);
                in_b = 0;
            }
        }
        if (!done) {
            const ivx_sort_field f[3] = {{0, 0, 32}, {0, 32, 64}, {1, 32, 64}};
            IVX_TRY(ivx_radix_sort(ctx, 2, a, b, n, f, 3, &in_b));
        }
    }
    u64 *const *r = in_b ? b : a, *const *y = in_b ? a : b;
    hipLaunchKernelGGL(k_unpack_rec, dim3(grid1(n)), dim3(OT), 0, ctx->stream, (const u64 *)r[0], (const u64 *)r[1], n, ks, rs, y[0], y[1]);
    const ivx_sort_field g[2] = {{0, 32, 64}, {1, 32, 64}};
    IVX_TRY(ivx_radix_sort(ctx, 2, y, r, n, g, 2, &in_b));
    u64 *const *r2 = in_b ? r : y;
    hipLaunchKernelGGL(k_unpack_rec, dim3(grid1(n)), dim3(OT), 0, ctx->stream, (const u64 *)r2[0], (const u64 *)r2[1], n, (u32 *)nullptr, re,
                       (u64 *)nullptr, (u64 *)nullptr);
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

// routing regions over rank grid g's per-key spans (ivx_route_view_build, ivx_join.hip)
ivx_status build_route_view(ivx_ctx *ctx, ivx_index *ix, const RankGridView &g) { return ivx_route_view_build(ctx, ix, g.origin, g.span, g.kcnt); }
void route_view_ready(ivx_ctx *ctx, ivx_index *ix) { ivx_route_view_ready(ctx, ix); }

}  // namespace

// ---------------------------------------------------------------------------- builds

ivx_status ivx_count_build(ivx_ctx *ctx, ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n)
{
    // the overlap index as well: #{starts <= qe} - #{ends < qs} equals the number of rows the overlap
    // predicate matches as long as no build row has end < start, and then big probe batches can take the
    // region-partitioned path (validates the key ids too)
    IVX_TRY(ivx_join_build(ctx, ix, key, s, e, n));
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, 2 * sizeof(u64), ctx->stream));
    if (n) hipLaunchKernelGGL(k_any_inverted, dim3(grid1(n)), dim3(OT), 0, ctx->stream, s, e, n, (u32 *)(ctx->d_scalars + 9));
    IVX_TRY(ivx_grid_build(ctx, ix, key, s, n, ix->nkeys, &ix->gs));     // starts, sorted independently (:35)
    IVX_TRY(ivx_grid_build(ctx, ix, key, e, n, ix->nkeys, &ix->ge));     // ends (:36)
    IVX_TRY(build_route_view(ctx, ix, ix->gs));
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 9, ctx->d_scalars + 9, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    IVX_TRY(check_keyflag(ctx));                                          // synchronises
    if (*(u32 *)(ctx->h_scalars + 9) == 0) ix->flags |= IVX_IXF_REGION_ROWVAL;
    route_view_ready(ctx, ix);
    return IVX_OK;
}

ivx_status ivx_coverage_build(ivx_ctx *ctx, ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n)
{
    hipStream_t st = ctx->stream;
    const u32 nkeys = ix->nkeys;
    u32 *flags = (u32 *)(ctx->d_scalars + 8);
    IVX_HIP(ctx, hipMemsetAsync(flags, 0, sizeof(u64), st));
    u64 m = 0;
    i64 *rs = nullptr, *re = nullptr; u32 *rk = nullptr;
    if (n) {
        // stable sort by (key, first): ties keep input order (interval_tree.rs:57 sort_by is stable)
        u64 *a[2], *b[2];
        IVX_TRY(ctx->get_scratch(WS_SA0, n * sizeof(u64), (void **)&a[0]));
        IVX_TRY(ctx->get_scratch(WS_SA1, n * sizeof(u64), (void **)&a[1]));
        IVX_TRY(ctx->get_scratch(WS_SB0, n * sizeof(u64), (void **)&b[0]));
        IVX_TRY(ctx->get_scratch(WS_SB1, n * sizeof(u64), (void **)&b[1]));
        hipLaunchKernelGGL(k_pack_first, dim3(grid1(n)), dim3(OT), 0, st, key, s, e, n, nkeys, a[0], a[1], flags);
        const ivx_sort_field f[2] = {{0, 32, 64}, {1, 32, 64}};
        int in_b = 0;
        IVX_TRY(ivx_radix_sort(ctx, 2, a, b, n, f, 2, &in_b));
        u64 *const *r = in_b ? b : a;
        u32 *ks; i64 *ss, *es;
        IVX_TRY(ctx->get_scratch(WS_T0, n * sizeof(u32), (void **)&ks));
        IVX_TRY(ctx->get_scratch(WS_T1, n * sizeof(i64), (void **)&ss));
        IVX_TRY(ctx->get_scratch(WS_T2, n * sizeof(i64), (void **)&es));
        hipLaunchKernelGGL(k_unpack_first, dim3(grid1(n)), dim3(OT), 0, st, (const u64 *)r[0], (const u64 *)r[1], n, ks, ss, es);
        IVX_TRY(ctx->get_scratch(WS_T3, n * sizeof(u32), (void **)&rk));
        IVX_TRY(ctx->get_scratch(WS_T4, n * sizeof(i64), (void **)&rs));
        IVX_TRY(ctx->get_scratch(WS_T8, n * sizeof(i64), (void **)&re));
        ivx_runs_out ro{rk, rs, re, nullptr};
        IVX_TRY(ivx_merge_runs(ctx, ks, ss, es, n, 0, 0, ro, &m));       // first <= current.last merges (:63)
    }
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 8, flags, sizeof(u64), hipMemcpyDeviceToHost, st));
    IVX_HIP(ctx, hipStreamSynchronize(st));
    const u32 *hf = (const u32 *)(ctx->h_scalars + 8);
    if (hf[0]) return ctx->fail(IVX_ERR_INVALID, "build key id >= n_keys");
    if (hf[1]) return ctx->fail(IVX_ERR_UNSUPPORTED, "coverage: a build interval has end < start (the reference's result for it is unspecified)");
    i32 *nfirst, *nlast; u64 *pw;
    IVX_TRY(ivx_index_alloc(ctx, ix, (m ? m : 1) * sizeof(i32), (void **)&nfirst));
    IVX_TRY(ivx_index_alloc(ctx, ix, (m ? m : 1) * sizeof(i32), (void **)&nlast));
    IVX_TRY(ivx_index_alloc(ctx, ix, (m + 1) * sizeof(u64), (void **)&pw));
    hipLaunchKernelGGL(k_nodes, dim3(grid1(m + 1)), dim3(OT), 0, st, (const i64 *)rs, (const i64 *)re, m, nfirst, nlast, pw);
    IVX_TRY(ivx_scan_exclusive_u64(ctx, pw, m + 1));
    IVX_TRY(ivx_grid_build(ctx, ix, rk, nfirst, m, nkeys, &ix->cv.first, true));   // merged nodes: disjoint, ascending
    IVX_TRY(ivx_grid_build(ctx, ix, rk, nlast, m, nkeys, &ix->cv.last, true));
    ix->cv.nfirst = nfirst; ix->cv.nlast = nlast; ix->cv.pw = (const i64 *)pw;
    IVX_HIP(ctx, hipGetLastError());
    // overlap index over the merged nodes for the region-partitioned probe of big batches
    IVX_TRY(build_route_view(ctx, ix, ix->cv.first));
    IVX_TRY(ivx_join_build(ctx, ix, rk, nfirst, nlast, m));             // (synchronises)
    route_view_ready(ctx, ix);
    ix->flags |= IVX_IXF_REGION_ROWVAL;
    return IVX_OK;
}

ivx_status ivx_nearest_build(ivx_ctx *ctx, ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n)
{
    hipStream_t st = ctx->stream;
    const u32 nkeys = ix->nkeys;
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(u64), st));
    const u64 na = n ? n : 1;
    ivx_nrec *rs, *re; u32 *ks;
    IVX_TRY(ivx_index_alloc(ctx, ix, na * sizeof(ivx_nrec), (void **)&rs));
    IVX_TRY(ivx_index_alloc(ctx, ix, na * sizeof(ivx_nrec), (void **)&re));
    IVX_TRY(ctx->get_scratch(WS_T0, na * 4, (void **)&ks));
    IVX_TRY(nearest_sorted_records(ctx, key, s, e, n, nkeys, ks, rs, re));
    if (n) {
        SegMaxIn in{ks, rs}; SegMaxOut out{rs};
        IVX_TRY((ivxscan::inclusive_f<SegMaxOp>(ctx, in, out, n)));                      // prefix_max_end :58-63
    }
    // the records' fields are the grids' value arrays (stride 4 words): sorted columns, prefix max: all ascending per key
    IVX_TRY(ivx_grid_build(ctx, ix, ks, &rs->a, n, nkeys, &ix->nv.by_start, true, 4));
    IVX_TRY(ivx_grid_build(ctx, ix, ks, &re->a, n, nkeys, &ix->nv.by_end, true, 4));
    IVX_TRY(ivx_grid_build(ctx, ix, ks, &rs->pmax, n, nkeys, &ix->nv.pmax, true, 4));
    ix->nv.rs = rs; ix->nv.re = re;
    IVX_TRY(build_route_view(ctx, ix, ix->nv.by_start));
    IVX_HIP(ctx, hipGetLastError());
    IVX_TRY(check_keyflag(ctx));                                          // synchronises
    route_view_ready(ctx, ix);
    return IVX_OK;
}

// ---------------------------------------------------------------------------- probes (called from ivx_capi.hip)

// big batches the LDS-slice pipeline cannot take (too many regions, or build rows with end < start): route the probe
// rows by coordinate region, gather from the rank grids in that order, put the values back
static bool rowval_routed_wanted(const ivx_index *ix, u64 n)
{
    if (ix->nroute_nreg == 0) return false;
    const char *f = getenv("IVX_ROWVAL_PATH");                          // tests: "direct" | "regions" | "routed"
    if (f && !strcmp(f, "routed")) return true;
    if (f) return false;
    return n >= (1u << 21);
}

static ivx_status rowval_routed(ivx_ctx *ctx, const ivx_index *ix, bool coverage, const u32 *key, const i32 *s, const i32 *e, u64 n, int strict, i64 *out)
{
    hipStream_t st = ctx->stream;
    ivx_routed R;
    IVX_TRY(ivx_route_rows(ctx, ix->nroute, key, s, e, n, strict ? 1u : 0u, &R));
    i64 *vd;
    IVX_TRY(ctx->get_scratch(WS_T3, n * sizeof(i64), (void **)&vd));
    const u32 grid = (ivx_stream_grid(n, OT * 4) + 7u) & ~7u;
    if (coverage) hipLaunchKernelGGL((k_rowval_routed<true>), dim3(grid), dim3(OT), 0, st, ix->gs, ix->ge, ix->cv, ix->nroute.rkey, ix->nroute_nreg, R.pse, R.hist, R.nblk, vd, R.unsorted);
    else hipLaunchKernelGGL((k_rowval_routed<false>), dim3(grid), dim3(OT), 0, st, ix->gs, ix->ge, ix->cv, ix->nroute.rkey, ix->nroute_nreg, R.pse, R.hist, R.nblk, vd, R.unsorted);
    IVX_TRY(ivx_unroute_pair(ctx, R, n, nullptr, vd, nullptr, nullptr, out, 0));      // rows that could not be routed: 0, as the reference answers
    // rows that came in region order were not moved: the plain kernels answer them in place
    if (coverage) hipLaunchKernelGGL(k_probe_coverage, dim3(ivx_stream_grid(n, OT * 4)), dim3(OT), 0, st, ix->cv, key, s, e, n, strict, out, R.unsorted);
    else hipLaunchKernelGGL(k_probe_count, dim3(ivx_stream_grid(n, OT * 4)), dim3(OT), 0, st, ix->gs, ix->ge, key, s, e, n, strict, out, R.unsorted);
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

ivx_status ivx_count_probe(ivx_ctx *ctx, const ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n, int strict, i64 *out)
{
    if (n == 0) return IVX_OK;
    if (rowval_regions_wanted(ix, n)) return ivx_rowval_probe_regions(ctx, ix->jv, ix->jv_nreg, IVX_RV_COUNT, key, s, e, n, strict, out, nullptr, ix->jv_filter, ix->jv_pk24, ix->jv_fast);
    if (rowval_routed_wanted(ix, n)) return rowval_routed(ctx, ix, false, key, s, e, n, strict, out);
    hipLaunchKernelGGL(k_probe_count, dim3(ivx_stream_grid(n, OT)), dim3(OT), 0, ctx->stream, ix->gs, ix->ge, key, s, e, n, strict, out, (const u32 *)nullptr);
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

ivx_status ivx_coverage_probe(ivx_ctx *ctx, const ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n, int strict, i64 *out)
{
    if (n == 0) return IVX_OK;
    if (rowval_regions_wanted(ix, n)) return ivx_rowval_probe_regions(ctx, ix->jv, ix->jv_nreg, IVX_RV_COVERAGE, key, s, e, n, strict, out, nullptr, ix->jv_filter, ix->jv_pk24, ix->jv_fast);
    if (rowval_routed_wanted(ix, n)) return rowval_routed(ctx, ix, true, key, s, e, n, strict, out);
    hipLaunchKernelGGL(k_probe_coverage, dim3(ivx_stream_grid(n, OT)), dim3(OT), 0, ctx->stream, ix->cv, key, s, e, n, strict, out, (const u32 *)nullptr);
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

ivx_status ivx_nearest_probe(ivx_ctx *ctx, const ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n,
                             int strict, u32 k, int include_overlaps, u32 *ob, u32 *op, i64 *od, u64 cap, u64 *rows)
{
    *rows = 0;
    if (n == 0) return IVX_OK;
    hipStream_t st = ctx->stream;
    if (k <= 1) {
        // k == 0 yields no candidate, i.e. one NULL row per probe row, like k == 1 on an empty index (nearest_index.rs:111)
        if (cap < n) { *rows = n; return ctx->fail(IVX_ERR_CAPACITY, "nearest: output buffers too small"); }
        bool routed = ix->nroute_nreg > 0 && n >= (1u << 21);
        if (const char *f = getenv("IVX_NEAREST_PATH")) routed = !strcmp(f, "routed") ? ix->nroute_nreg > 0 : (!strcmp(f, "direct") ? false : routed);
        if (k == 1 && routed) {
            // big batches: route the probe rows by coordinate region first, probe in that order, put the answers back
            ivx_routed R;
            IVX_TRY(ivx_route_rows(ctx, ix->nroute, key, s, e, n, strict ? 1u : 0u, &R));
            u32 *vb; i64 *vd = nullptr;
            IVX_TRY(ctx->get_scratch(WS_T2, n * sizeof(u32), (void **)&vb));
            if (od) IVX_TRY(ctx->get_scratch(WS_T3, n * sizeof(i64), (void **)&vd));
            const int pmax_first = getenv("IVX_NEAREST_PMAX_FIRST") ? atoi(getenv("IVX_NEAREST_PMAX_FIRST")) : 1;
            // five 256-thread workgroups per CU, not eight: the rows in flight on an XCD then span less of the index than its L2
            // holds (grid 2048 -> 1280: 3.75 -> 3.17 ms per 50M rows; 1024..1536 are within 3 %, 1792 and 2048 fall off)
            hipLaunchKernelGGL(k_nearest_routed, dim3((ivx_stream_grid(n, NR_T * 4, 1280u) + 7u) & ~7u), dim3(NR_T), 0, st, ix->nv, ix->nroute.rkey, ix->nroute_nreg, R.pse,
                               R.hist, R.nblk, strict ? 1u : 0u, include_overlaps, vb, vd, R.unsorted, pmax_first);
            IVX_TRY(ivx_unroute_pair(ctx, R, n, vb, vd, ob, op, od, -1));
            // rows that came in region order were not moved: the plain kernel answers them in place
            hipLaunchKernelGGL(k_probe_nearest1, dim3(ivx_stream_grid(n, OT * 4)), dim3(OT), 0, st, ix->nv, key, s, e, n, strict, include_overlaps, ob, op, od, R.unsorted);
        } else if (k == 1) {
            hipLaunchKernelGGL(k_probe_nearest1, dim3(ivx_stream_grid(n, OT)), dim3(OT), 0, st, ix->nv, key, s, e, n, strict, include_overlaps, ob, op, od, (const u32 *)nullptr);
        } else {
            u32 *cnt; u64 *offs;
            IVX_TRY(ctx->get_scratch(WS_T3, n * sizeof(u32), (void **)&cnt));
            IVX_TRY(ctx->get_scratch(WS_T4, (n + 1) * sizeof(u64), (void **)&offs));
            IVX_HIP(ctx, hipMemsetAsync(cnt, 0, n * sizeof(u32), st));
            hipLaunchKernelGGL(k_rows_of, dim3(grid1(n + 1)), dim3(OT), 0, st, (const u32 *)cnt, n, offs);
            IVX_TRY(ivx_scan_exclusive_u64(ctx, offs, n + 1));
            hipLaunchKernelGGL(k_nearest_emit, dim3(ivx_stream_grid(n, OT)), dim3(OT), 0, st, n, 1u, (const u32 *)cnt, (const i64 *)nullptr, (const u32 *)cnt, (const u64 *)offs, cap, ob, op, od);
        }
        IVX_HIP(ctx, hipGetLastError());
        *rows = n;
        return IVX_OK;
    }
    u32 *tmp, *cnt; u64 *offs; i64 *tmpd = nullptr;
    IVX_TRY(ctx->get_scratch(WS_T2, n * (u64)k * sizeof(u32), (void **)&tmp));
    if (od) IVX_TRY(ctx->get_scratch(WS_T5, n * (u64)k * sizeof(i64), (void **)&tmpd));
    IVX_TRY(ctx->get_scratch(WS_T3, n * sizeof(u32), (void **)&cnt));
    IVX_TRY(ctx->get_scratch(WS_T4, (n + 1) * sizeof(u64), (void **)&offs));
    hipLaunchKernelGGL(k_probe_nearestk, dim3(ivx_stream_grid(n, OT * 2)), dim3(OT), 0, st, ix->nv, key, s, e, n, strict, include_overlaps, k, tmp, tmpd, cnt);
    hipLaunchKernelGGL(k_rows_of, dim3(grid1(n + 1)), dim3(OT), 0, st, (const u32 *)cnt, n, offs);
    IVX_TRY(ivx_scan_exclusive_u64(ctx, offs, n + 1));
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 3, offs + n, sizeof(u64), hipMemcpyDeviceToHost, st));
    IVX_HIP(ctx, hipStreamSynchronize(st));
    const u64 total = ctx->h_scalars[3];
    *rows = total;
    if (total > cap) return ctx->fail(IVX_ERR_CAPACITY, "nearest: output buffers too small");
    hipLaunchKernelGGL(k_nearest_emit, dim3(ivx_stream_grid(n, OT * 2)), dim3(OT), 0, st, n, k, (const u32 *)tmp, (const i64 *)tmpd, (const u32 *)cnt, (const u64 *)offs, cap, ob, op, od);
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}
