// ivx_runs.hip -- interval-merge sweep as two device-wide scans (see ivx_runs.hpp).
#include <cstdlib>
#include "ivx_runs.hpp"
#include "ivx_scan.hpp"

namespace {

constexpr int RT = 256;

// cur_end transfer function:  konst ? (x -> c)  :  (x -> x < M ? c : x)
struct MState { i64 M; i64 c; u32 konst; u32 pad; };

struct MergeOp {
    using T = MState;
    __host__ __device__ static T identity() { T t; t.M = INT64_MIN; t.c = INT64_MIN; t.konst = 0; t.pad = 0; return t; }
    // a covers the earlier rows: result = b o a
    __device__ static T combine(const T &a, const T &b)
    {
        if (b.konst) return b;
        T r;
        r.pad = 0;
        if (a.konst) { r.konst = 1; r.M = 0; r.c = a.c < b.M ? b.c : a.c; return r; }
        r.konst = 0;
        if (b.M > a.M) { r.M = b.M; r.c = b.c; }
        else { r.M = a.M; r.c = a.c < b.M ? b.c : a.c; }
        return r;
    }
    __device__ static T shfl_up(const T &v, int d)
    {
        T r;
        r.M = __shfl_up(v.M, d, IVX_WAVE); r.c = __shfl_up(v.c, d, IVX_WAVE);
        r.konst = __shfl_up(v.konst, d, IVX_WAVE); r.pad = 0;
        return r;
    }
};

struct HeadAcc { u32 heads; u32 last_head; };       // #run heads so far, index of the latest one
struct HeadOp {
    using T = HeadAcc;
    __host__ __device__ static T identity() { T t; t.heads = 0; t.last_head = 0; return t; }
    __device__ static T combine(const T &a, const T &b)
    {
        T r; r.heads = a.heads + b.heads; r.last_head = a.last_head > b.last_head ? a.last_head : b.last_head; return r;
    }
    __device__ static T shfl_up(const T &v, int d)
    {
        T r; r.heads = __shfl_up(v.heads, d, IVX_WAVE); r.last_head = __shfl_up(v.last_head, d, IVX_WAVE); return r;
    }
};

__device__ __forceinline__ i64 sat_add(i64 a, i64 b)
{
    i64 r;
    if (__builtin_add_overflow(a, b, &r)) return b > 0 ? INT64_MAX : INT64_MIN;
    return r;
}
__device__ __forceinline__ i64 sat_sub_floor(i64 a, i64 b)     // b >= 0: only underflow is possible
{
    i64 r;
    if (__builtin_sub_overflow(a, b, &r)) return INT64_MIN;
    return r;
}

// merge.rs:291-296:  s <= cur_end (+) min_dist   (strict: <), (+) saturating
__device__ __forceinline__ bool merges(i64 s, i64 cur_end, i64 d, int strict)
{
    const i64 boundary = sat_add(cur_end, d);
    return strict ? (s < boundary) : (s <= boundary);
}

// element i of the cur_end scan: the transfer function of row i (computed from the sorted columns, never stored)
struct StateIn {
    SortedRows r; i64 d; int strict;
    __device__ MState operator()(u64 i) const { i64 s; return at(i, s); }
    // ... and the row's start, for callers that test the row against cur_end themselves
    __device__ MState at(u64 i, i64 &s) const
    {
        u32 k; i64 e; bool first;
        r.get(i, k, s, e, first);
        MState t; t.pad = 0;
        if (first || (strict && s == INT64_MAX)) {          // strict: s < anything never holds at i64::MAX
            t.konst = 1; t.M = 0; t.c = e;
        } else {
            // smallest cur_end that still merges row i
            const i64 T = strict ? sat_sub_floor(s + 1, d) : sat_sub_floor(s, d);
            t.konst = 0; t.M = T > e ? T : e; t.c = e;
        }
        return t;
    }
};
// after the inclusive scan the state of row i is a constant function: its value is cur_end after row i
// (cur_end is always some row's end: with narrow rows it is kept as a 32-bit offset from min_e as well)
struct CurEnd {
    i64 *wide; u32 *narrow; i64 min_e;
    __device__ __forceinline__ i64 at(u64 i) const { return narrow ? (i64)((u64)min_e + narrow[i]) : wide[i]; }
};
struct CurEndOut {
    CurEnd ce;
    __device__ void operator()(u64 i, const MState &t) const
    {
        if (ce.narrow) ce.narrow[i] = (u32)((u64)t.c - (u64)ce.min_e); else ce.wide[i] = t.c;
    }
};
// element i of the run-head scan: does row i start a run (merge.rs:291-296 against cur_end after row i-1)
struct HeadIn {
    SortedRows r; CurEnd cur_end; i64 d; int strict;
    __device__ HeadAcc operator()(u64 i) const
    {
        u32 k; i64 s, e; bool first;
        r.get(i, k, s, e, first);
        const bool head = first || !merges(s, cur_end.at(i - 1), d, strict);
        HeadAcc h; h.heads = head ? 1u : 0u; h.last_head = head ? (u32)i : 0u;
        return h;
    }
};
// the run-head scan's results: row i closes its run when row i + 1 starts one (or is missing) -- the run is emitted right
// here (its id is the number of heads so far, its first row the latest head); ha (nullable) keeps the per-row values for
// cluster()
struct HeadEmitOut {
    HeadAcc *ha;
    SortedRows r; CurEnd cur_end; i64 d; int strict; u64 n; ivx_runs_out out; u64 *m;
    __device__ void operator()(u64 i, const HeadAcc &h) const
    {
        if (ha) ha[i] = h;
        bool last = i + 1 == n;
        if (!last) {
            u32 k1; i64 s1, e1; bool first1;
            r.get(i + 1, k1, s1, e1, first1);
            last = first1 || !merges(s1, cur_end.at(i), d, strict);
        }
        if (last) {
            const u32 id = h.heads - 1;
            u32 kh; i64 sh, eh; bool fh;
            r.get(h.last_head, kh, sh, eh, fh);               // the run's first row: its key is the run's, its start the run's
            if (out.key) out.key[id] = kh;
            if (out.start) out.start[id] = sh;
            if (out.end) out.end[id] = cur_end.at(i);
            if (out.count) out.count[id] = (i64)(i - h.last_head + 1);
        }
        if (i + 1 == n) *m = h.heads;
    }
};

// The cur_end scan's second pass (ivxscan::k_apply_f over MergeOp) which, holding cur_end BEFORE every row of its tile in
// registers anyway, also decides the run heads and leaves the tile's head summary: the run-head scan then needs no
// reduce pass of its own over the columns.
__global__ __launch_bounds__(ivxscan::T_) void k_merge_apply(StateIn in, CurEndOut out, u64 n, const MState *__restrict__ offs, HeadAcc *__restrict__ hsums)
{
    using namespace ivxscan;
    __shared__ MState lds[T_ / IVX_WAVE + 1];
    __shared__ MState edge[T_ / IVX_WAVE];
    __shared__ HeadAcc hl[T_ / IVX_WAVE + 1];
    const u64 base = (u64)blockIdx.x * TILE_ + (u64)threadIdx.x * I_;
    MState v[I_];
    i64 rs[I_];
    MState s = MergeOp::identity();
#pragma unroll
    for (int i = 0; i < I_; i++) { rs[i] = 0; v[i] = base + i < n ? in.at(base + i, rs[i]) : MergeOp::identity(); s = MergeOp::combine(s, v[i]); }
    MState tot;
    MState inc = block_incl<MergeOp>(s, lds, &tot);
    MState run = offs ? offs[blockIdx.x] : MergeOp::identity();
    MState prev = MergeOp::shfl_up(inc, 1);
    if (lane_id() == IVX_WAVE - 1) edge[threadIdx.x / IVX_WAVE] = inc;
    __syncthreads();
    if (threadIdx.x == 0) prev = MergeOp::identity();
    else if (lane_id() == 0) prev = edge[threadIdx.x / IVX_WAVE - 1];
    run = MergeOp::combine(run, prev);
    HeadAcc hacc = HeadOp::identity();
#pragma unroll
    for (int i = 0; i < I_; i++) {
        const u64 idx = base + i;
        if (idx < n) {
            // (a key's first row -- and a strict row at i64::MAX -- is a constant state: a head; otherwise `run` covers row 0,
            //  hence is constant, and run.c is cur_end after row idx - 1)
            const bool head = v[i].konst || !merges(rs[i], run.c, in.d, in.strict);
            HeadAcc h; h.heads = head ? 1u : 0u; h.last_head = head ? (u32)idx : 0u;
            hacc = HeadOp::combine(hacc, h);
            run = MergeOp::combine(run, v[i]);
            out(idx, run);
        }
    }
    HeadAcc htot;
    block_incl<HeadOp>(hacc, hl, &htot);
    if (threadIdx.x == 0) hsums[blockIdx.x] = htot;
}

// The run-head scan's second pass with the tile's rows STRIPED over the threads (thread t takes rows t, t + 256, ... of
// the tile, one workgroup scan per stripe): consecutive lanes hold consecutive rows, so the runs they emit -- and the
// per-row values -- go out in whole cache lines (with a thread owning four consecutive rows every store is strided).
__global__ __launch_bounds__(ivxscan::T_) void k_head_apply(HeadIn in, HeadEmitOut out, u64 n, const HeadAcc *__restrict__ offs)
{
    using namespace ivxscan;
    __shared__ HeadAcc lds[T_ / IVX_WAVE + 1];
    HeadAcc carry = offs ? offs[blockIdx.x] : HeadOp::identity();
    const u64 base = (u64)blockIdx.x * TILE_ + threadIdx.x;
    HeadAcc v[I_];
#pragma unroll
    for (int k = 0; k < I_; k++) { const u64 i = base + (u64)k * T_; v[k] = i < n ? in(i) : HeadOp::identity(); }
#pragma unroll
    for (int k = 0; k < I_; k++) {
        const u64 i = base + (u64)k * T_;
        HeadAcc tot;
        const HeadAcc inc = block_incl<HeadOp>(v[k], lds, &tot);
        if (i < n) out(i, HeadOp::combine(carry, inc));
        carry = HeadOp::combine(carry, tot);
    }
}

}  // namespace

// cur_end scan -> run-head scan -> runs; both scans compute their elements from the sorted columns on the fly, the first
// one's second pass doubles as the second one's first, and the runs are emitted by the last pass itself: 68 bytes of
// traffic per row (232 with a materialised state array, 108 with two separate scans and an emit pass).  Leaves cur_end
// (n entries) in WS_T5 and, if want_ha, the per-row head counts in WS_T6.
static ivx_status sweep(ivx_ctx *ctx, const SortedRows &rows, u64 n,
                        i64 min_dist, int strict, const ivx_runs_out &out, const HeadAcc **ha_out, u64 *m, bool want_ha)
{
    using namespace ivxscan;
    hipStream_t stq = ctx->stream;
    i64 *cur_end; HeadAcc *ha = nullptr;
    IVX_TRY(ctx->get_scratch(WS_T5, n * sizeof(i64), (void **)&cur_end));
    if (want_ha) IVX_TRY(ctx->get_scratch(WS_T6, n * sizeof(HeadAcc), (void **)&ha));
    const u64 nblk = (n + TILE_ - 1) / TILE_;
    MState *sums; HeadAcc *hsums;
    IVX_TRY(ctx->get_scratch(WS_SCAN0, nblk * sizeof(MState), (void **)&sums));
    IVX_TRY(ctx->get_scratch(WS_SCAN2, nblk * sizeof(HeadAcc), (void **)&hsums));
    const StateIn in{rows, min_dist, strict};
    if (nblk > 1) {
        hipLaunchKernelGGL((k_reduce_f<MergeOp, StateIn>), dim3((u32)nblk), dim3(T_), 0, stq, in, n, sums);
        IVX_TRY((scan_rec<MergeOp, false>(ctx, sums, nblk, 1, WS_SCAN0)));
    }
    // (want_ha: cluster() reads cur_end's successor table, the run ends, as i64 -- only the plain merge narrows)
    const CurEnd ce{cur_end, (rows.s32 != nullptr && !want_ha) ? (u32 *)cur_end : nullptr, rows.min_e};
    hipLaunchKernelGGL(k_merge_apply, dim3((u32)nblk), dim3(T_), 0, stq, in, CurEndOut{ce}, n, nblk > 1 ? (const MState *)sums : (const MState *)nullptr, hsums);
    if (nblk > 1) IVX_TRY((scan_rec<HeadOp, false>(ctx, hsums, nblk, 1, WS_SCAN0)));
    u64 *d_m = ctx->d_scalars + 2;
    const HeadIn hin{rows, ce, min_dist, strict};
    const HeadEmitOut hout{ha, rows, ce, min_dist, strict, n, out, d_m};
    hipLaunchKernelGGL(k_head_apply, dim3((u32)nblk), dim3(T_), 0, stq, hin, hout, n, nblk > 1 ? (const HeadAcc *)hsums : (const HeadAcc *)nullptr);
    IVX_HIP(ctx, hipGetLastError());
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 2, d_m, sizeof(u64), hipMemcpyDeviceToHost, stq));
    *ha_out = ha;
    (void)m;
    return IVX_OK;
}

namespace {

// --------------------------------------------------------------------------------- the merge sweep over packed words
// The sweep over the sort's packed 8-byte words (Pack64) -- no unpack pass, no per-row state in memory -- for the inputs
// every caller meets: rows with start <= end, min_dist >= 0 (and, strict with min_dist = 0, no empty row).  For those
//   * cur_end before a row is simply the largest end among the earlier rows of its key -- a run head starts beyond every
//     earlier end, so the maximum over the key is the maximum over the current run;
//   * the order of rows with the same (key, start) does not matter: the first of them decides head-or-not by the same test,
//     the others merge into it, cur_end after the group is the maximum either way -- so the words need only be sorted on
//     their (key, start) bits, not repaired by end.
// With V = (key + 1) << bits_e | (end - min_e) the segmented running maximum is a plain one (a later key's V beats every
// earlier one).  Three kernels over tiles of 4096 words with two scans of one word per tile between them:
//   k_pk_max    the tile's largest V (only rows of the tile's last key can hold it: striped loads, no per-row key lookup)
//   k_pk_runs<0>  V-maximum before the tile -> run heads -> (heads, latest head) of the tile
//   k_pk_runs<1>  the same again, and every head closes the run before it (it holds cur_end before itself = that run's
//               end, and the previous head's row number) and opens its own; row n - 1 closes the last run.
//   k_pk_runs<2>  cluster(): instead of emitting runs, every row's run number and run start (+ k_pk_cluster_fin)
// A thread takes 8 CONSECUTIVE words (coalesced loads, turned through LDS per wavefront): the key of a word's linearised
// (key, start) is then a lookup for the first word and a compare for the others.
// (A single kernel with both scans chained through per-tile status words -- decoupled look-back -- was built first: 1.66 ms
// for 200 M rows against 1.2 ms for these three; its tiles wait on each other twice and it has to spin.  DESIGN.md section 3.)
constexpr int FT = 512, FI = 8, FTILE = FT * FI, FWV = FT / IVX_WAVE;

struct MaxPayOp {                                  // u64 maximum
    using T = u64;
    __host__ __device__ static T identity() { return 0; }
    __device__ static T combine(const T &a, const T &b) { return a > b ? a : b; }
    __device__ static T shfl_up(const T &v, int d) { return __shfl_up(v, d, IVX_WAVE); }
};
struct HeadPayOp {                                 // heads so far << 31 | (row number of the latest head) + 1
    using T = u64;
    __host__ __device__ static T identity() { return 0; }
    __device__ static T combine(const T &a, const T &b)
    {
        const u64 la = a & 0x7FFFFFFFull, lb = b & 0x7FFFFFFFull;
        return (((a >> 31) + (b >> 31)) << 31) | (la > lb ? la : lb);
    }
    __device__ static T shfl_up(const T &v, int d) { return __shfl_up(v, d, IVX_WAVE); }
};

__device__ __forceinline__ u64 f_shr(u64 x, u32 sh) { return sh >= 64 ? 0 : x >> sh; }
__device__ __forceinline__ u64 f_low(u64 x, u32 bits) { return bits >= 64 ? x : x & ((1ull << bits) - 1); }
// first k with base[k + 1] > lin (keys without rows are skipped)
__device__ __forceinline__ u32 f_key(const u64 *base, u32 nkeys, u64 lin)
{
    u32 a = 0, b = nkeys;
    while (a < b) { const u32 mid = (a + b) >> 1; if (base[mid + 1] > lin) b = mid; else a = mid + 1; }
    return a < nkeys ? a : (nkeys ? nkeys - 1 : 0);                    // (a key id >= n_keys -- reported by the caller -- packs garbage)
}

template <bool LIN>
__global__ __launch_bounds__(FT) void k_pk_max(const u64 *__restrict__ w, u64 n, Pack64 p, u64 *__restrict__ agg)
{
    extern __shared__ u64 s_tab[];                                      // LIN: base[nkeys + 1]
    __shared__ u64 red[FWV];
    if (LIN) for (u32 k = threadIdx.x; k <= p.nkeys; k += FT) s_tab[k] = p.base[k];
    const u64 t0 = (u64)blockIdx.x * FTILE;
    const u32 tn = (u32)(n - t0 < (u64)FTILE ? n - t0 : (u64)FTILE);
    const u32 be = p.bits_e, bs = p.bits_s;
    u64 x[FI];
#pragma unroll
    for (int q = 0; q < FI; q++) { const u32 j = (u32)q * FT + threadIdx.x; x[q] = j < tn ? w[t0 + j] : 0ull; }
    const u64 wl = w[t0 + tn - 1];                                      // the tile's last row: the tile's last key
    __syncthreads();
    u32 kl; u64 lo;                                                     // rows of that key: word >= lo
    if (LIN) { kl = f_key(s_tab, p.nkeys, f_shr(wl, be)); lo = s_tab[kl] << be; }
    else { kl = (u32)f_shr(wl, bs + be); lo = bs + be >= 64 ? 0 : (u64)kl << (bs + be); }
    u64 m = 0;
#pragma unroll
    for (int q = 0; q < FI; q++) { const u64 eo = f_low(x[q], be); if ((u32)q * FT + threadIdx.x < tn && x[q] >= lo && eo > m) m = eo; }
#pragma unroll
    for (int dd = IVX_WAVE / 2; dd > 0; dd >>= 1) { const u64 o = __shfl_xor(m, dd, IVX_WAVE); m = o > m ? o : m; }
    if (lane_id() == 0) red[threadIdx.x / IVX_WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 1; j < FWV; j++) m = red[j] > m ? red[j] : m;
        agg[blockIdx.x] = ((u64)(kl + 1) << be) | m;
    }
}

// NARROW (decided on the host: bits_e <= 32, the word's upper part -- `lin` -- fits 32 bits, every coordinate and min_dist within
// +-2^61 so that differences cannot overflow): a wavefront whose 512 rows share ONE key -- all but a few dozen of a launch --
// works on 32-bit (lin, end offset) pairs with the key's tables in scalar registers; the run-head test is the sign of
//   (start - cur_end) - min_dist [- 1]  =  (lin - base) - x + K ,   K = kmin - min_e - min_dist - (strict ? 0 : 1).
// The generic sections (any wavefront of a launch that is not NARROW, wavefronts that straddle a key or the end of the input)
// carry a key per row and rebuild 64-bit V and the i64 start from the word where they need them.
// MODE 0: the tile's head summary.  1: emit the runs (merge).  2: cluster() -- every row's run number and its run's start,
// the runs' ends by number, per key the runs before its first and up to its last.
struct PkCluster { u32 *rid; i64 *cstart; i64 *run_end; u32 *kfirst, *klast; };
template <bool LIN, int MODE, bool NARROW>
__global__ __launch_bounds__(FT, MODE == 0 ? 8 : 5) void k_pk_runs(const u64 *__restrict__ w, u64 n, Pack64 p, i64 d, int strict, const u64 *__restrict__ pre1,
                                                  u64 *__restrict__ agg2, const u64 *__restrict__ pre2, ivx_runs_out out, PkCluster cl, u64 *d_m)
{
    constexpr bool EMIT = MODE != 0;
    extern __shared__ u64 s_tab[];                                      // LIN: base[nkeys + 1], kmin[nkeys]
    __shared__ u64 s_x[FWV][IVX_WAVE * FI + IVX_WAVE];                  // a wavefront's 512 words, one spare slot per lane
    __shared__ u64 s_wmax[FWV];
    __shared__ u32 s_wh[FWV], s_wl[FWV];
    const u32 tid = threadIdx.x, wv = tid / IVX_WAVE, ln = lane_id();
    const u64 *s_base = s_tab;
    const u64 *s_kmin = s_tab + p.nkeys + 1;
    if (LIN) {
        for (u32 k = tid; k <= p.nkeys; k += FT) s_tab[k] = p.base[k];
        for (u32 k = tid; k < p.nkeys; k += FT) s_tab[p.nkeys + 1 + k] = (u64)p.kmin[k];
    }
    const u32 tile = blockIdx.x;
    const u64 w0 = (u64)tile * FTILE + (u64)wv * (IVX_WAVE * FI);
    const bool wfull = w0 + (u64)IVX_WAVE * FI <= n;                     // (wave-uniform)
    // ---- the wavefront's 512 consecutive words: coalesced loads, then every lane takes its 8 consecutive ones
    if (wfull) {
#pragma unroll
        for (int q = 0; q < FI; q++) { const u32 e = (u32)q * IVX_WAVE + ln; s_x[wv][e + (e >> 3)] = w[w0 + e]; }
    } else {
#pragma unroll
        for (int q = 0; q < FI; q++) { const u32 e = (u32)q * IVX_WAVE + ln; s_x[wv][e + (e >> 3)] = w0 + e < n ? w[w0 + e] : 0ull; }
    }
    const u64 P = pre1[tile], P2 = EMIT ? pre2[tile] : 0ull;
    __syncthreads();
    u64 r[FI];
#pragma unroll
    for (int i = 0; i < FI; i++) r[i] = s_x[wv][ln * (FI + 1) + i];
    const u64 row0 = w0 + (u64)ln * FI;
    const u32 nv = row0 >= n ? 0u : (n - row0 < (u64)FI ? (u32)(n - row0) : (u32)FI);
    const u32 be = p.bits_e, bs = p.bits_s;
    const u32 emask = be >= 32 ? 0xFFFFFFFFu : (1u << be) - 1u;

    // ---- does the whole wavefront sit in one key?  (wave-uniform; kA and its tables end up in scalar registers)
    bool fast = false;
    u32 kA = 0, baseA = 0;
    i64 koffA = 0;
    if (NARROW && wfull) {
        const u32 lin_a = (u32)f_shr(s_x[wv][0], be), lin_b = (u32)f_shr(s_x[wv][(IVX_WAVE - 1) * (FI + 1) + FI - 1], be);
        if (LIN) {
            // (the rows before this tile end in key (P >> be) - 1: the search starts there and usually stops at once)
            kA = P ? (u32)(P >> be) - 1u : 0u;
            while (kA + 1 < p.nkeys && (u64)lin_a >= s_base[kA + 1]) kA++;
            fast = kA + 1 >= p.nkeys || (u64)lin_b < s_base[kA + 1];
            baseA = (u32)s_base[kA]; koffA = (i64)s_kmin[kA];
        } else {
            kA = bs >= 32 ? 0u : lin_a >> bs;                            // (one key: no key bits at all)
            fast = (bs >= 32 ? 0u : lin_b >> bs) == kA;
            baseA = bs >= 32 ? 0u : kA << bs; koffA = p.min_s;
        }
        kA = __builtin_amdgcn_readfirstlane(kA); baseA = __builtin_amdgcn_readfirstlane(baseA);
        fast = __builtin_amdgcn_readfirstlane((u32)fast) != 0;
    }

    // generic rows: the key per row; V and the start come from the word
    u32 k[FI];
    auto Vof = [&](int i) -> u64 { return ((u64)(k[i] + 1) << be) | f_low(r[i], be); };
    auto sof = [&](int i) -> i64 {
        if (LIN) return (i64)(s_kmin[k[i]] + (f_shr(r[i], be) - s_base[k[i]]));
        return (i64)((u64)p.min_s + f_low(f_shr(r[i], be), bs));
    };

    // ---- section 1: the wavefront's largest V, each lane's inclusive maximum
    u32 inc32 = 0;                                                      // fast
    u64 inc = 0;                                                        // generic
    u64 wagg;
    if (fast) {
        u32 tm = 0;
#pragma unroll
        for (int i = 0; i < FI; i++) { k[i] = 0; const u32 eo = (u32)r[i] & emask; tm = eo > tm ? eo : tm; }
        inc32 = wave_incl_max32(tm);
        wagg = ((u64)(kA + 1) << be) | (u32)__builtin_amdgcn_readlane((int)inc32, IVX_WAVE - 1);
    } else {
        u32 kk = 0;
        if (LIN && nv) kk = f_key(s_base, p.nkeys, f_shr(r[0], be));
        u64 tm = 0;
#pragma unroll
        for (int i = 0; i < FI; i++) {
            k[i] = 0;
            if ((u32)i < nv) {
                if (LIN) { const u64 lin = f_shr(r[i], be); while (kk + 1 < p.nkeys && lin >= s_base[kk + 1]) kk++; }
                else kk = (u32)f_shr(r[i], bs + be);
                k[i] = kk;
                const u64 v = Vof(i);
                tm = v > tm ? v : tm;
            }
        }
        inc = tm;
#pragma unroll
        for (int dd = 1; dd < IVX_WAVE; dd <<= 1) { const u64 o = __shfl_up(inc, dd, IVX_WAVE); if (ln >= (u32)dd && o > inc) inc = o; }
        wagg = __shfl(inc, IVX_WAVE - 1, IVX_WAVE);
    }
    if (ln == 0) s_wmax[wv] = wagg;
    __syncthreads();
    u64 Xw = P;                                                          // V-maximum over every row before this wavefront's
#pragma unroll
    for (int j = 0; j < FWV; j++) { const u64 v = s_wmax[j]; if ((u32)j < wv && v > Xw) Xw = v; }

    // ---- section 2: run heads (merge.rs:291-296 against cur_end before the row)
    u32 hm = 0;
    u32 x32 = 0; bool hv = false;                                        // fast: cur_end offset before the thread's first row, is there one
    u64 X = Xw;                                                          // generic
    if (fast) {
        const bool hvw = (u32)(Xw >> be) == kA + 1;
        // (lane 0 of a wavefront that opens a key keeps the PREVIOUS key's cur_end in x32: its first row closes that run)
        x32 = (u32)Xw & emask;
        { const u32 up = wave_prev32(inc32); if (ln > 0) { x32 = hvw ? x32 : 0u; x32 = up > x32 ? up : x32; } }
        hv = hvw || ln > 0;
        const i64 K = koffA - p.min_e - d - (strict ? 0 : 1);
        u32 x = x32;
#pragma unroll
        for (int i = 0; i < FI; i++) {
            const u32 lin = (u32)(r[i] >> be), eo = (u32)r[i] & emask;
            const i64 t = (i64)(u64)(lin - baseA) - (i64)(u64)x + K;
            const bool head = (i == 0 && !hv) || t >= 0;
            hm |= head ? 1u << i : 0u;
            x = ((i == 0 && !hv) || eo > x) ? eo : x;
        }
    } else {
        { const u64 up = __shfl_up(inc, 1, IVX_WAVE); if (ln > 0 && up > X) X = up; }
        u64 x = X;
#pragma unroll
        for (int i = 0; i < FI; i++) {
            if ((u32)i < nv) {
                const bool same = f_shr(x, be) == (u64)k[i] + 1;
                const bool head = !same || !merges(sof(i), (i64)((u64)p.min_e + f_low(x, be)), d, strict);
                hm |= head ? 1u << i : 0u;
                const u64 v = Vof(i);
                x = v > x ? v : x;
            }
        }
    }
    const u32 h = (u32)__builtin_popcount(hm);
    const u32 lastp1 = hm ? (u32)(row0 + (31u - (u32)__builtin_clz(hm))) + 1u : 0u;
    const u32 hinc = wave_incl_sum32(h), linc = wave_incl_max32(lastp1);
    if (ln == IVX_WAVE - 1) { s_wh[wv] = hinc; s_wl[wv] = linc; }
    __syncthreads();
    u32 hpre = 0, lpre = 0, hagg = 0, lagg = 0;
#pragma unroll
    for (int j = 0; j < FWV; j++) {
        const u32 a = s_wh[j], b = s_wl[j];
        if ((u32)j < wv) { hpre += a; lpre = b > lpre ? b : lpre; }
        hagg += a; lagg = b > lagg ? b : lagg;
    }
    if (!EMIT) {
        if (tid == 0) agg2[tile] = ((u64)hagg << 31) | lagg;
        return;
    }
    u32 rid = (u32)(P2 >> 31) + hpre + (hinc - h);                        // runs that start before this thread's rows
    u32 prev = (u32)(P2 & 0x7FFFFFFFull);                                 // (row number of the latest head before them) + 1
    prev = lpre > prev ? lpre : prev;
    { const u32 up = wave_prev32(linc); prev = up > prev ? up : prev; }
    if (MODE == 2) {
        // ---- section 3, cluster(): a head closes the run before it (run_end) and, when it opens a key, settles the keys' run
        //      ranges; then per row the number of its run and the start of that run's head (rebuilt from the head mask row by
        //      row: nothing per row is kept in registers across the passes)
        const u32 rid0 = rid, prev0 = prev;
        if (fast) {
            u32 x = x32;
            if (__ballot(hm != 0)) {
#pragma unroll
                for (int i = 0; i < FI; i++) {
                    const u32 eo = (u32)r[i] & emask;
                    if ((hm >> i) & 1u) {
                        if (prev && cl.run_end) cl.run_end[rid - 1] = (i64)((u64)p.min_e + x);
                        if (i == 0 && !hv) { cl.kfirst[kA] = rid; if (prev) cl.klast[(u32)(Xw >> be) - 1u] = rid; }
                        prev = (u32)(row0 + i) + 1; rid++;
                    }
                    x = ((i == 0 && !hv) || eo > x) ? eo : x;
                }
            } else if (row0 + FI == n) {
#pragma unroll
                for (int i = 0; i < FI; i++) { const u32 eo = (u32)r[i] & emask; x = eo > x ? eo : x; }
            }
            if (row0 + FI == n) {
                if (cl.run_end) cl.run_end[rid - 1] = (i64)((u64)p.min_e + x);
                cl.klast[kA] = rid; *d_m = rid;
            }
        } else {
            u64 x = X;
            u32 kl = 0;                                                   // key of the thread's last row
#pragma unroll
            for (int i = 0; i < FI; i++) {
                if ((u32)i < nv) {
                    kl = k[i];
                    if ((hm >> i) & 1u) {
                        if (prev && cl.run_end) cl.run_end[rid - 1] = (i64)((u64)p.min_e + f_low(x, be));
                        if (f_shr(x, be) != (u64)k[i] + 1) { cl.kfirst[k[i]] = rid; if (prev) cl.klast[(u32)f_shr(x, be) - 1u] = rid; }
                        prev = (u32)(row0 + i) + 1; rid++;
                    }
                    const u64 v = Vof(i);
                    x = v > x ? v : x;
                }
            }
            if (nv && row0 + nv == n) {
                if (cl.run_end) cl.run_end[rid - 1] = (i64)((u64)p.min_e + f_low(x, be));
                cl.klast[kl] = rid; *d_m = rid;
            }
        }
        // the rows' values leave as the words came: turned through the wavefront's LDS slots into whole-line stores
        if (cl.rid) {
            __builtin_amdgcn_wave_barrier();
            u32 rr = rid0;
#pragma unroll
            for (int i = 0; i < FI; i++) { rr += (hm >> i) & 1u; s_x[wv][ln * (FI + 1) + i] = rr - 1; }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < FI; q++) { const u32 e = (u32)q * IVX_WAVE + ln; if (w0 + e < n) cl.rid[w0 + e] = (u32)s_x[wv][e + (e >> 3)]; }
        }
        if (cl.cstart) {
            __builtin_amdgcn_wave_barrier();
            if (fast) {
                u32 lin_h = prev0 ? (u32)(w[prev0 - 1] >> be) : 0u;      // (used only when this thread's first row continues that head's run)
#pragma unroll
                for (int i = 0; i < FI; i++) {
                    if ((hm >> i) & 1u) lin_h = (u32)(r[i] >> be);
                    s_x[wv][ln * (FI + 1) + i] = (u64)koffA + (lin_h - baseA);
                }
            } else {
                i64 cs_h = 0;
                if (prev0 && nv) {                                       // the head before this thread's rows
                    const u64 wp = w[prev0 - 1];
                    if (LIN) { const u64 lin = f_shr(wp, be); const u32 kh = f_key(s_base, p.nkeys, lin); cs_h = (i64)(s_kmin[kh] + (lin - s_base[kh])); }
                    else cs_h = (i64)((u64)p.min_s + f_low(f_shr(wp, be), bs));
                }
#pragma unroll
                for (int i = 0; i < FI; i++) {
                    if ((u32)i < nv && ((hm >> i) & 1u)) cs_h = sof(i);
                    s_x[wv][ln * (FI + 1) + i] = (u64)cs_h;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < FI; q++) { const u32 e = (u32)q * IVX_WAVE + ln; if (w0 + e < n) cl.cstart[w0 + e] = (i64)s_x[wv][e + (e >> 3)]; }
        }
        return;
    }
    // ---- section 3, emit: a head closes the run before it and opens its own
    if (fast) {
        u32 x = x32;
        if (__ballot(hm != 0)) {
#pragma unroll
            for (int i = 0; i < FI; i++) {
                const u32 lin = (u32)(r[i] >> be), eo = (u32)r[i] & emask;
                if ((hm >> i) & 1u) {
                    const u64 idx = row0 + i;
                    if (prev) {
                        if (out.end) out.end[rid - 1] = (i64)((u64)p.min_e + x);
                        if (out.count) out.count[rid - 1] = (i64)(idx - (prev - 1));
                    }
                    if (out.key) out.key[rid] = kA;
                    if (out.start) out.start[rid] = (i64)((u64)koffA + (lin - baseA));
                    prev = (u32)idx + 1;
                    rid++;
                }
                x = ((i == 0 && !hv) || eo > x) ? eo : x;
            }
        } else if (row0 + FI == n) {
#pragma unroll
            for (int i = 0; i < FI; i++) { const u32 eo = (u32)r[i] & emask; x = eo > x ? eo : x; }      // (no head here: hv holds)
        }
        if (row0 + FI == n) {                                            // the thread that holds the last row
            if (out.end) out.end[rid - 1] = (i64)((u64)p.min_e + x);
            if (out.count) out.count[rid - 1] = (i64)(n - (prev - 1));
            *d_m = rid;
        }
    } else {
        u64 x = X;
#pragma unroll
        for (int i = 0; i < FI; i++) {
            if ((u32)i < nv) {
                if ((hm >> i) & 1u) {
                    const u64 idx = row0 + i;
                    if (prev) {
                        if (out.end) out.end[rid - 1] = (i64)((u64)p.min_e + f_low(x, be));
                        if (out.count) out.count[rid - 1] = (i64)(idx - (prev - 1));
                    }
                    if (out.key) out.key[rid] = k[i];
                    if (out.start) out.start[rid] = sof(i);
                    prev = (u32)idx + 1;
                    rid++;
                }
                const u64 v = Vof(i);
                x = v > x ? v : x;
            }
        }
        if (nv && row0 + nv == n) {                                      // the thread that holds the last row
            if (out.end) out.end[rid - 1] = (i64)((u64)p.min_e + f_low(x, be));
            if (out.count) out.count[rid - 1] = (i64)(n - (prev - 1));
            *d_m = rid;
        }
    }
}

template <bool LIN>
ivx_status runs_packed(ivx_ctx *ctx, const u64 *w, u64 n, const Pack64 &p, i64 min_dist, int strict, const ivx_runs_out &out, const PkCluster *cl)
{
    hipStream_t stq = ctx->stream;
    const u64 ntile = (n + FTILE - 1) / FTILE;
    u64 *agg;                                                           // [0, ntile): V maxima, [ntile, 2 ntile): head summaries
    IVX_TRY(ctx->get_scratch(WS_T5, 2 * ntile * sizeof(u64), (void **)&agg));
    u64 *d_m = ctx->d_scalars + 2;
    const size_t tab = LIN ? ((size_t)p.nkeys * 2 + 1) * sizeof(u64) : 0;
    u32 bk = 0;
    for (u64 x = p.nkeys ? p.nkeys - 1 : 0; x; x >>= 1) bk++;
    const bool narrow = p.small && p.bits_e <= 32 && (LIN ? p.bits_s <= 32 : p.bits_s + bk <= 32) &&
                        min_dist < (1ll << 61) && !getenv("IVX_NO_NARROW_RUNS");
    const PkCluster none{nullptr, nullptr, nullptr, nullptr, nullptr};
    const dim3 grid((u32)ntile), blk(FT);
    const u64 *pre1 = agg, *pre2 = agg + ntile;
    hipLaunchKernelGGL((k_pk_max<LIN>), grid, blk, LIN ? ((size_t)p.nkeys + 1) * sizeof(u64) : 0, stq, w, n, p, agg);
    IVX_TRY((ivxscan::exclusive<MaxPayOp>(ctx, agg, ntile)));
    if (narrow) hipLaunchKernelGGL((k_pk_runs<LIN, 0, true>), grid, blk, tab, stq, w, n, p, min_dist, strict, pre1, agg + ntile, (const u64 *)nullptr, out, none, d_m);
    else hipLaunchKernelGGL((k_pk_runs<LIN, 0, false>), grid, blk, tab, stq, w, n, p, min_dist, strict, pre1, agg + ntile, (const u64 *)nullptr, out, none, d_m);
    IVX_TRY((ivxscan::exclusive<HeadPayOp>(ctx, agg + ntile, ntile)));
    if (cl) {
        if (narrow) hipLaunchKernelGGL((k_pk_runs<LIN, 2, true>), grid, blk, tab, stq, w, n, p, min_dist, strict, pre1, (u64 *)nullptr, pre2, out, *cl, d_m);
        else hipLaunchKernelGGL((k_pk_runs<LIN, 2, false>), grid, blk, tab, stq, w, n, p, min_dist, strict, pre1, (u64 *)nullptr, pre2, out, *cl, d_m);
    } else {
        if (narrow) hipLaunchKernelGGL((k_pk_runs<LIN, 1, true>), grid, blk, tab, stq, w, n, p, min_dist, strict, pre1, (u64 *)nullptr, pre2, out, none, d_m);
        else hipLaunchKernelGGL((k_pk_runs<LIN, 1, false>), grid, blk, tab, stq, w, n, p, min_dist, strict, pre1, (u64 *)nullptr, pre2, out, none, d_m);
    }
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

// cluster(), last pass: the id of every row's run -- global, or counted from the key's base (ClusterIdCoordinator) -- and the
// run's end, from the run numbers k_pk_runs<2> left.  ks: the (unpacked) key column; rows of one (key, start) may sit in another
// order there than in the packed words, but they share their run.
__global__ __launch_bounds__(RT) void k_pk_cluster_fin(const u32 *__restrict__ rid, const u32 *__restrict__ ks, u64 n, u32 nkeys,
                                                       const u32 *__restrict__ kfirst, const i64 *__restrict__ key_base,
                                                       const i64 *__restrict__ run_end, i64 *cluster, i64 *cend)
{
    const u64 i = (u64)blockIdx.x * RT + threadIdx.x;
    if (i >= n) return;
    const u32 r = rid[i];
    if (cluster) {
        i64 c = (i64)r;
        if (key_base) { const u32 k = ks[i]; if (k < nkeys) c = key_base[k] + (i64)(r - kfirst[k]); }
        cluster[i] = c;
    }
    if (cend) cend[i] = run_end[r];
}

}  // namespace

bool ivx_merge_packed_ok(const Pack64 &p, u64 n, u32 nkeys, i64 min_dist, int strict, bool malformed, bool has_empty)
{
    u32 bk = 0;
    for (u64 x = nkeys; x; x >>= 1) bk++;                                // bits of key + 1
    if (malformed || min_dist < 0 || (strict && min_dist == 0 && has_empty)) return false;
    if (n >= 0x7FFFFFFFull || bk + p.bits_e > 62) return false;
    if (p.lin && ((size_t)nkeys * 2 + 1) * sizeof(u64) > 40 * 1024) return false;
    return true;
}

ivx_status ivx_merge_runs_packed(ivx_ctx *ctx, const u64 *w, u64 n, const Pack64 &p, i64 min_dist, int strict, const ivx_runs_out &out, u64 *m)
{
    *m = 0;
    if (n == 0) return IVX_OK;
    if (p.lin) IVX_TRY(runs_packed<true>(ctx, w, n, p, min_dist, strict, out, nullptr));
    else IVX_TRY(runs_packed<false>(ctx, w, n, p, min_dist, strict, out, nullptr));
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 2, ctx->d_scalars + 2, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *m = ctx->h_scalars[2];
    return IVX_OK;
}

ivx_status ivx_merge_runs(ivx_ctx *ctx, const u32 *ks, const i64 *ss, const i64 *es, u64 n,
                          i64 min_dist, int strict, const ivx_runs_out &out, u64 *m)
{
    SortedRows rows{}; rows.ks = ks; rows.ss = ss; rows.es = es;
    return ivx_merge_runs_rows(ctx, rows, n, min_dist, strict, out, m);
}

ivx_status ivx_merge_runs_rows(ivx_ctx *ctx, const SortedRows &rows, u64 n, i64 min_dist, int strict, const ivx_runs_out &out, u64 *m)
{
    *m = 0;
    if (n == 0) return IVX_OK;
    const HeadAcc *ha;
    IVX_TRY(sweep(ctx, rows, n, min_dist, strict, out, &ha, m, false));
    IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *m = ctx->h_scalars[2];
    return IVX_OK;
}

namespace {

// kfirst[k] = runs before key k's first run, klast[k] = runs up to and including its last one
__global__ __launch_bounds__(RT) void k_key_runs(const u32 *__restrict__ ks, const HeadAcc *__restrict__ ha, u64 n, u32 nkeys,
                                                 u32 *kfirst, u32 *klast)
{
    const u64 i = (u64)blockIdx.x * RT + threadIdx.x;
    if (i >= n) return;
    const u32 k = ks[i];
    if (k >= nkeys) return;                                   // reported by the sort's key check
    if (i == 0 || ks[i - 1] != k) kfirst[k] = ha[i].heads - 1;
    if (i + 1 == n || ks[i + 1] != k) klast[k] = ha[i].heads;
}

__global__ __launch_bounds__(RT) void k_key_clusters(const u32 *__restrict__ kfirst, const u32 *__restrict__ klast, u32 nkeys, u64 *out)
{
    const u32 k = blockIdx.x * RT + threadIdx.x;
    if (k < nkeys) out[k] = (u64)(klast[k] - kfirst[k]);
}

__global__ __launch_bounds__(RT) void k_cluster_rows(const u32 *__restrict__ ks, const i64 *__restrict__ ss, const HeadAcc *__restrict__ ha,
                                                     const i64 *__restrict__ run_end, u64 n, u32 nkeys,
                                                     const u32 *__restrict__ kfirst, const i64 *__restrict__ key_base, ivx_cluster_out out)
{
    const u64 i = (u64)blockIdx.x * RT + threadIdx.x;
    if (i >= n) return;
    const HeadAcc h = ha[i];
    const u32 rid = h.heads - 1;
    if (out.cluster) {
        const u32 k = ks[i];
        out.cluster[i] = (key_base && k < nkeys) ? key_base[k] + (i64)(rid - kfirst[k]) : (i64)rid;
    }
    if (out.start) out.start[i] = ss[h.last_head];
    if (out.end) out.end[i] = run_end[rid];
}

}  // namespace

// cluster() over the packed words (see ivx_runs.hpp).  Scratch: WS_T5 (tile summaries), WS_T6 (run numbers), WS_T7 (run ends),
// WS_T8 / WS_T9 (per-key run ranges).
ivx_status ivx_cluster_rows_packed(ivx_ctx *ctx, const u64 *w, const Pack64 &p, const u32 *ks, u64 n, u32 nkeys,
                                   i64 min_dist, int strict, const i64 *key_base, const ivx_cluster_out &out, u64 *m)
{
    *m = 0;
    hipStream_t stq = ctx->stream;
    u32 *kfirst, *klast;
    IVX_TRY(ctx->get_scratch(WS_T8, (size_t)nkeys * sizeof(u32), (void **)&kfirst));
    IVX_TRY(ctx->get_scratch(WS_T9, (size_t)nkeys * sizeof(u32), (void **)&klast));
    IVX_HIP(ctx, hipMemsetAsync(kfirst, 0, (size_t)nkeys * sizeof(u32), stq));
    IVX_HIP(ctx, hipMemsetAsync(klast, 0, (size_t)nkeys * sizeof(u32), stq));
    if (n) {
        const bool rows_out = out.cluster || out.start || out.end;
        u32 *rid = nullptr; i64 *run_end = nullptr;
        if (out.cluster || out.end) IVX_TRY(ctx->get_scratch(WS_T6, n * sizeof(u32), (void **)&rid));
        if (out.end) IVX_TRY(ctx->get_scratch(WS_T7, n * sizeof(i64), (void **)&run_end));
        const PkCluster cl{rid, out.start, run_end, kfirst, klast};
        const ivx_runs_out none{nullptr, nullptr, nullptr, nullptr};
        if (p.lin) IVX_TRY(runs_packed<true>(ctx, w, n, p, min_dist, strict, none, &cl));
        else IVX_TRY(runs_packed<false>(ctx, w, n, p, min_dist, strict, none, &cl));
        if (rows_out && (out.cluster || out.end))
            hipLaunchKernelGGL(k_pk_cluster_fin, dim3((u32)((n + RT - 1) / RT)), dim3(RT), 0, stq, (const u32 *)rid, ks, n, nkeys, (const u32 *)kfirst, key_base,
                               (const i64 *)run_end, out.cluster, out.end);
    }
    if (out.key_clusters)
        hipLaunchKernelGGL(k_key_clusters, dim3((nkeys + RT - 1) / RT), dim3(RT), 0, stq, (const u32 *)kfirst, (const u32 *)klast, nkeys, out.key_clusters);
    IVX_HIP(ctx, hipGetLastError());
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 2, ctx->d_scalars + 2, sizeof(u64), hipMemcpyDeviceToHost, stq));
    IVX_HIP(ctx, hipStreamSynchronize(stq));
    *m = n ? ctx->h_scalars[2] : 0;
    return IVX_OK;
}

ivx_status ivx_cluster_rows(ivx_ctx *ctx, const u32 *ks, const i64 *ss, const i64 *es, u64 n, u32 nkeys,
                            i64 min_dist, int strict, const i64 *key_base, const ivx_cluster_out &out, u64 *m)
{
    *m = 0;
    hipStream_t stq = ctx->stream;
    u32 *kfirst, *klast;
    IVX_TRY(ctx->get_scratch(WS_T8, (size_t)nkeys * sizeof(u32), (void **)&kfirst));
    IVX_TRY(ctx->get_scratch(WS_T9, (size_t)nkeys * sizeof(u32), (void **)&klast));
    IVX_HIP(ctx, hipMemsetAsync(kfirst, 0, (size_t)nkeys * sizeof(u32), stq));
    IVX_HIP(ctx, hipMemsetAsync(klast, 0, (size_t)nkeys * sizeof(u32), stq));
    if (n) {
        i64 *run_end;
        IVX_TRY(ctx->get_scratch(WS_T7, n * sizeof(i64), (void **)&run_end));
        const HeadAcc *ha;
        const ivx_runs_out ro{nullptr, nullptr, run_end, nullptr};
        SortedRows rows{}; rows.ks = ks; rows.ss = ss; rows.es = es;
        IVX_TRY(sweep(ctx, rows, n, min_dist, strict, ro, &ha, m, true));
        const u32 grid = (u32)((n + RT - 1) / RT);
        hipLaunchKernelGGL(k_key_runs, dim3(grid), dim3(RT), 0, stq, ks, ha, n, nkeys, kfirst, klast);
        hipLaunchKernelGGL(k_cluster_rows, dim3(grid), dim3(RT), 0, stq, ks, ss, ha, (const i64 *)run_end, n, nkeys,
                           (const u32 *)kfirst, key_base, out);
    }
    if (out.key_clusters)
        hipLaunchKernelGGL(k_key_clusters, dim3((nkeys + RT - 1) / RT), dim3(RT), 0, stq, (const u32 *)kfirst, (const u32 *)klast, nkeys, out.key_clusters);
    IVX_HIP(ctx, hipGetLastError());
    IVX_HIP(ctx, hipStreamSynchronize(stq));
    *m = n ? ctx->h_scalars[2] : 0;
    return IVX_OK;
}
