// ivx_sweep.hip -- merge() and subtract() on int64 coordinates.
//
//   a7 StreamCollector / FullBatchCollector: per-contig sort_unstable of (start,end[,row])
//      (grouped_stream.rs:50-113, :163-237)      -> one device LSD radix sort by (key,start,end,row)
//   a8 MergeStream sweep (merge.rs:282-350)       -> two scans (ivx_runs.hpp)
//   a9 SubtractStream sweep (subtract.rs:390-462, :575-655) -> per-left-row binary searches over
//      the right side's "gap heads" + count/scan/fill.
//   f1 cluster (cluster.rs:443-477, :598-661)     -> the merge scans + one per-row pass (ivx_runs.hpp)
//   f2 complement (complement.rs:297-357, :394-465) -> merge, then per view interval two binary
//      searches over the merged runs + count/scan/fill by output row (see the complement section).
//
// subtract without the serial cursor.  For one left row [ls,le) the reference
// walks the rights of the contig in (start,end) order with cursor = ls:
//   a right j with rs_j > cursor emits [cursor, rs_j); cursor = max(cursor, re_j);
//   it stops at the first rs_j > le (strict: >= le); finally [cursor, le) if cursor < le.
// Rights skipped by the monotone right_cursor all end before ls, so they never
// change max(ls, .): cursor before j is max(ls, PM[j-1]) with PM the running max
// of right ends in the contig.  Hence j emits iff rs_j > PM[j-1] (a "gap head", a
// property of the right side alone), ls < rs_j <= le and j is not behind the
// right_cursor.  Because left starts ascend, the cursor for a left row is simply
// the first right with re_j >= ls (strict: > ls) = the first j with PM[j] >= ls,
// whatever came before.  The gap heads form a sorted array and each left row owns
// a contiguous slice of it.  (For well-formed rights the cursor bound is implied
// by rs_j > ls; it matters only when a right row has end < start.)
#include <cstdlib>
#include "ivx_runs.hpp"
#include "ivx_scan.hpp"
#include "ivx_sort.hpp"

namespace {

constexpr int ST = 256;
constexpr u64 SIGN64 = 0x8000000000000000ull;

u32 grid1(u64 n) { return (u32)((n + ST - 1) / ST); }

__global__ __launch_bounds__(ST) void k_pack64(const u32 *__restrict__ key, const i64 *__restrict__ s, const i64 *__restrict__ e,
                                               u64 n, u32 nkeys, u64 *w0, u64 *w1, u64 *w2, u32 *flags)
{
    const u64 i = (u64)blockIdx.x * ST + threadIdx.x;
    if (i >= n) return;
    const u32 k = key ? key[i] : 0u;
    if (k >= nkeys) flags[0] = 1;
    w0[i] = (u64)e[i] ^ SIGN64;
    w1[i] = (u64)s[i] ^ SIGN64;
    w2[i] = ((u64)k << 32) | (u32)i;
}

__global__ __launch_bounds__(ST) void k_unpack64(const u64 *__restrict__ w0, const u64 *__restrict__ w1, const u64 *__restrict__ w2,
                                                 u64 n, u32 *ks, i64 *ss, i64 *es, u32 *rows)
{
    const u64 i = (u64)blockIdx.x * ST + threadIdx.x;
    if (i >= n) return;
    es[i] = (i64)(w0[i] ^ SIGN64);
    ss[i] = (i64)(w1[i] ^ SIGN64);
    ks[i] = (u32)(w2[i] >> 32);
    if (rows) rows[i] = (u32)w2[i];
}

// ---- packed variant: when (key, start - min start, end - min end) fit 64 bits together -- genomic
// coordinates always do -- the sort key is ONE word and the record 16 bytes instead of 24, with fewer
// radix digits in total
struct Range64 { long long min_s, max_s, min_e, max_e; unsigned long long unsorted, odd; };   // unsorted: some row sorts before its predecessor;
                                                                                                // odd: bit 0 some end < start, bit 1 some end == start,
                                                                                                // bit 2 some row's (key, start) below its predecessor's

// kmin / kmax (nullable; nkeys <= LIN_KEYS): also every key's own range of starts, for the linearised sort word below --
// privatised in LDS, and a bound is touched by an atomic only when a row moves it (a plain read comes first)
constexpr u32 LIN_KEYS = 2048;
__global__ void k_init_keyrange64(i64 *kmin, i64 *kmax, u32 nkeys)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nkeys) { kmin[i] = INT64_MAX; kmax[i] = INT64_MIN; }
}
__global__ __launch_bounds__(ST) void k_range64(const u32 *__restrict__ key, const i64 *__restrict__ s, const i64 *__restrict__ e,
                                                u64 n, u32 nkeys, Range64 *out, u32 *flags, long long *kmin, long long *kmax)
{
    __shared__ i64 red[4][ST / IVX_WAVE];
    extern __shared__ long long s_kr[];                                 // [2 * nkeys] when kmin
    long long *smin = s_kr, *smax = s_kr + nkeys;
    if (kmin) {
        for (u32 k = threadIdx.x; k < nkeys; k += ST) { smin[k] = INT64_MAX; smax[k] = INT64_MIN; }
        __syncthreads();
    }
    i64 lo_s = INT64_MAX, hi_s = INT64_MIN, lo_e = INT64_MAX, hi_e = INT64_MIN;
    bool bad = false, inv = false, inv2 = false, mal = false, emp = false;
    constexpr int U = 4;                                                // rows per thread in flight (their loads depend on nothing)
    const u32 ln = lane_id();
    for (u64 i0 = (u64)blockIdx.x * (ST * U) + threadIdx.x; i0 - threadIdx.x < n; i0 += (u64)gridDim.x * (ST * U)) {
        i64 a[U], b[U]; u32 k[U];
        // the row before lane 0's (the other lanes take their neighbour's registers: half the loads of reading row i - 1 too)
        i64 pa0[U], pb0[U]; u32 pk0[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 i = i0 + (u64)u * ST;
            const bool in = i < n;
            a[u] = in ? s[i] : 0; b[u] = in ? e[i] : 0; k[u] = in ? (key ? key[i] : 0u) : 0u;
            pa0[u] = INT64_MIN; pb0[u] = INT64_MIN; pk0[u] = 0u;
            if (ln == 0 && in && i > 0) { pa0[u] = s[i - 1]; pb0[u] = e[i - 1]; pk0[u] = key ? key[i - 1] : 0u; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            // (every lane takes part in the lane shifts, also the ones past the end)
            u32 pk = wave_prev32(k[u]);
            i64 pa = (i64)(((u64)wave_prev32((u32)((u64)a[u] >> 32)) << 32) | wave_prev32((u32)(u64)a[u]));
            i64 pb = (i64)(((u64)wave_prev32((u32)((u64)b[u] >> 32)) << 32) | wave_prev32((u32)(u64)b[u]));
            if (ln == 0) { pk = pk0[u]; pa = pa0[u]; pb = pb0[u]; }
            if (i0 + (u64)u * ST >= n) continue;
            lo_s = a[u] < lo_s ? a[u] : lo_s; hi_s = a[u] > hi_s ? a[u] : hi_s;
            lo_e = b[u] < lo_e ? b[u] : lo_e; hi_e = b[u] > hi_e ? b[u] : hi_e;
            bad |= k[u] >= nkeys;
            mal |= b[u] < a[u]; emp |= b[u] == a[u];
            if (kmin && k[u] < nkeys) {
                if (a[u] < *(volatile long long *)&smin[k[u]]) atomicMin(&smin[k[u]], (long long)a[u]);
                if (a[u] > *(volatile long long *)&smax[k[u]]) atomicMax(&smax[k[u]], (long long)a[u]);
            }
            // (key,start,end) below the row before it?  (row 0 has none: pa = pb = INT64_MIN, pk = 0 compare as "not below")
            inv |= k[u] != pk ? k[u] < pk : (a[u] != pa ? a[u] < pa : b[u] < pb);
            inv2 |= k[u] != pk ? k[u] < pk : a[u] < pa;
        }
    }
#pragma unroll
    for (int d = IVX_WAVE / 2; d > 0; d >>= 1) {
        i64 t;
        t = __shfl_xor(lo_s, d, IVX_WAVE); lo_s = t < lo_s ? t : lo_s;
        t = __shfl_xor(hi_s, d, IVX_WAVE); hi_s = t > hi_s ? t : hi_s;
        t = __shfl_xor(lo_e, d, IVX_WAVE); lo_e = t < lo_e ? t : lo_e;
        t = __shfl_xor(hi_e, d, IVX_WAVE); hi_e = t > hi_e ? t : hi_e;
    }
    const u32 wv = threadIdx.x / IVX_WAVE;
    if (lane_id() == 0) { red[0][wv] = lo_s; red[1][wv] = hi_s; red[2][wv] = lo_e; red[3][wv] = hi_e; }
    if (bad) flags[0] = 1;
    if (inv) out->unsorted = 1;
    {
        const unsigned long long o = (__ballot(mal) ? 1ull : 0ull) | (__ballot(emp) ? 2ull : 0ull) | (__ballot(inv2) ? 4ull : 0ull);
        if (o && lane_id() == 0) atomicOr(&out->odd, o);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < ST / IVX_WAVE; w++) {
            lo_s = red[0][w] < lo_s ? red[0][w] : lo_s; hi_s = red[1][w] > hi_s ? red[1][w] : hi_s;
            lo_e = red[2][w] < lo_e ? red[2][w] : lo_e; hi_e = red[3][w] > hi_e ? red[3][w] : hi_e;
        }
        atomicMin(&out->min_s, (long long)lo_s); atomicMax(&out->max_s, (long long)hi_s);
        atomicMin(&out->min_e, (long long)lo_e); atomicMax(&out->max_e, (long long)hi_e);
    }
    if (kmin)
        for (u32 k = threadIdx.x; k < nkeys; k += ST)
            if (smin[k] <= smax[k]) { atomicMin(&kmin[k], smin[k]); atomicMax(&kmax[k], smax[k]); }
}

// one workgroup: base[k] = number of (key, start) positions before key k when every key spans just its own starts
// (base[nkeys] = all of them; hdr[0] = 0 when the count leaves 64 bits)
__global__ __launch_bounds__(1024) void k_lin_layout64(const long long *kmin, const long long *kmax, u32 nkeys, u64 *base, u64 *hdr)
{
    __shared__ u64 red[1024 / IVX_WAVE + 1];
    __shared__ u32 s_bad;
    const u32 t = threadIdx.x;
    if (t == 0) s_bad = 0;
    __syncthreads();
    u64 run = 0;
    for (u32 k0 = 0; k0 < nkeys; k0 += 1024) {
        const u32 k = k0 + t;
        u64 w = 0;
        if (k < nkeys && kmin[k] <= kmax[k]) {
            w = (u64)kmax[k] - (u64)kmin[k] + 1;
            if (w == 0 || w > (1ull << 62)) { s_bad = 1; w = 0; }
        }
        u64 tot;
        const u64 ex = block_excl_scan<u64, 1024>(w, red, &tot);
        if (k < nkeys) base[k] = run + ex;
        if (run + tot < run || run + tot > (1ull << 62)) s_bad = 1;
        run += tot;
    }
    __syncthreads();
    if (t == 0) { base[nkeys] = run; hdr[0] = s_bad ? 0 : 1; hdr[1] = run; }
}

// lin: the word's upper part is base[key] + (start - kmin[key]) -- bits_s bits, no separate key bits -- instead of
// key ‖ (start - min_s): a human genome's (contig, position) pairs number 3.1e9 = 32 bits = four radix digits, where
// 5 key bits + 28 position bits take five
// (Pack64: ivx_runs.hpp)
__device__ __forceinline__ u32 lin_key64(const u64 *base, u32 nkeys, u64 lin)
{
    u32 a = 0, b = nkeys;                                              // first k with base[k + 1] > lin (keys without rows are skipped)
    while (a < b) { const u32 m = (a + b) >> 1; if (base[m + 1] > lin) b = m; else a = m + 1; }
    return a;
}
__device__ __forceinline__ u64 shl64(u64 x, u32 sh) { return sh >= 64 ? 0 : x << sh; }
__device__ __forceinline__ u64 shr64(u64 x, u32 sh) { return sh >= 64 ? 0 : x >> sh; }
__device__ __forceinline__ u64 low64(u64 x, u32 bits) { return bits >= 64 ? x : x & ((1ull << bits) - 1); }

__global__ __launch_bounds__(ST) void k_pack1(const u32 *__restrict__ key, const i64 *__restrict__ s, const i64 *__restrict__ e,
                                              u64 n, Pack64 p, u64 *w0, u32 *w1)
{
    const u64 i = (u64)blockIdx.x * ST + threadIdx.x;
    if (i >= n) return;
    const u64 k = key ? key[i] : 0u;
    if (p.lin) w0[i] = shl64(p.base[k] + ((u64)s[i] - (u64)p.kmin[k]), p.bits_e) | ((u64)e[i] - (u64)p.min_e);
    else w0[i] = shl64(k, p.bits_s + p.bits_e) | shl64((u64)s[i] - (u64)p.min_s, p.bits_e) | ((u64)e[i] - (u64)p.min_e);
    if (w1) w1[i] = (u32)i;                                             // the row id rides along as a 32-bit payload (12-byte records); callers
}                                                                       // that do not ask for row ids sort the 8-byte words alone

// k_pack1 for a sort that follows at once: one workgroup per sort workgroup (ivx_sort_geometry1), which also counts the digits
// of the sort's first pass -- bits [shift, shift + 8) -- into hist[digit * nblk + block] as k_hist would: the words are not read
// a second time for that (1.6 GB of a 200 M-row sort)
__global__ __launch_bounds__(ST) void k_pack1h(const u32 *__restrict__ key, const i64 *__restrict__ s, const i64 *__restrict__ e,
                                               u64 n, Pack64 p, u64 *w0, u32 *w1, int shift, u32 nblk, u32 *__restrict__ hist, u32 chunk)
{
    __shared__ u32 cnt[ST / IVX_WAVE][256];
    for (int i = threadIdx.x; i < (ST / IVX_WAVE) * 256; i += ST) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const u64 lo = (u64)blockIdx.x * chunk;
    const u64 hi = lo + chunk < n ? lo + chunk : n;
    const u32 wv = threadIdx.x / IVX_WAVE;
    for (u64 i0 = lo; i0 < hi; i0 += ST * 4) {
        u64 x[4]; bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u64 i = i0 + (u64)u * ST + threadIdx.x;
            valid[u] = i < hi; x[u] = 0;
            if (valid[u]) {
                const u64 k = key ? key[i] : 0u;
                if (p.lin) x[u] = shl64(p.base[k] + ((u64)s[i] - (u64)p.kmin[k]), p.bits_e) | ((u64)e[i] - (u64)p.min_e);
                else x[u] = shl64(k, p.bits_s + p.bits_e) | shl64((u64)s[i] - (u64)p.min_s, p.bits_e) | ((u64)e[i] - (u64)p.min_e);
                w0[i] = x[u];
                if (w1) w1[i] = (u32)i;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {                                   // (as k_hist, ivx_sort.hip)
            const u32 d = (u32)((x[u] >> shift) & 0xFF);
            const u64 act = __ballot(valid[u]);
            if (act == 0) continue;
            const u32 first = (u32)__builtin_ctzll(act);
            const u32 d0 = __shfl(d, first, IVX_WAVE);
            if (__ballot(valid[u] && d == d0) == act) { if (lane_id() == first) cnt[wv][d0] += (u32)__popcll(act); }
            else if (valid[u]) atomicAdd(&cnt[wv][d], 1u);
        }
    }
    __syncthreads();
    {
        const u32 d = threadIdx.x;
        u32 t = 0;
#pragma unroll
        for (int k = 0; k < ST / IVX_WAVE; k++) t += cnt[k][d];
        hist[(u64)d * nblk + blockIdx.x] = t;
    }
}

// FIX: the words were sorted (stably) on their bits above lo_bits only -- (key, start) -- so rows of equal (key, start)
// sit together in input order and still have to be ordered by their low bits (end), then row.  Such runs are short
// for genomic data (two or three rows): every row finds its own place inside its run by counting the run's rows that
// sort before it, and is written there -- no pass over the words to repair them first.  A run beyond FIX_MAXRUN rows
// raises *toolong and the host falls back to the full-width sort.
constexpr u32 FIX_MAXRUN = 64;
constexpr u32 UNPACK_TILES = 8;
// NARROW: start and end leave as 32-bit offsets from p.min_s / p.min_e (the columns' storage is used as u32 arrays): 12
// instead of 20 bytes per row here and in every pass of the merge sweep behind it (SortedRows, ivx_runs.hpp)
// ... and the key as one byte when there are at most 256 keys (K8)
template <bool FIX, bool NARROW = false, bool K8 = false>
__global__ __launch_bounds__(ST) void k_unpack1(const u64 *__restrict__ w0, const u32 *__restrict__ w1, u64 n, Pack64 p,
                                                u32 *ks, i64 *ss, i64 *es, u32 *rows, u32 lo_bits, u32 *toolong)
{
    __shared__ u64 s_base[LIN_KEYS + 1];
    // the tile's words + FIX_MAXRUN on either side: a row looks at its neighbours in LDS (a chain of dependent global
    // loads per row in a run held every wavefront up)
    __shared__ u64 s_w[FIX ? ST + 2 * FIX_MAXRUN : 1];
    if (p.lin) for (u32 k = threadIdx.x; k <= p.nkeys; k += ST) s_base[k] = p.base[k];
    // slot x of s_w holds row t0 - FIX_MAXRUN + x; a thread fetches slots x = tid and (the first 2 * FIX_MAXRUN threads)
    // x = ST + tid -- for the NEXT tile while the current one is worked on (registers), so that a tile does not start with
    // a memory round trip
    static_assert(2 * FIX_MAXRUN <= ST, "halo slots are fetched by the first threads");
    u64 na = 0, nb = 0;
    auto fetch = [&](u64 t0) {
        if (!FIX || t0 >= n) return;
        const u64 g = t0 + threadIdx.x;
        na = (g >= FIX_MAXRUN && g - FIX_MAXRUN < n) ? w0[g - FIX_MAXRUN] : 0;
        if (threadIdx.x < 2 * FIX_MAXRUN) { const u64 g2 = g + ST; nb = (g2 - FIX_MAXRUN < n) ? w0[g2 - FIX_MAXRUN] : 0; }
    };
    fetch((u64)blockIdx.x * UNPACK_TILES * ST);
    // (UNPACK_TILES tiles per workgroup: the key table is staged once for all of them)
    for (u32 tile = 0; tile < UNPACK_TILES; tile++) {
        const u64 t0 = ((u64)blockIdx.x * UNPACK_TILES + tile) * ST;
        if (t0 >= n) break;                                             // (whole workgroup)
        const u64 i = t0 + threadIdx.x;
        u64 w = 0;
        if (FIX) {
            __syncthreads();                                            // (the previous tile's readers are done; first tile: s_base is written)
            s_w[FIX ? threadIdx.x : 0] = na;
            if (threadIdx.x < 2 * FIX_MAXRUN) s_w[FIX ? ST + threadIdx.x : 0] = nb;
            __syncthreads();
            if (tile + 1 < UNPACK_TILES) fetch(t0 + ST);
            w = s_w[FIX ? threadIdx.x + FIX_MAXRUN : 0];
        } else {
            if (tile == 0) __syncthreads();
            if (i < n) w = w0[i];
        }
        if (i >= n) continue;
        u64 pos = i;
        if (FIX) {
            const u64 hd = shr64(w, lo_bits);
            const u32 me = threadIdx.x + FIX_MAXRUN;                    // my slot; slot x is a row iff t0 + x - FIX_MAXRUN in [0, n)
            const u32 xlo = t0 >= FIX_MAXRUN ? 0u : (u32)(FIX_MAXRUN - t0);                               // first slot that is a row
            const u32 xhi = (u32)((n - t0 < (u64)(ST + FIX_MAXRUN) ? n - t0 : (u64)(ST + FIX_MAXRUN)) + FIX_MAXRUN);   // one past the last
            u32 h = me, t = me + 1;
            while (h > xlo && me - h < FIX_MAXRUN && shr64(s_w[FIX ? h - 1 : 0], lo_bits) == hd) h--;
            while (t < xhi && t - me < FIX_MAXRUN && shr64(s_w[FIX ? t : 0], lo_bits) == hd) t++;
            if (t - h > 1) {
                if (t - h > FIX_MAXRUN) { *toolong = 1; continue; }
                u32 below = 0;
                for (u32 j = h; j < t; j++) { const u64 x = s_w[FIX ? j : 0]; below += (x < w || (x == w && j < me)) ? 1u : 0u; }
                pos = t0 + h + below - FIX_MAXRUN;
            }
        }
        u64 so;                                                         // start - p.min_s
        if (p.lin) {
            const u64 lin = shr64(w, p.bits_e);
            const u32 k = lin_key64(s_base, p.nkeys, lin);
            if (K8) ((u8 *)ks)[pos] = (u8)k; else ks[pos] = k;
            so = (u64)p.kmin[k] + (lin - s_base[k]) - (u64)p.min_s;
        } else {
            const u32 k = (u32)shr64(w, p.bits_s + p.bits_e);
            if (K8) ((u8 *)ks)[pos] = (u8)k; else ks[pos] = k;
            so = low64(shr64(w, p.bits_e), p.bits_s);
        }
        if (NARROW) { ((u32 *)ss)[pos] = (u32)so; ((u32 *)es)[pos] = (u32)low64(w, p.bits_e); }
        else { ss[pos] = (i64)(so + (u64)p.min_s); es[pos] = (i64)(low64(w, p.bits_e) + (u64)p.min_e); }
        if (rows && w1) rows[pos] = w1[i];
    }
}

// rows that already are in (key,start,end) order: the sorted columns are the input columns, row ids 0..n-1
__global__ __launch_bounds__(ST) void k_copy_sorted(const u32 *__restrict__ key, const i64 *__restrict__ s, const i64 *__restrict__ e, u64 n,
                                                    u32 *ks, i64 *ss, i64 *es, u32 *rows)
{
    const u64 i = (u64)blockIdx.x * ST + threadIdx.x;
    if (i >= n) return;
    ks[i] = key ? key[i] : 0u; ss[i] = s[i]; es[i] = e[i];
    if (rows) rows[i] = (u32)i;
}

u32 bits_of(u64 x) { u32 b = 0; while (x) { b++; x >>= 1; } return b; }

// sort (key,start,end,row) ascending; rows of equal (key,start,end) keep input order = ascending row
// sw (nullable; needs rows == nullptr): the caller reads the sorted rows through a SortedRows -- narrow columns (32-bit
// offsets, stored in ss / es as u32 arrays) whenever the one-word form applies and both ranges fit 32 bits, else the wide ones.
// pk (nullable): the caller is a merge sweep with pk->d / pk->strict and takes the PACKED words, sorted on their (key, start)
// bits at least, whenever ivx_merge_packed_ok says the sweep over packed words applies: pk->ok.  With sw (merge, complement)
// nothing is unpacked then; without (cluster) the rows are unpacked as always and pk->w stays valid beside them.
struct PackedWant { i64 d; int strict; bool ok; const u64 *w; Pack64 p; };
ivx_status sort64(ivx_ctx *ctx, int slot_a, int slot_b, const u32 *key, const i64 *s, const i64 *e, u64 n, u32 nkeys,
                  u32 *ks, i64 *ss, i64 *es, u32 *rows, SortedRows *sw = nullptr, PackedWant *pk = nullptr)
{
    if (sw) { *sw = SortedRows{}; sw->ks = ks; sw->ss = ss; sw->es = es; }
    if (pk) pk->ok = false;
    if (n == 0) return IVX_OK;
    bool narrow = sw != nullptr && rows == nullptr && !getenv("IVX_NO_NARROW_SWEEP");
    hipStream_t st = ctx->stream;
    u32 *flags = (u32 *)(ctx->d_scalars + 8);
    Range64 *d_rng = (Range64 *)(ctx->d_scalars + 24);
    Range64 *h_init = (Range64 *)(ctx->h_scalars + 48);                   // pinned, so the async copy may read it later
    *h_init = Range64{INT64_MAX, INT64_MIN, INT64_MAX, INT64_MIN, 0ull, 0ull};
    IVX_HIP(ctx, hipMemcpyAsync(d_rng, h_init, sizeof(Range64), hipMemcpyHostToDevice, st));
    // every key's own range of starts as well, when the key table fits LDS (the linearised sort word, Pack64)
    long long *kmin = nullptr, *kmax = nullptr; u64 *base = nullptr;
    u64 *d_lin = ctx->d_scalars + 30;                                   // {usable, (key, start) positions in all}
    const bool try_lin = nkeys <= LIN_KEYS && !getenv("IVX_NO_LIN");
    if (try_lin) {
        // (the third record slots: free whenever the one-word form is in use, and owned by this call -- the callers keep
        //  their own tables in the WS_GRID / WS_T slots across it)
        IVX_TRY(ctx->get_scratch(slot_a + 2, (size_t)nkeys * 2 * sizeof(long long), (void **)&kmin));
        kmax = kmin + nkeys;
        IVX_TRY(ctx->get_scratch(slot_b + 2, ((size_t)nkeys + 1) * sizeof(u64), (void **)&base));
        hipLaunchKernelGGL(k_init_keyrange64, dim3((nkeys + ST - 1) / ST), dim3(ST), 0, st, (i64 *)kmin, (i64 *)kmax, nkeys);
    }
    hipLaunchKernelGGL(k_range64, dim3(ivx_stream_grid(n, ST * 4, 2048)), dim3(ST), try_lin ? (size_t)nkeys * 16 : 0, st, key, s, e, n, nkeys, d_rng, flags,
                       kmin, kmax);
    if (try_lin) hipLaunchKernelGGL(k_lin_layout64, dim3(1), dim3(1024), 0, st, (const long long *)kmin, (const long long *)kmax, nkeys, base, d_lin);
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 24, d_rng, 8 * sizeof(u64), hipMemcpyDeviceToHost, st));   // Range64 (6 words), d_lin (2 words)
    IVX_HIP(ctx, hipStreamSynchronize(st));
    const Range64 r = *(const Range64 *)(ctx->h_scalars + 24);
    // coordinate-sorted input (the usual state of BED / VCF / BAM-derived tables): nothing to sort, and equal
    // rows already are in ascending row order
    // (the sweep over packed words asks less: (key, start) order -- coordinate-sorted files rarely order the ends of equal starts)
    bool sorted_in = !r.unsorted && !getenv("IVX_FORCE_SORT");
    Pack64 p;
    p.min_s = r.min_s; p.min_e = r.min_e; p.base = base; p.kmin = kmin; p.lin = 0; p.nkeys = nkeys; p.pad = 0;
    {
        const long long lim = 1ll << 61;
        p.small = r.min_s > -lim && r.max_s < lim && r.min_e > -lim && r.max_e < lim;
    }
    p.bits_s = bits_of((u64)r.max_s - (u64)r.min_s); p.bits_e = bits_of((u64)r.max_e - (u64)r.min_e);
    u32 bits_k = bits_of(nkeys ? nkeys - 1 : 0);
    double positions = (double)(nkeys ? nkeys : 1) * (p.bits_s >= 62 ? 4.6e18 : (double)(1ull << p.bits_s));
    if (try_lin && ctx->h_scalars[30] && ctx->h_scalars[31]) {
        const u32 bits_lin = bits_of(ctx->h_scalars[31] - 1);
        if (bits_lin < bits_k + p.bits_s) { p.lin = 1; p.bits_s = bits_lin; bits_k = 0; positions = (double)ctx->h_scalars[31]; }
    }
    const u32 total = p.bits_s + p.bits_e + bits_k;
    u64 *a[3] = {nullptr, nullptr, nullptr}, *b[3] = {nullptr, nullptr, nullptr};
    // packed: one 64-bit sort word, plus the row ids -- a 32-bit payload -- only when the caller wants them back (merge /
    // complement / the right side of subtract do not: equal words are equal rows, and the record is 8 bytes instead of 12)
    const int nw = total <= 64 ? 1 : 3;
    // (the packed sweep takes sorted input as well: one pack pass instead of a 20-byte copy and three passes over wide rows)
    const bool packed = nw == 1 && pk && sw && !rows && !getenv("IVX_NO_FUSED_SWEEP") &&
                        ivx_merge_packed_ok(p, n, nkeys, pk->d, pk->strict, (r.odd & 1) != 0, (r.odd & 2) != 0);
    // (a caller that needs the unpacked rows AND sweeps over the packed words -- cluster -- gets both: pk->w stays valid)
    const bool also = nw == 1 && pk && !sw && !getenv("IVX_NO_FUSED_SWEEP") &&
                      ivx_merge_packed_ok(p, n, nkeys, pk->d, pk->strict, (r.odd & 1) != 0, (r.odd & 2) != 0);
    if (packed && (r.odd & 4) == 0 && !getenv("IVX_FORCE_SORT")) sorted_in = true;
    if (sorted_in && !packed) {
        hipLaunchKernelGGL(k_copy_sorted, dim3(grid1(n)), dim3(ST), 0, st, key, s, e, n, ks, ss, es, rows);
        if (also) {
            u64 *pw;
            IVX_TRY(ctx->get_scratch(slot_a, n * sizeof(u64), (void **)&pw));
            hipLaunchKernelGGL(k_pack1, dim3(grid1(n)), dim3(ST), 0, st, key, s, e, n, p, pw, (u32 *)nullptr);
            pk->ok = true; pk->w = pw; pk->p = p;
        }
        IVX_HIP(ctx, hipGetLastError());
        return IVX_OK;
    }
    narrow = narrow && nw == 1 && (u64)r.max_s - (u64)r.min_s <= 0xFFFFFFFFull && (u64)r.max_e - (u64)r.min_e <= 0xFFFFFFFFull;
    const bool k8 = narrow && nkeys <= 256 && !getenv("IVX_NO_K8");
    if (narrow) { sw->s32 = (const u32 *)ss; sw->e32 = (const u32 *)es; sw->min_s = r.min_s; sw->min_e = r.min_e; if (k8) sw->k8 = (const u8 *)ks; }
    for (int q = 0; q < nw; q++) {
        IVX_TRY(ctx->get_scratch(slot_a + q, n * sizeof(u64), (void **)&a[q]));
        if (!(packed && sorted_in)) IVX_TRY(ctx->get_scratch(slot_b + q, n * sizeof(u64), (void **)&b[q]));
    }
    u32 *pay[2] = {nullptr, nullptr};
    const bool with_rows = nw == 1 && rows != nullptr;
    if (with_rows) {
        IVX_TRY(ctx->get_scratch(slot_a + 1, n * sizeof(u32), (void **)&pay[0]));
        IVX_TRY(ctx->get_scratch(slot_b + 1, n * sizeof(u32), (void **)&pay[1]));
    }
    int in_b = 0;
    // pack for a sort whose first digit starts at bit `shift`: the pack kernel leaves that pass's histograms (k_pack1h)
    const bool pack_counts = !getenv("IVX_NO_PACK_HIST");
    auto pack_for_sort = [&](int shift, u32 *payload) -> ivx_status {
        if (!pack_counts) { hipLaunchKernelGGL(k_pack1, dim3(grid1(n)), dim3(ST), 0, st, key, s, e, n, p, a[0], payload); return IVX_OK; }
        u64 chunk; u32 nblk; u32 *hist;
        ivx_sort_geometry1(n, &chunk, &nblk);
        IVX_TRY(ctx->get_scratch(WS_SORTHIST, (size_t)256 * nblk * sizeof(u32), (void **)&hist));
        hipLaunchKernelGGL(k_pack1h, dim3(nblk), dim3(ST), 0, st, key, s, e, n, p, a[0], payload, shift, nblk, hist, (u32)chunk);
        return IVX_OK;
    };
    if (packed) {
        if (sorted_in) hipLaunchKernelGGL(k_pack1, dim3(grid1(n)), dim3(ST), 0, st, key, s, e, n, p, a[0], (u32 *)nullptr);
        else {
            const int lo = (int)p.bits_e;
            const ivx_sort_field f[1] = {{0, lo, lo + (int)((total - p.bits_e + 7) / 8 * 8)}};
            IVX_TRY(pack_for_sort(lo, nullptr));
            IVX_TRY(ivx_radix_sort(ctx, 1, a, b, n, f, 1, &in_b, true, nullptr, f[0].hi > f[0].lo && pack_counts));
        }
        pk->ok = true; pk->w = in_b ? b[0] : a[0]; pk->p = p;
        IVX_HIP(ctx, hipGetLastError());
        return IVX_OK;
    }
    if (nw == 1) {
        // Few rows share a (key,start) when the rows are sparse in the coordinate space: then sort on those bits
        // only -- the end bits would be three or four more digit passes -- and order the short runs of equal
        // (key,start) afterwards (k_fix_runs).
        const double per_pos = (double)n / positions;
        // rows already in (key, start) order -- coordinate-sorted files that do not order the ends of equal starts: the words
        // need no sort at all, only the repair of their equal-start runs
        const bool ks_sorted = (r.odd & 4) == 0 && !getenv("IVX_FORCE_SORT");
        bool two_step = ks_sorted || (p.bits_e >= 8 && n >= (1u << 16) && per_pos <= 0.25 && !getenv("IVX_FORCE_SORT"));
        u64 *const *o = a;
        if (two_step) {
            const int lo = (int)p.bits_e;
            const ivx_sort_field f[1] = {{0, lo, lo + (int)((total - p.bits_e + 7) / 8 * 8)}};
            if (ks_sorted) hipLaunchKernelGGL(k_pack1, dim3(grid1(n)), dim3(ST), 0, st, key, s, e, n, p, a[0], pay[0]);
            else {
                IVX_TRY(pack_for_sort(lo, pay[0]));
                IVX_TRY(ivx_radix_sort(ctx, 1, a, b, n, f, 1, &in_b, true, with_rows ? pay : nullptr, f[0].hi > f[0].lo && pack_counts));
            }
            o = in_b ? b : a;
            u32 *toolong = flags + 1;                                   // (upper half of the key-flag word; zeroed by the caller)
            if (narrow && k8)
            hipLaunchKernelGGL((k_unpack1<true, true, true>), dim3((grid1(n) + UNPACK_TILES - 1) / UNPACK_TILES), dim3(ST), 0, st, (const u64 *)o[0], (const u32 *)pay[in_b], n, p, ks, ss, es, rows,
                               p.bits_e, toolong);
            else if (narrow)
            hipLaunchKernelGGL((k_unpack1<true, true>), dim3((grid1(n) + UNPACK_TILES - 1) / UNPACK_TILES), dim3(ST), 0, st, (const u64 *)o[0], (const u32 *)pay[in_b], n, p, ks, ss, es, rows,
                               p.bits_e, toolong);
            else
            hipLaunchKernelGGL((k_unpack1<true>), dim3((grid1(n) + UNPACK_TILES - 1) / UNPACK_TILES), dim3(ST), 0, st, (const u64 *)o[0], (const u32 *)pay[in_b], n, p, ks, ss, es, rows,
                               p.bits_e, toolong);
            IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 8, flags, sizeof(u64), hipMemcpyDeviceToHost, st));
            IVX_HIP(ctx, hipStreamSynchronize(st));
            if (((const u32 *)(ctx->h_scalars + 8))[1]) {                // a long run of equal (key,start): the plain way after all
                IVX_HIP(ctx, hipMemsetAsync(toolong, 0, sizeof(u32), st));
                IVX_TRY(pack_for_sort(0, pay[0]));
                two_step = false;
            }
        }
        else IVX_TRY(pack_for_sort(0, pay[0]));
        if (!two_step) {
            const ivx_sort_field f[1] = {{0, 0, (int)((total + 7) / 8 * 8)}};
            IVX_TRY(ivx_radix_sort(ctx, 1, a, b, n, f, 1, &in_b, true, with_rows ? pay : nullptr, total > 0 && pack_counts));
            o = in_b ? b : a;
            if (narrow && k8)
            hipLaunchKernelGGL((k_unpack1<false, true, true>), dim3((grid1(n) + UNPACK_TILES - 1) / UNPACK_TILES), dim3(ST), 0, st, (const u64 *)o[0], (const u32 *)pay[in_b], n, p, ks, ss, es, rows,
                               0u, (u32 *)nullptr);
            else if (narrow)
            hipLaunchKernelGGL((k_unpack1<false, true>), dim3((grid1(n) + UNPACK_TILES - 1) / UNPACK_TILES), dim3(ST), 0, st, (const u64 *)o[0], (const u32 *)pay[in_b], n, p, ks, ss, es, rows,
                               0u, (u32 *)nullptr);
            else
            hipLaunchKernelGGL((k_unpack1<false>), dim3((grid1(n) + UNPACK_TILES - 1) / UNPACK_TILES), dim3(ST), 0, st, (const u64 *)o[0], (const u32 *)pay[in_b], n, p, ks, ss, es, rows,
                               0u, (u32 *)nullptr);
        }
        if (also) { pk->ok = true; pk->w = o[0]; pk->p = p; }
    } else {
        hipLaunchKernelGGL(k_pack64, dim3(grid1(n)), dim3(ST), 0, st, key, s, e, n, nkeys, a[0], a[1], a[2], flags);
        const ivx_sort_field f[3] = {{0, 0, 64}, {1, 0, 64}, {2, 32, 64}};
        IVX_TRY(ivx_radix_sort(ctx, 3, a, b, n, f, 3, &in_b));
        u64 *const *o = in_b ? b : a;
        hipLaunchKernelGGL(k_unpack64, dim3(grid1(n)), dim3(ST), 0, st, (const u64 *)o[0], (const u64 *)o[1], (const u64 *)o[2], n, ks, ss, es, rows);
    }
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

// ---------------------------------------------------------------- subtract

struct SegMax64 { i64 v; u32 head; u32 pad; };
struct SegMax64Op {
    using T = SegMax64;
    __host__ __device__ static T identity() { T t; t.v = INT64_MIN; t.head = 0; t.pad = 0; return t; }
    __device__ static T combine(const T &a, const T &b)
    {
        T r; r.pad = 0; r.head = a.head | b.head; r.v = b.head ? b.v : (a.v > b.v ? a.v : b.v); return r;
    }
    __device__ static T shfl_up(const T &x, int d)
    {
        T r; r.pad = 0; r.v = __shfl_up(x.v, d, IVX_WAVE); r.head = __shfl_up(x.head, d, IVX_WAVE); return r;
    }
};

__global__ __launch_bounds__(ST) void k_segmax64_in(const u32 *__restrict__ ks, const i64 *__restrict__ es, u64 n, SegMax64 *sm)
{
    const u64 i = (u64)blockIdx.x * ST + threadIdx.x;
    if (i >= n) return;
    SegMax64 t; t.pad = 0; t.v = es[i]; t.head = (i == 0 || ks[i] != ks[i - 1]) ? 1u : 0u;
    sm[i] = t;
}

// gap-head flag of every right row (as u32 for the sum scan); sm = inclusive running max of right ends per key
__global__ __launch_bounds__(ST) void k_gap_flags(const u32 *__restrict__ rk, const i64 *__restrict__ rs, const SegMax64 *__restrict__ sm,
                                                  u64 n, u32 *flag)
{
    const u64 j = (u64)blockIdx.x * ST + threadIdx.x;
    if (j > n) return;
    if (j == n) { flag[j] = 0; return; }
    const bool first = j == 0 || rk[j] != rk[j - 1];
    flag[j] = (first || rs[j] > sm[j - 1].v) ? 1u : 0u;
}

__global__ __launch_bounds__(ST) void k_gap_compact(const u32 *__restrict__ rk, const i64 *__restrict__ rs, const SegMax64 *__restrict__ sm,
                                                    const u32 *__restrict__ hid, u64 n, u32 *hk, i64 *hrs, i64 *hpm, u32 *hj)
{
    const u64 j = (u64)blockIdx.x * ST + threadIdx.x;
    if (j >= n) return;
    if (hid[j + 1] == hid[j]) return;
    const bool first = j == 0 || rk[j] != rk[j - 1];
    const u32 h = hid[j];
    hk[h] = rk[j]; hrs[h] = rs[j]; hpm[h] = first ? INT64_MIN : sm[j - 1].v; hj[h] = (u32)j;
}

// #elements with (key,val) < (k,x)  [strict_lt]  or  <= (k,x)
__device__ __forceinline__ u32 lex_rank(const u32 *__restrict__ kk, const i64 *__restrict__ vv, u32 n, u32 k, i64 x, bool le)
{
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        const u32 mk = kk[mid];
        bool before;
        if (mk != k) before = mk < k;
        else { const i64 mv = vv[mid]; before = le ? (mv <= x) : (mv < x); }
        if (before) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- partition points with a hint.  `before(i)` is monotone (true for i < answer).
template <class Before>
__device__ __forceinline__ u32 bisect(u32 lo, u32 hi, Before before)
{
    while (lo < hi) { const u32 mid = lo + ((hi - lo) >> 1); if (before(mid)) lo = mid + 1; else hi = mid; }
    return lo;
}

// answer expected at or a few elements after `start`: gallop upwards (1, 2, 4, ...), then bisect the last
// stride -- a handful of steps instead of log2(n).  If the answer lies before `start` after all, bisect [0,start).
template <class Before>
__device__ __forceinline__ u32 rank_near_up(u32 start, u32 n, Before before)
{
    if (start > n) start = n;
    if (start > 0 && !before(start - 1)) return bisect(0u, start - 1, before);
    u32 lo = start, hi = start, step = 1;
    while (hi < n && before(hi)) { lo = hi + 1; hi = (n - hi > step) ? hi + step : n; step <<= 1; }
    return bisect(lo, hi, before);
}

// answer expected at or a few elements before `start`
template <class Before>
__device__ __forceinline__ u32 rank_near_down(u32 start, u32 n, Before before)
{
    if (start > n) start = n;
    if (start < n && before(start)) return bisect(start + 1, n, before);
    u32 lo = start, hi = start, step = 1;
    while (lo > 0 && !before(lo - 1)) { hi = lo - 1; lo = lo > step ? lo - step : 0; step <<= 1; }
    return bisect(lo, hi, before);
}

struct SubPlan { u32 h_lo, h_hi; i64 tail_from; u32 has_tail; };

// One left row [ls,le) of key k against the rights of that key (sorted by start; sm = running max of their
// ends; heads = rights that start a new gap).  Two of the five partition points are real searches; the
// other three sit next to them (the row's end is a few rights after its start), so they gallop from there.
// [ha, hb]: bounds of the first search (the heads at or below the row's start): the caller's workgroup holds consecutive
// rows of the sorted left side, so the answers of its first and last row bracket everybody's (0 .. nh if unknown)
// the arrays plan_row searches, read from global memory ...
struct SubGlobal {
    const u32 *hk; const i64 *hrs; const u32 *hj; const u32 *rk; const i64 *rs; const SegMax64 *sm;
    __device__ __forceinline__ u32 Hk(u32 i) const { return hk[i]; }
    __device__ __forceinline__ i64 Hrs(u32 i) const { return hrs[i]; }
    __device__ __forceinline__ u32 Hj(u32 i) const { return hj[i]; }
    __device__ __forceinline__ u32 Rk(u32 i) const { return rk[i]; }
    __device__ __forceinline__ i64 Rs(u32 i) const { return rs[i]; }
    __device__ __forceinline__ i64 Pm(u32 i) const { return sm[i].v; }
};
// ... or from the workgroup's LDS copy of the stretch its rows need: heads [h0, h1), rights [r0, r1).  An index outside
// raises `esc` and reads as 0 -- the caller then plans that row again from global memory.  (Every read here is an LDS
// read; an accessor that falls back to global memory per access compiles to flat loads with full waits, 36 -> 46 ms.)
struct SubLds {
    const u32 *hk; const i64 *hrs; const u32 *hj; const u32 *rk; const i64 *rs; const i64 *pm;
    u32 h0, hn, r0, rn;
    mutable bool esc;
    __device__ __forceinline__ u32 hi_(u32 i) const { const u32 x = i - h0; if (x >= hn) { esc = true; return 0u; } return x; }
    __device__ __forceinline__ u32 ri_(u32 i) const { const u32 x = i - r0; if (x >= rn) { esc = true; return 0u; } return x; }
    __device__ __forceinline__ u32 Hk(u32 i) const { return hk[hi_(i)]; }
    __device__ __forceinline__ i64 Hrs(u32 i) const { return hrs[hi_(i)]; }
    __device__ __forceinline__ u32 Hj(u32 i) const { return hj[hi_(i)]; }
    __device__ __forceinline__ u32 Rk(u32 i) const { return rk[ri_(i)]; }
    __device__ __forceinline__ i64 Rs(u32 i) const { return rs[ri_(i)]; }
    __device__ __forceinline__ i64 Pm(u32 i) const { return pm[ri_(i)]; }
};

// WF: every right row has start <= end (k_range64 says so).  Then the gap heads are the merged runs of the rights, the right_cursor
// can never be behind the first head above ls (that head's own row ends beyond ls), and the walk's last right with rs <= le lies in
// the run of the last head with rs <= le, whose final end is either the cursor or beyond le: TWO searches and one lookup instead of
// five partition points.
template <class A, bool WF>
__device__ __forceinline__ SubPlan plan_row(u32 k, i64 ls, i64 le, int strict, const A &a, u32 nh, u32 nr, u32 ha, u32 hb)
{
    SubPlan p;
    const bool incl = !strict;
    // heads with rs <= ls never emit
    const u32 h_ls = bisect(ha, hb, [&](u32 i) { const u32 mk = a.Hk(i); if (mk != k) return mk < k; return a.Hrs(i) <= ls; });
    if (WF) {
        p.h_lo = h_ls;
        p.h_hi = rank_near_up(h_ls, nh, [&](u32 i) { const u32 mk = a.Hk(i); if (mk != k) return mk < k; const i64 mv = a.Hrs(i); return incl ? (mv <= le) : (mv < le); });
        if (p.h_hi < p.h_lo) p.h_hi = p.h_lo;
        i64 cursor = ls;
        if (p.h_hi > 0 && a.Hk(p.h_hi - 1) == k) {
            const u32 jl = (p.h_hi < nh ? a.Hj(p.h_hi) : nr) - 1u;      // the last right of that head's run
            const i64 pm = a.Pm(jl);
            if (pm > cursor) cursor = pm;
        }
        p.tail_from = cursor;
        p.has_tail = cursor < le ? 1u : 0u;
        return p;
    }
    // right_cursor (subtract.rs:401-412): first right whose running max end reaches ls.  Everything before the
    // last head at or below ls ends below that head's start, so the search starts there; it normally ends
    // before the next head (checked, not assumed: rights with end < start break it)
    const bool pm_le = strict != 0;
    auto pm_before = [&](u32 i) { const u32 mk = a.Rk(i); if (mk != k) return mk < k; const i64 mv = a.Pm(i); return pm_le ? (mv <= ls) : (mv < ls); };
    const u32 rc_lo = h_ls > 0 ? a.Hj(h_ls - 1) : 0u;
    u32 rc_hi = h_ls < nh ? a.Hj(h_ls) : nr;
    if (rc_hi < nr && pm_before(rc_hi)) rc_hi = nr;
    const u32 rc = bisect(rc_lo, rc_hi, pm_before);
    // heads behind the cursor are skipped: first head at or after it, at or just before h_ls
    const u32 hc = rank_near_down(h_ls, nh, [&](u32 i) { return a.Hj(i) < rc; });
    p.h_lo = hc > h_ls ? hc : h_ls;
    // heads with rs <= le (strict: rs < le)
    p.h_hi = rank_near_up(h_ls, nh, [&](u32 i) { const u32 mk = a.Hk(i); if (mk != k) return mk < k; const i64 mv = a.Hrs(i); return incl ? (mv <= le) : (mv < le); });
    if (p.h_hi < p.h_lo) p.h_hi = p.h_lo;
    // rights visited by the walk: rs <= le (strict: <); they begin at the cursor
    const u32 jhi = rank_near_up(rc, nr, [&](u32 i) { const u32 mk = a.Rk(i); if (mk != k) return mk < k; const i64 mv = a.Rs(i); return incl ? (mv <= le) : (mv < le); });
    i64 cursor = ls;
    if (jhi > 0 && a.Rk(jhi - 1) == k) { const i64 pm = a.Pm(jhi - 1); if (pm > cursor) cursor = pm; }
    p.tail_from = cursor;
    p.has_tail = cursor < le ? 1u : 0u;                                 // subtract.rs:435
    return p;
}

// The workgroup's rows are consecutive rows of the sorted left side: the heads at or below its first and last row's start
// bracket everybody's first search, and the heads / rights all its searches touch are a short stretch around that bracket
// -- staged in LDS (SUB_HW heads, SUB_RW rights), so that the ~70 dependent reads of a row's five partition points are LDS
// reads (11.7 -> ... ms for 200 M rows); a row whose searches leave the stretch (a long row, rights with end < start) is
// planned again from global memory, and a workgroup whose stretch does not fit plans all its rows there.
constexpr u32 SUB_HW = 256, SUB_RW = 640, SUB_MARGIN = 48;      // (17 KB of LDS: eight workgroups per CU)
// brk[b] = heads at or below the start of the first left row of workgroup b (brk[#workgroups]: of the last row): one thread
// per boundary, all searches in flight together.  (Done by two threads at the head of every k_sub_count workgroup, the two
// full-length searches -- ~24 dependent reads -- held the other 254 threads up: 9 of the kernel's 11.7 ms.)
__global__ __launch_bounds__(ST) void k_sub_brackets(const u32 *__restrict__ lk, const i64 *__restrict__ lsv, u64 nl, const u32 *hk, const i64 *hrs, u32 nh,
                                                     u32 nblk, u32 *__restrict__ brk)
{
    const u32 b = blockIdx.x * ST + threadIdx.x;
    if (b > nblk) return;
    const u64 r = b < nblk ? (u64)b * ST : nl - 1;
    brk[b] = r < nl ? lex_rank(hk, hrs, nh, lk[r], lsv[r], true) : nh;
}

__global__ __launch_bounds__(ST) void k_sub_count(const u32 *__restrict__ lk, const i64 *__restrict__ lsv, const i64 *__restrict__ lev, u64 nl,
                                                  int strict, const u32 *hk, const i64 *hrs, const u32 *hj, u32 nh,
                                                  const u32 *rk, const i64 *rs, const SegMax64 *sm, u32 nr, u64 *cnt,
                                                  u32 *__restrict__ plan_hlo, i64 *__restrict__ plan_tail, const u32 *__restrict__ brk, int wf)
{
    __shared__ u32 s_h[2], s_win[4];
    __shared__ u32 l_hk[SUB_HW], l_hj[SUB_HW], l_rk[SUB_RW];
    __shared__ i64 l_hrs[SUB_HW], l_rs[SUB_RW], l_pm[SUB_RW];
    const u64 i0 = (u64)blockIdx.x * ST;
    // the answers of the workgroup's first row and of the next workgroup's first row (the last row's, for the last
    // workgroup) bracket everybody's first search: the left side is sorted
    if (threadIdx.x < 2 && i0 < nl) s_h[threadIdx.x] = brk[blockIdx.x + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0 && i0 < nl) {
        const u32 h0 = s_h[0] > SUB_MARGIN + 1 ? s_h[0] - SUB_MARGIN - 1 : 0u;
        const u32 h1 = nh - s_h[1] > SUB_MARGIN ? s_h[1] + SUB_MARGIN : nh;
        const u32 r0 = h0 < nh ? hj[h0] : nr;
        u32 r1 = h1 < nh ? hj[h1] : nr;
        r1 = nr - r1 > SUB_MARGIN ? r1 + SUB_MARGIN : nr;
        s_win[0] = h0; s_win[1] = h1 - h0; s_win[2] = r0 < r1 ? r0 : r1; s_win[3] = r1 > r0 ? r1 - r0 : 0u;
    }
    __syncthreads();
    const bool lds = i0 < nl && s_win[1] <= SUB_HW && s_win[3] <= SUB_RW;          // (the same for the whole workgroup)
    if (lds) {
        const u32 h0 = s_win[0], hn = s_win[1], r0 = s_win[2], rn = s_win[3];
        for (u32 x = threadIdx.x; x < hn; x += ST) { l_hk[x] = hk[h0 + x]; l_hrs[x] = hrs[h0 + x]; l_hj[x] = hj[h0 + x]; }
        for (u32 x = threadIdx.x; x < rn; x += ST) { l_rk[x] = rk[r0 + x]; l_rs[x] = rs[r0 + x]; l_pm[x] = sm[r0 + x].v; }
        __syncthreads();
    }
    const u64 i = i0 + threadIdx.x;
    if (i > nl) return;
    if (i == nl) { cnt[i] = 0; return; }
    const u32 k = lk[i]; const i64 ls = lsv[i], le = lev[i];
    const SubGlobal g{hk, hrs, hj, rk, rs, sm};
    SubPlan p;
    bool redo = !lds;
    if (lds) {
        SubLds a{l_hk, l_hrs, l_hj, l_rk, l_rs, l_pm, s_win[0], s_win[1], s_win[2], s_win[3], false};
        p = wf ? plan_row<SubLds, true>(k, ls, le, strict, a, nh, nr, s_h[0], s_h[1]) : plan_row<SubLds, false>(k, ls, le, strict, a, nh, nr, s_h[0], s_h[1]);
        redo = a.esc;
    }
    if (redo) p = wf ? plan_row<SubGlobal, true>(k, ls, le, strict, g, nh, nr, s_h[0], s_h[1]) : plan_row<SubGlobal, false>(k, ls, le, strict, g, nh, nr, s_h[0], s_h[1]);
    cnt[i] = (u64)(p.h_hi - p.h_lo) + p.has_tail;
    // the row's plan stays for the fill pass (12 bytes per row instead of the five searches again): its first head and
    // where its tail fragment starts; the number of heads follows from the scanned counts, has_tail from tail < end
    plan_hlo[i] = p.h_lo; plan_tail[i] = p.tail_from;
}

// Fragments of ST consecutive left rows, one thread per OUTPUT row: the rows' plans (k_sub_count) and scanned counts
// are staged in LDS, a thread finds the left row of its fragment there by bisection and writes the fragment -- the
// stores of a wavefront are consecutive whatever the rows' fragment counts are (a thread per left row looping over its
// fragments wrote strided, divergent runs: 9.1 ms for 564 M fragments).
__global__ __launch_bounds__(ST) void k_sub_fill(const u32 *__restrict__ lk, const i64 *__restrict__ lsv, const i64 *__restrict__ lev,
                                                 const u32 *__restrict__ lrow, u64 nl, int strict,
                                                 const u32 *hk, const i64 *hrs, const i64 *hpm, const u32 *hj, u32 nh,
                                                 const u32 *rk, const i64 *rs, const SegMax64 *sm, u32 nr,
                                                 const u64 *__restrict__ offs, u64 cap,
                                                 u32 *ok, i64 *os, i64 *oe, u32 *orow,
                                                 const u32 *__restrict__ plan_hlo, const i64 *__restrict__ plan_tail)
{
    __shared__ u64 s_off[ST + 1];
    __shared__ i64 s_ls[ST], s_le[ST], s_tail[ST];
    __shared__ u32 s_k[ST], s_row[ST], s_hlo[ST];
    const u64 i0 = (u64)blockIdx.x * ST;
    const u32 nrows = (u32)(nl - i0 < (u64)ST ? nl - i0 : (u64)ST);
    const u32 t = threadIdx.x;
    if (t < nrows) {
        const u64 i = i0 + t;
        s_off[t] = offs[i]; s_k[t] = lk[i]; s_ls[t] = lsv[i]; s_le[t] = lev[i]; s_row[t] = lrow ? lrow[i] : 0u;
        s_hlo[t] = plan_hlo[i]; s_tail[t] = plan_tail[i];
    }
    if (t == 0) s_off[nrows] = offs[i0 + nrows];
    __syncthreads();
    const u64 base = s_off[0];
    const u64 total = s_off[nrows] - base;
    for (u64 o = t; o < total; o += ST) {
        const u64 at = base + o;
        if (at >= cap) return;
        u32 lo = 0, hi = nrows - 1;                                     // last row r with s_off[r] <= at
        while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (s_off[mid] <= at) lo = mid; else hi = mid - 1; }
        const u32 r = lo;
        const u32 j = (u32)(at - s_off[r]);
        const i64 ls = s_ls[r], le = s_le[r];
        const u32 has_tail = s_tail[r] < le ? 1u : 0u;
        const u32 nheads = (u32)(s_off[r + 1] - s_off[r]) - has_tail;
        i64 fs, fe;
        if (j < nheads) {                                               // [cursor, rs)  subtract.rs:423-428
            const u32 h = s_hlo[r] + j;
            const i64 pm = hpm[h];
            fs = pm > ls ? pm : ls; fe = hrs[h];
        } else { fs = s_tail[r]; fe = le; }                             // [cursor, le)  :435-440
        if (ok) ok[at] = s_k[r];
        if (os) os[at] = fs;
        if (oe) oe[at] = fe;
        if (orow) orow[at] = s_row[r];
    }
}

ivx_status keyflag(ivx_ctx *ctx, const char *what)
{
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 8, ctx->d_scalars + 8, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*(u32 *)(ctx->h_scalars + 8)) return ctx->fail(IVX_ERR_INVALID, what);
    return IVX_OK;
}

// ---------------------------------------------------------------- complement
//
// For one view interval [vs,ve) the reference walks the merged runs of the contig with cursor = vs:
// runs with me <= vs are skipped, the walk stops at the first ms >= ve, a run emits [cursor, max(ms,vs))
// if that is non-empty and sets cursor = min(me,ve); finally [cursor, ve) if cursor < ve
// (complement.rs:320-356).  Merged runs satisfy ms[j] > me[j-1] (strict: >=), and for well-formed
// input me ascends too, so the runs a view touches are the contiguous range
//     a = first run with me > vs,   b = first run with ms >= ve        (two binary searches)
// and its output is: a head [vs, ms[a]) if ms[a] > vs, the non-empty gaps [me[j-1], ms[j]) for
// a < j < b -- a property of the merged list alone, counted with a prefix sum G -- and a tail
// [min(me[b-1],ve), ve).  Output rows are then filled one thread per ROW (binary search for the view),
// so one chromosome-wide view is as parallel as a million small ones.  If the merged ends do not
// ascend (input rows with end < start), every view walks its contig serially instead: same answers as
// the reference for any input, speed only matters for sane data.

__global__ __launch_bounds__(ST) void k_mark_keys(const u32 *__restrict__ key, u64 n, u32 nkeys, u32 *flag, u32 *bad)
{
    const u64 i = (u64)blockIdx.x * ST + threadIdx.x;
    if (i >= n) return;
    const u32 k = key ? key[i] : 0u;
    if (k >= nkeys) { *bad = 1; return; }
    flag[k] = 1;
}

// need[k] = key has input rows but no view row: gets the implicit view [0, i64::MAX) (complement.rs:401-403)
__global__ __launch_bounds__(ST) void k_need_implicit(const u32 *__restrict__ has_in, const u32 *__restrict__ has_view, u32 nkeys, u32 *need)
{
    const u32 k = blockIdx.x * ST + threadIdx.x;
    if (k > nkeys) return;
    need[k] = (k < nkeys && has_in[k] && !has_view[k]) ? 1u : 0u;
}

__global__ __launch_bounds__(ST) void k_views_gather(const u32 *__restrict__ vkey, const i64 *__restrict__ vs, const i64 *__restrict__ ve, u64 nv,
                                                     const u32 *__restrict__ has_in, const u32 *__restrict__ has_view, const u32 *__restrict__ ioff,
                                                     u32 nkeys, u32 *wk, i64 *ws, i64 *we)
{
    const u64 i = (u64)blockIdx.x * ST + threadIdx.x;
    if (i < nv) { wk[i] = vkey ? vkey[i] : 0u; ws[i] = vs[i]; we[i] = ve[i]; }
    if (i < nkeys && has_in[i] && !has_view[i]) { const u64 at = nv + ioff[i]; wk[at] = (u32)i; ws[at] = 0; we[at] = INT64_MAX; }
}

// per merged run: gap flag (a non-empty gap in front of it inside its key) and the regularity check
__global__ __launch_bounds__(ST) void k_run_gaps(const u32 *__restrict__ mk, const i64 *__restrict__ ms, const i64 *__restrict__ me, u64 m,
                                                 u32 *gflag, u32 *irregular)
{
    const u64 j = (u64)blockIdx.x * ST + threadIdx.x;
    if (j > m) return;
    if (j == m) { gflag[j] = 0; return; }
    const bool same = j > 0 && mk[j] == mk[j - 1];
    gflag[j] = (same && ms[j] > me[j - 1]) ? 1u : 0u;
    if (same && me[j] < me[j - 1]) *irregular = 1;
}

__global__ __launch_bounds__(ST) void k_gap_index(const u32 *__restrict__ G, u64 m, u32 *cg)
{
    const u64 j = (u64)blockIdx.x * ST + threadIdx.x;
    if (j < m && G[j + 1] != G[j]) cg[G[j]] = (u32)j;
}

__global__ __launch_bounds__(ST) void k_view_flags(const u32 *__restrict__ wk, u64 nw, const u32 *__restrict__ has_in, u32 *f)
{
    const u64 i = (u64)blockIdx.x * ST + threadIdx.x;
    if (i > nw) return;
    f[i] = (i < nw && has_in[wk[i]]) ? 1u : 0u;
}

struct CompRuns { const u32 *mk; const i64 *ms, *me; u32 m; const u32 *G; const u32 *cg; };

// the reference's loop for one view, verbatim; emit(start,end) per output row; returns the row count
template <class F>
__device__ __forceinline__ u64 view_walk_serial(const CompRuns &R, u32 k, i64 vs, i64 ve, F &&emit)
{
    const u32 lo = lex_rank(R.mk, R.ms, R.m, k, INT64_MIN, false), hi = lex_rank(R.mk, R.ms, R.m, k, INT64_MAX, true);
    u64 rows = 0;
    i64 cursor = vs;
    for (u32 j = lo; j < hi; j++) {
        if (R.me[j] <= vs) continue;
        if (R.ms[j] >= ve) break;
        const i64 is = R.ms[j] > vs ? R.ms[j] : vs, ie = R.me[j] < ve ? R.me[j] : ve;
        if (is > cursor) { emit(rows, cursor, is); rows++; }
        cursor = ie;
    }
    if (cursor < ve) { emit(rows, cursor, ve); rows++; }
    return rows;
}

// count pass: one thread per view (sorted order i); results stored at the view's OUTPUT position:
// views of keys with input rows first, then view-only keys (complement.rs:431-456), both in (key,start,end) order
__global__ __launch_bounds__(ST) void k_comp_count(const u32 *__restrict__ wk, const i64 *__restrict__ ws, const i64 *__restrict__ we, u64 nw,
                                                   const u32 *__restrict__ has_in, const u32 *__restrict__ A, CompRuns R, int irregular,
                                                   u64 *cnt, u32 *inv, u32 *pa, u32 *pb)
{
    const u64 i = (u64)blockIdx.x * ST + threadIdx.x;
    if (i > nw) return;
    if (i == nw) { cnt[nw] = 0; return; }
    const u32 k = wk[i];
    const bool in = has_in[k] != 0;
    const u32 totalA = A[nw];
    const u32 pos = in ? A[i] : totalA + ((u32)i - A[i]);
    inv[pos] = (u32)i;
    const i64 vs = ws[i], ve = we[i];
    u64 c; u32 a = 0, b = 0;
    if (!in) c = 1;                                                     // the whole view, as it is
    else if (irregular) c = view_walk_serial(R, k, vs, ve, [](u64, i64, i64) {});
    else {
        a = lex_rank(R.mk, R.me, R.m, k, vs, true);                     // first run with me > vs
        b = lex_rank(R.mk, R.ms, R.m, k, ve, false);                    // first run with ms >= ve
        if (a >= b) { b = a; c = vs < ve ? 1 : 0; }
        else c = (R.ms[a] > vs ? 1u : 0u) + (u64)(R.G[b] - R.G[a + 1]) + (R.me[b - 1] < ve ? 1u : 0u);
    }
    cnt[pos] = c; pa[pos] = a; pb[pos] = b;
}

__global__ __launch_bounds__(ST) void k_comp_fill_rows(const u32 *__restrict__ wk, const i64 *__restrict__ ws, const i64 *__restrict__ we, u32 nw,
                                                       const u32 *__restrict__ has_in, CompRuns R, const u64 *__restrict__ offs,
                                                       const u32 *__restrict__ inv, const u32 *__restrict__ pa, const u32 *__restrict__ pb,
                                                       u64 total, u32 *ok, i64 *os, i64 *oe)
{
    const u64 o = (u64)blockIdx.x * ST + threadIdx.x;
    if (o >= total) return;
    u32 lo = 0, hi = nw;                                               // last view position with offs <= o
    while (lo < hi) { const u32 mid = lo + ((hi - lo) >> 1); if (offs[mid] <= o) lo = mid + 1; else hi = mid; }
    const u32 pos = lo - 1, i = inv[pos];
    const u32 k = wk[i];
    const i64 vs = ws[i], ve = we[i];
    i64 s0, e0;
    if (!has_in[k]) { s0 = vs; e0 = ve; }
    else {
        const u32 a = pa[pos], b = pb[pos];
        u64 r = o - offs[pos];
        if (a >= b) { s0 = vs; e0 = ve; }
        else {
            const u32 head = R.ms[a] > vs ? 1u : 0u;
            const u32 inter = R.G[b] - R.G[a + 1];
            if (head && r == 0) { s0 = vs; e0 = R.ms[a]; }
            else if (r - head < inter) { const u32 j = R.cg[R.G[a + 1] + (u32)(r - head)]; s0 = R.me[j - 1]; e0 = R.ms[j]; }
            else { s0 = R.me[b - 1]; e0 = ve; }                          // me[b-1] < ve here
        }
    }
    if (ok) ok[o] = k;
    if (os) os[o] = s0;
    if (oe) oe[o] = e0;
}

__global__ __launch_bounds__(ST) void k_comp_fill_serial(const u32 *__restrict__ wk, const i64 *__restrict__ ws, const i64 *__restrict__ we, u32 nw,
                                                         const u32 *__restrict__ has_in, CompRuns R, const u64 *__restrict__ offs,
                                                         const u32 *__restrict__ inv, u32 *ok, i64 *os, i64 *oe)
{
    const u32 pos = blockIdx.x * ST + threadIdx.x;
    if (pos >= nw) return;
    const u32 i = inv[pos], k = wk[i];
    const u64 at = offs[pos];
    auto put = [&](u64 r, i64 a, i64 b) { if (ok) ok[at + r] = k; if (os) os[at + r] = a; if (oe) oe[at + r] = b; };
    if (!has_in[k]) put(0, ws[i], we[i]);
    else view_walk_serial(R, k, ws[i], we[i], put);
}

}  // namespace

// ------------------------------------------------------------------------------------ device entry points

ivx_status ivx_merge_device(ivx_ctx *ctx, const u32 *key, const i64 *s, const i64 *e, u64 n, u32 nkeys,
                            i64 min_dist, int strict, u32 *ok, i64 *os, i64 *oe, i64 *on, u64 *m)
{
    *m = 0;
    if (n == 0) return IVX_OK;
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(u64), ctx->stream));
    u32 *ks; i64 *ss, *es;
    IVX_TRY(ctx->get_scratch(WS_T0, n * sizeof(u32), (void **)&ks));
    IVX_TRY(ctx->get_scratch(WS_T1, n * sizeof(i64), (void **)&ss));
    IVX_TRY(ctx->get_scratch(WS_T2, n * sizeof(i64), (void **)&es));
    SortedRows rows;
    PackedWant pk{min_dist, strict, false, nullptr, Pack64{}};
    IVX_TRY(sort64(ctx, WS_SA0, WS_SB0, key, s, e, n, nkeys, ks, ss, es, nullptr, &rows, &pk));
    ivx_runs_out ro{ok, os, oe, on};
    if (pk.ok) IVX_TRY(ivx_merge_runs_packed(ctx, pk.w, n, pk.p, min_dist, strict, ro, m));
    else IVX_TRY(ivx_merge_runs_rows(ctx, rows, n, min_dist, strict, ro, m));
    return keyflag(ctx, "merge: key id >= n_keys");
}

ivx_status ivx_subtract_fill_planned(ivx_ctx *ctx, u32 *ok, i64 *os, i64 *oe, u32 *orow, u64 cap, u64 *n_out);

ivx_status ivx_subtract_device(ivx_ctx *ctx, const u32 *lkey, const i64 *ls, const i64 *le, u64 nl,
                               const u32 *rkey, const i64 *rs, const i64 *re, u64 nr, u32 nkeys, int strict,
                               u32 *ok, i64 *os, i64 *oe, u32 *orow, u64 cap, u64 *n_out)
{
    *n_out = 0;
    if (nl == 0) return IVX_OK;
    hipStream_t st = ctx->stream;
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(u64), st));
    u32 *lk, *lrow, *rk; i64 *lsv, *lev, *rsv, *rev;
    IVX_TRY(ctx->get_scratch(WS_T0, nl * sizeof(u32), (void **)&lk));
    IVX_TRY(ctx->get_scratch(WS_T1, nl * sizeof(i64), (void **)&lsv));
    IVX_TRY(ctx->get_scratch(WS_T2, nl * sizeof(i64), (void **)&lev));
    IVX_TRY(ctx->get_scratch(WS_T3, nl * sizeof(u32), (void **)&lrow));
    IVX_TRY(sort64(ctx, WS_SA0, WS_SB0, lkey, ls, le, nl, nkeys, lk, lsv, lev, lrow));
    const u64 nra = nr ? nr : 1;
    IVX_TRY(ctx->get_scratch(WS_T4, nra * sizeof(u32), (void **)&rk));
    IVX_TRY(ctx->get_scratch(WS_T5, nra * sizeof(i64), (void **)&rsv));
    IVX_TRY(ctx->get_scratch(WS_T6, nra * sizeof(i64), (void **)&rev));
    IVX_TRY(sort64(ctx, WS_RA0, WS_RB0, rkey, rs, re, nr, nkeys, rk, rsv, rev, nullptr));
    // (sort64 left the right side's Range64 in the pinned scalars: odd bit 0 = some right row has end < start)
    const int wf = (nr == 0 || (ctx->h_scalars[29] & 1ull) == 0) && !getenv("IVX_SUB_GENERAL");
    IVX_TRY(keyflag(ctx, "subtract: key id >= n_keys"));

    // right side: running max of ends per key, gap heads
    SegMax64 *sm; u32 *hid, *hk, *hj; i64 *hrs, *hpm;
    IVX_TRY(ctx->get_scratch(WS_T7, nra * sizeof(SegMax64), (void **)&sm));
    IVX_TRY(ctx->get_scratch(WS_T8, (nr + 1) * sizeof(u32), (void **)&hid));
    u64 nh = 0;
    if (nr) {
        hipLaunchKernelGGL(k_segmax64_in, dim3(grid1(nr)), dim3(ST), 0, st, (const u32 *)rk, (const i64 *)rev, nr, sm);
        IVX_TRY(ivxscan::inclusive<SegMax64Op>(ctx, sm, nr));
        hipLaunchKernelGGL(k_gap_flags, dim3(grid1(nr + 1)), dim3(ST), 0, st, (const u32 *)rk, (const i64 *)rsv, (const SegMax64 *)sm, nr, hid);
        IVX_TRY(ivx_scan_exclusive_u32(ctx, hid, nr + 1));
        IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 4, hid + nr, sizeof(u32), hipMemcpyDeviceToHost, st));
        IVX_HIP(ctx, hipStreamSynchronize(st));
        nh = *(u32 *)(ctx->h_scalars + 4);
    }
    const u64 nha = nh ? nh : 1;
    // the sorted right ends are no longer needed: reuse their slot for the heads' keys
    IVX_TRY(ctx->get_scratch(WS_T9, nha * sizeof(i64), (void **)&hrs));
    IVX_TRY(ctx->get_scratch(WS_RA0, nha * sizeof(i64), (void **)&hpm));
    IVX_TRY(ctx->get_scratch(WS_RA1, nha * sizeof(u32), (void **)&hk));
    IVX_TRY(ctx->get_scratch(WS_RB0, nha * sizeof(u32), (void **)&hj));
    if (nr) hipLaunchKernelGGL(k_gap_compact, dim3(grid1(nr)), dim3(ST), 0, st, (const u32 *)rk, (const i64 *)rsv, (const SegMax64 *)sm, (const u32 *)hid, nr, hk, hrs, hpm, hj);

    u64 *cnt; u32 *plan_hlo; i64 *plan_tail;
    IVX_TRY(ctx->get_scratch(WS_RA2, (nl + 1) * sizeof(u64), (void **)&cnt));
    IVX_TRY(ctx->get_scratch(WS_RB1, nl * sizeof(u32), (void **)&plan_hlo));
    IVX_TRY(ctx->get_scratch(WS_RB2, nl * sizeof(i64), (void **)&plan_tail));
    const u32 nblk_c = grid1(nl + 1);
    u32 *brk;
    IVX_TRY(ctx->get_scratch(WS_GRID0, ((size_t)nblk_c + 2) * sizeof(u32), (void **)&brk));
    hipLaunchKernelGGL(k_sub_brackets, dim3(grid1((u64)nblk_c + 1)), dim3(ST), 0, st, (const u32 *)lk, (const i64 *)lsv, nl, (const u32 *)hk, (const i64 *)hrs, (u32)nh, nblk_c, brk);
    hipLaunchKernelGGL(k_sub_count, dim3(nblk_c), dim3(ST), 0, st, (const u32 *)lk, (const i64 *)lsv, (const i64 *)lev, nl, strict,
                       (const u32 *)hk, (const i64 *)hrs, (const u32 *)hj, (u32)nh, (const u32 *)rk, (const i64 *)rsv, (const SegMax64 *)sm, (u32)nr, cnt,
                       plan_hlo, plan_tail, (const u32 *)brk, wf);
    IVX_TRY(ivx_scan_exclusive_u64(ctx, cnt, nl + 1));
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 5, cnt + nl, sizeof(u64), hipMemcpyDeviceToHost, st));
    IVX_HIP(ctx, hipStreamSynchronize(st));
    const u64 total = ctx->h_scalars[5];
    *n_out = total;
    ivx_sub_plan &pl = ctx->sub_plan;
    pl.nl = nl; pl.nr = nr; pl.nh = nh; pl.total = total; pl.nkeys = nkeys; pl.strict = strict; pl.stream = st;
    pl.lk = lk; pl.lsv = lsv; pl.lev = lev; pl.lrow = lrow; pl.rk = rk; pl.rsv = rsv; pl.sm = sm;
    pl.hk = hk; pl.hrs = hrs; pl.hpm = hpm; pl.hj = hj; pl.offs = cnt; pl.plan_hlo = plan_hlo; pl.plan_tail = plan_tail;
    if (cap == 0 && !ok && !os && !oe && !orow) {                       // count only: the fill call that follows reuses all of it
        pl.slots = 0;
        for (int slot : {WS_T0, WS_T1, WS_T2, WS_T3, WS_T4, WS_T5, WS_T7, WS_T9, WS_RA0, WS_RA1, WS_RA2, WS_RB0, WS_RB1, WS_RB2}) pl.slots |= 1ull << slot;
        pl.valid = true;
        return IVX_OK;
    }
    return ivx_subtract_fill_planned(ctx, ok, os, oe, orow, cap, n_out);
}

// the output pass alone, from the plan a sizing call (or the lines above) left in the context
ivx_status ivx_subtract_fill_planned(ivx_ctx *ctx, u32 *ok, i64 *os, i64 *oe, u32 *orow, u64 cap, u64 *n_out)
{
    const ivx_sub_plan &pl = ctx->sub_plan;
    *n_out = pl.total;
    if (pl.total > cap) return ctx->fail(IVX_ERR_CAPACITY, "subtract: output buffers too small");
    hipLaunchKernelGGL(k_sub_fill, dim3(grid1(pl.nl)), dim3(ST), 0, ctx->stream, pl.lk, pl.lsv, pl.lev, pl.lrow, pl.nl, pl.strict,
                       pl.hk, pl.hrs, pl.hpm, pl.hj, (u32)pl.nh, pl.rk, pl.rsv, (const SegMax64 *)pl.sm, (u32)pl.nr,
                       pl.offs, cap, ok, os, oe, orow, pl.plan_hlo, pl.plan_tail);
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

ivx_status ivx_cluster_device(ivx_ctx *ctx, const u32 *key, const i64 *s, const i64 *e, u64 n, u32 nkeys,
                              i64 min_dist, int strict, const i64 *key_base,
                              u32 *ok, i64 *os, i64 *oe, u32 *orow, i64 *oc, i64 *ocs, i64 *oce, u64 *key_clusters, u64 *m)
{
    *m = 0;
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(u64), ctx->stream));
    // the sorted rows ARE output columns; scratch only for the ones the caller did not ask for
    u32 *ks = ok, *rows = orow; i64 *ss = os, *es = oe;
    if (n) {
        if (!ks) IVX_TRY(ctx->get_scratch(WS_T0, n * sizeof(u32), (void **)&ks));
        if (!ss) IVX_TRY(ctx->get_scratch(WS_T1, n * sizeof(i64), (void **)&ss));
        if (!es) IVX_TRY(ctx->get_scratch(WS_T2, n * sizeof(i64), (void **)&es));
    }
    PackedWant pk{min_dist, strict, false, nullptr, Pack64{}};
    if (n) IVX_TRY(sort64(ctx, WS_SA0, WS_SB0, key, s, e, n, nkeys, ks, ss, es, rows, nullptr, &pk));
    const ivx_cluster_out co{oc, ocs, oce, key_clusters};
    if (pk.ok) IVX_TRY(ivx_cluster_rows_packed(ctx, pk.w, pk.p, ks, n, nkeys, min_dist, strict, key_base, co, m));
    else IVX_TRY(ivx_cluster_rows(ctx, ks, ss, es, n, nkeys, min_dist, strict, key_base, co, m));
    return keyflag(ctx, "cluster: key id >= n_keys");
}

ivx_status ivx_complement_device(ivx_ctx *ctx, const u32 *key, const i64 *s, const i64 *e, u64 n,
                                 const u32 *vkey, const i64 *vs, const i64 *ve, u64 nv, u32 nkeys, int strict,
                                 u32 *ok, i64 *os, i64 *oe, u64 cap, u64 *n_out)
{
    *n_out = 0;
    hipStream_t st = ctx->stream;
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, 2 * sizeof(u64), st));
    u32 *bad = (u32 *)(ctx->d_scalars + 8), *irregular = (u32 *)(ctx->d_scalars + 9);

    // which keys have input rows / view rows; implicit views for input keys without one
    u32 *has_in, *has_view, *need;
    IVX_TRY(ctx->get_scratch(WS_GRID0, (size_t)nkeys * sizeof(u32), (void **)&has_in));
    IVX_TRY(ctx->get_scratch(WS_GRID1, (size_t)nkeys * sizeof(u32), (void **)&has_view));
    IVX_TRY(ctx->get_scratch(WS_GRID2, ((size_t)nkeys + 1) * sizeof(u32), (void **)&need));
    IVX_HIP(ctx, hipMemsetAsync(has_in, 0, (size_t)nkeys * sizeof(u32), st));
    IVX_HIP(ctx, hipMemsetAsync(has_view, 0, (size_t)nkeys * sizeof(u32), st));
    if (n) hipLaunchKernelGGL(k_mark_keys, dim3(grid1(n)), dim3(ST), 0, st, key, n, nkeys, has_in, bad);
    if (nv) hipLaunchKernelGGL(k_mark_keys, dim3(grid1(nv)), dim3(ST), 0, st, vkey, nv, nkeys, has_view, bad);
    hipLaunchKernelGGL(k_need_implicit, dim3(grid1((u64)nkeys + 1)), dim3(ST), 0, st, (const u32 *)has_in, (const u32 *)has_view, nkeys, need);
    IVX_TRY(ivx_scan_exclusive_u32(ctx, need, (u64)nkeys + 1));
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 4, need + nkeys, sizeof(u32), hipMemcpyDeviceToHost, st));
    IVX_TRY(keyflag(ctx, "complement: key id >= n_keys"));                  // synchronises
    const u64 nw = nv + *(u32 *)(ctx->h_scalars + 4);
    if (nw >= 0xFFFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "complement: more than 2^32-1 view rows");
    if (nw == 0) return IVX_OK;

    // merged runs of the input: sort, then the merge sweep with min_dist = 0 (complement.rs:297-317)
    u32 *mk; i64 *ms, *me; u64 m = 0;
    const u64 na = n ? n : 1;
    IVX_TRY(ctx->get_scratch(WS_T3, na * sizeof(u32), (void **)&mk));
    IVX_TRY(ctx->get_scratch(WS_T4, na * sizeof(i64), (void **)&ms));
    IVX_TRY(ctx->get_scratch(WS_T7, na * sizeof(i64), (void **)&me));
    if (n) {
        u32 *ks; i64 *ss, *es;
        IVX_TRY(ctx->get_scratch(WS_T0, n * sizeof(u32), (void **)&ks));
        IVX_TRY(ctx->get_scratch(WS_T1, n * sizeof(i64), (void **)&ss));
        IVX_TRY(ctx->get_scratch(WS_T2, n * sizeof(i64), (void **)&es));
        SortedRows rows;
        PackedWant pk{0, strict, false, nullptr, Pack64{}};
        IVX_TRY(sort64(ctx, WS_SA0, WS_SB0, key, s, e, n, nkeys, ks, ss, es, nullptr, &rows, &pk));
        const ivx_runs_out ro{mk, ms, me, nullptr};
        if (pk.ok) IVX_TRY(ivx_merge_runs_packed(ctx, pk.w, n, pk.p, 0, strict, ro, &m));
        else IVX_TRY(ivx_merge_runs_rows(ctx, rows, n, 0, strict, ro, &m));
    }
    u32 *G, *cg;
    IVX_TRY(ctx->get_scratch(WS_T5, (m + 1) * sizeof(u32), (void **)&G));       // the sweep's scratch is free again
    IVX_TRY(ctx->get_scratch(WS_T6, (m + 1) * sizeof(u32), (void **)&cg));
    hipLaunchKernelGGL(k_run_gaps, dim3(grid1(m + 1)), dim3(ST), 0, st, (const u32 *)mk, (const i64 *)ms, (const i64 *)me, m, G, irregular);
    IVX_TRY(ivx_scan_exclusive_u32(ctx, G, m + 1));
    if (m) hipLaunchKernelGGL(k_gap_index, dim3(grid1(m)), dim3(ST), 0, st, (const u32 *)G, m, cg);

    // views (+ implicit ones) sorted by (key, start, end): the per-contig view_bounds order
    u32 *uk, *wk; i64 *us, *ue, *ws, *we;
    IVX_TRY(ctx->get_scratch(WS_RA0, nw * sizeof(u32), (void **)&uk));
    IVX_TRY(ctx->get_scratch(WS_RA1, nw * sizeof(i64), (void **)&us));
    IVX_TRY(ctx->get_scratch(WS_RA2, nw * sizeof(i64), (void **)&ue));
    const u64 gmax = nv > nkeys ? nv : nkeys;
    hipLaunchKernelGGL(k_views_gather, dim3(grid1(gmax)), dim3(ST), 0, st, vkey, vs, ve, nv, (const u32 *)has_in, (const u32 *)has_view,
                       (const u32 *)need, nkeys, uk, us, ue);
    IVX_TRY(ctx->get_scratch(WS_T0, nw * sizeof(u32), (void **)&wk));            // the sorted input rows are dead by now
    IVX_TRY(ctx->get_scratch(WS_T1, nw * sizeof(i64), (void **)&ws));
    IVX_TRY(ctx->get_scratch(WS_T2, nw * sizeof(i64), (void **)&we));
    IVX_TRY(sort64(ctx, WS_SA0, WS_SB0, uk, us, ue, nw, nkeys, wk, ws, we, nullptr));

    u32 *A, *inv, *pa, *pb; u64 *cnt;
    IVX_TRY(ctx->get_scratch(WS_T8, (nw + 1) * sizeof(u32), (void **)&A));
    IVX_TRY(ctx->get_scratch(WS_T9, nw * sizeof(u32), (void **)&inv));
    IVX_TRY(ctx->get_scratch(WS_RB0, nw * sizeof(u32), (void **)&pa));
    IVX_TRY(ctx->get_scratch(WS_RB1, nw * sizeof(u32), (void **)&pb));
    IVX_TRY(ctx->get_scratch(WS_RB2, (nw + 1) * sizeof(u64), (void **)&cnt));
    hipLaunchKernelGGL(k_view_flags, dim3(grid1(nw + 1)), dim3(ST), 0, st, (const u32 *)wk, nw, (const u32 *)has_in, A);
    IVX_TRY(ivx_scan_exclusive_u32(ctx, A, nw + 1));
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 9, irregular, sizeof(u64), hipMemcpyDeviceToHost, st));
    IVX_HIP(ctx, hipStreamSynchronize(st));
    const int irr = *(u32 *)(ctx->h_scalars + 9) != 0;
    const CompRuns R{mk, ms, me, (u32)m, G, cg};
    hipLaunchKernelGGL(k_comp_count, dim3(grid1(nw + 1)), dim3(ST), 0, st, (const u32 *)wk, (const i64 *)ws, (const i64 *)we, nw,
                       (const u32 *)has_in, (const u32 *)A, R, irr, cnt, inv, pa, pb);
    IVX_TRY(ivx_scan_exclusive_u64(ctx, cnt, nw + 1));
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 5, cnt + nw, sizeof(u64), hipMemcpyDeviceToHost, st));
    IVX_HIP(ctx, hipStreamSynchronize(st));
    const u64 total = ctx->h_scalars[5];
    *n_out = total;
    if (cap == 0 && !ok && !os && !oe) return IVX_OK;                       // count only
    if (total > cap) return ctx->fail(IVX_ERR_CAPACITY, "complement: output buffers too small");
    if (total == 0) return IVX_OK;
    if (irr)
        hipLaunchKernelGGL(k_comp_fill_serial, dim3(grid1(nw)), dim3(ST), 0, st, (const u32 *)wk, (const i64 *)ws, (const i64 *)we, (u32)nw,
                           (const u32 *)has_in, R, (const u64 *)cnt, (const u32 *)inv, ok, os, oe);
    else
        hipLaunchKernelGGL(k_comp_fill_rows, dim3(grid1(total)), dim3(ST), 0, st, (const u32 *)wk, (const i64 *)ws, (const i64 *)we, (u32)nw,
                           (const u32 *)has_in, R, (const u64 *)cnt, (const u32 *)inv, (const u32 *)pa, (const u32 *)pb, total, ok, os, oe);
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}
