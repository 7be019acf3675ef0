// ivx_sort.hip -- stable LSD radix sort of multi-word records (hand-written;
// rocPRIM/hipCUB are not used).  Replaces the reference's per-contig
// comparison sorts: `sort_unstable` on (start,end[,row]) tuples
// (grouped_stream.rs:93-95, :213-215), by_start / by_end ordering
// (nearest_index.rs:50-55, :77-82), merge_intervals' stable sort by first
// (interval_tree.rs:57).
//
// A record is NW 64-bit words held as NW separate arrays (SoA).  One pass
// sorts by an 8-bit digit of one word:
//   k_hist    per-workgroup digit histograms (wave ballot "match" + LDS counters)
//   scan      exclusive prefix over [digit][workgroup]
//   k_scatter stable local ranking per wavefront (ballot + popcount), records
//             re-ordered through LDS so that global writes are contiguous runs
// HBM-bound: per pass 8 B/row read for the histogram + 8*NW B/row read and
// written by the scatter.  Digits whose bits are constant over the whole input
// are skipped (one OR-reduction up front).
#include "ivx_device.hpp"
#include "ivx_sort.hpp"

namespace {

// One workgroup per CU (the tile's records sit in LDS): big tiles make the per-digit runs written to HBM
// long (8192 records / 256 digits = 32 records = 256 B per word array), which is what the scatter lives on.
constexpr int RS_T = 1024;
constexpr int RS_WAVES = RS_T / IVX_WAVE;       // 16
constexpr int RS_HT = 256;                      // histogram kernel: threads (= digits)
constexpr int RS_HWAVES = RS_HT / IVX_WAVE;
constexpr u64 RS_CHUNK = 65536;                 // records per workgroup of a big sort (histogram and scatter agree on it; sort_impl
                                                // takes fewer for inputs that would not fill the chip: whole tiles, at least one)
template <int NW> struct Tile {
    static constexpr int I = NW == 3 ? 4 : NW == 1 ? 8 : 8;   // records per thread per tile: 8192-record tiles, 4096 for 24-byte records, 16384 for 8-byte ones (LDS)
    static constexpr int N = RS_T * I;
    static constexpr int WT = N / RS_WAVES;     // consecutive records per wavefront
};

template <int NW> struct Ptrs { u64 *w[NW]; u32 *p; };            // p: optional 32-bit payload riding along (12-byte records for NW = 1)
template <int NW> struct CPtrs { const u64 *w[NW]; const u32 *p; };

// word `word` of a record held in registers, without a runtime-indexed array
// (a runtime index would send the record to scratch memory)
template <int NW>
__device__ __forceinline__ u64 pick_word(const u64 (&x)[NW], int word)
{
    u64 v = x[0];
    if (NW > 1 && word == 1) v = x[NW > 1 ? 1 : 0];
    if (NW > 2 && word == 2) v = x[NW > 2 ? 2 : 0];
    return v;
}

// lanes of the wave holding the same 8-bit digit (valid lanes only)
__device__ __forceinline__ u64 match_digit(u32 d, bool valid)
{
    u64 peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1u;
        const u64 bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    return peers;
}

__global__ __launch_bounds__(RS_HT) void k_varbits(const u64 *__restrict__ w, u64 n, unsigned long long *out)
{
    __shared__ u64 lds[RS_HT / IVX_WAVE];
    const u64 x0 = n ? w[0] : 0;
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * RS_HT + threadIdx.x; i < n; i += (u64)gridDim.x * RS_HT) acc |= w[i] ^ x0;
#pragma unroll
    for (int d = IVX_WAVE / 2; d > 0; d >>= 1) acc |= __shfl_xor(acc, d, IVX_WAVE);
    if (lane_id() == 0) lds[threadIdx.x / IVX_WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 a = 0;
        for (int i = 0; i < RS_HT / IVX_WAVE; i++) a |= lds[i];
        if (a) atomicOr(out, (unsigned long long)a);
    }
}

__global__ __launch_bounds__(RS_HT) void k_hist(const u64 *__restrict__ w, u64 n, int shift, u32 nblk, u32 *__restrict__ hist, u32 chunk)
{
    __shared__ u32 cnt[RS_HWAVES][256];
    for (int i = threadIdx.x; i < RS_HWAVES * 256; i += RS_HT) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const u64 lo = (u64)blockIdx.x * chunk;
    const u64 hi = lo + chunk < n ? lo + chunk : n;
    const u32 wv = threadIdx.x / IVX_WAVE;
    // four independent loads per thread in flight; per-wavefront LDS counters.  A digit shared by the whole
    // wavefront (sorted or clustered keys) is counted once by one lane, anything else by LDS atomics, which
    // random digits rarely make collide.
    for (u64 i0 = lo; i0 < hi; i0 += RS_HT * 4) {
        u64 x[4]; bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const u64 i = i0 + (u64)u * RS_HT + threadIdx.x; valid[u] = i < hi; x[u] = valid[u] ? w[i] : 0; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
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
        u32 s = 0;
#pragma unroll
        for (int k = 0; k < RS_HWAVES; k++) s += cnt[k][d];
        hist[(u64)d * nblk + blockIdx.x] = s;
    }
}

template <int NW, bool PAY>
__global__ __launch_bounds__(RS_T) void k_scatter(CPtrs<NW> in, Ptrs<NW> out, u64 n, int word, int shift, u32 nblk,
                                                 const u32 *__restrict__ offs, u32 chunk)
{
    constexpr int RS_I = Tile<NW>::I, RS_TILE = Tile<NW>::N, RS_WTILE = Tile<NW>::WT;
    __shared__ u64 rec[NW][RS_TILE];
    __shared__ u32 recp[PAY ? RS_TILE : 1];
    // 8-byte records: the lanes that share a digit find each other through one 64-bit word per (wavefront, digit) -- every
    // lane ORs its bit in, reads the word, writes zero back (LDS operations of a wavefront execute in order) -- instead of
    // eight ballots and sixteen selects per record: the kernel is bound by the instructions it issues (122 VALU per 64
    // records with the ballots), not by its 3.2 GB.  The wider records' tiles leave no room for the 32 KB.
#ifdef IVX_SORT_NO_LMATCH
    constexpr bool LMATCH = false;
#else
    constexpr bool LMATCH = NW == 1;
#endif
    __shared__ u64 pmask[LMATCH ? RS_WAVES : 1][LMATCH ? 256 : 1];
    __shared__ u32 wcnt[RS_WAVES][256];          // per-wave digit counts -> exclusive offsets across waves
    __shared__ u32 dstart[256];                  // first local slot of the digit in this tile
    __shared__ u32 tcnt[256];                    // records of the digit in this tile
    __shared__ u32 gbase[256];                   // next global slot of the digit for this workgroup
    __shared__ u32 scan_lds[RS_T / IVX_WAVE + 1];

    const u32 tid = threadIdx.x, wv = tid / IVX_WAVE, ln = lane_id();
    if (tid < 256) gbase[tid] = offs[(u64)tid * nblk + blockIdx.x];
    if (LMATCH) for (int i = tid; i < RS_WAVES * 256; i += RS_T) (&pmask[0][0])[i] = 0;     // (the first tile's barrier orders this)
    const u64 lanebit = 1ull << ln;
    const u64 lo = (u64)blockIdx.x * chunk;
    const u64 hi = lo + chunk < n ? lo + chunk : n;

    // 8-byte records: the next tile's records are loaded while this one goes through LDS (registers allow it)
    // (not with the 32-bit payload: 24 more registers per thread spilled 156 bytes per lane and cost 7 %)
    constexpr bool PREFETCH = NW == 1 && !PAY;
    u64 nxt[PREFETCH ? RS_I : 1][NW];
    u32 nxp[PREFETCH && PAY ? RS_I : 1];
    auto load_tile = [&](u64 t0, u64 (&dst)[PREFETCH ? RS_I : 1][NW], u32 (&dp)[PREFETCH && PAY ? RS_I : 1]) {
        if (!PREFETCH) return;
        const u32 tn = (u32)(hi - t0 < RS_TILE ? hi - t0 : RS_TILE);
        if (tn == RS_TILE) {
#pragma unroll
            for (int k = 0; k < RS_I; k++) {
#pragma unroll
                for (int q = 0; q < NW; q++) dst[PREFETCH ? k : 0][q] = in.w[q][t0 + wv * RS_WTILE + k * IVX_WAVE + ln];
                if (PAY) dp[PREFETCH && PAY ? k : 0] = in.p[t0 + wv * RS_WTILE + k * IVX_WAVE + ln];
            }
        } else {
#pragma unroll
            for (int k = 0; k < RS_I; k++) {
                const u32 j = wv * RS_WTILE + k * IVX_WAVE + ln;
#pragma unroll
                for (int q = 0; q < NW; q++) dst[PREFETCH ? k : 0][q] = j < tn ? in.w[q][t0 + j] : 0;
                if (PAY) dp[PREFETCH && PAY ? k : 0] = j < tn ? in.p[t0 + j] : 0u;
            }
        }
    };
    if (PREFETCH && lo < hi) load_tile(lo, nxt, nxp);
    for (u64 t0 = lo; t0 < hi; t0 += RS_TILE) {
        for (int i = tid; i < RS_WAVES * 256; i += RS_T) (&wcnt[0][0])[i] = 0;
        __syncthreads();
        const u32 tile_n = (u32)(hi - t0 < RS_TILE ? hi - t0 : RS_TILE);
        u64 r[RS_I][NW];
        u32 rp[PAY ? RS_I : 1];
        u32 dig[RS_I], lrank[RS_I];
        // ---- the wave's records of this tile: every load is issued before the first one is used (a load inside the
        //      ranking loop below would be a full memory round trip per record)
        if (PREFETCH) {
#pragma unroll
            for (int k = 0; k < RS_I; k++)
#pragma unroll
                for (int q = 0; q < NW; q++) r[k][q] = nxt[PREFETCH ? k : 0][q];
            if (PAY) {
#pragma unroll
                for (int k = 0; k < RS_I; k++) rp[PAY ? k : 0] = nxp[PREFETCH && PAY ? k : 0];
            }
            if (t0 + RS_TILE < hi) load_tile(t0 + RS_TILE, nxt, nxp);
        } else if (tile_n == RS_TILE) {
#pragma unroll
            for (int k = 0; k < RS_I; k++) {
#pragma unroll
                for (int q = 0; q < NW; q++) r[k][q] = in.w[q][t0 + wv * RS_WTILE + k * IVX_WAVE + ln];
                if (PAY) rp[PAY ? k : 0] = in.p[t0 + wv * RS_WTILE + k * IVX_WAVE + ln];
            }
        } else {
#pragma unroll
            for (int k = 0; k < RS_I; k++) {
                const u32 j = wv * RS_WTILE + k * IVX_WAVE + ln;
#pragma unroll
                for (int q = 0; q < NW; q++) r[k][q] = j < tile_n ? in.w[q][t0 + j] : 0;
                if (PAY) rp[PAY ? k : 0] = j < tile_n ? in.p[t0 + j] : 0u;
            }
        }
        // ---- stable rank inside the wave's slice (index order = round, lane)
#pragma unroll
        for (int k = 0; k < RS_I; k++) {
            const u32 j = wv * RS_WTILE + k * IVX_WAVE + ln;        // slot in tile
            const bool valid = j < tile_n;
            const u32 d = valid ? (u32)((pick_word<NW>(r[k], word) >> shift) & 0xFF) : 0u;
            u64 peers;
            if (LMATCH) {
                peers = 0;
                if (valid) {
                    atomicOr((unsigned long long *)&pmask[LMATCH ? wv : 0][LMATCH ? d : 0], (unsigned long long)lanebit);
                    peers = pmask[LMATCH ? wv : 0][LMATCH ? d : 0];
                    pmask[LMATCH ? wv : 0][LMATCH ? d : 0] = 0;
                }
            } else peers = match_digit(d, valid);
            u32 base = 0;
            if (valid) base = wcnt[wv][d];
            const u32 rk = mask_rank(peers);
            if (valid && rk == 0) wcnt[wv][d] = base + (u32)__popcll(peers);
            dig[k] = valid ? d : 0xFFFFFFFFu;
            lrank[k] = base + rk;
        }
        __syncthreads();
        // ---- digit totals, offsets of each wave inside the digit, digit starts
        {
            const u32 d = tid;
            u32 run = 0;
            if (d < 256) {
#pragma unroll
                for (int k = 0; k < RS_WAVES; k++) { const u32 c = wcnt[k][d]; wcnt[k][d] = run; run += c; }
                tcnt[d] = run;
            }
            u32 tot;
            const u32 ds = block_excl_scan<u32, RS_T>(run, scan_lds, &tot);
            if (d < 256) dstart[d] = ds;
        }
        __syncthreads();
        // ---- local reorder through LDS
#pragma unroll
        for (int k = 0; k < RS_I; k++) {
            if (dig[k] != 0xFFFFFFFFu) {
                const u32 pos = dstart[dig[k]] + wcnt[wv][dig[k]] + lrank[k];
#pragma unroll
                for (int q = 0; q < NW; q++) rec[q][pos] = r[k][q];
                if (PAY) recp[PAY ? pos : 0] = rp[PAY ? k : 0];
            }
        }
        __syncthreads();
        // ---- contiguous runs to global
#pragma unroll
        for (int k = 0; k < RS_I; k++) {
            const u32 j = k * RS_T + tid;
            if (j < tile_n) {
                u64 x[NW];
#pragma unroll
                for (int q = 0; q < NW; q++) x[q] = rec[q][j];
                const u32 d = (u32)((pick_word<NW>(x, word) >> shift) & 0xFF);
                const u64 g = (u64)gbase[d] + (j - dstart[d]);
#pragma unroll
                for (int q = 0; q < NW; q++) out.w[q][g] = x[q];
                if (PAY) out.p[g] = recp[PAY ? j : 0];
            }
        }
        __syncthreads();
        if (tid < 256) gbase[tid] += tcnt[tid];
        __syncthreads();
    }
}

// records per workgroup: RS_CHUNK for big inputs; smaller ones are spread over ~4 workgroups per CU's worth of chunks
// (whole tiles), or a 1 M-row sort would run on 16 of the 256 CUs
template <int NW>
u64 sort_chunk(u64 n)
{
    u64 chunk = RS_CHUNK;
    const u64 tile = (u64)Tile<NW>::N;
    const u64 want = (n + 1023) / 1024;                                // ~1024 chunks
    const u64 c = (want + tile - 1) / tile * tile;
    if (c < chunk) chunk = c < tile ? tile : c;
    return chunk;
}

template <int NW, bool PAY>
ivx_status sort_impl(ivx_ctx *ctx, u64 *const *a, u64 *const *b, u64 n, const ivx_sort_field *fields, int nfields, int *in_b, bool tight,
                     u32 *const *pay, bool first_hist)
{
    *in_b = 0;
    if (n <= 1) return IVX_OK;
    if (n >= 0xFFFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "sort: more than 2^32-1 records");
    hipStream_t st = ctx->stream;
    // which bits vary at all (per sorted word)
    // (tight: the caller packed the fields from the columns' value ranges, every digit varies -- no need to look)
    unsigned long long *d_var = (unsigned long long *)(ctx->d_scalars + 16);
    if (tight) {
        for (int f = 0; f < nfields; f++) ctx->h_scalars[16 + f] = ~0ull;
    } else {
        IVX_HIP(ctx, hipMemsetAsync(d_var, 0, 8 * sizeof(u64), st));
        for (int f = 0; f < nfields; f++)
            hipLaunchKernelGGL(k_varbits, dim3(ivx_stream_grid(n, RS_HT * 16, 1024)), dim3(RS_HT), 0, st, (const u64 *)a[fields[f].word], n, d_var + f);
        IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 16, d_var, 8 * sizeof(u64), hipMemcpyDeviceToHost, st));
        IVX_HIP(ctx, hipStreamSynchronize(st));
    }

    const u64 chunk = sort_chunk<NW>(n);
    const u32 nblk = (u32)((n + chunk - 1) / chunk);
    u32 *hist;
    IVX_TRY(ctx->get_scratch(WS_SORTHIST, (size_t)256 * nblk * sizeof(u32), (void **)&hist));
    int cur = 0;
    for (int f = 0; f < nfields; f++) {
        const u64 var = ctx->h_scalars[16 + f];
        for (int sh = fields[f].lo; sh < fields[f].hi; sh += 8) {
            const u64 window = (sh + 8 >= 64 ? ~0ull : ((1ull << (sh + 8)) - 1)) & ~((1ull << sh) - 1);
            u64 fieldmask = fields[f].hi >= 64 ? ~0ull : ((1ull << fields[f].hi) - 1);
            if (((var & window) & fieldmask) == 0) continue;          // digit constant over the input
            u64 *const *src = cur ? b : a;
            u64 *const *dst = cur ? a : b;
            CPtrs<NW> ci; Ptrs<NW> po;
            for (int q = 0; q < NW; q++) { ci.w[q] = src[q]; po.w[q] = dst[q]; }
            ci.p = PAY ? pay[cur] : nullptr; po.p = PAY ? pay[cur ^ 1] : nullptr;
            if (!first_hist) hipLaunchKernelGGL(k_hist, dim3(nblk), dim3(RS_HT), 0, st, (const u64 *)src[fields[f].word], n, sh, nblk, hist, (u32)chunk);
            first_hist = false;
            IVX_TRY(ivx_scan_exclusive_u32(ctx, hist, (u64)256 * nblk));
            hipLaunchKernelGGL((k_scatter<NW, PAY>), dim3(nblk), dim3(RS_T), 0, st, ci, po, n, fields[f].word, sh, nblk, (const u32 *)hist, (u32)chunk);
            cur ^= 1;
        }
    }
    IVX_HIP(ctx, hipGetLastError());
    *in_b = cur;
    return IVX_OK;
}

}  // namespace

void ivx_sort_geometry1(u64 n, u64 *chunk, u32 *nblk)
{
    *chunk = sort_chunk<1>(n);
    *nblk = (u32)((n + *chunk - 1) / *chunk);
}

ivx_status ivx_radix_sort(ivx_ctx *ctx, int nw, u64 *const *a, u64 *const *b, u64 n,
                          const ivx_sort_field *fields, int nfields, int *in_b, bool tight, u32 *const *pay, bool first_hist)
{
    if (pay && nw != 1) return ctx->fail(IVX_ERR_INVALID, "sort: a 32-bit payload goes with one-word records");
    if (first_hist && (nw != 1 || !tight)) return ctx->fail(IVX_ERR_INVALID, "sort: a prepared first histogram goes with tight one-word sorts");
    switch (nw) {
    case 1: return pay ? sort_impl<1, true>(ctx, a, b, n, fields, nfields, in_b, tight, pay, first_hist) : sort_impl<1, false>(ctx, a, b, n, fields, nfields, in_b, tight, pay, first_hist);
    case 2: return sort_impl<2, false>(ctx, a, b, n, fields, nfields, in_b, tight, pay, false);
    case 3: return sort_impl<3, false>(ctx, a, b, n, fields, nfields, in_b, tight, pay, false);
    default: return ctx->fail(IVX_ERR_INVALID, "sort: unsupported record width");
    }
}

// test hook (not part of include/ivx.h): sort (w0[,w1]) device or host arrays by the given fields
extern "C" ivx_status ivx_debug_sort(ivx_ctx *ctx, int nw, u64 *w0, u64 *w1, u64 *w2, u64 n,
                                     const int *field_words, const int *field_lo, const int *field_hi, int nfields)
{
    if (!ctx || nw < 1 || nw > 3 || nfields > 8) return IVX_ERR_INVALID;
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    u64 *host[3] = {w0, w1, w2};
    u64 *a[3], *b[3];
    for (int q = 0; q < nw; q++) {
        IVX_TRY(ctx->get_scratch(WS_SA0 + q, n * sizeof(u64), (void **)&a[q]));
        IVX_TRY(ctx->get_scratch(WS_SB0 + q, n * sizeof(u64), (void **)&b[q]));
        IVX_HIP(ctx, hipMemcpyAsync(a[q], host[q], n * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    }
    ivx_sort_field f[8];
    for (int i = 0; i < nfields; i++) { f[i].word = field_words[i]; f[i].lo = field_lo[i]; f[i].hi = field_hi[i]; }
    int in_b = 0;
    IVX_TRY(ivx_radix_sort(ctx, nw, a, b, n, f, nfields, &in_b));
    for (int q = 0; q < nw; q++)
        IVX_HIP(ctx, hipMemcpyAsync(host[q], in_b ? b[q] : a[q], n * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

// test hook: one-word records with a 32-bit payload (pay may be NULL), fields at any bit position, `tight` as the library passes it
extern "C" ivx_status ivx_debug_sort_pay(ivx_ctx *ctx, u64 *w0, u32 *pay, u64 n, int lo, int hi, int tight)
{
    if (!ctx || !w0) return IVX_ERR_INVALID;
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    u64 *a[1], *b[1]; u32 *p[2] = {nullptr, nullptr};
    IVX_TRY(ctx->get_scratch(WS_SA0, n * sizeof(u64), (void **)&a[0]));
    IVX_TRY(ctx->get_scratch(WS_SB0, n * sizeof(u64), (void **)&b[0]));
    IVX_HIP(ctx, hipMemcpyAsync(a[0], w0, n * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    if (pay) {
        IVX_TRY(ctx->get_scratch(WS_SA1, n * sizeof(u32), (void **)&p[0]));
        IVX_TRY(ctx->get_scratch(WS_SB1, n * sizeof(u32), (void **)&p[1]));
        IVX_HIP(ctx, hipMemcpyAsync(p[0], pay, n * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
    }
    const ivx_sort_field f[1] = {{0, lo, hi}};
    int in_b = 0;
    IVX_TRY(ivx_radix_sort(ctx, 1, a, b, n, f, 1, &in_b, tight != 0, pay ? p : nullptr));
    IVX_HIP(ctx, hipMemcpyAsync(w0, in_b ? b[0] : a[0], n * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    if (pay) IVX_HIP(ctx, hipMemcpyAsync(pay, p[in_b], n * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}
