// ivx_grid.hip -- per-key statistics and the rank-grid build (see ivx_grid.hpp).
// Counting sort with atomics: histogram of cells, exclusive scan, scatter.
#include <cstdlib>
#include "ivx_grid.hpp"
#include "ivx_scan.hpp"

namespace {

constexpr int GT = 256;
constexpr u32 KEYS_IN_LDS = 2048;

// (+ up to three short word ranges of the caller's to zero: flags and headers that would each be a memset launch)
__global__ void k_init_keystats(i32 *kmin, i32 *kmax, u32 *kcnt, u32 nkeys, ivx_zero_ranges z)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nkeys) { kmin[i] = INT32_MAX; kmax[i] = INT32_MIN; kcnt[i] = 0; }
    if (blockIdx.x == 0)
        for (int r = 0; r < 3; r++)
            for (u32 j = threadIdx.x; j < z.n[r]; j += blockDim.x) z.p[r][j] = 0;
}

// per-key min / max of v and row counts, privatised in LDS; key ids >= nkeys raise *errflag.
// vend + lenhist (both nullable; vs == 1): also the histogram of the rows' length classes for the overlap
// index layout, lenhist[b] = rows whose (vend - v) has b significant bits (0 for vend <= v), b = 0..32
__global__ __launch_bounds__(GT) void k_keystats(const u32 *__restrict__ key, const i32 *__restrict__ v, u64 n,
                                                 u32 nkeys, i32 *kmin, i32 *kmax, u32 *kcnt, u32 *errflag, u32 vs,
                                                 const i32 *__restrict__ vend, u32 *lenhist)
{
    extern __shared__ i32 sh[];
    __shared__ u32 s_len[33];
    if (lenhist && threadIdx.x < 33) s_len[threadIdx.x] = 0;
    const bool priv = nkeys <= KEYS_IN_LDS;
    if (!priv && lenhist) __syncthreads();
    i32 *smin = sh, *smax = sh + nkeys;
    u32 *scnt = (u32 *)(sh + 2 * nkeys);
    if (priv) {
        for (u32 k = threadIdx.x; k < nkeys; k += GT) { smin[k] = INT32_MAX; smax[k] = INT32_MIN; scnt[k] = 0; }
        __syncthreads();
    }
    // a thread keeps the statistics of its current key in registers and hands them over when the key changes: rows
    // that come grouped by key (sorted tables) cost no atomics at all, random keys what they cost before
    u32 ck = 0xFFFFFFFFu, ccnt = 0; i32 cmin = INT32_MAX, cmax = INT32_MIN;
    u32 cl = 0xFFFFFFFFu, clcnt = 0;
    auto flush = [&]() {
        if (!ccnt) return;
        if (priv) { atomicMin(&smin[ck], cmin); atomicMax(&smax[ck], cmax); atomicAdd(&scnt[ck], ccnt); }
        else { atomicMin(&kmin[ck], cmin); atomicMax(&kmax[ck], cmax); atomicAdd(&kcnt[ck], ccnt); }
    };
    for (u64 i = (u64)blockIdx.x * GT + threadIdx.x; i < n; i += (u64)gridDim.x * GT) {
        const u32 k = key ? key[i] : 0u;
        if (k >= nkeys) { *errflag = 1; continue; }
        const i32 x = v[i * vs];
        if (lenhist) {
            const i64 len = (i64)vend[i] - (i64)x;
            const u32 c = len <= 0 ? 0u : 64u - (u32)__clzll((u64)len);
            if (c != cl) { if (clcnt) atomicAdd(&s_len[cl], clcnt); cl = c; clcnt = 0; }
            clcnt++;
        }
        if (k != ck) { flush(); ck = k; ccnt = 0; cmin = INT32_MAX; cmax = INT32_MIN; }
        cmin = x < cmin ? x : cmin; cmax = x > cmax ? x : cmax; ccnt++;
    }
    flush();
    if (lenhist && clcnt) atomicAdd(&s_len[cl], clcnt);
    if (priv) {
        __syncthreads();
        for (u32 k = threadIdx.x; k < nkeys; k += GT)
            if (scnt[k]) { atomicMin(&kmin[k], smin[k]); atomicMax(&kmax[k], smax[k]); atomicAdd(&kcnt[k], scnt[k]); }
    }
    if (lenhist) {
        __syncthreads();
        if (threadIdx.x < 33 && s_len[threadIdx.x]) atomicAdd(&lenhist[threadIdx.x], s_len[threadIdx.x]);
    }
}

// rows grouped by ascending key with non-decreasing v inside a key: a key's rows are found by bisecting the key
// column, its min / max are its first / last value (one thread per key)
__global__ __launch_bounds__(GT) void k_keystats_sorted(const u32 *__restrict__ key, const i32 *__restrict__ v, u64 n, u32 nkeys,
                                                        i32 *kmin, i32 *kmax, u32 *kcnt, u32 vs)
{
    const u32 k = blockIdx.x * GT + threadIdx.x;
    if (k >= nkeys) return;
    auto lower = [&](u32 x) {                       // first row whose key is >= x
        u64 a = 0, b = n;
        while (a < b) { const u64 m = (a + b) >> 1; if ((key ? key[m] : 0u) < x) a = m + 1; else b = m; }
        return a;
    };
    const u64 lo = lower(k), hi = k + 1 == 0u ? n : lower(k + 1);
    kcnt[k] = (u32)(hi - lo);
    kmin[k] = hi > lo ? v[lo * vs] : INT32_MAX;
    kmax[k] = hi > lo ? v[(hi - 1) * vs] : INT32_MIN;
}

__device__ __forceinline__ u32 gcells(u32 cnt, u32 span, u32 sh) { return cnt ? (span >> sh) + 1u : 0u; }

// one workgroup: origin/span, row offsets per key, cell width, first cell per key
__global__ __launch_bounds__(1024) void k_grid_layout(const i32 *kmin, const i32 *kmax, const u32 *kcnt, u32 nkeys, u64 n,
                                                      i32 *origin, u32 *span, u32 *koff, u32 *kbase, u32 *hdr, u64 budget)
{
    __shared__ u64 red[1024 / IVX_WAVE + 1];
    __shared__ u32 s_sh;
    const u32 t = threadIdx.x;
    for (u32 k = t; k < nkeys; k += 1024) {
        const u32 c = kcnt[k];
        origin[k] = c ? kmin[k] : 0;
        span[k] = c ? (u32)((i64)kmax[k] - (i64)kmin[k]) : 0u;
    }
    __syncthreads();
    u32 lo = 0, hi = 31;
    while (lo < hi) {
        const u32 mid = (lo + hi) / 2;
        u64 s = 0;
        for (u32 k = t; k < nkeys; k += 1024) s += gcells(kcnt[k], span[k], mid);
        const u64 tot = block_sum<u64, 1024>(s, red);
        if (tot <= budget) hi = mid; else lo = mid + 1;
    }
    if (t == 0) s_sh = lo;
    __syncthreads();
    const u32 sh = s_sh;
    u64 run_c = 0, run_r = 0;
    for (u32 k0 = 0; k0 < nkeys; k0 += 1024) {
        const u32 k = k0 + t;
        const u64 c = k < nkeys ? gcells(kcnt[k], span[k], sh) : 0u;
        const u64 r = k < nkeys ? kcnt[k] : 0u;
        u64 totc, totr;
        const u64 exc = block_excl_scan<u64, 1024>(c, red, &totc);
        const u64 exr = block_excl_scan<u64, 1024>(r, red, &totr);
        if (k < nkeys) { kbase[k] = (u32)(run_c + exc); koff[k] = (u32)(run_r + exr); }
        run_c += totc; run_r += totr;
    }
    if (t == 0) { koff[nkeys] = (u32)run_r; hdr[0] = sh; hdr[1] = (u32)run_c; }
}

__global__ __launch_bounds__(GT) void k_grid_count(const u32 *__restrict__ key, const i32 *__restrict__ v, u64 n, u32 nkeys,
                                                   const i32 *origin, const u32 *kbase, const u32 *hdr, u32 *bincnt)
{
    const u32 sh = hdr[0];
    for (u64 i = (u64)blockIdx.x * GT + threadIdx.x; i < n; i += (u64)gridDim.x * GT) {
        const u32 k = key ? key[i] : 0u;
        if (k >= nkeys) continue;
        const u32 c = kbase[k] + ((u32)((i64)v[i] - (i64)origin[k]) >> sh);
        atomicAdd(&bincnt[c], 1u);
    }
}

__global__ __launch_bounds__(GT) void k_grid_scatter(const u32 *__restrict__ key, const i32 *__restrict__ v, u64 n, u32 nkeys,
                                                     const i32 *origin, const u32 *kbase, const u32 *hdr,
                                                     const u32 *binstart, u32 *cursor, i32 *val)
{
    const u32 sh = hdr[0];
    for (u64 i = (u64)blockIdx.x * GT + threadIdx.x; i < n; i += (u64)gridDim.x * GT) {
        const u32 k = key ? key[i] : 0u;
        if (k >= nkeys) continue;
        const i32 x = v[i];
        const u32 c = kbase[k] + ((u32)((i64)x - (i64)origin[k]) >> sh);
        val[binstart[c] + atomicAdd(&cursor[c], 1u)] = x;
    }
}

// input already sorted by (key, value): the cell table needs no atomics, no scan and the values no copy.
// binstart[c] = rows in cells below c = (index of the last row of the nearest non-empty cell below c) + 1, so the row
// that ends its cell writes that number into every cell up to and including the next row's (cells ascend with the
// rows: keys ascend and their cell ranges are laid out in key order); the first row also writes the zeros before it,
// the last one the n's after it.  Gaps are a cell or two on average; a long one (an empty stretch of a contig) is
// filled by the whole wavefront.
__global__ __launch_bounds__(GT) void k_grid_bounds(const u32 *__restrict__ key, const i32 *__restrict__ v, u64 n, u32 nkeys,
                                                    const i32 *origin, const u32 *kbase, const u32 *hdr, u32 *binstart, u32 vs)
{
    const u32 sh = hdr[0], ncells = hdr[1];
    const u64 nround = (n + GT - 1) / GT * GT;                          // whole wavefronts stay in the loop together
    for (u64 i = (u64)blockIdx.x * GT + threadIdx.x; i < nround; i += (u64)gridDim.x * GT) {
        u32 lo = 1, hi = 0, val = 0;                                    // cells [lo, hi] <- val
        const u32 k = i < n ? (key ? key[i] : 0u) : 0xFFFFFFFFu;
        // the row's own cell; the NEXT row's cell comes from the lane above (every lane of the wavefront is in the loop),
        // only the last lane looks its successor up itself: half the loads and cell computations
        const u32 myc = k < nkeys ? kbase[k] + ((u32)((i64)v[i * vs] - (i64)origin[k]) >> sh) : ncells;
        u32 cn = wave_next32(myc);
        if (lane_id() == IVX_WAVE - 1) {
            cn = ncells;
            if (i + 1 < n) {
                const u32 k2 = key ? key[i + 1] : 0u;
                if (k2 < nkeys) cn = kbase[k2] + ((u32)((i64)v[(i + 1) * vs] - (i64)origin[k2]) >> sh);
            }
        }
        if (k < nkeys) {
            const u32 c = myc;
            if (i == 0) { for (u32 x = 0; x <= c; x++) binstart[x] = 0; }       // (a key's first cell holds its first row: c is small)
            lo = c + 1; hi = cn; val = (u32)(i + 1);
        }
        u64 big = __ballot(lo <= hi && hi - lo >= 64u);
        if (lo <= hi && hi - lo < 64u) for (u32 x = lo; x <= hi; x++) binstart[x] = val;
        while (big) {
            const int src = __builtin_ctzll(big);
            big &= big - 1;
            const u32 a = __shfl(lo, src, IVX_WAVE), b = __shfl(hi, src, IVX_WAVE), w = __shfl(val, src, IVX_WAVE);
            for (u64 x = (u64)a + lane_id(); x <= b; x += IVX_WAVE) binstart[x] = w;
        }
    }
}

}  // namespace

ivx_status ivx_keystats(ivx_ctx *ctx, const u32 *key, const i32 *v, u64 n, u32 nkeys,
                        i32 *kmin, i32 *kmax, u32 *kcnt, u32 *errflag)
{
    return ivx_keystats(ctx, key, v, n, nkeys, kmin, kmax, kcnt, errflag, 1u);
}

ivx_status ivx_keystats(ivx_ctx *ctx, const u32 *key, const i32 *v, u64 n, u32 nkeys,
                        i32 *kmin, i32 *kmax, u32 *kcnt, u32 *errflag, u32 vstride)
{
    return ivx_keystats_len(ctx, key, v, n, nkeys, kmin, kmax, kcnt, errflag, vstride, nullptr, nullptr);
}

ivx_status ivx_keystats_len(ivx_ctx *ctx, const u32 *key, const i32 *v, u64 n, u32 nkeys,
                            i32 *kmin, i32 *kmax, u32 *kcnt, u32 *errflag, u32 vstride, const i32 *vend, u32 *lenhist,
                            const ivx_zero_ranges *zero)
{
    hipStream_t st = ctx->stream;
    ivx_zero_ranges z{};
    if (zero) z = *zero;
    hipLaunchKernelGGL(k_init_keystats, dim3((nkeys + GT - 1) / GT), dim3(GT), 0, st, kmin, kmax, kcnt, nkeys, z);
    if (n) {
        const u32 grid = ivx_stream_grid(n, GT * 8, 1024);
        const size_t shm = nkeys <= KEYS_IN_LDS ? (size_t)nkeys * 12 : 0;
        hipLaunchKernelGGL(k_keystats, dim3(grid), dim3(GT), shm, st, key, v, n, nkeys, kmin, kmax, kcnt, errflag, vstride, vend, lenhist);
    }
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

ivx_status ivx_grid_build(ivx_ctx *ctx, ivx_index *ix, const u32 *key, const i32 *v, u64 n, u32 nkeys, RankGridView *out, bool sorted, u32 vstride)
{
    if (!sorted && vstride != 1) return ctx->fail(IVX_ERR_INVALID, "grid: strided values need a sorted column");
    hipStream_t st = ctx->stream;
    const u64 maxcells = 2 * n + nkeys + 64;
    if (maxcells + 1 >= 0xFFFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "column too large for 32-bit cell ids");
    i32 *origin, *val; u32 *span, *kcnt, *koff, *kbase, *binstart, *hdr;
    IVX_TRY(ivx_index_alloc(ctx, ix, nkeys * sizeof(i32), (void **)&origin));
    IVX_TRY(ivx_index_alloc(ctx, ix, nkeys * sizeof(u32), (void **)&span));
    IVX_TRY(ivx_index_alloc(ctx, ix, nkeys * sizeof(u32), (void **)&kcnt));
    IVX_TRY(ivx_index_alloc(ctx, ix, ((size_t)nkeys + 1) * sizeof(u32), (void **)&koff));
    IVX_TRY(ivx_index_alloc(ctx, ix, nkeys * sizeof(u32), (void **)&kbase));
    IVX_TRY(ivx_index_alloc(ctx, ix, (maxcells + 1) * sizeof(u32), (void **)&binstart));
    IVX_TRY(ivx_index_alloc(ctx, ix, 4 * sizeof(u32), (void **)&hdr));
    if (sorted) val = const_cast<i32 *>(v);                 // the caller's column (index memory) is the value array as it is
    else IVX_TRY(ivx_index_alloc(ctx, ix, (n ? n : 1) * sizeof(i32), (void **)&val));
    i32 *kmin, *kmax; u32 *cursor;
    IVX_TRY(ctx->get_scratch(WS_GRID0, nkeys * sizeof(i32), (void **)&kmin));
    IVX_TRY(ctx->get_scratch(WS_GRID1, nkeys * sizeof(i32), (void **)&kmax));
    cursor = nullptr;
    if (!sorted) IVX_TRY(ctx->get_scratch(WS_GRID2, (maxcells + 1) * sizeof(u32), (void **)&cursor));
    u32 *errflag = (u32 *)(ctx->d_scalars + 8);
    if (!sorted || !n) IVX_HIP(ctx, hipMemsetAsync(binstart, 0, (maxcells + 1) * sizeof(u32), st));
    if (!sorted) IVX_HIP(ctx, hipMemsetAsync(cursor, 0, (maxcells + 1) * sizeof(u32), st));
    if (sorted && n) hipLaunchKernelGGL(k_keystats_sorted, dim3((nkeys + GT - 1) / GT), dim3(GT), 0, st, key, v, n, nkeys, kmin, kmax, kcnt, vstride);
    else IVX_TRY(ivx_keystats(ctx, key, v, n, nkeys, kmin, kmax, kcnt, errflag, vstride));
    // cells: two per value (fewer, e.g. one per two values for the nearest index's grids over record fields, was measured:
    // the cell tables shrink but the cell scans grow, k=1 probe 3.10 -> 3.28 ms per 50M rows at one per two)
    const u64 budget = 2 * n + nkeys;
    hipLaunchKernelGGL(k_grid_layout, dim3(1), dim3(1024), 0, st, kmin, kmax, kcnt, nkeys, n, origin, span, koff, kbase, hdr, budget);
    if (n && sorted) {
        const u32 grid = ivx_stream_grid(n, GT * 2, 1u << 20);          // (a row's loads depend on nothing: many short threads hide their latency)
        hipLaunchKernelGGL(k_grid_bounds, dim3(grid), dim3(GT), 0, st, key, v, n, nkeys, origin, kbase, hdr, binstart, vstride);
    } else if (n) {
        const u32 grid = ivx_stream_grid(n, GT * 8, 1024);
        hipLaunchKernelGGL(k_grid_count, dim3(grid), dim3(GT), 0, st, key, v, n, nkeys, origin, kbase, hdr, binstart);
        IVX_TRY(ivx_scan_exclusive_u32(ctx, binstart, maxcells + 1));
        hipLaunchKernelGGL(k_grid_scatter, dim3(grid), dim3(GT), 0, st, key, v, n, nkeys, origin, kbase, hdr, binstart, cursor, val);
    }
    IVX_HIP(ctx, hipGetLastError());
    out->origin = origin; out->span = span; out->kcnt = kcnt; out->koff = koff; out->kbase = kbase;
    out->binstart = binstart; out->val = val; out->hdr = hdr; out->nkeys = nkeys;
    return IVX_OK;
}
