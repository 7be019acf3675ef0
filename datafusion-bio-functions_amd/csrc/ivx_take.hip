// ivx_take.hip -- f3: `compute::take` of payload columns on the device, the step right after the probe
// (interval_join.rs:1655-1667 takes every projected build/probe column with left_idx / right_idx;
// nearest.rs:469-482 does the same with a nullable left index).
//
//   fixed-width columns: one thread per output element; the index stream and the output are read and
//     written coalesced, the source is a gather (payload columns of the BUILD side are small and stay
//     in L2 / Infinity Cache; probe-side indices ascend, so that gather streams).
//   Utf8 / LargeUtf8: lengths -> exclusive scan (= the output offsets) -> byte copy.  The copy is
//     wavefront-cooperative: a wavefront takes 64 rows, prefix-sums their lengths and its lanes then
//     walk the BYTES of those 64 strings, so output stores are consecutive bytes whatever the mix of
//     string lengths, and one long string is spread over all lanes.
//
// Null handling as arrow's take: a null index (IVX_NULL_IDX) or a null source slot gives a null output
// slot; the value bytes of a null-index slot are zero / the string is empty.
#include "ivx_device.hpp"
#include "ivx_scan.hpp"

namespace {

constexpr int TK = 256;

__device__ __forceinline__ bool bit_at(const u8 *bits, u64 i) { return (bits[i >> 3] >> (i & 7)) & 1u; }

template <typename V>
__global__ __launch_bounds__(TK) void k_take_fixed(const V *__restrict__ src, u64 n_src, const u8 *__restrict__ src_valid,
                                                   const u32 *__restrict__ idx, u64 n, V *__restrict__ out, u8 *__restrict__ out_valid, u32 *bad)
{
    for (u64 i = (u64)blockIdx.x * TK + threadIdx.x; i < n; i += (u64)gridDim.x * TK) {
        const u32 j = idx[i];
        V v{};
        bool ok = j != IVX_NULL_IDX;
        if (ok && j >= n_src) { *bad = 1; ok = false; }
        if (ok) v = src[j];
        out[i] = v;
        if (out_valid) out_valid[i] = (ok && (!src_valid || bit_at(src_valid, j))) ? 1 : 0;
    }
}

struct alignas(16) B32 { uint4 a, b; };

// the inverse of take: out[idx[i]] = src[i].  Puts a per-row column computed on a shard of the rows (idx = the rows'
// numbers in the whole job) back in input order; the value stream is read coalesced, the stores are a scatter.
template <typename V>
__global__ __launch_bounds__(TK) void k_scatter_fixed(const V *__restrict__ src, const u32 *__restrict__ idx, u64 n, V *__restrict__ out, u64 n_out, u32 *bad)
{
    for (u64 i = (u64)blockIdx.x * TK + threadIdx.x; i < n; i += (u64)gridDim.x * TK) {
        const u32 j = idx[i];
        if (j >= n_out) { *bad = 1; continue; }
        out[j] = src[i];
    }
}

// Boolean columns are bitmaps: one thread packs eight output bits into one byte
__global__ __launch_bounds__(TK) void k_take_bits(const u8 *__restrict__ src, u64 n_src, const u8 *__restrict__ src_valid,
                                                  const u32 *__restrict__ idx, u64 n, u8 *__restrict__ out, u8 *__restrict__ out_valid, u32 *bad)
{
    const u64 nbytes = (n + 7) / 8;
    for (u64 b = (u64)blockIdx.x * TK + threadIdx.x; b < nbytes; b += (u64)gridDim.x * TK) {
        u32 byte = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const u64 i = b * 8 + t;
            if (i >= n) break;
            const u32 j = idx[i];
            bool ok = j != IVX_NULL_IDX;
            if (ok && j >= n_src) { *bad = 1; ok = false; }
            if (ok && bit_at(src, j)) byte |= 1u << t;
            if (out_valid) out_valid[i] = (ok && (!src_valid || bit_at(src_valid, j))) ? 1 : 0;
        }
        out[b] = (u8)byte;
    }
}

template <typename O>
__global__ __launch_bounds__(TK) void k_take_len(const O *__restrict__ off, u64 n_src, const u8 *__restrict__ src_valid,
                                                 const u32 *__restrict__ idx, u64 n, u64 *__restrict__ len, u8 *__restrict__ out_valid, u32 *bad)
{
    for (u64 i = (u64)blockIdx.x * TK + threadIdx.x; i <= n; i += (u64)gridDim.x * TK) {
        if (i == n) { len[i] = 0; continue; }
        const u32 j = idx[i];
        bool ok = j != IVX_NULL_IDX;
        if (ok && j >= n_src) { *bad = 1; ok = false; }
        len[i] = ok ? (u64)(off[j + 1] - off[j]) : 0;
        if (out_valid) out_valid[i] = (ok && (!src_valid || bit_at(src_valid, j))) ? 1 : 0;
    }
}

template <typename O>
__global__ __launch_bounds__(TK) void k_take_offsets(const u64 *__restrict__ pos, u64 n, O *__restrict__ out_off)
{
    for (u64 i = (u64)blockIdx.x * TK + threadIdx.x; i <= n; i += (u64)gridDim.x * TK) out_off[i] = (O)pos[i];
}

template <typename O>
__global__ __launch_bounds__(TK) void k_take_bytes(const O *__restrict__ off, const u8 *__restrict__ data, const u32 *__restrict__ idx, u64 n,
                                                   const u64 *__restrict__ pos, u8 *__restrict__ out)
{
    __shared__ u64 s_ex[TK / IVX_WAVE][IVX_WAVE];
    __shared__ u64 s_src[TK / IVX_WAVE][IVX_WAVE];
    const u32 wv = threadIdx.x / IVX_WAVE, ln = lane_id();
    const u64 ngroups = (n + IVX_WAVE - 1) / IVX_WAVE;
    for (u64 g = (u64)blockIdx.x * (TK / IVX_WAVE) + wv; g < ngroups; g += (u64)gridDim.x * (TK / IVX_WAVE)) {
        const u64 i = g * IVX_WAVE + ln;
        const u64 p0 = pos[g * IVX_WAVE];                               // output byte position of the group
        const u64 hi_row = (g + 1) * IVX_WAVE < n ? (g + 1) * IVX_WAVE : n;
        const u64 total = pos[hi_row] - p0;                             // bytes of the group's strings
        u64 so = 0, ex = ~0ull;
        if (i < n) {
            const u32 j = idx[i];
            if (j != IVX_NULL_IDX) so = (u64)off[j];
            ex = pos[i] - p0;
        }
        s_ex[wv][ln] = ex; s_src[wv][ln] = so;
        __builtin_amdgcn_wave_barrier();
        for (u64 t = ln; t < total; t += IVX_WAVE) {
            u32 lo = 0, hi = IVX_WAVE - 1;                              // last row whose first byte is <= t
            while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (s_ex[wv][mid] <= t) lo = mid; else hi = mid - 1; }
            out[p0 + t] = data[s_src[wv][lo] + (t - s_ex[wv][lo])];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- Utf8View / BinaryView: 16-byte views {len, prefix | inline bytes, buffer index, offset}.  Strings of at
// most 12 bytes live inside the view; longer ones are gathered into ONE new data buffer (the output never
// refers to the source's buffers) and their views are rewritten to (buffer 0, new offset).
struct View16 { u32 len; u32 w1; u32 w2; u32 w3; };

__global__ __launch_bounds__(TK) void k_view_len(const View16 *__restrict__ views, u64 n_src, const u8 *__restrict__ src_valid,
                                                 const u32 *__restrict__ idx, u64 n, u64 *__restrict__ len, u8 *__restrict__ out_valid, u32 *bad)
{
    for (u64 i = (u64)blockIdx.x * TK + threadIdx.x; i <= n; i += (u64)gridDim.x * TK) {
        if (i == n) { len[i] = 0; continue; }
        const u32 j = idx[i];
        bool ok = j != IVX_NULL_IDX;
        if (ok && j >= n_src) { *bad = 1; ok = false; }
        const u32 l = ok ? views[j].len : 0u;
        len[i] = l > 12 ? l : 0;                                     // bytes this row needs in the new data buffer
        if (out_valid) out_valid[i] = (ok && (!src_valid || bit_at(src_valid, j))) ? 1 : 0;
    }
}

__global__ __launch_bounds__(TK) void k_view_copy(const View16 *__restrict__ views, const u8 *const *__restrict__ bufs, const u32 *__restrict__ idx, u64 n,
                                                  const u64 *__restrict__ pos, View16 *__restrict__ out_views, u8 *__restrict__ out_data)
{
    __shared__ u64 s_ex[TK / IVX_WAVE][IVX_WAVE];
    __shared__ const u8 *s_src[TK / IVX_WAVE][IVX_WAVE];
    const u32 wv = threadIdx.x / IVX_WAVE, ln = lane_id();
    const u64 ngroups = (n + IVX_WAVE - 1) / IVX_WAVE;
    for (u64 g = (u64)blockIdx.x * (TK / IVX_WAVE) + wv; g < ngroups; g += (u64)gridDim.x * (TK / IVX_WAVE)) {
        const u64 i = g * IVX_WAVE + ln;
        const u64 p0 = pos[g * IVX_WAVE];
        const u64 hi_row = (g + 1) * IVX_WAVE < n ? (g + 1) * IVX_WAVE : n;
        const u64 total = pos[hi_row] - p0;
        u64 ex = ~0ull; const u8 *src = nullptr;
        if (i < n) {
            const u32 j = idx[i];
            View16 v{0, 0, 0, 0};
            if (j != IVX_NULL_IDX) v = views[j];
            ex = pos[i] - p0;
            if (v.len > 12) { src = bufs[v.w2] + v.w3; v.w2 = 0; v.w3 = (u32)pos[i]; }   // (the prefix word stays)
            out_views[i] = v;
        }
        s_ex[wv][ln] = ex; s_src[wv][ln] = src;
        __builtin_amdgcn_wave_barrier();
        for (u64 t = ln; t < total; t += IVX_WAVE) {
            u32 lo = 0, hi = IVX_WAVE - 1;
            while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (s_ex[wv][mid] <= t) lo = mid; else hi = mid - 1; }
            out_data[p0 + t] = s_src[wv][lo][t - s_ex[wv][lo]];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

u32 take_grid(u64 n) { return ivx_stream_grid(n, TK * 4, 256 * 16); }

ivx_status take_flag(ivx_ctx *ctx, const char *what = "take: index out of bounds")
{
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 8, ctx->d_scalars + 8, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*(u32 *)(ctx->h_scalars + 8)) return ctx->fail(IVX_ERR_INVALID, what);
    return IVX_OK;
}

}  // namespace

ivx_status ivx_take_fixed_device(ivx_ctx *ctx, const void *src, u32 width, u64 n_src, const u8 *src_valid,
                                 const u32 *idx, u64 n, void *out, u8 *out_valid)
{
    hipStream_t st = ctx->stream;
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(u64), st));
    u32 *bad = (u32 *)(ctx->d_scalars + 8);
    if (n) {
        const u32 grid = take_grid(n);
#define IVX_TAKE(V) hipLaunchKernelGGL(k_take_fixed<V>, dim3(grid), dim3(TK), 0, st, (const V *)src, n_src, src_valid, idx, n, (V *)out, out_valid, bad)
        switch (width) {
        case 1: IVX_TAKE(u8); break;
        case 2: IVX_TAKE(unsigned short); break;
        case 4: IVX_TAKE(u32); break;
        case 8: IVX_TAKE(u64); break;
        case 16: IVX_TAKE(uint4); break;
        case 32: IVX_TAKE(B32); break;
        default: return ctx->fail(IVX_ERR_UNSUPPORTED, "take: fixed width must be 1, 2, 4, 8, 16 or 32 bytes");
        }
#undef IVX_TAKE
        IVX_HIP(ctx, hipGetLastError());
    }
    return take_flag(ctx);
}

ivx_status ivx_scatter_fixed_device(ivx_ctx *ctx, const void *src, u32 width, const u32 *idx, u64 n, void *out, u64 n_out)
{
    hipStream_t st = ctx->stream;
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(u64), st));
    u32 *bad = (u32 *)(ctx->d_scalars + 8);
    if (n) {
        const u32 grid = take_grid(n);
#define IVX_SCAT(V) hipLaunchKernelGGL(k_scatter_fixed<V>, dim3(grid), dim3(TK), 0, st, (const V *)src, idx, n, (V *)out, n_out, bad)
        switch (width) {
        case 1: IVX_SCAT(u8); break;
        case 2: IVX_SCAT(unsigned short); break;
        case 4: IVX_SCAT(u32); break;
        case 8: IVX_SCAT(u64); break;
        case 16: IVX_SCAT(uint4); break;
        case 32: IVX_SCAT(B32); break;
        default: return ctx->fail(IVX_ERR_UNSUPPORTED, "scatter: fixed width must be 1, 2, 4, 8, 16 or 32 bytes");
        }
#undef IVX_SCAT
        IVX_HIP(ctx, hipGetLastError());
    }
    return take_flag(ctx, "scatter: index out of bounds");
}

ivx_status ivx_take_bits_device(ivx_ctx *ctx, const u8 *src_bits, u64 n_src, const u8 *src_valid, const u32 *idx, u64 n, u8 *out_bits, u8 *out_valid)
{
    hipStream_t st = ctx->stream;
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(u64), st));
    if (n) hipLaunchKernelGGL(k_take_bits, dim3(take_grid((n + 7) / 8)), dim3(TK), 0, st, src_bits, n_src, src_valid, idx, n, out_bits, out_valid, (u32 *)(ctx->d_scalars + 8));
    IVX_HIP(ctx, hipGetLastError());
    return take_flag(ctx);
}

// large != 0: 64-bit offsets (LargeUtf8 / LargeBinary).  out_offsets (n+1) is always written; the bytes only
// when out_data is given and data_cap suffices (else IVX_ERR_CAPACITY with *data_bytes set).
ivx_status ivx_take_utf8_device(ivx_ctx *ctx, int large, const void *offsets, const u8 *data, u64 n_src, const u8 *src_valid,
                                const u32 *idx, u64 n, void *out_offsets, u8 *out_data, u64 data_cap, u64 *data_bytes, u8 *out_valid)
{
    hipStream_t st = ctx->stream;
    *data_bytes = 0;
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(u64), st));
    u32 *bad = (u32 *)(ctx->d_scalars + 8);
    u64 *pos;
    IVX_TRY(ctx->get_scratch(WS_T0, (n + 1) * sizeof(u64), (void **)&pos));
    const u32 grid = take_grid(n + 1);
    if (large) hipLaunchKernelGGL(k_take_len<i64>, dim3(grid), dim3(TK), 0, st, (const i64 *)offsets, n_src, src_valid, idx, n, pos, out_valid, bad);
    else hipLaunchKernelGGL(k_take_len<i32>, dim3(grid), dim3(TK), 0, st, (const i32 *)offsets, n_src, src_valid, idx, n, pos, out_valid, bad);
    IVX_TRY(ivx_scan_exclusive_u64(ctx, pos, n + 1));
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 5, pos + n, sizeof(u64), hipMemcpyDeviceToHost, st));
    IVX_TRY(take_flag(ctx));                                                    // synchronises
    const u64 total = ctx->h_scalars[5];
    *data_bytes = total;
    if (!large && total > 0x7FFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "take: Utf8 offsets overflow i32 (use LargeUtf8)");
    if (out_offsets) {
        if (large) hipLaunchKernelGGL(k_take_offsets<i64>, dim3(grid), dim3(TK), 0, st, (const u64 *)pos, n, (i64 *)out_offsets);
        else hipLaunchKernelGGL(k_take_offsets<i32>, dim3(grid), dim3(TK), 0, st, (const u64 *)pos, n, (i32 *)out_offsets);
    }
    if (!out_data) return IVX_OK;                                              // sizing call
    if (total > data_cap) return ctx->fail(IVX_ERR_CAPACITY, "take: string data buffer too small");
    if (total && n) {
        const u64 ngroups = (n + IVX_WAVE - 1) / IVX_WAVE;
        const u32 g2 = ivx_stream_grid(ngroups, TK / IVX_WAVE, 256 * 16);
        if (large) hipLaunchKernelGGL(k_take_bytes<i64>, dim3(g2), dim3(TK), 0, st, (const i64 *)offsets, data, idx, n, (const u64 *)pos, out_data);
        else hipLaunchKernelGGL(k_take_bytes<i32>, dim3(g2), dim3(TK), 0, st, (const i32 *)offsets, data, idx, n, (const u64 *)pos, out_data);
    }
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

// views: [n_src] 16-byte views; bufs: device array of the n_bufs variadic data buffer pointers (device memory).
// out_views [n] always written when given; out_data = NULL sizes only (*data_bytes = bytes of the long strings).
ivx_status ivx_take_view_device(ivx_ctx *ctx, const void *views, const u8 *const *bufs, u64 n_src, const u8 *src_valid,
                                const u32 *idx, u64 n, void *out_views, u8 *out_data, u64 data_cap, u64 *data_bytes, u8 *out_valid)
{
    hipStream_t st = ctx->stream;
    *data_bytes = 0;
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars + 8, 0, sizeof(u64), st));
    u32 *bad = (u32 *)(ctx->d_scalars + 8);
    u64 *pos;
    IVX_TRY(ctx->get_scratch(WS_T0, (n + 1) * sizeof(u64), (void **)&pos));
    const u32 grid = take_grid(n + 1);
    hipLaunchKernelGGL(k_view_len, dim3(grid), dim3(TK), 0, st, (const View16 *)views, n_src, src_valid, idx, n, pos, out_valid, bad);
    IVX_TRY(ivx_scan_exclusive_u64(ctx, pos, n + 1));
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 5, pos + n, sizeof(u64), hipMemcpyDeviceToHost, st));
    IVX_TRY(take_flag(ctx));
    const u64 total = ctx->h_scalars[5];
    *data_bytes = total;
    if (total > 0x7FFFFFFFull) return ctx->fail(IVX_ERR_UNSUPPORTED, "take: gathered view data exceeds one 2 GiB buffer");
    if (!out_views || (total && !out_data)) return IVX_OK;                      // sizing call
    if (total > data_cap) return ctx->fail(IVX_ERR_CAPACITY, "take: string data buffer too small");
    if (n) {
        const u64 ngroups = (n + IVX_WAVE - 1) / IVX_WAVE;
        hipLaunchKernelGGL(k_view_copy, dim3(ivx_stream_grid(ngroups, TK / IVX_WAVE, 256 * 16)), dim3(TK), 0, st, (const View16 *)views, bufs, idx, n,
                           (const u64 *)pos, (View16 *)out_views, out_data);
    }
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}
