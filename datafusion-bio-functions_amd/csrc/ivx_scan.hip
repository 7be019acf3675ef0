// ivx_scan.hip -- device-wide exclusive prefix sum (reduce / recurse / apply).
// HBM-bound: reads the array twice and writes it once; tiles of 2048 elements
// per 256-thread workgroup, 8 contiguous elements per thread.
#include "ivx_device.hpp"

namespace {

constexpr int SC_T = 256;
constexpr int SC_I = 8;
constexpr int SC_TILE = SC_T * SC_I;

template <typename T>
__global__ __launch_bounds__(SC_T) void k_scan_reduce(const T *__restrict__ in, u64 n, T *__restrict__ sums)
{
    __shared__ T lds[SC_T / IVX_WAVE + 1];
    const u64 base = (u64)blockIdx.x * SC_TILE + (u64)threadIdx.x * SC_I;
    T s = 0;
#pragma unroll
    for (int i = 0; i < SC_I; i++)
        if (base + i < n) s += in[base + i];
    T tot = block_sum<T, SC_T>(s, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

template <typename T>
__global__ __launch_bounds__(SC_T) void k_scan_apply(T *__restrict__ data, u64 n, const T *__restrict__ offs)
{
    __shared__ T lds[SC_T / IVX_WAVE + 1];
    const u64 base = (u64)blockIdx.x * SC_TILE + (u64)threadIdx.x * SC_I;
    T v[SC_I];
    T s = 0;
#pragma unroll
    for (int i = 0; i < SC_I; i++) { v[i] = base + i < n ? data[base + i] : (T)0; s += v[i]; }
    T tot;
    T ex = block_excl_scan<T, SC_T>(s, lds, &tot) + (offs ? offs[blockIdx.x] : (T)0);
#pragma unroll
    for (int i = 0; i < SC_I; i++) {
        if (base + i < n) data[base + i] = ex;
        ex += v[i];
    }
}

// one workgroup walks the whole (small) array carrying the running total
template <typename T>
__global__ __launch_bounds__(SC_T) void k_scan_single(T *__restrict__ data, u64 n)
{
    __shared__ T lds[SC_T / IVX_WAVE + 1];
    T carry = 0;
    for (u64 t0 = 0; t0 < n; t0 += SC_TILE) {
        const u64 base = t0 + (u64)threadIdx.x * SC_I;
        T v[SC_I];
        T s = 0;
#pragma unroll
        for (int i = 0; i < SC_I; i++) { v[i] = base + i < n ? data[base + i] : (T)0; s += v[i]; }
        T tot;
        T ex = block_excl_scan<T, SC_T>(s, lds, &tot) + carry;
#pragma unroll
        for (int i = 0; i < SC_I; i++) {
            if (base + i < n) data[base + i] = ex;
            ex += v[i];
        }
        carry += tot;
    }
}

template <typename T>
ivx_status scan_rec(ivx_ctx *ctx, T *data, u64 n, int level)
{
    if (n == 0) return IVX_OK;
    if (n <= 8 * SC_TILE || level >= 2) {
        hipLaunchKernelGGL(k_scan_single<T>, dim3(1), dim3(SC_T), 0, ctx->stream, data, n);
        return IVX_OK;
    }
    const u64 nblk = (n + SC_TILE - 1) / SC_TILE;
    T *sums;
    IVX_TRY(ctx->get_scratch(WS_SCAN0 + level, nblk * sizeof(T), (void **)&sums));
    hipLaunchKernelGGL(k_scan_reduce<T>, dim3((u32)nblk), dim3(SC_T), 0, ctx->stream, data, n, sums);
    IVX_TRY(scan_rec<T>(ctx, sums, nblk, level + 1));
    hipLaunchKernelGGL(k_scan_apply<T>, dim3((u32)nblk), dim3(SC_T), 0, ctx->stream, data, n, sums);
    return IVX_OK;
}

}  // namespace

ivx_status ivx_scan_exclusive_u32(ivx_ctx *ctx, u32 *data, u64 n) { return scan_rec<u32>(ctx, data, n, 0); }
ivx_status ivx_scan_exclusive_u64(ivx_ctx *ctx, u64 *data, u64 n) { return scan_rec<u64>(ctx, data, n, 0); }
