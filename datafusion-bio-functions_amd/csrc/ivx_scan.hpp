// ivx_scan.hpp -- device-wide scan over an arbitrary associative operator
// (reduce / recurse / apply; HBM-bound: 2 reads + 1 write of the array).
//
//   struct Op { using T = ...; __host__ __device__ static T identity();
//               __device__ static T combine(const T& earlier, const T& later);
//               __device__ static T shfl_up(const T&, int delta); };
//
// The operator need not be commutative (interval-merge state functions are
// not): combine(a,b) always has `a` covering the lower indices.
#pragma once
#include "ivx_device.hpp"

namespace ivxscan {

constexpr int T_ = 256;
constexpr int I_ = 4;
constexpr int TILE_ = T_ * I_;

template <class Op>
__device__ __forceinline__ typename Op::T wave_incl(typename Op::T v)
{
    const u32 l = lane_id();
#pragma unroll
    for (int d = 1; d < IVX_WAVE; d <<= 1) {
        typename Op::T o = Op::shfl_up(v, d);
        if (l >= (u32)d) v = Op::combine(o, v);
    }
    return v;
}

// inclusive scan of one value per thread across the workgroup; lds needs T_/64+1 slots
template <class Op>
__device__ __forceinline__ typename Op::T block_incl(typename Op::T v, typename Op::T *lds, typename Op::T *total)
{
    using T = typename Op::T;
    constexpr int NW = T_ / IVX_WAVE;
    const u32 l = lane_id(), w = threadIdx.x / IVX_WAVE;
    T inc = wave_incl<Op>(v);
    if (l == IVX_WAVE - 1) lds[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = Op::identity();
        for (int i = 0; i < NW; i++) { T t = lds[i]; lds[i] = run; run = Op::combine(run, t); }
        lds[NW] = run;
    }
    __syncthreads();
    T res = Op::combine(lds[w], inc);
    *total = lds[NW];
    __syncthreads();
    return res;
}

template <class Op>
__global__ __launch_bounds__(T_) void k_reduce(const typename Op::T *__restrict__ in, u64 n, typename Op::T *__restrict__ sums)
{
    using T = typename Op::T;
    __shared__ T lds[T_ / IVX_WAVE + 1];
    const u64 base = (u64)blockIdx.x * TILE_ + (u64)threadIdx.x * I_;
    T s = Op::identity();
#pragma unroll
    for (int i = 0; i < I_; i++)
        if (base + i < n) s = Op::combine(s, in[base + i]);
    T tot;
    block_incl<Op>(s, lds, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// inclusive (INCL) or exclusive scan of the tile, seeded with offs[block] (exclusive prefix of earlier tiles)
template <class Op, bool INCL>
__global__ __launch_bounds__(T_) void k_apply(typename Op::T *__restrict__ data, u64 n, const typename Op::T *__restrict__ offs)
{
    using T = typename Op::T;
    __shared__ T lds[T_ / IVX_WAVE + 1];
    const u64 base = (u64)blockIdx.x * TILE_ + (u64)threadIdx.x * I_;
    T v[I_];
    T s = Op::identity();
#pragma unroll
    for (int i = 0; i < I_; i++) { v[i] = base + i < n ? data[base + i] : Op::identity(); s = Op::combine(s, v[i]); }
    T tot;
    T inc = block_incl<Op>(s, lds, &tot);
    // exclusive prefix of this thread = everything before its first item
    T run = offs ? offs[blockIdx.x] : Op::identity();
    {
        // inc includes this thread's items; rebuild the exclusive part from the wave/block pieces:
        // shuffle the inclusive value of the previous thread through LDS-free path
        T prev = Op::shfl_up(inc, 1);
        __shared__ T edge[T_ / IVX_WAVE];
        if (lane_id() == IVX_WAVE - 1) edge[threadIdx.x / IVX_WAVE] = inc;
        __syncthreads();
        if (threadIdx.x == 0) prev = Op::identity();
        else if (lane_id() == 0) prev = edge[threadIdx.x / IVX_WAVE - 1];
        run = Op::combine(run, prev);
    }
#pragma unroll
    for (int i = 0; i < I_; i++) {
        if (INCL) { run = Op::combine(run, v[i]); if (base + i < n) data[base + i] = run; }
        else { if (base + i < n) data[base + i] = run; run = Op::combine(run, v[i]); }
    }
}

template <class Op, bool INCL>
__global__ __launch_bounds__(T_) void k_single(typename Op::T *__restrict__ data, u64 n)
{
    using T = typename Op::T;
    __shared__ T lds[T_ / IVX_WAVE + 1];
    __shared__ T edge[T_ / IVX_WAVE];
    T carry = Op::identity();
    for (u64 t0 = 0; t0 < n; t0 += TILE_) {
        const u64 base = t0 + (u64)threadIdx.x * I_;
        T v[I_];
        T s = Op::identity();
#pragma unroll
        for (int i = 0; i < I_; i++) { v[i] = base + i < n ? data[base + i] : Op::identity(); s = Op::combine(s, v[i]); }
        T tot;
        T inc = block_incl<Op>(s, lds, &tot);
        T prev = Op::shfl_up(inc, 1);
        if (lane_id() == IVX_WAVE - 1) edge[threadIdx.x / IVX_WAVE] = inc;
        __syncthreads();
        if (threadIdx.x == 0) prev = Op::identity();
        else if (lane_id() == 0) prev = edge[threadIdx.x / IVX_WAVE - 1];
        T run = Op::combine(carry, prev);
#pragma unroll
        for (int i = 0; i < I_; i++) {
            if (INCL) { run = Op::combine(run, v[i]); if (base + i < n) data[base + i] = run; }
            else { if (base + i < n) data[base + i] = run; run = Op::combine(run, v[i]); }
        }
        carry = Op::combine(carry, tot);
        __syncthreads();
    }
}

template <class Op, bool INCL>
ivx_status scan_rec(ivx_ctx *ctx, typename Op::T *data, u64 n, int level, int slot0)
{
    using T = typename Op::T;
    if (n == 0) return IVX_OK;
    if (n <= 4 * TILE_ || level >= 2) {
        hipLaunchKernelGGL((k_single<Op, INCL>), dim3(1), dim3(T_), 0, ctx->stream, data, n);
        return IVX_OK;
    }
    const u64 nblk = (n + TILE_ - 1) / TILE_;
    T *sums;
    IVX_TRY(ctx->get_scratch(slot0 + level, nblk * sizeof(T), (void **)&sums));
    hipLaunchKernelGGL((k_reduce<Op>), dim3((u32)nblk), dim3(T_), 0, ctx->stream, (const T *)data, n, sums);
    IVX_TRY((scan_rec<Op, false>(ctx, sums, nblk, level + 1, slot0)));     // block offsets are always exclusive
    hipLaunchKernelGGL((k_apply<Op, INCL>), dim3((u32)nblk), dim3(T_), 0, ctx->stream, data, n, (const T *)sums);
    return IVX_OK;
}

// ---- the same scans over elements that are COMPUTED, not stored: `in(i)` yields element i (from whatever columns it is
// a function of) and `out(i, v)` takes the scanned value (and stores only what later passes need) -- no materialised
// element array, which for a 24-byte state is most of the traffic of the scan
template <class Op, class In>
__global__ __launch_bounds__(T_) void k_reduce_f(In in, u64 n, typename Op::T *__restrict__ sums)
{
    using T = typename Op::T;
    __shared__ T lds[T_ / IVX_WAVE + 1];
    const u64 base = (u64)blockIdx.x * TILE_ + (u64)threadIdx.x * I_;
    T s = Op::identity();
#pragma unroll
    for (int i = 0; i < I_; i++)
        if (base + i < n) s = Op::combine(s, in(base + i));
    T tot;
    block_incl<Op>(s, lds, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

template <class Op, class In, class Out>
__global__ __launch_bounds__(T_) void k_apply_f(In in, Out out, u64 n, const typename Op::T *__restrict__ offs)
{
    using T = typename Op::T;
    __shared__ T lds[T_ / IVX_WAVE + 1];
    __shared__ T edge[T_ / IVX_WAVE];
    const u64 base = (u64)blockIdx.x * TILE_ + (u64)threadIdx.x * I_;
    T v[I_];
    T s = Op::identity();
#pragma unroll
    for (int i = 0; i < I_; i++) { v[i] = base + i < n ? in(base + i) : Op::identity(); s = Op::combine(s, v[i]); }
    T tot;
    T inc = block_incl<Op>(s, lds, &tot);
    T run = offs ? offs[blockIdx.x] : Op::identity();
    T prev = Op::shfl_up(inc, 1);
    if (lane_id() == IVX_WAVE - 1) edge[threadIdx.x / IVX_WAVE] = inc;
    __syncthreads();
    if (threadIdx.x == 0) prev = Op::identity();
    else if (lane_id() == 0) prev = edge[threadIdx.x / IVX_WAVE - 1];
    run = Op::combine(run, prev);
#pragma unroll
    for (int i = 0; i < I_; i++) { run = Op::combine(run, v[i]); if (base + i < n) out(base + i, run); }
}

// inclusive scan of in(0..n) into out; the block partials take the stored-array path (they are few)
template <class Op, class In, class Out>
ivx_status inclusive_f(ivx_ctx *ctx, In in, Out out, u64 n)
{
    using T = typename Op::T;
    if (n == 0) return IVX_OK;
    const u64 nblk = (n + TILE_ - 1) / TILE_;
    T *sums;
    IVX_TRY(ctx->get_scratch(WS_SCAN0, nblk * sizeof(T), (void **)&sums));
    if (nblk > 1) {
        hipLaunchKernelGGL((k_reduce_f<Op, In>), dim3((u32)nblk), dim3(T_), 0, ctx->stream, in, n, sums);
        IVX_TRY((scan_rec<Op, false>(ctx, sums, nblk, 1, WS_SCAN0)));
    }
    hipLaunchKernelGGL((k_apply_f<Op, In, Out>), dim3((u32)nblk), dim3(T_), 0, ctx->stream, in, out, n, nblk > 1 ? (const T *)sums : (const T *)nullptr);
    return IVX_OK;
}

// in-place scans; scratch slots WS_SCAN0.. are used for the block partials
template <class Op>
ivx_status inclusive(ivx_ctx *ctx, typename Op::T *data, u64 n) { return scan_rec<Op, true>(ctx, data, n, 0, WS_SCAN0); }
template <class Op>
ivx_status exclusive(ivx_ctx *ctx, typename Op::T *data, u64 n) { return scan_rec<Op, false>(ctx, data, n, 0, WS_SCAN0); }

}  // namespace ivxscan
