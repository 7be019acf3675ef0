// ivx_device.hpp -- wave64 / workgroup primitives for gfx950 (CDNA4).
// Wavefronts are 64 lanes: every cross-lane idiom below is written for 64.
#pragma once
#include "ivx_internal.hpp"

__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// base + number of set bits of a 64-bit ballot below this lane (the bit counter takes an addend: no separate add)
__device__ __forceinline__ u32 mask_rank_from(u64 mask, u32 base)
{
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, base));
}

// number of set bits of a 64-bit ballot below this lane
__device__ __forceinline__ u32 mask_rank(u64 mask)
{
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

// 32-bit wavefront scans on the DPP path (row shifts, then the row broadcasts of gfx9): no LDS round trip per step as with
// ds_bpermute.  Lanes without a source take `old` = 0, the identity of both operators used here.
template <int CTRL, int ROWS>
__device__ __forceinline__ u32 dpp0(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false); }
__device__ __forceinline__ u32 wave_incl_max32(u32 v)
{
    u32 o;
    o = dpp0<0x111, 0xf>(v); v = o > v ? o : v;                          // row_shr:1, 2, 4, 8
    o = dpp0<0x112, 0xf>(v); v = o > v ? o : v;
    o = dpp0<0x114, 0xf>(v); v = o > v ? o : v;
    o = dpp0<0x118, 0xf>(v); v = o > v ? o : v;
    o = dpp0<0x142, 0xa>(v); v = o > v ? o : v;                          // row_bcast:15 into rows 1 and 3
    o = dpp0<0x143, 0xc>(v); v = o > v ? o : v;                          // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ u32 wave_incl_sum32(u32 v)
{
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ u32 wave_prev32(u32 v) { return dpp0<0x138, 0xf>(v); }      // wave_shr:1 (lane 0: 0)
__device__ __forceinline__ u32 wave_next32(u32 v) { return dpp0<0x130, 0xf>(v); }      // wave_shl:1 (lane 63: 0)

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v)
{
    const u32 l = lane_id();
#pragma unroll
    for (int d = 1; d < IVX_WAVE; d <<= 1) {
        T o = __shfl_up(v, d, IVX_WAVE);
        if (l >= (u32)d) v += o;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int d = IVX_WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, IVX_WAVE);
    return v;
}

// Workgroup exclusive scan of one value per thread.  NT threads (multiple of 64).
// `lds` needs NT/64 + 1 elements.  Returns the exclusive prefix; *total gets the sum.
template <typename T, int NT>
__device__ __forceinline__ T block_excl_scan(T v, T *lds, T *total)
{
    constexpr int NW = NT / IVX_WAVE;
    const u32 l = lane_id();
    const u32 w = threadIdx.x / IVX_WAVE;
    T inc = wave_incl_scan(v);
    if (l == IVX_WAVE - 1) lds[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) { T t = lds[i]; lds[i] = run; run += t; }
        lds[NW] = run;
    }
    __syncthreads();
    T res = inc - v + lds[w];
    *total = lds[NW];
    __syncthreads();        // lds may be reused by the caller right away
    return res;
}

template <typename T, int NT>
__device__ __forceinline__ T block_sum(T v, T *lds)
{
    constexpr int NW = NT / IVX_WAVE;
    T s = wave_sum(v);
    if (lane_id() == 0) lds[threadIdx.x / IVX_WAVE] = s;
    __syncthreads();
    T r = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) r += lds[i];
    __syncthreads();
    return r;
}

// grid size for a grid-stride streaming kernel: enough workgroups to fill
// 256 CUs x 8 without a long tail (cdna_hip_programming.md Guideline 11)
static inline u32 ivx_stream_grid(u64 n, u32 per_block, u32 max_blocks = 2048)
{
    u64 b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (u32)b;
}
