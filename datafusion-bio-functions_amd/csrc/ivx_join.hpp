// ivx_join.hpp -- device helpers shared by the overlap-index kernels (ivx_join.hip, ivx_join_regions.hip).
#pragma once
#include "ivx_device.hpp"

__device__ __forceinline__ u32 level_of(i32 s, i32 e, u32 sh0, u32 nlev)
{
    i64 len = (i64)e - (i64)s;
    if (len <= 0) return 0;
    u32 bits = 64 - __clzll((u64)len);                 // len < 2^bits
    u32 l = bits <= sh0 ? 0 : (bits - sh0 + IVX_LSTEP - 1) / IVX_LSTEP;
    return l < nlev ? l : nlev - 1;                    // top level has sh >= 32 >= bits
}

__device__ __forceinline__ u32 cell_of(const i32 *origin, const u32 *lbase, u32 nkeys, u32 k, i32 s, u32 l, u32 sh0)
{
    const u32 sh = sh0 + IVX_LSTEP * l;
    const u32 off = (u32)((i64)s - (i64)origin[k]);
    return lbase[(u64)l * nkeys + k] + (sh >= 32 ? 0u : off >> sh);
}

// visit every build row of key k overlapping [qs,qe] that sits in levels lev0..nlev-1: f(entry)
template <class F>
__device__ __forceinline__ void walk_ent(const JoinIndexView &ix, u32 sh0, u32 lev0, u32 nlev, u32 k, i32 qs, i32 qe, F &&f)
{
    if (k >= ix.nkeys) return;
    if (ix.kcnt[k] == 0) return;
    const i32 origin = ix.origin[k];
    const u32 span = ix.span[k];
    const i64 hi64 = (i64)qe - (i64)origin;
    if (hi64 < 0) return;                                 // every start of this key is > qe
    for (u32 l = lev0; l < nlev; l++) {
        if (ix.hdr[HDR_LEVCNT + l] == 0) continue;        // wave-uniform
        const u32 sh = sh0 + IVX_LSTEP * l;
        u32 blo = 0, bhi = 0;
        if (sh < 32) {
            const u32 ncell = (span >> sh) + 1u;
            const i64 lo64 = (i64)qs - ((i64)1 << sh) + 1 - (i64)origin;   // starts below this cannot reach qs
            const i64 bl = lo64 <= 0 ? 0 : (lo64 >> sh);
            const i64 bh = hi64 >> sh;
            if (bl >= (i64)ncell) continue;
            blo = (u32)bl;
            bhi = bh >= (i64)ncell ? ncell - 1u : (u32)bh;
            if (blo > bhi) continue;
        }
        const u32 base = ix.lbase[(u64)l * ix.nkeys + k];
        const u32 a = ix.binstart[base + blo], b = ix.binstart[base + bhi + 1];
        for (u32 j = a; j < b; j++) {
            const ivx_ent x = ix.ent[j];
            if (x.s <= qe && x.e >= qs) f(x);
        }
    }
}
// ... f(build row)
template <class F>
__device__ __forceinline__ void walk(const JoinIndexView &ix, u32 sh0, u32 lev0, u32 nlev, u32 k, i32 qs, i32 qe, F &&f)
{
    walk_ent(ix, sh0, lev0, nlev, k, qs, qe, [&](const ivx_ent &x) { f(x.row); });
}

