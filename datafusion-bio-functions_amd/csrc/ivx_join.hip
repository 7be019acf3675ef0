// ivx_join.hip -- binned overlap index (build) and the overlap-pairs probe.
//
// Replaces, for Algorithm::Coitrees, the reference's
//   build : update_hashmap + COITree::new per key   (interval_join.rs:745-763, :903-931)
//   probe : IntervalJoinAlgorithm::get + RLE expand (interval_join.rs:849-862, :1614-1653)
//
// Index layout in HBM (all integer/index work, HBM/L2-latency bound, no MFMA):
//   Build rows are classed by length into levels l = 0..L-1; level l holds the
//   rows with (end-start) < 2^(sh0 + IVX_LSTEP*l) and is cut into bins 2^(sh0 + IVX_LSTEP*l) wide
//   per key, so a row starts at most one bin before the bin its end falls in.
//   All (level,key,bin) cells are laid out in one CSR: binstart[] -> ent[]
//   (12-byte {start,end,row} entries grouped by cell by a counting sort with
//   atomics; order inside a cell is arbitrary -- the reference pins only the
//   pair multiset).  A probe row [qs,qe] reads, per non-empty level, the two
//   binstart words bracketing cells [bin(qs - 2^sh + 1) .. bin(qe)] and tests
//   the entries between them with the literal predicate
//   start <= qe && end >= qs.
//   sh0 is picked on the device: at least the shift at which the finest level
//   has ~2 cells per build row (per-key coordinate spans), and above that the
//   one with the smallest expected probe cost given the rows' length classes
//   (k_join_layout) -- level 0 is the level the region probe keeps in LDS.
#include <cstdlib>
#include <cstring>
#include "ivx_device.hpp"
#include "ivx_grid.hpp"
#include "ivx_join.hpp"

namespace {

constexpr int BT = 256;              // build kernels

// ------------------------------------------------------------------ build

__device__ __forceinline__ u32 cells_of(u32 cnt, u32 span, u32 sh) { return cnt ? (sh >= 32 ? 1u : (span >> sh) + 1u) : 0u; }

// One workgroup: turn per-key (min,max,count) into origin/span, pick sh0 and
// lay out the (level,key) cell ranges.  hdr: sh0, #levels, #cells.
__global__ __launch_bounds__(1024) void k_join_layout(const i32 *kmin, const i32 *kmax, const u32 *kcnt, u32 nkeys, u64 n,
                                                      i32 *origin, u32 *span, u32 *lbase, u32 *hdr, u64 maxcells,
                                                      u32 *kreg, u32 *rkey, const u32 *lenhist, u32 regmax, u32 *fbase, int filter_mode)
{
    __shared__ u64 red[1024 / IVX_WAVE + 1];
    __shared__ u32 s_sh0;
    const u32 t = threadIdx.x;
    for (u32 k = t; k < nkeys; k += 1024) {
        u32 c = kcnt[k];
        origin[k] = c ? kmin[k] : 0;
        span[k] = c ? (u32)((i64)kmax[k] - (i64)kmin[k]) : 0u;
    }
    __syncthreads();
    // Every candidate of a search is evaluated at once, one wavefront per candidate, instead of a bisection
    // with a workgroup reduction per step (this kernel is a serial 1-workgroup stage of every index build).
    __shared__ u64 s_tot[40];
    __shared__ u64 s_loff[IVX_MAXL + 1];
    const u32 wv = t / IVX_WAVE, ln = lane_id();
    constexpr u32 NWV = 1024 / IVX_WAVE;
    // ---- smallest shift whose finest level fits the budget (cells are monotone in sh)
    const u64 budget0 = 2 * n + nkeys;
    for (u32 sh = IVX_SH_MIN + wv; sh <= 31; sh += NWV) {
        u64 a = 0;
        for (u32 k = ln; k < nkeys; k += IVX_WAVE) a += cells_of(kcnt[k], span[k], sh);
        a = wave_sum(a);
        if (ln == 0) s_tot[sh] = a;
    }
    __syncthreads();
    __shared__ float s_occ0;                                           // level-0 rows per level-0 cell at the chosen shift
    __shared__ u64 s_cum[34];                                          // s_cum[b + 1] = build rows with at most b length bits
    __shared__ float s_cost[32], s_cocc[32];
    __shared__ u32 s_shb;
    if (t <= 32) s_cum[t + 1] = lenhist[t];
    __syncthreads();
    __shared__ u32 s_fg;
    if (t == 0) {
        u32 shb = IVX_SH_MIN;
        while (shb < 31 && s_tot[shb] > budget0) shb++;
        s_shb = shb;
        // ---- occupancy bitmap: the finest block width whose bitmap (blocks of every key's start span + one overflow
        //      block, padded to whole words per key) fits IVX_FBITS_MAX.  Looking a probe row up in it is a random
        //      8-byte gather (one lane per cycle and CU: ~0.3 ms per 100 M rows), about what routing the row costs, so
        //      it only pays when it rejects most rows: built when the build rows touch under 15 % of the blocks
        //      (measured on 100M x 1M rows over the human genome, 36 % of the blocks touched, 59 % of the rows rejected:
        //      partition 0.52 -> 0.86 ms, probe 0.66 -> 0.41 ms).  filter_mode: 0 never, 1 by that rule, 2 whenever it fits (tests)
        u32 fg = 0xFFFFFFFFu;
        if (filter_mode != 0 && (u64)nkeys * 34 < IVX_FBITS_MAX / 2) {
            u32 g = IVX_SH_MIN;
            while (g < 31 && s_tot[g] + (u64)nkeys * 34 > IVX_FBITS_MAX) g++;
            const double bits = (double)(s_tot[g] + nkeys);
            double touched = 0.0;                                       // blocks the build rows touch, by length class
            for (u32 b = 0; b <= 32; b++) touched += (double)s_cum[b + 1] * ((b > g ? (double)(3ull << (b - g)) * 0.25 : 0.0) + 1.5);
            if (n && s_tot[g] + (u64)nkeys * 34 <= IVX_FBITS_MAX && (filter_mode == 2 || touched < 0.15 * bits) && touched < 64.0 * (double)n + 1.0e6) fg = g;
        }
        s_fg = fg;
        u64 run = 0;
        s_cum[0] = 0;
        for (u32 b = 0; b <= 32; b++) { run += s_cum[b + 1]; s_cum[b + 1] = run; }
    }
    __syncthreads();
    // The budget gives the FINEST usable grid (shb).  When most build rows are longer than its cells they all land
    // in the upper levels, which a probe row reads with dependent global gathers, while level 0 is what the
    // region probe stages in LDS.  So among the shifts >= shb take the one with the smallest expected probe cost:
    // per non-empty level a visit plus ~2.2 cells' worth of candidates (a short query reaches into the cell before
    // its own), upper levels weighted as global reads.  One thread per candidate shift.
    if (t < 32) {
        const u32 sh = s_shb + t;
        float cost = 3.0e38f, occ0 = 0.f;
        if (sh <= 31) {
            cost = 0.f;
            u32 lo_bits = 0;                                            // smallest length-bit count not yet covered by a lower level
            for (u32 l = 0; l < IVX_MAXL; l++) {
                const u32 lsh = sh + IVX_LSTEP * l;
                const bool last = lsh >= 32 || l + 1 == IVX_MAXL;
                const u32 hi_bits = last ? 32u : lsh;
                const u64 rows = hi_bits >= lo_bits ? s_cum[hi_bits + 1] - s_cum[lo_bits] : 0;
                const float cells = lsh >= 32 ? (float)(nkeys ? nkeys : 1u) : (float)(s_tot[lsh] ? s_tot[lsh] : 1);
                const float occ = (float)rows / cells;
                if (l == 0) { occ0 = occ; cost += 1.f + 2.2f * occ; }
                else if (rows) cost += 8.f + 2.2f * occ * 4.f;
                if (last) break;
                lo_bits = lsh + 1;
            }
        }
        s_cost[t] = cost; s_cocc[t] = occ0;
    }
    __syncthreads();
    if (t == 0) {
        u32 bi = 0;
        for (u32 c = 1; c < 32; c++) if (s_cost[c] < s_cost[bi]) bi = c;
#ifdef IVX_SH_BIAS                                              // experiments: shift the choice by a fixed amount
        { const int b2 = (int)bi + (IVX_SH_BIAS); bi = b2 < 0 ? 0u : (s_shb + (u32)b2 > 31 ? 31 - s_shb : (u32)b2); }
#endif
        s_sh0 = s_shb + bi; s_occ0 = s_cocc[bi];
    }
    __syncthreads();
    const u32 sh0 = s_sh0;
    u32 nlev = 0;
    for (u32 l = 0; l < IVX_MAXL; l++) { nlev = l + 1; if (sh0 + IVX_LSTEP * l >= 32) break; }
    // ---- exclusive prefix over (level, key) cells: wavefront l scans level l, then the level offsets are added
    for (u32 l = wv; l < nlev; l += NWV) {
        const u32 sh = sh0 + IVX_LSTEP * l;
        u64 carry = 0;
        for (u32 k0 = 0; k0 < nkeys; k0 += IVX_WAVE) {
            const u32 k = k0 + ln;
            const u64 c = k < nkeys ? cells_of(kcnt[k], span[k], sh) : 0u;
            const u64 inc = wave_incl_scan(c);
            if (k < nkeys) lbase[(u64)l * nkeys + k] = (u32)(carry + inc - c);
            carry += __shfl(inc, IVX_WAVE - 1, IVX_WAVE);
        }
        if (ln == 0) s_tot[l] = carry;
    }
    __syncthreads();
    if (t == 0) {
        u64 run = 0;
        for (u32 l = 0; l < nlev; l++) { s_loff[l] = run; run += s_tot[l]; }
        hdr[HDR_SH0] = sh0;
        hdr[HDR_NLEV] = nlev;
        hdr[HDR_NBINS] = (u32)(run <= maxcells ? run : maxcells);   // never exceeds the budget by construction
    }
    __syncthreads();
    for (u32 l = 1 + wv; l < nlev; l += NWV) {
        const u32 add = (u32)s_loff[l];
        for (u32 k = ln; k < nkeys; k += IVX_WAVE) lbase[(u64)l * nkeys + k] += add;
    }
    // ---- probe regions: runs of 2^cs level-0 cells that never straddle a key, at most IVX_MAXREG of them
    //      (one radix digit of the probe partition pass); smallest cs that fits (monotone)
    for (u32 mid = wv; mid <= 32; mid += NWV) {
        u64 a = 0;
        for (u32 k = ln; k < nkeys; k += IVX_WAVE) {
            const u64 c = cells_of(kcnt[k], span[k], sh0);
            a += mid >= 32 ? (c ? 1u : 0u) : ((c + (1ull << mid) - 1) >> mid);
        }
        a = wave_sum(a);
        if (ln == 0) s_tot[mid] = a;
    }
    __syncthreads();
    u32 clo = 0;
    while (clo < 32 && s_tot[clo] > IVX_MAXREG) clo++;
    // regions wider than 2^IVX_REG_CS_MAX cells cannot be staged in LDS: take more, narrower regions instead
    // (the probe rows are then routed by a two-digit sort), as long as their number stays within IVX_MAXREG2
    if (clo > IVX_REG_CS_MAX && clo <= 32 && s_tot[IVX_REG_CS_MAX] <= IVX_MAXREG2) clo = IVX_REG_CS_MAX;
    // Cells per region R: 2^clo, unless an average region of that size holds more level-0 rows than the LDS slice
    // has room for (IVX_RP_ECAP).  Then R is whatever fills ~0.85 of the slice -- any integer, not a power of two
    // (a power of two would waste up to half of the capacity and double the number of regions, and 1023 regions
    // is where the one-pass routing of the probe rows ends): region = cell / R by multiplication with
    // M = ceil(2^40 / R), exact while cell * R < 2^40.
    u64 R = clo < 32 ? (1ull << clo) : 0;
    bool pow2 = true;
    if (clo < 32 && s_occ0 * (float)R > 0.9f * (float)IVX_RP_ECAP) {
        const float want = 0.85f * (float)IVX_RP_ECAP / s_occ0;
        R = want < 1.f ? 1ull : (u64)want;
        pow2 = false;
    }
    auto regions_for = [&](u64 r) -> u64 {                          // sum over keys of ceil(cells / r); workgroup-uniform
        u64 a = 0;
        for (u32 k = t; k < nkeys; k += 1024) { const u64 c = cells_of(kcnt[k], span[k], sh0); a += (c + r - 1) / r; }
        return block_sum<u64, 1024>(a, red);
    };
    if (R && nkeys > regmax && regions_for(1ull << 40) > regmax) R = 0; // more keys with rows than region slots: no region probe for this index
    if (R) {
        // (a power-of-two R straight from the search above is known to fit: s_tot[clo] regions)
        if (!(pow2 && clo < 32 && s_tot[clo] <= regmax)) while (regions_for(R) > regmax) { R *= 2; }   // the caller's tables hold regmax regions
        if (!pow2) {
            u64 mc = 0;
            for (u32 k = t; k < nkeys; k += 1024) { const u64 c = cells_of(kcnt[k], span[k], sh0); mc = c > mc ? c : mc; }
            __shared__ u64 s_mc;
            if (t == 0) s_mc = 0;
            __syncthreads();
            atomicMax((unsigned long long *)&s_mc, (unsigned long long)mc);
            __syncthreads();
            const u64 M = ((1ull << 40) + R - 1) / R;
            if (s_mc * R >= (1ull << 40) || s_mc > (1ull << 63) / M || (R & (R - 1)) == 0) {   // not exact / overflow (or R is a power of two after all)
                u64 p = 1; while (p * 2 <= R) p *= 2;
                R = p; pow2 = true;
                while (regions_for(R) > regmax) R *= 2;
            }
        }
    }
    u32 cs = 32;
    if (R && pow2) { cs = 0; while ((1ull << cs) < R) cs++; }
    u64 rrun = 0;
    for (u32 k0 = 0; k0 < nkeys; k0 += 1024) {
        const u32 k = k0 + t;
        u64 c = k < nkeys ? cells_of(kcnt[k], span[k], sh0) : 0u;
        c = R ? (c + R - 1) / R : 0;
        u64 tot;
        const u64 ex = block_excl_scan<u64, 1024>(c, red, &tot);
        if (k < nkeys && R) {
            kreg[k] = (u32)(rrun + ex);
            for (u64 r = 0; r < c; r++) rkey[rrun + ex + r] = k;
        }
        rrun += tot;
    }
    {   // first bit of every key in the occupancy bitmap (whole words per key)
        const u32 fg = s_fg;
        u64 frun = 0;
        for (u32 k0 = 0; k0 < nkeys; k0 += 1024) {
            const u32 k = k0 + t;
            u64 c = (fg != 0xFFFFFFFFu && k < nkeys && kcnt[k]) ? (u64)(((span[k] >> fg) + 2u + 31u) & ~31u) : 0u;
            u64 tot;
            const u64 ex = block_excl_scan<u64, 1024>(c, red, &tot);
            if (k < nkeys) fbase[k] = (u32)(frun + ex);
            frun += tot;
        }
        if (t == 0) { hdr[HDR_FG] = frun <= IVX_FBITS_MAX ? fg : 0xFFFFFFFFu; hdr[HDR_FBITS] = (u32)frun; }
    }
    if (t == 0) {
        kreg[nkeys] = (u32)rrun;
        hdr[HDR_CS] = pow2 ? cs : 0xFFFFFFFFu;
        hdr[HDR_NREG] = R ? (u32)rrun : 0u;
        hdr[HDR_RCELLS] = (u32)(R > 0xFFFFFFFFull ? 0xFFFFFFFFull : R);
        const u64 M = R ? ((1ull << 40) + R - 1) / R : 0;
        hdr[HDR_RMUL_LO] = (u32)M; hdr[HDR_RMUL_HI] = (u32)(M >> 32);
        hdr[HDR_PK24] = (R && sh0 < 24 && R <= (1ull << (24 - sh0))) ? 1u : 0u;
    }
}

// one thread per probe region: the level-0 cell window a workgroup stages for it and its entry range
__global__ __launch_bounds__(256) void k_join_regdesc(const i32 *origin, const u32 *span, const u32 *lbase, u32 *hdr,
                                                      const u32 *kreg, const u32 *rkey, const u32 *binstart, ivx_regdesc *rdesc)
{
    const u32 r = threadIdx.x + blockIdx.x * 256;
    if (r == 0) {                                               // rows outside level 0: the regions are not one LDS-resident level
        bool upper = hdr[HDR_LEVCNT] == 0;
        for (u32 l = 1; l < hdr[HDR_NLEV]; l++) upper |= hdr[HDR_LEVCNT + l] != 0;
        if (upper) hdr[HDR_SLOW] = 1u;
    }
    if (r >= hdr[HDR_NREG]) return;
    const u32 sh0 = hdr[HDR_SH0];
    const u64 R = hdr[HDR_RCELLS];
    ivx_regdesc d;
    d.k = rkey[r]; d.origin = origin[d.k]; d.span = span[d.k]; d.lb = lbase[d.k];
    const u32 cells0 = (d.span >> sh0) + 1u;
    const u64 rc0w = (u64)(r - kreg[d.k]) * R;                  // < cells0: the region exists
    const u32 rc0 = (u32)rc0w;
    const u64 rc1w = rc0w + R;
    const u32 rc1 = rc1w < cells0 ? (u32)rc1w : cells0;
    d.slo = rc0 ? rc0 - 1u : 0u;                      // a build row starts at most one cell before the cell it reaches into
    d.shi = rc1 + IVX_RP_HALO < cells0 ? rc1 + IVX_RP_HALO : cells0;
    d.e0 = binstart[d.lb + d.slo];
    d.ne = binstart[d.lb + d.shi] - d.e0;
    d.rbase = (i32)((i64)d.origin + (i64)(rc0w << sh0));           // (rc0 < cells0: at most the key's largest start)
    if (d.ne > IVX_RP_ECAP || d.shi - d.slo + 1u > 8192u + IVX_RP_HALO + 2u) hdr[HDR_SLOW] = 1u;   // the slice does not fit LDS (ivx_join_regions.hip)
    rdesc[r] = d;
}

__global__ __launch_bounds__(BT) void k_join_count(const u32 *__restrict__ key, const i32 *__restrict__ s,
                                                   const i32 *__restrict__ e, u64 n, u32 nkeys,
                                                   const i32 *origin, const u32 *lbase, u32 *hdr, u32 *bincnt,
                                                   u32 *__restrict__ cellid, u32 *__restrict__ rank)
{
    const u32 sh0 = hdr[HDR_SH0], nlev = hdr[HDR_NLEV];
    u32 levels = 0;                                        // levels this thread put a row into
    for (u64 i = (u64)blockIdx.x * BT + threadIdx.x; i < n; i += (u64)gridDim.x * BT) {
        u32 k = key ? key[i] : 0u;
        if (k >= nkeys) { cellid[i] = 0xFFFFFFFFu; continue; }
        i32 si = s[i], ei = e[i];
        u32 l = level_of(si, ei, sh0, nlev);
        const u32 c = cell_of(origin, lbase, nkeys, k, si, l, sh0);
        cellid[i] = c;
        rank[i] = atomicAdd(&bincnt[c], 1u);               // the row's slot inside its cell: the scatter needs no second atomic
        levels |= 1u << l;
    }
#pragma unroll
    for (int d = IVX_WAVE / 2; d > 0; d >>= 1) levels |= __shfl_xor(levels, d, IVX_WAVE);
    if (lane_id() == 0)
        for (u32 l = 0; l < nlev; l++)
            if ((levels >> l) & 1u) hdr[HDR_LEVCNT + l] = 1u;   // "level holds rows" flag (same value from every writer)
}

__global__ __launch_bounds__(BT) void k_join_scatter(const i32 *__restrict__ s, const i32 *__restrict__ e, u64 n,
                                                     const u32 *__restrict__ binstart, const u32 *__restrict__ cellid,
                                                     const u32 *__restrict__ rank, ivx_ent *ent)
{
    for (u64 i = (u64)blockIdx.x * BT + threadIdx.x; i < n; i += (u64)gridDim.x * BT) {
        const u32 c = cellid[i];
        if (c == 0xFFFFFFFFu) continue;
        ivx_ent x; x.s = s[i]; x.e = e[i]; x.row = (u32)i;
        ent[binstart[c] + rank[i]] = x;
    }
}

// occupancy bitmap: every build row sets the bits of the blocks [block(start), block(max(start, end))] of its key
// (block(x) = min((x - origin) >> g, blocks of the start span) -- the last block stands for everything behind the largest
// start).  A probe row [qs, qe] can only match if one of the blocks [block(max(qs, origin)), block(qe)] is set: a build
// row with start <= qe and end >= qs holds a point of [max(start, qs), min(end, qe)], and block() is monotone.
// (rows with end < start match probe rows that contain [end, start]: those contain start.)
__global__ __launch_bounds__(BT) void k_join_filter(const u32 *__restrict__ key, const i32 *__restrict__ s, const i32 *__restrict__ e, u64 n, u32 nkeys,
                                                    const i32 *origin, const u32 *span, const u32 *fbase, const u32 *hdr, u32 *fbits)
{
    const u32 g = hdr[HDR_FG];
    if (g == 0xFFFFFFFFu) return;
    for (u64 i = (u64)blockIdx.x * BT + threadIdx.x; i < n; i += (u64)gridDim.x * BT) {
        const u32 k = key ? key[i] : 0u;
        if (k >= nkeys) continue;
        const i32 si = s[i], ei = e[i];
        const u32 last = (span[k] >> g) + 1u;                         // the overflow block
        const i64 o = origin[k];
        const u32 b0 = (u32)(((i64)si - o) >> g);                     // start >= origin, within the span
        const i64 hi = ((i64)(ei > si ? ei : si) - o) >> g;
        const u32 b1 = hi > (i64)last ? last : (u32)hi;
        const u32 p0 = fbase[k] + b0, p1 = fbase[k] + b1;
        for (u32 w = p0 >> 5; w <= (p1 >> 5); w++) {
            u32 m = 0xFFFFFFFFu;
            if (w == (p0 >> 5)) m &= 0xFFFFFFFFu << (p0 & 31);
            if (w == (p1 >> 5)) m &= 0xFFFFFFFFu >> (31 - (p1 & 31));
            if ((fbits[w] & m) != m) atomicOr(&fbits[w], m);
        }
    }
}

// ------------------------------------------------------------------ probe

constexpr int PT = 256;     // probe workgroup
constexpr int PI = 4;       // probe rows per thread per tile
constexpr int PTILE = PT * PI;

// routing regions: 2^sh0-wide cells over every key's span of starts, R (a power of two) cells per
// region, at most IVX_MAXREG_WIDE regions that never straddle a key; nreg = 0 if more keys than that have rows
__global__ __launch_bounds__(1024) void k_nroute_layout(const i32 *origin, const u32 *span, const u32 *kcnt, u32 nkeys, u32 *kreg, u32 *rkey, u32 *hdr)
{
    __shared__ u64 red[1024 / IVX_WAVE + 1];
    const u32 t = threadIdx.x;
    const u32 sh0 = 10;
    auto regions_for = [&](u32 cs) -> u64 {
        u64 a = 0;
        for (u32 k = t; k < nkeys; k += 1024) if (kcnt[k]) a += (((u64)(span[k] >> sh0) + 1) + ((1ull << cs) - 1)) >> cs;
        return block_sum<u64, 1024>(a, red);
    };
    u32 cs = 0;
    while (cs < 32 && regions_for(cs) > IVX_MAXREG_WIDE) cs++;
    const bool ok = regions_for(cs) <= IVX_MAXREG_WIDE;
    u64 rrun = 0;
    for (u32 k0 = 0; k0 < nkeys; k0 += 1024) {
        const u32 k = k0 + t;
        u64 c = (ok && k < nkeys && kcnt[k]) ? ((((u64)(span[k] >> sh0) + 1) + ((1ull << cs) - 1)) >> cs) : 0;
        u64 tot;
        const u64 ex = block_excl_scan<u64, 1024>(c, red, &tot);
        if (k < nkeys) { kreg[k] = (u32)(rrun + ex); for (u64 r = 0; r < c; r++) rkey[rrun + ex + r] = k; }
        rrun += tot;
    }
    if (t == 0) {
        kreg[nkeys] = (u32)rrun;
        hdr[HDR_SH0] = sh0; hdr[HDR_CS] = cs < 31 ? cs : 31; hdr[HDR_NREG] = ok ? (u32)rrun : 0u; hdr[HDR_RCELLS] = 1u << (cs < 31 ? cs : 31);
        hdr[HDR_RMUL_LO] = 0; hdr[HDR_RMUL_HI] = 0;
    }
}

// per-row match counts over probe rows ROUTED by coordinate region (ivx_route_rows): build sides with too many regions
// for the LDS-slice pipeline.  XCD x sweeps the x-th eighth of the routed rows (gridDim.x is a multiple of 8); the key of
// a row follows from its routing region; counts at the routed positions, their sum onto *total.
__global__ __launch_bounds__(PT) void k_overlap_rowval_routed(JoinIndexView ix, const u32 *__restrict__ rkey, u32 nreg, const u64 *__restrict__ pse,
                                                              const u32 *__restrict__ offs, u32 nblk, int exists_only, u32 *__restrict__ vb,
                                                              unsigned long long *total, const u32 *unsorted)
{
    __shared__ u32 s_rfirst[IVX_MAXREG_WIDE + 2];
    __shared__ u64 lds64[PT / IVX_WAVE];
    if (*unsorted == 0) return;
    for (u32 t = threadIdx.x; t <= nreg; t += PT) s_rfirst[t] = offs[(u64)t * nblk];
    __syncthreads();
    const u64 nrows = s_rfirst[nreg];
    const u32 sh0 = ix.hdr[HDR_SH0], nlev = ix.hdr[HDR_NLEV];
    const u32 xcd = blockIdx.x & 7u, nb = gridDim.x >> 3, bi = blockIdx.x >> 3;
    const u64 seg_lo = nrows * xcd / 8, seg_hi = nrows * (xcd + 1) / 8;
    u64 acc = 0;
    for (u64 i = seg_lo + (u64)bi * PT + threadIdx.x; i < seg_hi; i += (u64)nb * PT) {
        u32 a = 0, b = nreg;                                            // last region whose first row is <= i
        while (a < b) { const u32 m = (a + b + 1) >> 1; if (s_rfirst[m] <= i) a = m; else b = m - 1; }
        const u64 w = pse[i];
        u32 m = 0;
        walk(ix, sh0, 0, nlev, rkey[a], (i32)(u32)w, (i32)(u32)(w >> 32), [&](u32) { m++; });
        vb[i] = m;
        acc += m;
    }
    if (!exists_only && total) {
        const u64 tot = block_sum<u64, PT>(acc, lds64);
        if (threadIdx.x == 0 && tot) atomicAdd(total, (unsigned long long)tot);
    }
}

// PI_: probe rows per thread per tile.  A thread walks its rows one after the other (each walk is a chain of dependent index
// reads), so a DataFusion-sized batch takes ONE row per thread -- 8192 rows are 32 workgroups and one walk of latency, not 8 and four
template <int MODE, int PI_>
__global__ __launch_bounds__(PT) void k_probe_overlap(JoinIndexView ix, const u32 *__restrict__ pkey,
                                                      const i32 *__restrict__ ps, const i32 *__restrict__ pe, u64 n,
                                                      u32 *__restrict__ per_row, u8 *__restrict__ exists,
                                                      u32 *__restrict__ ob, u32 *__restrict__ op, u64 cap,
                                                      unsigned long long *cursor, const u32 *gate)
{
    __shared__ u32 lds[PT / IVX_WAVE + 1];
    __shared__ unsigned long long s_base;
    if (gate && *gate != 0) return;                        // (routed callers: only when the probe rows were left in place)
    const u32 sh0 = ix.hdr[HDR_SH0], nlev = ix.hdr[HDR_NLEV];
    constexpr int PTILE_ = PT * PI_;
    const u64 ntiles = (n + PTILE_ - 1) / PTILE_;
    u64 acc = 0;                                           // MODE COUNT / PER_ROW: pairs seen by this thread
    for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        u32 cnt[PI_], st0[PI_], st1[PI_];
        u32 tsum = 0;
#pragma unroll
        for (int it = 0; it < PI_; it++) {
            const u64 i = tile * PTILE_ + (u64)it * PT + threadIdx.x;
            u32 m = 0, a0 = 0, a1 = 0;
            if (i < n) {
                const u32 k = pkey ? pkey[i] : 0u;
                const i32 qs = ps[i], qe = pe[i];
                walk(ix, sh0, 0, nlev, k, qs, qe, [&](u32 row) {
                    if (m == 0) a0 = row; else if (m == 1) a1 = row;
                    m++;
                });
                if (MODE == JP_PER_ROW) per_row[i] = m;
                if (MODE == JP_EXISTS) exists[i] = m != 0;
            }
            cnt[it] = m; st0[it] = a0; st1[it] = a1;
            tsum += m;
        }
        if (MODE != JP_FILL) { acc += tsum; continue; }
        // ---- wavefront + workgroup prefix sum -> one cursor bump per tile
        u32 total;
        const u32 ex = block_excl_scan<u32, PT>(tsum, lds, &total);
        if (threadIdx.x == 0) s_base = total ? atomicAdd(cursor, (unsigned long long)total) : 0ull;
        __syncthreads();
        const u64 base = s_base;
        if (total && base + total <= cap) {
            u64 at = base + ex;
#pragma unroll
            for (int it = 0; it < PI_; it++) {
                const u32 m = cnt[it];
                if (m == 0) continue;
                const u64 i = tile * PTILE_ + (u64)it * PT + threadIdx.x;
                if (m <= 2) {
                    ob[at] = st0[it]; op[at] = (u32)i;
                    if (m == 2) { ob[at + 1] = st1[it]; op[at + 1] = (u32)i; }
                    at += m;
                } else {
                    const u32 k = pkey ? pkey[i] : 0u;
                    walk(ix, sh0, 0, nlev, k, ps[i], pe[i], [&](u32 row) { ob[at] = row; op[at] = (u32)i; at++; });
                }
            }
        }
        __syncthreads();                                   // s_base is rewritten next tile
    }
    if (MODE == JP_COUNT || MODE == JP_PER_ROW) {
        __shared__ u64 lds64[PT / IVX_WAVE];
        u64 tot = block_sum<u64, PT>(acc, lds64);
        if (threadIdx.x == 0 && tot) atomicAdd(cursor, (unsigned long long)tot);
    }
}

}  // namespace

// ---------------------------------------------------------------------------

// Regions that only ROUTE big unsorted probe batches (no cells of their own) over per-key (origin, span, count)
// tables that outlive the index.  The region count arrives on the host with the caller's next synchronisation
// (ivx_route_view_ready).
ivx_status ivx_route_view_build(ivx_ctx *ctx, ivx_index *ix, const i32 *origin, const u32 *span, const u32 *kcnt)
{
    hipStream_t st = ctx->stream;
    const u32 nkeys = ix->nkeys;
    u32 *kreg, *rkey, *rhdr;
    IVX_TRY(ivx_index_alloc(ctx, ix, ((size_t)nkeys + 1) * sizeof(u32), (void **)&kreg));
    IVX_TRY(ivx_index_alloc(ctx, ix, ((size_t)IVX_MAXREG_WIDE + 1) * sizeof(u32), (void **)&rkey));
    IVX_TRY(ivx_index_alloc(ctx, ix, HDR_WORDS * sizeof(u32), (void **)&rhdr));
    IVX_HIP(ctx, hipMemsetAsync(rhdr, 0, HDR_WORDS * sizeof(u32), st));
    hipLaunchKernelGGL(k_nroute_layout, dim3(1), dim3(1024), 0, st, origin, span, kcnt, nkeys, kreg, rkey, rhdr);
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 48, rhdr, HDR_WORDS * sizeof(u32), hipMemcpyDeviceToHost, st));
    ix->nroute = JoinIndexView{};
    ix->nroute.origin = origin; ix->nroute.span = span; ix->nroute.kcnt = kcnt; ix->nroute.kreg = kreg; ix->nroute.rkey = rkey;
    ix->nroute.hdr = rhdr; ix->nroute.nkeys = nkeys;
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}
void ivx_route_view_ready(ivx_ctx *ctx, ivx_index *ix) { ix->nroute_nreg = ((const u32 *)(ctx->h_scalars + 48))[HDR_NREG]; }

// rle_right / exists of a big probe batch on an index with too many regions for the LDS-slice pipeline
ivx_status ivx_join_rowval_routed(ivx_ctx *ctx, const ivx_index *ix, int mode, const u32 *key, const i32 *s, const i32 *e, u64 n,
                                  u32 *per_row, u8 *exists, u64 *d_total)
{
    hipStream_t st = ctx->stream;
    ivx_routed R;
    IVX_TRY(ivx_route_rows(ctx, ix->nroute, key, s, e, n, 0u, &R));
    u32 *vb;
    IVX_TRY(ctx->get_scratch(WS_T2, n * sizeof(u32), (void **)&vb));
    // (five workgroups per CU: the rows in flight on an XCD span less of the index than its L2 holds -- 100M x 10M rle_right
    //  11.0 -> 9.5 ms; see k_nearest_routed)
    const u32 grid = (ivx_stream_grid(n, PT * 4, 1280u) + 7u) & ~7u;
    hipLaunchKernelGGL(k_overlap_rowval_routed, dim3(grid), dim3(PT), 0, st, ix->jv, ix->nroute.rkey, ix->nroute_nreg, R.pse, R.hist, R.nblk,
                       mode == JP_EXISTS ? 1 : 0, vb, (unsigned long long *)d_total, R.unsorted);
    IVX_TRY(ivx_unroute_u32(ctx, R, n, vb, per_row, exists));
    // rows that came in region order were not moved: the plain kernel answers them in place (gated on the flag)
    const u32 g2 = ivx_stream_grid(n, PTILE, 256 * 8);
    if (mode == JP_PER_ROW) hipLaunchKernelGGL((k_probe_overlap<JP_PER_ROW, PI>), dim3(g2), dim3(PT), 0, st, ix->jv, key, s, e, n, per_row, exists, (u32 *)nullptr, (u32 *)nullptr, (u64)0, (unsigned long long *)d_total, R.unsorted);
    else hipLaunchKernelGGL((k_probe_overlap<JP_EXISTS, PI>), dim3(g2), dim3(PT), 0, st, ix->jv, key, s, e, n, per_row, exists, (u32 *)nullptr, (u32 *)nullptr, (u64)0, (unsigned long long *)d_total, R.unsorted);
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

ivx_status ivx_join_build(ivx_ctx *ctx, ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n, bool overlap)
{
    const u32 nkeys = ix->nkeys;
    hipStream_t st = ctx->stream;
    const u64 maxcells = 2 * n + (IVX_LSTEP >= 4 ? n / 4 : n) + (u64)IVX_MAXL * nkeys + 64;   // geometric sum over the levels
    if (maxcells + 1 >= 0xFFFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "build side too large for 32-bit cell ids");

    i32 *origin; u32 *span, *kcnt, *lbase, *binstart, *hdr, *kreg, *rkey, *fbase, *fbits; ivx_ent *ent; ivx_regdesc *rdesc;
    IVX_TRY(ivx_index_alloc(ctx, ix, nkeys * sizeof(i32), (void **)&origin));
    IVX_TRY(ivx_index_alloc(ctx, ix, nkeys * sizeof(u32), (void **)&span));
    IVX_TRY(ivx_index_alloc(ctx, ix, nkeys * sizeof(u32), (void **)&kcnt));
    IVX_TRY(ivx_index_alloc(ctx, ix, (size_t)IVX_MAXL * nkeys * sizeof(u32), (void **)&lbase));
    IVX_TRY(ivx_index_alloc(ctx, ix, (maxcells + 1) * sizeof(u32), (void **)&binstart));
    IVX_TRY(ivx_index_alloc(ctx, ix, HDR_WORDS * sizeof(u32), (void **)&hdr));
    IVX_TRY(ivx_index_alloc(ctx, ix, ((size_t)nkeys + 1) * sizeof(u32), (void **)&kreg));
    // region tables: 256 entries cover the one-digit scheme; big build sides may need up to IVX_MAXREG2
    // region tables: regions hold ~0.85 * IVX_RP_ECAP level-0 rows or 2^IVX_REG_CS_MAX cells, and at least one per key
    size_t regcap = (size_t)(n / (IVX_RP_ECAP / 2)) + (size_t)(maxcells >> IVX_REG_CS_MAX) + 2 * (size_t)nkeys + 64;
    regcap = regcap <= IVX_MAXREG + 1 ? (size_t)IVX_MAXREG + 1 : (regcap > (size_t)IVX_MAXREG2 + 1 ? (size_t)IVX_MAXREG2 + 1 : regcap);
    IVX_TRY(ivx_index_alloc(ctx, ix, regcap * sizeof(u32), (void **)&rkey));
    IVX_TRY(ivx_index_alloc(ctx, ix, regcap * sizeof(ivx_regdesc), (void **)&rdesc));
    IVX_TRY(ivx_index_alloc(ctx, ix, (n ? n : 1) * sizeof(ivx_ent), (void **)&ent));
    // occupancy bitmap: as many words as the layout kernel may ask for with this many rows (it picks the block width on the
    // device), + padding for the probe's window reads
    const size_t fwords = IVX_FBITS_MAX / 32 + 4;
    IVX_TRY(ivx_index_alloc(ctx, ix, nkeys * sizeof(u32), (void **)&fbase));
    IVX_TRY(ivx_index_alloc(ctx, ix, fwords * sizeof(u32), (void **)&fbits));

    i32 *kmin, *kmax; u32 *cellid, *rank, *errflag;
    IVX_TRY(ctx->get_scratch(WS_GRID0, nkeys * sizeof(i32), (void **)&kmin));
    IVX_TRY(ctx->get_scratch(WS_GRID1, nkeys * sizeof(i32), (void **)&kmax));
    IVX_TRY(ctx->get_scratch(WS_GRID2, (n ? n : 1) * sizeof(u32), (void **)&cellid));
    IVX_TRY(ctx->get_scratch(WS_T9, (n ? n : 1) * sizeof(u32), (void **)&rank));
    errflag = (u32 *)(ctx->d_scalars + 8);
    u32 *lenhist = (u32 *)(ctx->d_scalars + 32);                        // 33 counters

    const ivx_zero_ranges zr{{errflag, lenhist, hdr}, {1u, 34u, (u32)HDR_WORDS}};     // cleared by the key-statistics' first kernel
    // IVX_FILTER=0: no occupancy bitmap; =force: whenever it fits (tests); default: when it would reject most probe rows
    const char *fenv = getenv("IVX_FILTER");
    const int filter_mode = !fenv ? 1 : !strcmp(fenv, "0") ? 0 : !strcmp(fenv, "force") ? 2 : 1;
    if (filter_mode) IVX_HIP(ctx, hipMemsetAsync(fbits, 0, fwords * sizeof(u32), st));
    IVX_TRY(ivx_keystats_len(ctx, key, s, n, nkeys, kmin, kmax, kcnt, errflag, 1u, e, lenhist, &zr));   // + the length classes for the layout
    // (every row's atomic returns a value, its loads depend on nothing: a row or two per thread, not a loop of eight round trips)
    const u32 grid = ivx_stream_grid(n, BT * 2, 16384);
    hipLaunchKernelGGL(k_join_layout, dim3(1), dim3(1024), 0, st, kmin, kmax, kcnt, nkeys, n, origin, span, lbase, hdr, maxcells, kreg, rkey, (const u32 *)lenhist, (u32)(regcap - 1), fbase, filter_mode);
    if (filter_mode) hipLaunchKernelGGL(k_join_filter, dim3(grid), dim3(BT), 0, st, key, s, e, n, nkeys, (const i32 *)origin, (const u32 *)span, (const u32 *)fbase, (const u32 *)hdr, fbits);
    // routing regions for the per-row modes of build sides that outgrow the LDS-slice pipeline (more than 1023 regions
    // takes > 5 M rows); count / coverage / nearest indexes route on their rank grids
    const bool want_route = ix->kind == IVX_KIND_OVERLAP && n >= (4u << 20);
    // What follows -- cell count, scan, scatter, region descriptors -- writes binstart / ent / rdesc and two header flags;
    // the per-key tables, the region layout and the occupancy bitmap are final here.  With the build overlap on it goes to
    // the aux stream and the caller gets its index back as soon as the layout has reached the host.
    overlap = overlap && ctx->overlap && ctx->aux != nullptr && !want_route && n >= (1u << 16);
    hipStream_t tail_st = st;
    if (overlap) {
        ctx->join_tail();                                       // (an earlier tail may still read the scratch this one is about to write)
        IVX_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
        IVX_HIP(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
        tail_st = ctx->aux;
    }
    IVX_HIP(ctx, hipMemsetAsync(binstart, 0, (maxcells + 1) * sizeof(u32), tail_st));    // (8 MB per million rows: part of the tail, not of what the caller waits for)
    hipLaunchKernelGGL(k_join_count, dim3(grid), dim3(BT), 0, tail_st, key, s, e, n, nkeys, origin, lbase, hdr, binstart, cellid, rank);
    {
        const hipStream_t keep = ctx->stream;
        ctx->stream = tail_st;                                  // (the scan launches on the context's stream)
        const ivx_status sst = ivx_scan_exclusive_u32(ctx, binstart, maxcells + 1);
        ctx->stream = keep;
        IVX_TRY(sst);
    }
    hipLaunchKernelGGL(k_join_scatter, dim3(grid), dim3(BT), 0, tail_st, s, e, n, (const u32 *)binstart, (const u32 *)cellid, (const u32 *)rank, ent);
    hipLaunchKernelGGL(k_join_regdesc, dim3((u32)((regcap + 255) / 256)), dim3(256), 0, tail_st, origin, span, lbase, hdr, kreg, rkey, binstart, rdesc);
    IVX_HIP(ctx, hipGetLastError());
    if (overlap) {
        if (ix->ready == nullptr) IVX_HIP(ctx, hipEventCreateWithFlags(&ix->ready, hipEventDisableTiming));
        IVX_HIP(ctx, hipEventRecord(ix->ready, ctx->aux));
        IVX_HIP(ctx, hipEventRecord(ctx->tail_ev, ctx->aux));    // (the context's own event: the index, and its event, may be freed first)
        ctx->tail_pending = true;
        ix->jv_fast_unknown = true;
    }
    if (want_route) IVX_TRY(ivx_route_view_build(ctx, ix, origin, span, kcnt));

    // key ids are validated on the device; surface the flag (one small D2H).  (With the tail overlapped the header copy
    // carries the layout's words -- regions, bitmap, packed rows -- but not yet the two flags the tail sets.)
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 8, errflag, sizeof(u32), hipMemcpyDeviceToHost, st));
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 32, hdr, HDR_WORDS * sizeof(u32), hipMemcpyDeviceToHost, st));
    IVX_HIP(ctx, hipStreamSynchronize(st));
    if (*(u32 *)(ctx->h_scalars + 8)) return ctx->fail(IVX_ERR_INVALID, "build key id >= n_keys");
    ix->jv_nreg = ((const u32 *)(ctx->h_scalars + 32))[HDR_NREG];
    ix->jv_filter = ((const u32 *)(ctx->h_scalars + 32))[HDR_FG] != 0xFFFFFFFFu;
    ix->jv_pk24 = ((const u32 *)(ctx->h_scalars + 32))[HDR_PK24] != 0;
    ix->jv_fast = ((const u32 *)(ctx->h_scalars + 32))[HDR_SLOW] == 0 && ix->jv_nreg > 0;
    if (want_route) ivx_route_view_ready(ctx, ix);

    ix->jv.origin = origin; ix->jv.span = span; ix->jv.kcnt = kcnt; ix->jv.lbase = lbase;
    ix->jv.binstart = binstart; ix->jv.ent = ent; ix->jv.hdr = hdr; ix->jv.nkeys = nkeys;
    ix->jv.kreg = kreg; ix->jv.rkey = rkey; ix->jv.rdesc = rdesc;
    ix->jv.fbits = fbits; ix->jv.fbase = fbase;
    return IVX_OK;
}

ivx_status ivx_join_probe(ivx_ctx *ctx, const JoinIndexView &jv, int mode,
                          const u32 *key, const i32 *s, const i32 *e, u64 n,
                          u32 *per_row, u8 *exists, u32 *ob, u32 *op, u64 cap, u64 *d_cursor)
{
    if (n == 0) return IVX_OK;
    unsigned long long *cur = (unsigned long long *)d_cursor;
    const bool small = n <= (1u << 18);                                // one row per thread: latency, not throughput, is what a small batch costs
    const u32 grid = small ? (u32)((n + PT - 1) / PT) : ivx_stream_grid(n, PTILE, 256 * 8);
#define IVX_PROBE_LAUNCH(M_) do { \
        if (small) hipLaunchKernelGGL((k_probe_overlap<M_, 1>), dim3(grid), dim3(PT), 0, ctx->stream, jv, key, s, e, n, per_row, exists, ob, op, cap, cur, (const u32 *)nullptr); \
        else hipLaunchKernelGGL((k_probe_overlap<M_, PI>), dim3(grid), dim3(PT), 0, ctx->stream, jv, key, s, e, n, per_row, exists, ob, op, cap, cur, (const u32 *)nullptr); \
    } while (0)
    switch (mode) {
    case JP_COUNT: IVX_PROBE_LAUNCH(JP_COUNT); break;
    case JP_PER_ROW: IVX_PROBE_LAUNCH(JP_PER_ROW); break;
    case JP_EXISTS: IVX_PROBE_LAUNCH(JP_EXISTS); break;
    default: IVX_PROBE_LAUNCH(JP_FILL); break;
    }
#undef IVX_PROBE_LAUNCH
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}
