// ivx_join_regions.hip -- the overlap probe without random HBM gathers.
//
// Unsorted probe rows gathered straight from the index cost ~1.6 fabric reads of
// 64 B per row (profiles/r1_a_probe_count_v1_pmc.txt): the index (tens of MB) does
// not fit a 4 MB XCD L2.  Here the probe side is first radix-partitioned by index
// REGION (a run of 2^cs level-0 cells of one key, at most 255 regions = one 8-bit
// digit), then every region is probed by workgroups that stage its slice of the
// index -- cell offsets and (start,end) of its entries -- in LDS and stream the
// region's probe rows through it with coalesced reads:
//
//   k_part_hist     region histogram per workgroup (LDS counters, 16-byte row loads)
//   scan            exclusive prefix over [region][workgroup]
//   k_part_scatter  (qs,qe,row) records re-ordered through LDS, contiguous runs out
//   k_probe_regions LDS-resident slice; matches are compacted with a wavefront
//                   prefix sum into per-wavefront LDS staging and written with ONE
//                   global atomicAdd per workgroup and round (~3000 pairs); no
//                   LDS or global atomics per pair.  Rows longer than the slice
//                   halo, regions whose slice exceeds LDS and the long-interval
//                   levels fall back to global reads inside the same kernel.
//
// HBM traffic per probe row: 8 B (hist) + 12 B + 12 B (scatter) + 12 B (probe) +
// 8 B per pair, all streaming (measured: profiles/r1_d_regions_pipeline_pmc.txt).
#include <type_traits>
#include "ivx_join.hpp"
#include <cstdlib>
#include <cstring>

namespace {

// kernels carry an `int dbg` of ablation switches (skip stores, synthetic loads, ...) that only means
// something in profiling builds; elsewhere it is forced to 0 and the branches fold away
#ifdef IVX_ABLATE
#define IVX_DBG_ARG(dbg) (dbg)
#else
#define IVX_DBG_ARG(dbg) 0
#endif

// ------------------------------------------------------------------ partition pass

constexpr int PA_T = 1024;
#ifndef IVX_PA_I
#define IVX_PA_I 12
#endif
constexpr int PA_I = IVX_PA_I;                    // rows per thread and tile (multiple of 4)
constexpr int PA_TILE = PA_T * PA_I;              // 12288 rows: ~63 rows per region and tile = the length of the runs written to HBM
constexpr u32 NO_REGION = 0xFFFFFFFFu;
constexpr u32 KT_MAX = 256;                       // per-key tables cached in LDS up to this many keys

// per-key lookup for "which region does a probe row start in", cached in LDS (up to KT_MAX keys; lastcell = 0xFFFFFFFF: key
// has no build rows).  The LDS tables are passed to region_of as the kernel's own arrays, never through a pointer that may
// also be null / global: such a pointer makes every lookup a flat load with full waits.
struct KeyTab {
    bool lds;
    u32 nkeys, sh0, cs;          // cs = log2(cells per region), or ~0u: divide by multiplying with rmul
    u64 rmul;
};

__device__ __forceinline__ void keytab_load(const JoinIndexView &ix, i32 *s_origin, u32 *s_last, u32 *s_kreg, KeyTab &kt)
{
    kt.nkeys = ix.nkeys; kt.sh0 = ix.hdr[HDR_SH0]; kt.cs = ix.hdr[HDR_CS];
    kt.rmul = (u64)ix.hdr[HDR_RMUL_LO] | ((u64)ix.hdr[HDR_RMUL_HI] << 32);
    kt.lds = ix.nkeys <= KT_MAX;
    if (kt.lds) {
        for (u32 k = threadIdx.x; k < ix.nkeys; k += blockDim.x) {
            s_origin[k] = ix.origin[k];
            s_last[k] = ix.kcnt[k] ? (ix.span[k] >> kt.sh0) : 0xFFFFFFFFu;
            s_kreg[k] = ix.kreg[k];
        }
    }
}

#define KEYTAB_DISPATCH(kt_, body_) do { if ((kt_).lds) body_(std::true_type{}); else body_(std::false_type{}); } while (0)

// region of a probe row = region of the level-0 cell its START falls in (clamped into the key).  KLDS is a template
// parameter, not a run-time choice next to the loads: "LDS table or index column" in one expression compiles to flat loads.
// The kernels run their main loop once per case (KEYTAB_DISPATCH).
template <bool KLDS>
__device__ __forceinline__ u32 region_of(const JoinIndexView &ix, const KeyTab &kt, const i32 *s_origin, const u32 *s_last, const u32 *s_kreg, u32 k, i32 qs)
{
    if (k >= kt.nkeys) return NO_REGION;
    i32 origin; u32 last, kreg;
    if (KLDS) { origin = s_origin[k]; last = s_last[k]; kreg = s_kreg[k]; }
    else { origin = ix.origin[k]; last = ix.kcnt[k] ? (ix.span[k] >> kt.sh0) : 0xFFFFFFFFu; kreg = ix.kreg[k]; }
    if (last == 0xFFFFFFFFu) return NO_REGION;                    // cannot match anything
    const i64 d = (i64)qs - (i64)origin;
    const i64 c64 = d <= 0 ? 0 : (d >> kt.sh0);
    const u32 c = c64 > (i64)last ? last : (u32)c64;
    return kreg + (kt.cs != 0xFFFFFFFFu ? c >> kt.cs : (u32)(((u64)c * kt.rmul) >> 40));   // k_join_layout guarantees exactness
}

// LDS counter bump that returns the old value.  Sorted / clustered probe input sends a whole wavefront to
// the same counter; then one lane adds the wavefront's count and the lanes rank themselves by ballot,
// instead of 64 serialised same-address atomics.
__device__ __forceinline__ u32 lds_count_up(u32 *cnt, u32 d, bool active)
{
    const u64 act = __ballot(active);
    if (act == 0) return 0;
    const u32 first = (u32)__builtin_ctzll(act);
    const u32 d0 = __shfl(d, first, IVX_WAVE);
    const u64 same = __ballot(active && d == d0);
    if (same == act) {                                          // wave-uniform digit
        u32 base = 0;
        if (lane_id() == first) base = atomicAdd(&cnt[d0], (u32)__popcll(act));
        base = __shfl(base, first, IVX_WAVE);
        return base + mask_rank(act);
    }
    return active ? atomicAdd(&cnt[d], 1u) : 0u;
}

// The same for the FOUR consecutive rows a lane loads in one go (lane l of a wavefront holds rows 4l .. 4l + 3 of a 256-row
// stretch of the input): coordinate-sorted input puts the whole stretch into one region, and then one lane bumps the
// counter by 256 and every row's rank follows from its place in the stretch; anything else takes four plain LDS atomics per
// lane.  ONE test and one branch per four rows (lds_count_up spends two ballots, a shuffle and two or three branches on
// every row, and the partition is bound by the instructions it issues).
__device__ __forceinline__ void lds_count_up4(u32 *cnt, const u32 (&d)[4], u32 (&rank)[4])
{
    const u32 d0 = __builtin_amdgcn_readfirstlane(d[0]);
    const bool uni = d[0] == d0 && d[1] == d0 && d[2] == d0 && d[3] == d0 && d0 != NO_REGION;
    if (__builtin_amdgcn_ballot_w64(!uni) == 0) {
        u32 base = 0;
        const u32 ln = lane_id();
        if (ln == 0) base = atomicAdd(&cnt[d0], 4u * IVX_WAVE);
        base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
        for (int k = 0; k < 4; k++) rank[k] = base + ln * 4u + (u32)k;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) rank[k] = d[k] != NO_REGION ? atomicAdd(&cnt[d[k]], 1u) : 0u;
}

// four consecutive probe rows per lane: 16-byte loads when the columns are 16-byte aligned
template <bool VEC>
__device__ __forceinline__ void load4(const u32 *__restrict__ pkey, const i32 *__restrict__ ps, const i32 *__restrict__ pe,
                                      u64 i, u64 hi, u32 (&k)[4], i32 (&s)[4], i32 (&e)[4])
{
    if (VEC && i + 4 <= hi) {
        const uint4 kv = pkey ? *reinterpret_cast<const uint4 *>(pkey + i) : make_uint4(0, 0, 0, 0);
        const int4 sv = *reinterpret_cast<const int4 *>(ps + i);
        k[0] = kv.x; k[1] = kv.y; k[2] = kv.z; k[3] = kv.w;
        s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w;
        if (pe) { const int4 ev = *reinterpret_cast<const int4 *>(pe + i); e[0] = ev.x; e[1] = ev.y; e[2] = ev.z; e[3] = ev.w; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool ok = i + j < hi;
            k[j] = ok ? (pkey ? pkey[i + j] : 0u) : 0xFFFFFFFFu;       // key id 0xFFFFFFFF never matches
            s[j] = ok ? ps[i + j] : 0;
            e[j] = (ok && pe) ? pe[i + j] : 0;
        }
    }
}

// adj = 1 for the UDTFs' strict mode: the query is shrunk to [start+1, end-1] before anything else
// (interval_tree.rs:185-188, :253-256; i32 wrapping like the reference's release build)
// `unsorted` (device flag, preset 0) is raised when the rows are NOT already grouped by ascending region --
// a row of a smaller region after a larger one, or a row that cannot be routed.  If it stays 0 (probe input
// sorted by contig id and start: the usual state of genomic files) the partitioned order IS the input order
// and the scatter pass is skipped altogether.
// ND = digits of the pass: 256 for up to IVX_MAXREG regions, 1024 for up to IVX_MAXREG_WIDE (build sides of a few
// million rows: four times the table, shorter runs in the scatter, still one pass)
// SPLIT (more than IVX_MAXREG_WIDE regions): the digit is the SUPER-region = region / G (G = split.x, as a
// multiplication by split.y = ceil(2^32 / G), exact for region * G < 2^32); a second pass orders each super-region's
// rows by region % G (k_p2_*).  Sortedness is still judged on the regions themselves.
template <bool VEC, int ND, bool SPLIT = false>
__global__ __launch_bounds__(PA_T) void k_part_hist(JoinIndexView ix, const u32 *__restrict__ pkey, const i32 *__restrict__ ps,
                                                    u64 n, u32 nblk, u32 chunk, u32 *__restrict__ hist, u32 adj, u32 *unsorted,
                                                    uint2 split = make_uint2(1u, 0u))
{
    __shared__ u32 cnt[ND];
    __shared__ i32 s_origin[KT_MAX];
    __shared__ u32 s_last[KT_MAX], s_kreg[KT_MAX];
    __shared__ u32 s_unsorted;
    KeyTab kt;
    keytab_load(ix, s_origin, s_last, s_kreg, kt);
    if (threadIdx.x < ND) cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_unsorted = 0;
    __syncthreads();
    const u64 lo = (u64)blockIdx.x * chunk;
    const u64 hi = lo + chunk < n ? lo + chunk : n;
    auto body = [&](auto klds_tag) {
    constexpr bool KLDS = decltype(klds_tag)::value;
    for (u64 i0 = lo; i0 < hi; i0 += (u64)PA_T * 4) {
        u32 k[4]; i32 q[4], unused[4];
        const u64 i = i0 + (u64)threadIdx.x * 4;
        if (i >= hi) continue;
        load4<VEC>(pkey, ps, nullptr, i, hi, k, q, unused);
        u32 d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            d[u] = region_of<KLDS>(ix, kt, s_origin, s_last, s_kreg, k[u], (i32)((u32)q[u] + adj));
            lds_count_up(cnt, SPLIT ? (u32)(((u64)d[u] * split.y) >> 32) : d[u], d[u] != NO_REGION);
        }
        if (!s_unsorted) {                                  // (once raised nobody needs to look any further)
            bool bad = false;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (i + u >= hi) break;
                bad |= d[u] == NO_REGION || (u && d[u] < d[u - 1]);
            }
            if (i + 3 < hi && i + 4 < n) {                  // the row after this thread's four: next thread, wavefront, loop step or workgroup
                const u32 dn = region_of<KLDS>(ix, kt, s_origin, s_last, s_kreg, pkey ? pkey[i + 4] : 0u, (i32)((u32)ps[i + 4] + adj));
                bad |= dn < d[3];
            }
            if (bad) s_unsorted = 1;
        }
    }
    };
    KEYTAB_DISPATCH(kt, body);
    __syncthreads();
    if (threadIdx.x < ND) hist[(u64)threadIdx.x * nblk + blockIdx.x] = cnt[threadIdx.x];
    if (threadIdx.x == 0 && s_unsorted) *unsorted = 1;
}

// order inside a region is irrelevant (the reference pins only the pair multiset), so the local
// rank of a row is just the value an LDS counter held when the row arrived
// RowT = u32: the row's index in the probe batch (join: it goes into the pair list);
// RowT = u16: its index inside this workgroup's chunk of at most two tiles (per-row-output operators: k_unpermute
// puts the chunk back in input order through LDS, so the chunk-local index is all that is needed)
template <bool VEC, typename RowT, int ND, bool SPLIT = false>
__global__ __launch_bounds__(PA_T) void k_part_scatter(JoinIndexView ix, const u32 *__restrict__ pkey, const i32 *__restrict__ ps,
                                                       const i32 *__restrict__ pe, u64 n, u32 nblk, const u32 *__restrict__ offs,
                                                       u64 *__restrict__ out_se, RowT *__restrict__ out_row, u32 chunk, u32 adj, const u32 *unsorted, int dbg,
                                                       uint2 split = make_uint2(1u, 0u), unsigned char *__restrict__ out_sub = nullptr)
{
    __shared__ u64 r_se[PA_TILE];
    __shared__ unsigned short r_slot[PA_TILE];          // the row's slot in the tile (its row id follows from it)
    using DigT = typename std::conditional<(ND > 256), unsigned short, unsigned char>::type;
    __shared__ DigT r_dig[PA_TILE];
    __shared__ u32 dstart[ND], gbase[ND];              // dstart: the tile's counters first, then (in place) their exclusive scan
    __shared__ u32 scan_lds[PA_T / IVX_WAVE + 1];
    __shared__ i32 s_origin[KT_MAX];
    __shared__ u32 s_last[KT_MAX], s_kreg[KT_MAX];

    const u32 tid = threadIdx.x;
    dbg = IVX_DBG_ARG(dbg);
    if (*unsorted == 0) return;                             // input already in region order: nothing to move
    KeyTab kt;
    keytab_load(ix, s_origin, s_last, s_kreg, kt);
    if (tid < ND) gbase[tid] = offs[(u64)tid * nblk + blockIdx.x];
    const u64 lo = (u64)blockIdx.x * chunk;
    const u64 hi = lo + chunk < n ? lo + chunk : n;
    auto body = [&](auto klds_tag) {
    constexpr bool KLDS = decltype(klds_tag)::value;
    for (u64 t0 = lo; t0 < hi; t0 += PA_TILE) {
        if (tid < ND) dstart[tid] = 0;
        __syncthreads();
        u64 se[PA_I]; u32 dig[PA_I], lrank[PA_I];
        u32 kk[PA_I]; i32 qs[PA_I], qe[PA_I];
#pragma unroll
        for (int v = 0; v < PA_I / 4; v++) {
            u32 k4[4]; i32 s4[4], e4[4];
            if (dbg & 2) { for (int j = 0; j < 4; j++) { k4[j] = (tid + j) % 24; s4[j] = (i32)((tid * 977u + j * 131071u + (u32)t0 * 7u) % 40000000u); e4[j] = s4[j] + 100; } }
            else load4<VEC>(pkey, ps, pe, t0 + ((u64)v * PA_T + tid) * 4, hi, k4, s4, e4);
#pragma unroll
            for (int j = 0; j < 4; j++) { kk[v * 4 + j] = k4[j]; qs[v * 4 + j] = s4[j]; qe[v * 4 + j] = e4[j]; }
        }
#pragma unroll
        for (int k = 0; k < PA_I; k++) {
            qs[k] = (i32)((u32)qs[k] + adj); qe[k] = (i32)((u32)qe[k] - adj);
            se[k] = (u64)(u32)qs[k] | ((u64)(u32)qe[k] << 32);
            u32 d = region_of<KLDS>(ix, kt, s_origin, s_last, s_kreg, kk[k], qs[k]);
            if (SPLIT && d != NO_REGION) {                       // digit = super-region; region % G rides along in the bits above it
                const u32 sup = (u32)(((u64)d * split.y) >> 32);
                d = sup | ((d - sup * split.x) << 10);
            }
            dig[k] = d;
            lrank[k] = lds_count_up(dstart, SPLIT ? (d & 1023u) : d, d != NO_REGION);
        }
        __syncthreads();
        u32 tot;
        const u32 mine = tid < ND ? dstart[tid] : 0u;
        const u32 ds = block_excl_scan<u32, PA_T>(mine, scan_lds, &tot);     // (barriers inside: every counter is read before any is overwritten)
        if (tid < ND) dstart[tid] = ds;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PA_I; k++) {
            if (dig[k] != NO_REGION) {
                const u32 pos = dstart[SPLIT ? (dig[k] & 1023u) : dig[k]] + lrank[k];
                r_se[pos] = se[k];
                r_slot[pos] = (unsigned short)(((k / 4) * PA_T + tid) * 4 + (k % 4));
                r_dig[pos] = (DigT)dig[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PA_I; k++) {
            const u32 j = k * PA_T + tid;
            if (j < tot) {
                const u32 dd = r_dig[j];
                const u32 d = SPLIT ? (dd & 1023u) : dd;
                const u64 g = (u64)gbase[d] + (j - dstart[d]);
                if (SPLIT) out_sub[g] = (unsigned char)(dd >> 10);
                if (!(dbg & 1)) {
                    const RowT row = (RowT)(t0 - (sizeof(RowT) == 2 ? lo : 0) + r_slot[j]);
                    if (dbg & 8) { __builtin_nontemporal_store(r_se[j], &out_se[g]); __builtin_nontemporal_store(row, &out_row[g]); }
                    else { out_se[g] = r_se[j]; out_row[g] = row; }
                }
            }
        }
        __syncthreads();
        if (tid < ND) gbase[tid] += mine;
    }
    };
    KEYTAB_DISPATCH(kt, body);
}

// ------------------------------------------------------------------ one-pass partition into region pages
// The two-pass partition above reads the probe rows twice (histogram, then scatter) because a region's rows must end
// up contiguous, and where a region starts depends on every other tile.  Here a region's rows go to PAGES instead
// (2^lgpg rows each, taken from one pool on demand), so a tile needs nothing from the others but a position in its
// regions' row streams:
//   - the tile's rows are ranked by region in LDS as before; rows that cannot match anything are dropped first:
//     unknown keys, keys without build rows, rows that end before the key's first start, and rows whose blocks of
//     the build side's occupancy bitmap are all empty (filter_pass; the bitmap sits in L2, the rows stream past it);
//   - the thread that owns region r reserves the tile's run in r's stream with ONE returning atomicAdd on rcur[r]
//     (virtual row numbers v .. v+c-1); virtual page p of region r lives wherever ptab[r][p] says.  The tile whose run
//     holds a page's first row takes a page from the pool (atomicAdd on *pool_next) and publishes it; a tile that finds
//     the entry still empty polls it -- the publisher never waits for anything between its reservation and the
//     publication, so the poll ends (the entries are 4-byte granules written and read at agent scope);
//   - rows leave LDS as contiguous runs as before, into their pages.
// The probe kernels read region r's rows through the same table (PAGED).  Order inside a region is free, as everywhere.
// per-key tables of the one-pass partition.  KLDS (at most KT_MAX keys): cached in LDS; otherwise read from the index.
// The two sets are never mixed in one pointer: a pointer that may be LDS or global becomes flat loads with full waits.
struct KeyTab2 {
    const i32 *s_origin; const u32 *s_span, *s_kreg, *s_fbase;      // LDS; kreg = 0xFFFFFFFF: key has no build rows
    const u32 *fbits;
    u32 nkeys, sh0, cs, fg, rcells;
    u64 rmul;
};

template <bool KLDS>
__device__ __forceinline__ void keytab2_load(const JoinIndexView &ix, i32 *s_origin, u32 *s_span, u32 *s_kreg, u32 *s_fbase, KeyTab2 &kt, bool use_filter)
{
    kt.nkeys = ix.nkeys; kt.sh0 = ix.hdr[HDR_SH0]; kt.cs = ix.hdr[HDR_CS]; kt.fg = use_filter ? ix.hdr[HDR_FG] : 0xFFFFFFFFu;
    kt.rmul = (u64)ix.hdr[HDR_RMUL_LO] | ((u64)ix.hdr[HDR_RMUL_HI] << 32);
    kt.rcells = ix.hdr[HDR_RCELLS];
    kt.fbits = ix.fbits;
    if (KLDS) {
        for (u32 k = threadIdx.x; k < ix.nkeys; k += blockDim.x) {
            s_origin[k] = ix.origin[k]; s_span[k] = ix.span[k];
            s_kreg[k] = ix.kcnt[k] ? ix.kreg[k] : 0xFFFFFFFFu;
            s_fbase[k] = kt.fg != 0xFFFFFFFFu ? ix.fbase[k] : 0u;
        }
    }
    kt.s_origin = s_origin; kt.s_span = s_span; kt.s_kreg = s_kreg; kt.s_fbase = s_fbase;
}

// Which region a row is routed to, in two steps so that a thread can have the bitmap gathers of all its rows in flight
// together (a branch on one row's looked-up word would make the next row's gather wait for it):
//   route_prep   region of the row (NO_REGION: unknown key, key without build rows, row ends before the key's first
//                start) and the position of its window of the occupancy bitmap: first bit | (blocks - 1) << 26
//                (0x3F blocks-1 = more than 32 blocks: not tested)
//   route_test   whether any block of the window is set, given the 8 bytes that start at the 4-byte word holding the
//                window's first bit (up to 32 blocks always fit)
// PK (8-byte routed rows, hdr[HDR_PK24]): a routed row is ONE word,
//     bits  0..23  start inside its region          bits 24..31  length (end - start), low 8 bits
//     bits 32..    row id (rowbits bits)             bits 32+rowbits..63  length, the bits above the low 8
// so the fewer rows a batch has, the longer a row may be (100 M rows: 27 bits of row id, lengths up to 8190).  A length
// field of all ones marks a row that does not fit -- it starts outside its region's coordinates (before the key's first
// or behind its last start), is too long, or has end < start: the probe reads such a row's coordinates from the input
// columns by its row id.  `packed` = start | length << 24 as a 64-bit value, or PK_ESCAPE.
constexpr u64 PK_ESCAPE = ~0ull;
// the length field's all-ones value: 8 bits plus the row id's spare bits, at most 16 (the host passes rowbits = 32, i.e. no
// spare bits, when the occupancy bitmap is in use: the partition kernel then has no register to carry the upper bits in)
__host__ __device__ __forceinline__ u32 pk_maxlen(u32 rowbits) { const u32 spare = rowbits >= 32 ? 0u : 32u - rowbits; return (1u << (8u + (spare > 8u ? 8u : spare))) - 1u; }
template <bool KLDS, bool FILT, bool PK>
__device__ __forceinline__ u32 route_prep(const JoinIndexView &ix, const KeyTab2 &kt, u32 k, i32 qs, i32 qe, u32 &fpos, u64 &packed, u32 maxlen)
{
    const bool kok = k < kt.nkeys;
    const u32 kk = kok ? k : 0u;
    const u32 kreg = KLDS ? kt.s_kreg[kk] : (ix.kcnt[kk] ? ix.kreg[kk] : 0xFFFFFFFFu);
    const i32 o = KLDS ? kt.s_origin[kk] : ix.origin[kk];
    const u32 span = KLDS ? kt.s_span[kk] : ix.span[kk];
    // qe - o and qs - o in 32 bits: exact as unsigned numbers whenever they are not negative, which one signed compare tells
    // (the partition is bound by the instructions it issues as much as by its LDS phases; 64-bit differences, shifts and
    // clamps were a sixth of them)
    const bool hi_ok = qe >= o;                                     // else: every build row of the key starts behind qe
    const bool d_ok = qs >= o;
    const u32 hi32 = (u32)qe - (u32)o, d32 = (u32)qs - (u32)o;
    const bool ok = kok & (kreg != 0xFFFFFFFFu) & hi_ok;
    fpos = 0;
    if (FILT) {
        const u32 lastb = (span >> kt.fg) + 1u;                     // the overflow block
        const u32 x0 = d_ok ? d32 >> kt.fg : 0u, x1 = hi_ok ? hi32 >> kt.fg : 0u;
        const u32 c0 = x0 > lastb ? lastb : x0, c1 = x1 > lastb ? lastb : x1;
        const u32 b0 = c0 < c1 ? c0 : c1, b1 = c0 < c1 ? c1 : c0;   // (a row with end < start matches build rows that contain [end, start])
        const u32 nb1 = b1 - b0;                                    // blocks - 1
        const u32 fb = KLDS ? kt.s_fbase[kk] : ix.fbase[kk];
        fpos = ok ? (fb + b0) | ((nb1 > 31u ? 0x3Fu : nb1) << 26) : 0u;
    }
    const u32 last = span >> kt.sh0;
    const u32 cc = d_ok ? d32 >> kt.sh0 : 0u;
    const u32 c = cc > last ? last : cc;
    const u32 rin = kt.cs != 0xFFFFFFFFu ? c >> kt.cs : (u32)(((u64)c * kt.rmul) >> 40);    // region inside the key
    packed = PK_ESCAPE;
    if (PK) {
        // for d >= 0 the region's first coordinate (rin * R << sh0 <= span) is at most d, and for end >= start the length
        // is below 2^32
        const u32 rel = d32 - ((rin * kt.rcells) << kt.sh0);
        const u32 len = (u32)qe - (u32)qs;
        if (d_ok && rel < (1u << 24) && qe >= qs && len < maxlen) packed = (u64)rel | ((u64)len << 24);
    }
    return ok ? kreg + rin : NO_REGION;
}

__device__ __forceinline__ bool route_test(u32 fpos, u64 win)
{
    const u32 nb1 = fpos >> 26;
    const u64 m = (2ull << (nb1 & 31u)) - 1ull;
    return (nb1 == 0x3Fu) | (((win >> (fpos & 31u)) & m) != 0);
}

struct PageTab { u32 *ptab; u32 pstride, lgpg; };                 // [region][page slot] -> pool page + 1 (0 = not there yet)

__device__ __forceinline__ u32 page_wait(u32 *slot)
{
    u32 v;
    while ((v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(2);
    return v - 1u;
}

// four consecutive rows per lane, streamed past the caches' keep-lists (the occupancy bitmap should stay in L2)
template <bool VEC>
__device__ __forceinline__ void load4nt(const u32 *__restrict__ pkey, const i32 *__restrict__ ps, const i32 *__restrict__ pe,
                                        u64 i, u64 hi, u32 (&k)[4], i32 (&s)[4], i32 (&e)[4])
{
    if (VEC && i + 4 <= hi) {
        typedef u32 __attribute__((ext_vector_type(4))) v4u;
        typedef i32 __attribute__((ext_vector_type(4))) v4i;
        const v4u kv = pkey ? __builtin_nontemporal_load(reinterpret_cast<const v4u *>(pkey + i)) : v4u{0u, 0u, 0u, 0u};
        const v4i sv = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(ps + i));
        const v4i ev = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(pe + i));
        k[0] = kv.x; k[1] = kv.y; k[2] = kv.z; k[3] = kv.w;
        s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w;
        e[0] = ev.x; e[1] = ev.y; e[2] = ev.z; e[3] = ev.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool ok = i + j < hi;
            k[j] = ok ? (pkey ? pkey[i + j] : 0u) : 0xFFFFFFFFu;
            s[j] = ok ? ps[i + j] : 0;
            e[j] = ok ? pe[i + j] : 0;
        }
    }
}

// ND digits (256 / 1024), T threads, I rows per thread and tile (tile = T * I rows <= one page).  Packed rows need no
// slot array in LDS: two 512-thread workgroups with 8192-row tiles then share a CU, and one's loads overlap the other's
// LDS phases (the kernel is bound by those phases, not by HBM: 8-byte instead of 12-byte rows alone changed nothing)
// PK: a routed row is ONE 8-byte word, (start inside its region | length) and the row id (route_prep), instead of
// (start, end) in one array and the row id in another: a third fewer bytes written here and read by the probe
template <bool VEC, int ND, int I, bool KLDS, bool FILT, bool PK, int T = PA_T>
__global__ __launch_bounds__(T, 4) void k_part_onepass(JoinIndexView ix, const u32 *__restrict__ pkey, const i32 *__restrict__ ps,
                                                       const i32 *__restrict__ pe, u64 n, u32 chunk, u32 *__restrict__ rcur, PageTab pt,
                                                       u32 *pool_next, u64 *__restrict__ out_se, u32 *__restrict__ out_row, u32 rowbits,
                                                       u32 adj = 0, uint2 *__restrict__ vtab = nullptr)
{
    // adj = 1: the UDTFs' strict mode shrinks the query to [start+1, end-1] before anything else (interval_tree.rs:185-188)
    // vtab (per-row-value operators): [tile][region] -> (virtual start, rows) of the tile's run, for the un-permute
    constexpr int TILE = T * I;
    static_assert(ND <= T, "one thread per region");
    __shared__ u64 r_se[TILE];
    __shared__ unsigned short r_slot[PK ? 1 : TILE];            // (packed rows carry their row id with them)
    using DigT = typename std::conditional<(ND > 256), unsigned short, unsigned char>::type;
    __shared__ DigT r_dig[TILE];
    __shared__ u32 dstart[ND];
    __shared__ uint2 wtab[ND];                                   // per region, for the write-out: {virtual row of the run - its LDS start, first pool page | its page slot << 16}
    __shared__ u32 scan_lds[T / IVX_WAVE + 1];
    __shared__ i32 s_origin[KT_MAX];
    __shared__ u32 s_span[KT_MAX], s_kreg[KT_MAX], s_fbase[KT_MAX];

    const u32 tid = threadIdx.x;
    KeyTab2 kt;
    keytab2_load<KLDS>(ix, s_origin, s_span, s_kreg, s_fbase, kt, FILT);
    const u32 pmask = (1u << pt.lgpg) - 1u;
    const u64 lo = (u64)blockIdx.x * chunk;
    const u64 hi = lo + chunk < n ? lo + chunk : n;
    for (u64 t0 = lo; t0 < hi; t0 += TILE) {
        if (tid < ND) dstart[tid] = 0;
        __syncthreads();
        u64 se[PK ? 1 : I]; u32 plo[PK ? I : 1], dig[I];     // PK: low word of the packed row (the length's bits above its low 8 ride in dig)
        // the tile's rows in chunks of CH per thread (all of them, or eight at a time when a thread holds sixteen: the raw
        // columns of sixteen rows plus their routed form do not fit the registers)
        constexpr int CH = (I % 8 == 0 && I > 8) ? 8 : I;
#pragma unroll
        for (int c0 = 0; c0 < I; c0 += CH) {
            // a chunk's row loads first, then all of its bitmap gathers: straight-line code (no per-row branches), so that
            // the loads of a stage are in flight together
            u32 kk[CH]; i32 qs[CH], qe[CH];
            if (VEC && t0 + TILE <= hi) {
                typedef u32 __attribute__((ext_vector_type(4))) v4u;
                typedef i32 __attribute__((ext_vector_type(4))) v4i;
#pragma unroll
                for (int v = 0; v < CH / 4; v++) {
                    const u64 i = t0 + ((u64)(c0 / 4 + v) * T + tid) * 4;
                    const v4u kv = pkey ? __builtin_nontemporal_load(reinterpret_cast<const v4u *>(pkey + i)) : v4u{0u, 0u, 0u, 0u};
                    const v4i sv = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(ps + i));
                    const v4i ev = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(pe + i));
                    kk[v * 4] = kv.x; kk[v * 4 + 1] = kv.y; kk[v * 4 + 2] = kv.z; kk[v * 4 + 3] = kv.w;
                    qs[v * 4] = sv.x; qs[v * 4 + 1] = sv.y; qs[v * 4 + 2] = sv.z; qs[v * 4 + 3] = sv.w;
                    qe[v * 4] = ev.x; qe[v * 4 + 1] = ev.y; qe[v * 4 + 2] = ev.z; qe[v * 4 + 3] = ev.w;
                }
            } else {
#pragma unroll
                for (int v = 0; v < CH / 4; v++) {
                    u32 k4[4]; i32 s4[4], e4[4];
                    load4nt<false>(pkey, ps, pe, t0 + ((u64)(c0 / 4 + v) * T + tid) * 4, hi, k4, s4, e4);
#pragma unroll
                    for (int j = 0; j < 4; j++) { kk[v * 4 + j] = k4[j]; qs[v * 4 + j] = s4[j]; qe[v * 4 + j] = e4[j]; }
                }
            }
            u32 fpos[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) {
                u64 packed;
                const u32 maxlen = pk_maxlen(rowbits);
                qs[k] = (i32)((u32)qs[k] + adj); qe[k] = (i32)((u32)qe[k] - adj);
                dig[c0 + k] = route_prep<KLDS, FILT, PK>(ix, kt, kk[k], qs[k], qe[k], fpos[k], packed, maxlen);
                if (PK) {
                    const u32 lenf = packed == PK_ESCAPE ? maxlen : (u32)(packed >> 24);
                    plo[PK ? c0 + k : 0] = (packed == PK_ESCAPE ? 0u : (u32)packed & 0xFFFFFFu) | (lenf << 24);
                    fpos[k] = FILT ? fpos[k] : (lenf >> 8);       // (parked until the row's rank is known; with the bitmap in use lengths keep to 8 bits)
                } else se[PK ? 0 : c0 + k] = (u64)(u32)qs[k] | ((u64)(u32)qe[k] << 32);
            }
            if (FILT) {
                u64 win[CH];
#pragma unroll
                for (int k = 0; k < CH; k++) __builtin_memcpy(&win[k], kt.fbits + ((fpos[k] & 0x3FFFFFFu) >> 5), sizeof(u64));
#pragma unroll
                for (int k = 0; k < CH; k++) dig[c0 + k] = route_test(fpos[k], win[k]) ? dig[c0 + k] : NO_REGION;
            }
#pragma unroll
            for (int v = 0; v < CH / 4; v++) {                  // region and rank share a register from here on (10 + 14 bits)
                const u32 d4[4] = {dig[c0 + v * 4], dig[c0 + v * 4 + 1], dig[c0 + v * 4 + 2], dig[c0 + v * 4 + 3]};
                u32 lr[4];
                lds_count_up4(dstart, d4, lr);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int k = v * 4 + u;
                    const u32 lhi = (PK && !FILT) ? fpos[k] << 24 : 0u;
                    dig[c0 + k] = dig[c0 + k] == NO_REGION ? NO_REGION : (dig[c0 + k] | (lr[u] << 10) | lhi);
                }
            }
            if (CH != I) asm volatile("" ::: "memory");         // (keeps the next chunk's loads from being hoisted above this chunk's work)
        }
        __syncthreads();
        // ---- reserve the tile's run in every region's row stream; take / look up the pages it touches
        const u32 mine = tid < ND ? dstart[tid] : 0u;
        u32 v = 0;
        if (mine) v = atomicAdd(&rcur[tid], mine);
        if (vtab && tid < ND) vtab[(u64)(t0 / TILE) * ND + tid] = make_uint2(v, mine);
        u32 tot;
        const u32 ds = block_excl_scan<u32, T>(mine, scan_lds, &tot);     // (barriers inside: every counter is read before any is overwritten)
        if (tid < ND) dstart[tid] = ds;
        __syncthreads();
        // (the returned v is first needed after the LDS re-order below, which runs while the atomics are in flight)
#pragma unroll
        for (int k = 0; k < I; k++) {
            if (dig[k] != NO_REGION) {
                const u32 d = dig[k] & 1023u;
                const u32 pos = dstart[d] + ((dig[k] >> 10) & 0x3FFFu);
                if (PK) {
                    const u32 row = (u32)(t0 + (u32)(((k / 4) * T + tid) * 4 + (k % 4)));
                    r_se[pos] = (u64)plo[PK ? k : 0] | ((u64)(row | (rowbits < 32 ? (dig[k] >> 24) << rowbits : 0u)) << 32);
                } else {
                    r_se[pos] = se[PK ? 0 : k];
                    r_slot[pos] = (unsigned short)(((k / 4) * T + tid) * 4 + (k % 4));
                }
                r_dig[pos] = (DigT)d;
            }
        }
        if (mine) {
            u32 *row = pt.ptab + (u64)tid * pt.pstride;
            const u32 p0 = v >> pt.lgpg, p1 = (v + mine - 1u) >> pt.lgpg;       // TILE <= page: at most one page border inside the run
            const bool own0 = (v & pmask) == 0u, own1 = p1 != p0;
            u32 got = 0;
            if (own0 || own1) {
                got = atomicAdd(pool_next, (own0 ? 1u : 0u) + (own1 ? 1u : 0u));
                if (own0) __hip_atomic_store(&row[p0], got + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (own1) __hip_atomic_store(&row[p1], got + (own0 ? 2u : 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wtab[tid] = make_uint2(v - ds, (own0 ? got : page_wait(&row[p0])) | (p0 << 16));    // (pool pages and page slots number at most ~5000: host)
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < I; k++) {
            const u32 j = k * T + tid;
            if (j < tot) {
                const u32 d = r_dig[j];
                const uint2 w = wtab[d];                                        // (one 8-byte LDS read instead of three lookups)
                const u32 x = w.x + j;                                          // virtual row number in region d
                u32 pg = w.y & 0xFFFFu;
                if ((x >> pt.lgpg) != (w.y >> 16)) pg = page_wait(pt.ptab + (u64)d * pt.pstride + (x >> pt.lgpg));   // the run's second page
                const u64 g = ((u64)pg << pt.lgpg) + (x & pmask);
                out_se[g] = r_se[j];
                if (!PK) out_row[g] = (u32)(t0 + r_slot[j]);
            }
        }
        __syncthreads();
    }
}

// first routed row of every region (exclusive scan of the regions' row counts; nreg <= IVX_MAXREG_WIDE), as the probe
// kernels' region table
__global__ __launch_bounds__(1024) void k_page_bounds(const u32 *__restrict__ rcur, u32 nreg, u32 *__restrict__ rfirst)
{
    __shared__ u32 red[1024 / IVX_WAVE + 1];
    const u32 t = threadIdx.x;
    u32 tot;
    const u32 ex = block_excl_scan<u32, 1024>(t < nreg ? rcur[t] : 0u, red, &tot);
    if (t < nreg) rfirst[t] = ex;
    if (t == 0) rfirst[nreg] = tot;
}

// ------------------------------------------------------------------ region probe

constexpr int RP_T = 1024;                 // one workgroup per CU (LDS-bound), 16 wavefronts
constexpr int RP_W = RP_T / IVX_WAVE;
#ifndef IVX_RP_B
#define IVX_RP_B 8
#endif
#ifndef IVX_RP_RING
#define IVX_RP_RING 512
#endif
#ifndef IVX_RP_ECAP
#define IVX_RP_ECAP 6144
#endif
constexpr int RP_B = IVX_RP_B;             // probe rows per lane per wavefront batch (fill: 8, 4, 2 or 1 by match density)
constexpr u32 RP_HALO = IVX_RP_HALO;       // slice cells past the region's last cell
constexpr u32 RP_CCAP = 8192 + RP_HALO + 2;
constexpr u32 RP_ECAP = IVX_RP_ECAP;       // entries staged per slice
constexpr u32 RP_RING = IVX_RP_RING;       // per-wavefront ring of staged pairs: two consecutive rounds must fit (power of two)
constexpr u32 RP_NSLOT = 4;                // rounds whose reservation state is kept (see round_publish)
constexpr u32 RP_NPG = 64;                 // page ids of one region segment kept in LDS (paged rows)
constexpr u32 RP_GRID = 256;                // fill pass: one workgroup per CU (LDS-bound)
#ifndef IVX_RP_VGRID
#define IVX_RP_VGRID 512
#endif
constexpr u32 RP_VGRID = IVX_RP_VGRID;       // row shares ("virtual workgroups"); the count pass runs two per CU

struct Slice {
    const JoinIndexView *ix;
    const unsigned short *s_off; const u64 *s_ent; const u32 *s_row;
    u32 sh0, nlev, k, lb, slo, shi, e0, ncell0; bool inlds, upper, lev0;
    i32 origin; u32 span;
    i32 rbase;                  // coordinate of the region's first cell (packed rows hold their start relative to it)
    // packed rows (start relative to rbase, length) find their cells in 32-bit arithmetic when `fast`: the whole region is
    // one LDS-resident level.  off = cells between the slice's first cell and the region's (0 or 1), cmax = the key's last
    // cell and ncm1 = the slice's last cell, both relative to the slice
    bool fast; u32 off, cmax, ncm1;
};

// every match of one probe row: f(v, is_slot, start, end) -- v is a slot of the staged slice (build row =
// s_row[v]) when is_slot, else the build row itself (general path); start/end are the match's coordinates
template <class F>
__device__ __forceinline__ void probe_row(const Slice &S, i32 qs, i32 qe, F &&f)
{
    const JoinIndexView &ix = *S.ix;
    const i64 hi64 = (i64)qe - (i64)S.origin;
    if (hi64 < 0) return;
    if (S.lev0) {
        const u32 ncell = S.ncell0;
        const i64 lo64 = (i64)qs - ((i64)1 << S.sh0) + 1 - (i64)S.origin;
        const i64 bl = lo64 <= 0 ? 0 : (lo64 >> S.sh0);
        if (bl < (i64)ncell) {
            const u32 blo = (u32)bl;
            const i64 bh = hi64 >> S.sh0;
            const u32 bhi = bh >= (i64)ncell ? ncell - 1u : (u32)bh;
            if (blo <= bhi) {
                if (S.inlds && blo >= S.slo && bhi < S.shi) {
                    const u32 a = S.s_off[blo - S.slo], b = S.s_off[bhi + 1 - S.slo];
                    for (u32 j = a; j < b; j++) {
                        const u64 x = S.s_ent[j];
                        if ((i32)(u32)x <= qe && (i32)(u32)(x >> 32) >= qs) f(j, true, (i32)(u32)x, (i32)(u32)(x >> 32));
                    }
                } else {
                    const u32 a = ix.binstart[S.lb + blo], b = ix.binstart[S.lb + bhi + 1];
                    for (u32 j = a; j < b; j++) {
                        const ivx_ent x = ix.ent[j];
                        if (x.s <= qe && x.e >= qs) f(x.row, false, x.s, x.e);
                    }
                }
            }
        }
    }
    if (S.upper) {                                              // long-interval levels: global reads
        for (u32 l = 1; l < S.nlev; l++) {
            if (ix.hdr[HDR_LEVCNT + l] == 0) continue;
            const u32 sh = S.sh0 + IVX_LSTEP * l;
            u32 blo = 0, bhi = 0;
            if (sh < 32) {
                const u32 ncell = (S.span >> sh) + 1u;
                const i64 lo64 = (i64)qs - ((i64)1 << sh) + 1 - (i64)S.origin;
                const i64 bl = lo64 <= 0 ? 0 : (lo64 >> sh);
                const i64 bh = hi64 >> sh;
                if (bl >= (i64)ncell) continue;
                blo = (u32)bl;
                bhi = bh >= (i64)ncell ? ncell - 1u : (u32)bh;
                if (blo > bhi) continue;
            }
            const u32 base = ix.lbase[(u64)l * ix.nkeys + S.k];
            const u32 a = ix.binstart[base + blo], b = ix.binstart[base + bhi + 1];
            for (u32 j = a; j < b; j++) {
                const ivx_ent x = ix.ent[j];
                if (x.s <= qe && x.e >= qs) f(x.row, false, x.s, x.e);
            }
        }
    }
}

// The same for a packed row (rel = start - S.rbase < 2^24, len = end - start; ok = the row really is in that form): on a
// `fast` slice its cells follow from two shifts -- (start - origin) = (region's first cell) * cell + rel -- instead of
// the 64-bit coordinate arithmetic above (the walk is bound by the instructions it issues, and those were a fifth of them)
template <class F>
__device__ __forceinline__ void probe_row_rel(const Slice &S, u32 rel, u32 len, bool ok, i32 qs, i32 qe, F &&f)
{
    if (S.fast && ok) {
        const u32 t = ((rel + 1u) >> S.sh0) + S.off;                  // first cell a matching build row can start in: one cell back
        const u32 bl = t ? t - 1u : 0u;
        u32 bh = ((rel + len) >> S.sh0) + S.off;
        bh = bh < S.cmax ? bh : S.cmax;
        if (bh < S.ncm1) {
            if (bl <= bh) {
                const u32 a = S.s_off[bl], b = S.s_off[bh + 1u];
                for (u32 j = a; j < b; j++) {
                    const u64 x = S.s_ent[j];
                    if ((i32)(u32)x <= qe && (i32)(u32)(x >> 32) >= qs) f(j, true, (i32)(u32)x, (i32)(u32)(x >> 32));
                }
            }
            return;
        }
    }
    probe_row(S, qs, qe, f);
}

// ------------------------------------------------------------------ shared pieces of the probe kernels

struct ProbeLds {
    unsigned short *s_off; u64 *s_ent; u32 *s_row;
    // fill pass only
    u64 (*s_q)[RP_RING];                          // per-wavefront ring of staged (build row, probe row) pairs
    u32 *s_wpos;                                  // per-wavefront ring write position (running, never reset)
    u32 (*s_wcnt)[RP_W];                          // [RP_NSLOT] pairs each wavefront staged in a round
    unsigned long long *s_base;                   // [RP_NSLOT] output position reserved for the round
    u32 *s_arrive, *s_ready;                      // [RP_NSLOT] wavefronts arrived / round tag once s_base is valid
};

__device__ __forceinline__ void slice_init(const JoinIndexView &ix, Slice &S, const ProbeLds &L)
{
    S.ix = &ix; S.s_off = L.s_off; S.s_ent = L.s_ent; S.s_row = L.s_row;
    S.sh0 = ix.hdr[HDR_SH0]; S.nlev = ix.hdr[HDR_NLEV];
    S.upper = false;
    for (u32 l = 1; l < S.nlev; l++) S.upper |= ix.hdr[HDR_LEVCNT + l] != 0;
    S.lev0 = ix.hdr[HDR_LEVCNT] != 0;
}

// stage region r's slice of the index in LDS (all threads of the workgroup; barriers inside)
__device__ __forceinline__ void slice_load(const JoinIndexView &ix, Slice &S, const ProbeLds &L, u32 r, bool reload)
{
    const u32 tid = threadIdx.x;
    __syncthreads();
    const ivx_regdesc d = ix.rdesc[r];                                  // built by k_join_regdesc
    S.k = d.k; S.origin = d.origin; S.span = d.span; S.lb = d.lb;
    S.ncell0 = (S.span >> S.sh0) + 1u;
    S.slo = d.slo; S.shi = d.shi; S.e0 = d.e0; S.rbase = d.rbase;
    const u32 ne = d.ne;
    const u32 nc = S.shi - S.slo + 1u;
    S.inlds = ne <= RP_ECAP && nc <= RP_CCAP;
    S.fast = S.inlds && S.lev0 && !S.upper;
    S.off = (u32)(((i64)S.rbase - (i64)S.origin) >> S.sh0) - S.slo;
    S.cmax = S.ncell0 - 1u - S.slo; S.ncm1 = S.shi - S.slo;
    if (S.inlds && reload) {
        for (u32 c0 = 0; c0 < nc; c0 += RP_T * 4) {
            u32 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const u32 c = c0 + u * RP_T + tid; v[u] = c < nc ? ix.binstart[S.lb + S.slo + c] : 0u; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const u32 c = c0 + u * RP_T + tid; if (c < nc) L.s_off[c] = (unsigned short)(v[u] - S.e0); }
        }
        for (u32 j0 = 0; j0 < ne; j0 += RP_T * 4) {
            ivx_ent x[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const u32 j = j0 + u * RP_T + tid; if (j < ne) x[u] = ix.ent[S.e0 + j]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const u32 j = j0 + u * RP_T + tid;
                if (j < ne) { L.s_ent[j] = (u64)(u32)x[u].s | ((u64)(u32)x[u].e << 32); L.s_row[j] = x[u].row; }
            }
        }
    }
    __syncthreads();
}

// One wave batch: B rows per lane (bit q of okmask says whether the lane holds a row in slot q).
// Count pass: returns the lane's pair count.
// Fill pass: every match takes the next slot of the wavefront's LDS staging buffer with an LDS atomic on
// the wavefront's own running counter (lanes of one instruction are serialised by the LDS unit and get
// distinct slots), so no per-row match stash, prefix sum or second walk is needed.  Returns the batch's
// pair count; direct = true if it did not fit the ring (see batch_write_direct).
// PK: rel / len hold the rows' packed form (valid where relmask has the row's bit)
template <bool FILL, int B, bool PK = false>
__device__ __forceinline__ u32 batch_walk(const Slice &S, const ProbeLds &L, const i32 (&qs)[B], const i32 (&qe)[B],
                                          const u32 (&rowv)[B], u32 okmask, u32 wv, u32 ring_tail, u32 &ring_start, bool &direct, int dbg,
                                          const u32 (&rel)[B], const u32 (&len)[B], u32 relmask)
{
    auto walk = [&](int q, auto &&f) {
        if (PK) probe_row_rel(S, rel[q], len[q], (relmask >> q) & 1u, qs[q], qe[q], f);
        else probe_row(S, qs[q], qe[q], f);
    };
    if (!FILL) {
        u32 tsum = 0;
#pragma unroll
        for (int q = 0; q < B; q++) {
            if (!((okmask >> q) & 1u)) continue;
            if (dbg & 4) tsum += (u32)(qs[q] ^ qe[q]) & 1u;
            else walk(q, [&](u32, bool, i32, i32) { tsum++; });
        }
        return tsum;
    }
    u32 *cp = &L.s_wpos[wv];
    const u32 base = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ring_start = base;
#pragma unroll
    for (int q = 0; q < B; q++) {
        if (!((okmask >> q) & 1u)) continue;
        if (dbg & 128) continue;
        walk(q, [&](u32 v, bool sl, i32, i32) {
            const u32 pos = __hip_atomic_fetch_add(cp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (pos - ring_tail < RP_RING && !(dbg & 64))
                L.s_q[wv][pos & (RP_RING - 1)] = (u64)(sl ? S.s_row[v] : v) | ((u64)rowv[q] << 32);
        });
    }
    const u32 wtot = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - base;
    // this round and the previous one must fit the ring together; if not, nothing of this batch counts as
    // staged (the previous round's pairs were never overwritten) and the caller writes the batch directly
    // once the round's output range is known (batch_write_direct)
    direct = base + wtot - ring_tail > RP_RING;
    if (direct) ring_start = base + wtot;
    return wtot;
}

// second walk of a batch that did not fit the staging ring: pairs go straight to their place in the
// output, [g, g + wtot) of the round's reserved range; slots again by LDS atomic
template <int B>
__device__ __forceinline__ void batch_write_direct(const Slice &S, const ProbeLds &L, const i32 (&qs)[B], const i32 (&qe)[B],
                                                   const u32 (&rowv)[B], u32 okmask, u32 wv, u64 g, bool ok, u32 *ob, u32 *op)
{
    u32 *cp = &L.s_wpos[wv];
    const u32 base = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int q = 0; q < B; q++) {
        if (!((okmask >> q) & 1u)) continue;
        probe_row(S, qs[q], qe[q], [&](u32 v, bool sl, i32, i32) {
            const u32 pos = __hip_atomic_fetch_add(cp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - base;
            if (ok) { ob[g + pos] = sl ? S.s_row[v] : v; op[g + pos] = rowv[q]; }
        });
    }
}

// Per-row-output operators on the same slices: the row's value instead of its pairs.
//   RV_COUNT     count_overlaps: 0 if qe < qs (interval_tree.rs:42-44), else the number of build rows the
//                literal predicate matches -- equal to #{starts <= qe} - #{ends < qs} (:45-48) whenever no
//                build row has end < start, which is when the host takes this path
//   RV_COVERAGE  get_coverage over the merged nodes: sum of max(1, min(qe+1, last) - max(qs-1, first)) in
//                wrapping i32 arithmetic (:145-152)
enum { RV_COUNT = 2, RV_COVERAGE = 3, RV_MATCHES = 4 };   // RV_MATCHES: the join's rle_right / exists (plain match count)

__device__ __forceinline__ i32 rv_wadd(i32 a, i32 b) { return (i32)((u32)a + (u32)b); }
__device__ __forceinline__ i32 rv_wsub(i32 a, i32 b) { return (i32)((u32)a - (u32)b); }

template <int KIND, int B, bool PK = false>
__device__ __forceinline__ void batch_rowval(const Slice &S, const i32 (&qs)[B], const i32 (&qe)[B], u32 okmask, u32 (&val)[B],
                                             const u32 (&rel)[B], const u32 (&len)[B], u32 relmask)
{
    auto walk = [&](int q, auto &&f) {
        if (PK) probe_row_rel(S, rel[q], len[q], (relmask >> q) & 1u, qs[q], qe[q], f);
        else probe_row(S, qs[q], qe[q], f);
    };
#pragma unroll
    for (int q = 0; q < B; q++) {
        u32 v = 0;
        if ((okmask >> q) & 1u) {
            if (KIND == RV_COUNT) {
                if (!(qe[q] < qs[q])) walk(q, [&](u32, bool, i32, i32) { v++; });
            } else if (KIND == RV_MATCHES) {
                walk(q, [&](u32, bool, i32, i32) { v++; });
            } else {
                const i32 a = rv_wadd(qe[q], 1), b = rv_wsub(qs[q], 1);
                walk(q, [&](u32, bool, i32 first, i32 last) {
                    const i32 d = rv_wsub(a < last ? a : last, b > first ? b : first);
                    v = (u32)rv_wadd((i32)v, d > 1 ? d : 1);
                });
            }
        }
        val[q] = v;
    }
}

// Fill pass, output reservation without workgroup barriers.  Wavefronts run the rounds of a workgroup
// independently: after its batch of round r a wavefront publishes its staged count and "arrives"; the
// wavefront that arrives LAST adds the 16 counts, reserves the round's output range with ONE global
// atomicAdd and publishes the base with the round's tag.  Nobody waits for that: a wavefront copies the
// pairs of round r out only after it has walked round r+1 (round_copy_out), when the base has long
// arrived.  Every wavefront waits for round r-1's tag before it starts round r+1, so wavefronts are never
// more than one round apart and RP_NSLOT = 4 reservation slots cannot be overwritten while still read.
__device__ __forceinline__ void round_publish(const ProbeLds &L, u32 mine, u32 round, u32 wv, unsigned long long *cursor)
{
    if (lane_id() != 0) return;
    const u32 sl = round % RP_NSLOT;
    L.s_wcnt[sl][wv] = mine;
    const u32 before = __hip_atomic_fetch_add(&L.s_arrive[sl], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (before != RP_W - 1) return;
    u32 tot = 0;
#pragma unroll
    for (int w = 0; w < RP_W; w++) tot += L.s_wcnt[sl][w];
    L.s_base[sl] = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
    __hip_atomic_store(&L.s_arrive[sl], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&L.s_ready[sl], round + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// wait for round `round`'s reservation; returns where this wavefront's pairs of that round go and whether
// the whole round fits the caller's buffers
__device__ __forceinline__ u64 round_wait(const ProbeLds &L, u32 round, u32 wv, u64 cap, bool &fits)
{
    const u32 sl = round % RP_NSLOT;
    while (__hip_atomic_load(&L.s_ready[sl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != round + 1u)
        __builtin_amdgcn_s_sleep(1);
    const u64 base = L.s_base[sl];
    u64 g = base;
    u32 tot = 0;
#pragma unroll
    for (int w = 0; w < RP_W; w++) { const u32 c = L.s_wcnt[sl][w]; if (w < (int)wv) g += c; tot += c; }
    fits = base + tot <= cap;
    return g;
}

// copy this wavefront's `mine` staged pairs of round `round` (ring positions start, start+1, ...) to their
// place in the output with full-width stores
__device__ __forceinline__ void round_copy_out(const ProbeLds &L, u32 mine, u32 start, u32 round, u32 wv,
                                               u32 *ob, u32 *op, u64 cap, int dbg)
{
    bool fits;
    const u64 g = round_wait(L, round, wv, cap, fits);
    if (!mine || !fits || (dbg & 16)) return;
    for (u32 t = lane_id(); t < mine; t += IVX_WAVE) {
        const u64 x = L.s_q[wv][(start + t) & (RP_RING - 1)];
        ob[g + t] = (u32)x; op[g + t] = (u32)(x >> 32);
    }
}

#define IVX_PROBE_LDS(FILL)                                                                          \
    __shared__ unsigned short s_off[RP_CCAP];                                                        \
    __shared__ u64 s_ent[RP_ECAP];                                                                   \
    __shared__ u32 s_row[RP_ECAP];                                                                   \
    __shared__ u64 s_q[FILL ? RP_W : 1][RP_RING];                                                    \
    __shared__ u32 s_wpos[RP_W];                                                                     \
    __shared__ u32 s_wcnt[RP_NSLOT][RP_W];                                                           \
    __shared__ unsigned long long s_base[RP_NSLOT];                                                  \
    __shared__ u32 s_arrive[RP_NSLOT], s_ready[RP_NSLOT];                                            \
    if (threadIdx.x < RP_W) s_wpos[threadIdx.x] = 0;                                                 \
    if (threadIdx.x < RP_NSLOT) { s_arrive[threadIdx.x] = 0; s_ready[threadIdx.x] = 0; }             \
    ProbeLds L{s_off, s_ent, s_row, s_q, s_wpos, s_wcnt, s_base, s_arrive, s_ready};

// ------------------------------------------------------------------ region-major probe (rows scattered by region)
// Persistent workgroups: the partitioned probe rows are cut into equal row shares ("virtual
// workgroups"), a workgroup walks its share region segment by region segment, and inside a segment
// wavefront w owns batches w, w+16, ... of RP_WB rows.
//   MODE 0 (ivx_probe_overlap_count): wavefronts never synchronise; one atomicAdd of the wavefront's
//           total at the end.
//   MODE 1 (fill): single walk, see batch_walk / round_publish / round_copy_out.
//   MODE RV_COUNT / RV_COVERAGE: one 32-bit value per row, written at the row's partitioned position
//           (`ob`), no synchronisation at all; k_unpermute puts the values back in input order.
// rows per lane and wavefront batch of the fill pass, chosen on the device from the pairs expected per ROUTED row
// (`hint` pairs over the rows the partition kept): the same rule as fill_rows_per_lane below
__global__ void k_pick_rows(const u32 *__restrict__ rfirst, u32 nreg, u64 hint, u32 force, u32 *bsel)
{
    const u32 routed = rfirst[nreg];
    const float per_row = (float)((double)hint / (double)(routed ? routed : 1u));
    *bsel = force ? force : per_row <= 0.40f ? 8u : per_row <= 0.8f ? 4u : per_row <= 1.6f ? 2u : 1u;
}

template <int MODE, int B, bool IDENT, bool PAGED = false, bool PK = false>
__global__ __launch_bounds__(RP_T) void k_probe_regions(JoinIndexView ix, const void *__restrict__ rows_a, const void *__restrict__ rows_b,
                                                        const u32 *__restrict__ offs, u32 nblk, u32 vpb,
                                                        u32 *__restrict__ ob, u32 *__restrict__ op, u64 cap,
                                                        unsigned long long *cursor, u32 prow_stride, u32 adj,
                                                        const u32 *unsorted, int dbg, PageTab pt = PageTab{nullptr, 0u, 0u},
                                                        const u32 *bsel = nullptr, const i32 *__restrict__ ps_in = nullptr, const i32 *__restrict__ pe_in = nullptr,
                                                        u32 rowbits = 32, const u32 *only_if_set = nullptr)
{
    constexpr bool FILL = MODE == 1;
    if (only_if_set != nullptr && *only_if_set == 0u) return;         // (the lean fill kernel took this index: see ivx_join_probe_regions)
    const u32 rowmask = rowbits >= 32 ? 0xFFFFFFFFu : (1u << rowbits) - 1u;
    const u32 maxlen = pk_maxlen(rowbits);
    static_assert(!PK || PAGED, "packed rows come from the one-pass partition");
    // PK: rows_a holds 8-byte words (start inside the region | length << 24, row id << 32; PK_ESCAPE: the coordinates are
    // read from the input columns ps_in / pe_in by the row id); rows_b is not used
    if (bsel != nullptr && *bsel != (u32)B) return;                   // (every B is launched; k_pick_rows chose one)
    // IDENT: the input already is in region order (k_part_hist left `unsorted` at 0): the partitioned arrays
    // were never written and row i IS input row i.  Both instantiations are launched; the one whose case
    // does not apply returns at once (the host never waits for the flag).
    if ((unsorted != nullptr && *unsorted == 0) != IDENT) return;
    // rows_a / rows_b: the partitioned (qs,qe) and row-id arrays, or -- IDENT -- the input start and end columns, or --
    // PAGED -- the page pool of the one-pass partition: routed row i of region r (whose first routed row is rf) sits at
    // row_at(i, r, rf) of the pool
    static_assert(!(IDENT && PAGED), "paged rows are never read in place");
    __shared__ u32 s_pg[RP_NPG];                                      // the pages of the region segment being walked
    u32 pg_first = 0;
    // (the page id always comes from LDS: a choice between an LDS and a global pointer becomes a flat load with
    //  vmcnt(0) waits that serialise the row loads)
    auto row_at = [&](u64 i, u32, u64 rf) -> u64 {
        if (!PAGED) return i;
        const u32 x = (u32)(i - rf);
        const u32 pg = s_pg[(x >> pt.lgpg) - pg_first] - 1u;
        return ((u64)pg << pt.lgpg) + (x & ((1u << pt.lgpg) - 1u));
    };
    // (all threads; barriers inside) the page ids of routed rows [lo, c_hi) of region r; c_hi is cut back to what RP_NPG
    // pages hold (the caller then walks the rest of the region as another segment)
    auto pages_load = [&](u64 lo, u64 &c_hi, u32 r, u64 rf) {
        if (!PAGED) return;
        __syncthreads();
        pg_first = (u32)(lo - rf) >> pt.lgpg;
        const u64 lim = (rf + ((u64)(pg_first + RP_NPG) << pt.lgpg)) & ~63ull;   // (a cut falls between two 64-row granules)
        if (c_hi > lim) c_hi = lim;
        const u32 npg = ((u32)(c_hi - 1 - rf) >> pt.lgpg) - pg_first + 1u;
        if (threadIdx.x < npg) {
            u32 v = pt.ptab[(u64)r * pt.pstride + pg_first + threadIdx.x];
            // 0 = the partition never published this page (cannot happen once k_part_onepass has completed; it did in a
            // profiling build whose switch skipped the publication, and the page "0 - 1" then was a wild address: round 2's
            // fault).  Read page 0 instead and tell the host.
            if (v == 0u) { v = 1u; if (cursor != nullptr) atomicOr((unsigned int *)(cursor + 1), 1u); }
            s_pg[threadIdx.x] = v;
        }
        __syncthreads();
    };
    auto row_se = [&](u64 i) -> u64 {
        if (IDENT) return (u64)(u32)((u32)((const i32 *)rows_a)[i] + adj) | ((u64)(u32)((u32)((const i32 *)rows_b)[i] - adj) << 32);
        return ((const u64 *)rows_a)[i];
    };
    auto row_id = [&](u64 i, u64 at) -> u32 { return IDENT ? (u32)i : ((const u32 *)rows_b)[at * prow_stride]; };
    dbg = IVX_DBG_ARG(dbg);
    constexpr u32 WB = IVX_WAVE * B;                                  // rows per wavefront batch
    IVX_PROBE_LDS(FILL)
    const u32 wv = threadIdx.x / IVX_WAVE, ln = lane_id();
    const u32 nreg = ix.hdr[HDR_NREG];
    // first partitioned row of every region, once, in LDS (the share boundaries below search it)
    __shared__ u32 s_rfirst[IVX_MAXREG_WIDE + 2];
    const bool rf_lds = nreg <= IVX_MAXREG_WIDE;                           // (the two-digit scheme has up to 65025 regions: global table)
    if (rf_lds) for (u32 t = threadIdx.x; t <= nreg; t += RP_T) s_rfirst[t] = offs[(u64)t * nblk];
    __syncthreads();
    auto rfirst = [&](u32 r) -> u32 { return rf_lds ? s_rfirst[r] : offs[(u64)r * nblk]; };
    const u64 total_rows = rfirst(nreg);
    const u32 nvb = gridDim.x * vpb;
    u32 loaded_r = 0xFFFFFFFFu;                                       // region whose slice currently sits in LDS
    Slice S;
    slice_init(ix, S, L);
    u32 round = 0;                                                    // fill pass: rounds of this workgroup so far
    u32 pend_mine = 0, pend_start = 0;                                // the previous round's staged pairs (ring range)
    for (u32 vb = blockIdx.x * vpb; vb < (blockIdx.x + 1) * vpb; vb++) {
        u64 lo = total_rows * vb / nvb;
        const u64 hi = total_rows * (vb + 1) / nvb;
        u64 wcur = 0;                                                 // count pass: pairs seen by this wavefront
        if (lo < hi) {
            u32 r;
            {   // last region whose first row is <= lo
                u32 a = 0, b = nreg;
                while (a < b) { const u32 m = (a + b + 1) >> 1; if (rfirst(m) <= lo) a = m; else b = m - 1; }
                r = a;
            }
            for (; lo < hi; r++) {
                const u64 rend = rfirst(r + 1);
                u64 c_hi = hi < rend ? hi : rend;
                if (c_hi <= lo) continue;
                // every wavefront streams one batch of WB rows per round; the next round's rows are in
                // flight while the current batch walks the LDS slice (the first batch while the slice loads)
                u64 nx[B]; u32 nxr[B];
                // a round is RP_W * WB consecutive rows; wavefront w takes the 64-row granules w, w+16, ... of it, so
                // that dense and empty stretches of sorted input are shared evenly by the 16 wavefronts
                u64 b0 = lo + (u64)wv * IVX_WAVE;
                const u64 rf = PAGED ? rfirst(r) : 0;
                pages_load(lo, c_hi, r, rf);
                // PAGED: when the lane's B rows of the round lie inside the segment and inside one pool page -- nearly always --
                // the first row's place in the pool gives the others' (they are RP_W * 64 rows apart): one bounds test and one
                // page lookup per lane and round instead of B.  (Doing the same with a scalar base per wavefront was measured
                // and dropped, fill 641 -> 703 us: the scalar page lookup waits where the vector one overlaps.)
                u64 nat0 = 0, cat0 = 0;                               // pool position of the lane's first row: of the round in flight / being walked
                auto prefetch = [&](u64, u64 bl) -> bool {
                    if (PAGED) {
                        const u64 i0 = bl + ln, i7 = i0 + (u64)(B - 1) * (RP_W * IVX_WAVE);
                        const u32 x0 = (u32)(i0 - rf), x7 = (u32)(i7 - rf);
                        const bool one = i7 < c_hi && (x0 >> pt.lgpg) == (x7 >> pt.lgpg);
                        if (__ballot(!one) == 0) {                     // (uniform: every lane of the wavefront)
                            const u64 at0 = row_at(i0, r, rf);
                            nat0 = at0;
#pragma unroll
                            for (int q = 0; q < B; q++) {
                                const u64 at = at0 + (u64)q * (RP_W * IVX_WAVE);
                                nx[q] = row_se(at);
                                nxr[q] = (FILL && !PK) ? row_id(i0 + (u64)q * (RP_W * IVX_WAVE), at) : 0u;
                            }
                            return true;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < B; q++) {
                        const u64 i = bl + (u64)q * (RP_W * IVX_WAVE) + ln;
                        const u64 at = i < c_hi ? row_at(i, r, rf) : 0;
                        nx[q] = i < c_hi ? row_se(at) : 0;
                        nxr[q] = (FILL && !PK && i < c_hi) ? row_id(i, at) : 0u;
                    }
                    return false;
                };
                bool nfull = prefetch(lo, b0);
                slice_load(ix, S, L, r, r != loaded_r);
                loaded_r = r;
                for (u64 r0 = lo; r0 < c_hi; r0 += (u64)RP_W * WB, b0 += (u64)RP_W * WB) {
                    i32 qs[B], qe[B]; u32 rowv[B];
                    u32 prel[PK ? B : 1], plen[PK ? B : 1], relmask = 0;   // the packed form, for the walk's 32-bit cell arithmetic
                    u32 okmask = 0;
#pragma unroll
                    for (int q = 0; q < B; q++) {
                        if (PK) {
                            const u32 lo32 = (u32)nx[q], hi32 = (u32)(nx[q] >> 32);
                            const u32 len = (lo32 >> 24) | ((rowbits < 32 ? (hi32 >> rowbits) & 0xFFu : 0u) << 8);
                            prel[PK ? q : 0] = lo32 & 0xFFFFFFu; plen[PK ? q : 0] = len;
                            qs[q] = (i32)((u32)S.rbase + (lo32 & 0xFFFFFFu)); qe[q] = (i32)((u32)qs[q] + len); rowv[q] = hi32 & rowmask;
                            nxr[q] = len;                               // (kept for the escape test below; the row id prefetch slot is free here)
                        } else { qs[q] = (i32)(u32)nx[q]; qe[q] = (i32)(u32)(nx[q] >> 32); rowv[q] = nxr[q]; }
                        if (!nfull && b0 + (u64)q * (RP_W * IVX_WAVE) + ln < c_hi) okmask |= 1u << q;
                    }
                    if (nfull) okmask = (1u << B) - 1u;
                    const bool cfull = nfull;
                    cat0 = nat0;
                    if (PK) {                                           // rows that did not fit the packed form (rare)
                        u32 esc = 0;
#pragma unroll
                        for (int q = 0; q < B; q++) if (((okmask >> q) & 1u) && nxr[q] == maxlen) esc |= 1u << q;
                        relmask = okmask & ~esc;
                        if (__any(esc != 0)) {                          // all the gathers first, then their uses: one round trip, not 2 * B
                            i32 ts[B], te[B];
#pragma unroll
                            for (int q = 0; q < B; q++) { const u32 rr = ((esc >> q) & 1u) ? rowv[q] : 0u; ts[q] = ps_in[rr]; te[q] = pe_in[rr]; }
#pragma unroll
                            for (int q = 0; q < B; q++) if ((esc >> q) & 1u) { qs[q] = (i32)((u32)ts[q] + adj); qe[q] = (i32)((u32)te[q] - adj); }
                        }
                    }
                    nfull = prefetch(r0 + (u64)RP_W * WB, b0 + (u64)RP_W * WB);
                    if (MODE >= RV_COUNT) {
                        u32 val[B];
                        if constexpr (PK) batch_rowval<MODE, B, true>(S, qs, qe, okmask, val, (const u32 (&)[B])prel, (const u32 (&)[B])plen, relmask);
                        else batch_rowval<MODE, B, false>(S, qs, qe, okmask, val, (const u32 (&)[B])qs, (const u32 (&)[B])qs, 0u);
#pragma unroll
                        for (int q = 0; q < B; q++) {
                            if (!((okmask >> q) & 1u)) continue;
                            const u64 i = b0 + (u64)q * (RP_W * IVX_WAVE) + ln;
                            if (PAGED) {    // in place: the value takes the low half of the row's packed word, the row id stays above it
                                const u64 at = cfull ? cat0 + (u64)q * (RP_W * IVX_WAVE) : row_at(i, r, rf);
                                ((u64 *)ob)[at] = (u64)val[q] | ((u64)rowv[q] << 32);      // (ob = the page pool itself)
                            } else ob[i] = val[q];
                        }
                        continue;
                    }
                    u32 start = 0;
                    bool direct = false;
                    u32 got;
                    if constexpr (PK) got = batch_walk<FILL, B, true>(S, L, qs, qe, rowv, okmask, wv, pend_start, start, direct, dbg, (const u32 (&)[B])prel, (const u32 (&)[B])plen, relmask);
                    else got = batch_walk<FILL, B, false>(S, L, qs, qe, rowv, okmask, wv, pend_start, start, direct, dbg, (const u32 (&)[B])qs, (const u32 (&)[B])qs, 0u);
                    if (MODE == 0) { wcur += got; continue; }
                    if (!(dbg & 32)) {
                        round_publish(L, got, round, wv, cursor);
                        if (direct) {                          // needs this round's range now: wait for the 16th wavefront
                            bool fits;
                            const u64 g = round_wait(L, round, wv, cap, fits);
                            batch_write_direct<B>(S, L, qs, qe, rowv, okmask, wv, g, fits, ob, op);
                            start += got; got = 0;             // nothing staged; the second walk moved the ring position too
                        }
                        if (round) round_copy_out(L, pend_mine, pend_start, round - 1, wv, ob, op, cap, dbg);
                    }
                    pend_mine = got; pend_start = start; round++;
                }
                lo = c_hi;
                if (PAGED && lo < hi && lo < rend) r--;               // the segment was cut at the page window: same region again
            }
        }
        if (MODE == 0) {
            const u64 tot = wave_sum(wcur);
            if (ln == 0 && tot) atomicAdd(cursor, (unsigned long long)tot);
        }
    }
    if (FILL && round && !(dbg & 32)) round_copy_out(L, pend_mine, pend_start, round - 1, wv, ob, op, cap, dbg);
}

// ------------------------------------------------------------------ lean fill probe (round 3)
// The headline's fill pass again, for the case it always meets: packed 8-byte rows in region pages (k_part_onepass) and an
// index whose every region is one LDS-resident level (hdr[HDR_FAST]).  Same slices, same staging ring and round-level
// output reservation as k_probe_regions<fill>; what differs is what a wavefront executes per row -- that kernel is bound
// by the instructions it issues, 224 per 64 rows:
//  * work is dealt in CHUNKS of 8192 routed rows that never straddle a pool page (chunk k of region r = its virtual rows
//    [8192 k, 8192 (k + 1)); wavefront w owns rows [512 w, 512 (w + 1)) of it), so the rows of a wavefront batch are
//    64 * B consecutive words behind ONE wave-uniform pointer: loads are `scalar base + lane offset + immediate`, with no
//    per-row bounds test, page lookup or 64-bit address arithmetic (a third of the old kernel's instructions);
//  * the walk is the 32-bit packed-row form only, staged as (slot, probe row): the slot -> build row lookup happens once per
//    64 pairs in the copy-out, not per match inside the divergent loop; the cell offsets of all B rows are fetched before
//    the first candidate loop runs;
//  * rows the packed form cannot carry (escapes, rows reaching past the slice's halo) and batches that overflow the ring
//    send their whole batch through the generic walk + direct write of k_probe_regions (cold code, out of line);
//  * the last wavefront to arrive in a round leaves every wavefront's output position, not just the round's base.
constexpr u32 FP_CHUNK = (u32)RP_W * IVX_WAVE * 8u;              // 8192 rows

// first routed row (rfirst) and first chunk (cfirst) of every region; nreg <= IVX_MAXREG_WIDE
__global__ __launch_bounds__(1024) void k_chunk_bounds(const u32 *__restrict__ rcur, u32 nreg, u32 *__restrict__ rfirst, u32 *__restrict__ cfirst)
{
    __shared__ u32 red[1024 / IVX_WAVE + 1];
    const u32 t = threadIdx.x;
    const u32 rows = t < nreg ? rcur[t] : 0u;
    u32 tot;
    const u32 ex = block_excl_scan<u32, 1024>(rows, red, &tot);
    if (t < nreg) rfirst[t] = ex;
    if (t == 0) rfirst[nreg] = tot;
    __syncthreads();
    const u32 ec = block_excl_scan<u32, 1024>((rows + FP_CHUNK - 1) / FP_CHUNK, red, &tot);
    if (t < nreg) cfirst[t] = ec;
    if (t == 0) cfirst[nreg] = tot;
}

// Rows the lean kernel does not take: region, first virtual row, rows -- ONE row the packed form cannot carry (an escape, or
// a row that reaches past its slice's halo; listed by its lane, the rest of its batch goes the fast way), or a whole batch
// that found more pairs than the ring holds (nothing of it stays staged).  k_fill_fast appends them to a list; k_fill_rest
// walks the listed rows afterwards with the generic gather walk -- the lean kernel holds no generic code (and no scratch).
struct FpRest { u32 r, first, cnt, pad; };
// the two lists live in one scratch buffer: batches first (at most one per wavefront batch), then single rows (at most n)
__host__ __device__ __forceinline__ u64 fp_max_batches(u64 n, u32 nreg) { return ((n >> 13) + nreg + 1) * (u64)(RP_W * 8); }

template <int B>
__global__ __launch_bounds__(RP_T) void k_fill_fast(JoinIndexView ix, const u64 *__restrict__ pool, const u32 *__restrict__ rcur,
                                                    const u32 *__restrict__ cfirst, PageTab pt, u32 *__restrict__ ob, u32 *__restrict__ op, u64 cap,
                                                    unsigned long long *cursor, const u32 *bsel, u32 rowbits, FpRest *__restrict__ rest, u64 *__restrict__ rest_rows, u32 *rest_n)
{
    if (bsel != nullptr && *bsel != (u32)B) return;                   // (every B is launched; k_pick_rows chose one)
    if (ix.hdr[HDR_SLOW] != 0u) return;                               // not every region is one LDS-resident level: the general kernel's
    constexpr u32 WB = IVX_WAVE * B;                                  // rows of a wavefront batch
    constexpr u32 SUB = 8u / B;                                       // batches a wavefront makes of its 512 rows of a chunk
    IVX_PROBE_LDS(true)
    __shared__ u32 s_cfirst[IVX_MAXREG_WIDE + 2];
    __shared__ u32 s_wat[RP_NSLOT][RP_W];                             // a wavefront's output position inside its round's range
    const u32 wv = __builtin_amdgcn_readfirstlane(threadIdx.x / IVX_WAVE), ln = lane_id();
    const u32 nreg = ix.hdr[HDR_NREG];
    for (u32 t = threadIdx.x; t <= nreg; t += RP_T) s_cfirst[t] = cfirst[t];
    __syncthreads();
    const u32 nchunk = s_cfirst[nreg];
    const u32 c_lo = (u32)((u64)nchunk * blockIdx.x / gridDim.x), c_hi = (u32)((u64)nchunk * (blockIdx.x + 1) / gridDim.x);
    if (c_lo >= c_hi) return;
    const u32 rowmask = rowbits >= 32 ? 0xFFFFFFFFu : (1u << rowbits) - 1u;
    const u32 maxlen = pk_maxlen(rowbits);
    const u32 pmask = (1u << pt.lgpg) - 1u;
    Slice S;
    slice_init(ix, S, L);
    u32 r_next;                                                       // region of the batch in flight
    {   // last region whose first chunk is <= c_lo
        u32 a = 0, b = nreg;
        while (a < b) { const u32 m = (a + b + 1) >> 1; if (s_cfirst[m] <= c_lo) a = m; else b = m - 1; }
        r_next = a;
    }
    // batch i of this wavefront: its region, first virtual row and row count are wave-uniform, its rows 64 * B consecutive
    // words of one pool page
    const u32 nbatch = (c_hi - c_lo) * SUB;
    u64 nx[B];
    u32 ncnt = 0, nfirst = 0;
    auto prefetch = [&](u32 i) {
        const u32 c = c_lo + i / SUB, sb = i % SUB;
        while (c >= s_cfirst[r_next + 1]) r_next++;
        nfirst = (c - s_cfirst[r_next]) * FP_CHUNK + wv * (8u * IVX_WAVE) + sb * WB;
        const u32 rows = rcur[r_next];
        ncnt = rows > nfirst ? (rows - nfirst < WB ? rows - nfirst : WB) : 0u;
        if (ncnt) {
            u32 pg = pt.ptab[(u64)r_next * pt.pstride + (nfirst >> pt.lgpg)];
            if (pg == 0u) { pg = 1u; if (ln == 0) atomicOr((unsigned int *)(cursor + 1), 1u); }     // never published (see pages_load): page 0, and the host is told
            pg -= 1u;
            const u64 *src = pool + (((u64)pg << pt.lgpg) + (nfirst & pmask));
#pragma unroll
            for (int q = 0; q < B; q++) nx[q] = src[q * IVX_WAVE + ln];      // (inside the page whatever ncnt is: pages are whole)
        }
    };
    prefetch(0);
    u32 loaded_r = 0xFFFFFFFFu;
    u32 round = 0;
    u32 pend_mine = 0;                                                // pairs the previous round staged (in its half of the ring)
    // The ring is two halves of RP_RING / 2 pairs; round r stages into half r & 1 from the half's start, so nothing a round
    // does can touch the previous round's pairs (which wait in the other half for their copy-out) and the walk needs no
    // per-pair room test: a round that finds more pairs than a half holds wraps over its own pairs, is recognised by its count
    // and goes to the rest list.  The position inside the half is wave-private state in a scalar register (ranks by ballot: no
    // LDS atomic, no wait per match); the candidate loops are wave-uniform -- every lane stays in until the longest list is
    // done, its steps predicated -- so that the position is one value for the wavefront by construction.
    constexpr u32 HALF = RP_RING / 2;
    // copy this wavefront's staged pairs of round pr out: (slice slot as the LDS address of its entry, packed row word) ->
    // (build row, probe row)
    auto copy_out = [&](u32 pr) {
        const u32 sl = pr % RP_NSLOT;
        while (__hip_atomic_load(&L.s_ready[sl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != pr + 1u) __builtin_amdgcn_s_sleep(1);
        if (!pend_mine) return;
        const u64 base = L.s_base[sl];
        const u32 tot = s_wat[sl][RP_W - 1] + L.s_wcnt[sl][RP_W - 1];
        if (base + tot > cap) return;                                 // the caller's buffers are too small: nothing of the round is written
        const u64 g = base + s_wat[sl][wv];
        u32 *ob_w = ob + g, *op_w = op + g;
        const uint2 *half = (const uint2 *)L.s_q[wv] + (pr & 1u) * HALF;
        for (u32 t = ln; t < pend_mine; t += IVX_WAVE) {
            const uint2 x = half[t];
            ob_w[t] = L.s_row[x.x]; op_w[t] = x.y & rowmask;
        }
    };
    for (u32 i = 0; i < nbatch; i++) {
        const u32 r = r_next, cnt = ncnt, first = nfirst;
        if (r != loaded_r) {
            // the ring holds slice slots: whatever is still staged leaves before the slice changes
            if (round) { copy_out(round - 1u); pend_mine = 0; }
            slice_load(ix, S, L, r, true);
            loaded_r = r;
        }
        // ---- decode the batch in flight, start the next one
        const bool full = cnt == WB;
        u32 rel[B], len[B], roww[B];
#pragma unroll
        for (int q = 0; q < B; q++) {
            const u32 lo32 = (u32)nx[q], hi32 = (u32)(nx[q] >> 32);
            rel[q] = lo32 & 0xFFFFFFu;
            len[q] = (lo32 >> 24) | ((rowbits < 32 ? (hi32 >> rowbits) & 0xFFu : 0u) << 8);
            roww[q] = hi32;                                              // (the row id is masked out of it in the copy-out, 64 pairs at a time)
        }
        if (i + 1 < nbatch) prefetch(i + 1);
        // ---- cells: of all B rows first (their LDS reads are in flight together), then the candidate loops
        const u32 sh0 = S.sh0, off = S.off, cmax = S.cmax, ncm1 = S.ncm1;
        const i32 rbase = S.rbase;
        u32 ca[B], cb[B], slow = 0;
        // (two copies of the block, chosen once per batch: a per-row choice between "every lane holds B rows" and "test the
        //  lane's row number" was two branches per row)
        auto cells = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int q = 0; q < B; q++) {
                const u32 t = ((rel[q] + 1u) >> sh0) + off;              // first cell a matching build row can start in: one cell back
                const u32 bl0 = (t > 1u ? t : 1u) - 1u;
                const u32 bh0 = ((rel[q] + len[q]) >> sh0) + off;
                const u32 bh = bh0 < cmax ? bh0 : cmax;
                bool bad = len[q] == maxlen || bh >= ncm1;               // escape, or past the slice's halo: the rest list's
                const bool ok = FULL || (u32)q * IVX_WAVE + ln < cnt;    // (only a region's last batch is short)
                if (ok && bad) slow |= 1u << q;
                bad |= !ok;
                // no row to walk: an empty range (twice the same offset); a row behind the key's last cell gets one by the clamp
                const u32 e1 = bad ? 0u : bh + 1u;
                const u32 bl = bl0 < e1 ? bl0 : e1;
                ca[q] = L.s_off[bl];
                cb[q] = L.s_off[e1];
            }
        };
        if (full) cells(std::true_type{}); else cells(std::false_type{});
        u32 wpos = 0;                                                    // pairs of this round so far (scalar)
        uint2 *half = (uint2 *)L.s_q[wv] + (round & 1u) * HALF;
#pragma unroll
        for (int q = 0; q < B; q++) {
            const i32 qs = (i32)((u32)rbase + rel[q]), qe = (i32)((u32)qs + len[q]);
            u32 j = ca[q];
#if defined(IVX_FP_ABL) && IVX_FP_ABL >= 2
            const u32 jend = j + ((cb[q] ^ roww[q]) == 0x12345u ? 1u : 0u);     // (profiling: no candidate loop; the cell lookups stay live)
#else
            const u32 jend = cb[q];
#endif
            // (walking two or four rows' lists side by side -- their slice reads in flight together, one wait for all -- was
            //  measured: no change.  The walk is bound by the instructions it issues, not by those waits.)
            // One backward branch per step; the match block sits on the fall-through path (a taken branch empties the
            // wavefront's instruction buffer, and the kernel retired 31 branches per 64 rows).
            // (lane masks straight from the compares -- uicmp / sicmp -- and back into a predicate -- inverse_ballot: the
            //  bool-to-ballot round trip of `ballot(act && ...)` cost two VALU instructions per step, the recomputed loop
            //  test one more: 13 -> 10 per step, and the walk is what the kernel's time goes into)
            u64 actm = __builtin_amdgcn_uicmp(j, jend, 36 /* ULT */);   // lanes whose list is not done
            if (actm != 0) {
                do {
#if defined(IVX_FP_BCAST)
                    const u64 x = L.s_ent[__builtin_amdgcn_readfirstlane(j) & 4095u];                  // (profiling: one address per wavefront, no bank conflicts)
#else
                    const u64 x = L.s_ent[j];                            // (a lane past its list reads on inside LDS; its result is not used)
#endif
#if defined(IVX_FP_DUP)
                    { u64 y = *(const volatile u64 *)(L.s_ent + j + 1); asm volatile("" :: "v"(y)); }  // (profiling: every slice read twice)
#endif
                    const u64 mm = actm & __builtin_amdgcn_sicmp((i32)(u32)x, qe, 41 /* SLE */) & __builtin_amdgcn_sicmp((i32)(u32)(x >> 32), qs, 39 /* SGE */);
#if !(defined(IVX_FP_ABL) && IVX_FP_ABL == 1)
                    if (__builtin_amdgcn_inverse_ballot_w64(mm)) half[mask_rank_from(mm, wpos) & (HALF - 1)] = make_uint2(j, roww[q]);      // (the counter's addend carries the position)
#endif
                    wpos += (u32)__popcll(mm);
                    j++;
                    actm = __builtin_amdgcn_uicmp(j, jend, 36);
                } while (actm != 0);
            }
        }
        u32 got = wpos;
        // more pairs than the half holds: nothing of this batch counts as staged
        const bool skip = got > HALF;
        if (__builtin_expect(skip, 0)) {
            // (as 64-row pieces: each gets a wavefront of its own in k_fill_rest)
            if (ln < B && ln * IVX_WAVE < cnt) rest[atomicAdd(rest_n, 1u)] = FpRest{r, first + ln * IVX_WAVE, cnt - ln * IVX_WAVE < IVX_WAVE ? cnt - ln * IVX_WAVE : (u32)IVX_WAVE, 0u};
            got = 0;
        } else if (__builtin_expect(slow != 0, 0)) {                     // this lane's slow rows, one by one (their cell ranges were empty)
#pragma unroll
            for (int q = 0; q < B; q++)
                if ((slow >> q) & 1u) rest_rows[atomicAdd(rest_n + 1, 1u)] = (u64)(first + (u32)q * IVX_WAVE + ln) | ((u64)r << 32);
        }
        // ---- publish the round's count; the last wavefront to arrive reserves the round's output range
#if defined(IVX_FP_ABL) && IVX_FP_ABL >= 3
        if (got == 0x7654321u) ob[got] = got;
        continue;
#endif
        if (ln == 0) {
            const u32 sl = round % RP_NSLOT;
            L.s_wcnt[sl][wv] = got;
            const u32 before = __hip_atomic_fetch_add(&L.s_arrive[sl], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (before == RP_W - 1) {
                u32 tot = 0;
#pragma unroll
                for (int w = 0; w < RP_W; w++) { s_wat[sl][w] = tot; tot += L.s_wcnt[sl][w]; }
                L.s_base[sl] = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
                __hip_atomic_store(&L.s_arrive[sl], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&L.s_ready[sl], round + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (round) copy_out(round - 1u);                              // the previous round's pairs: its base has long arrived
        pend_mine = got; round++;
    }
    if (round) copy_out(round - 1u);
}

// What k_fill_fast left: single rows (one lane each) and whole batches (one wavefront each, 64 rows at a time); coordinates
// from the packed word or -- escapes -- the input columns, the generic walk over the index in global memory (count, one
// output reservation per 64 rows, write).
__global__ __launch_bounds__(256) void k_fill_rest(JoinIndexView ix, const u64 *__restrict__ pool, PageTab pt, const FpRest *__restrict__ rest,
                                                   const u64 *__restrict__ rest_rows, const u32 *__restrict__ rest_n, u32 *__restrict__ ob,
                                                   u32 *__restrict__ op, u64 cap, unsigned long long *cursor, const i32 *__restrict__ ps_in,
                                                   const i32 *__restrict__ pe_in, u32 rowbits)
{
    const u32 nbat = rest_n[0], nrow = rest_n[1];
    if (nbat == 0 && nrow == 0) return;
    const u32 wpb = blockDim.x / IVX_WAVE, ln = lane_id();
    const u32 rowmask = rowbits >= 32 ? 0xFFFFFFFFu : (1u << rowbits) - 1u;
    const u32 maxlen = pk_maxlen(rowbits);
    const u32 pmask = (1u << pt.lgpg) - 1u;
    const u32 sh0 = ix.hdr[HDR_SH0], nlev = ix.hdr[HDR_NLEV];
    // 64 routed rows, one per lane: (region, virtual row) or nothing
    auto rows64 = [&](bool ok, u32 r, u32 v) {
        i32 qs = 0, qe = -1; u32 row = 0, k = 0;
        if (ok) {
            u32 pg = pt.ptab[(u64)r * pt.pstride + (v >> pt.lgpg)];
            if (pg == 0u) { pg = 1u; atomicOr((unsigned int *)(cursor + 1), 1u); }     // never published (see pages_load)
            pg -= 1u;
            const u64 x = pool[((u64)pg << pt.lgpg) + (v & pmask)];
            const u32 lo32 = (u32)x, hi32 = (u32)(x >> 32);
            const u32 len = (lo32 >> 24) | ((rowbits < 32 ? (hi32 >> rowbits) & 0xFFu : 0u) << 8);
            row = hi32 & rowmask;
            k = ix.rkey[r];
            if (len == maxlen) { qs = ps_in[row]; qe = pe_in[row]; }
            else { qs = (i32)((u32)ix.rdesc[r].rbase + (lo32 & 0xFFFFFFu)); qe = (i32)((u32)qs + len); }
        }
        u32 m = 0;
        if (ok) walk(ix, sh0, 0, nlev, k, qs, qe, [&](u32) { m++; });
        const u32 inc = wave_incl_scan(m);
        const u32 tot = __shfl(inc, IVX_WAVE - 1, IVX_WAVE);
        if (tot == 0) return;
        unsigned long long g = 0;
        if (ln == 0) g = atomicAdd(cursor, (unsigned long long)tot);
        g = __shfl(g, 0, IVX_WAVE);
        if (g + tot > cap) return;                                       // (the count still tells the caller what it needs)
        u64 at = g + inc - m;
        if (m) walk(ix, sh0, 0, nlev, k, qs, qe, [&](u32 brow) { ob[at] = brow; op[at] = row; at++; });
    };
    const u32 wave = blockIdx.x * wpb + threadIdx.x / IVX_WAVE, nwave = gridDim.x * wpb;
    for (u32 i0 = wave * IVX_WAVE; i0 < nrow; i0 += nwave * IVX_WAVE) {
        const bool ok = i0 + ln < nrow;
        const u64 e = ok ? rest_rows[i0 + ln] : 0ull;
        rows64(ok, (u32)(e >> 32), (u32)e);
    }
    for (u32 b = wave; b < nbat; b += nwave) {
        const FpRest w = rest[b];
        for (u32 t0 = 0; t0 < w.cnt; t0 += IVX_WAVE) rows64(t0 + ln < w.cnt, w.r, w.first + t0 + ln);
    }
}

// ------------------------------------------------------------------ lean per-row-value probe (round 3)
// count_overlaps / the join's rle_right and exists (RV_COUNT, RV_MATCHES) and coverage (RV_COVERAGE) over packed rows in region
// pages when every region is one LDS-resident level: k_fill_fast's row streaming and walk without a ring, rounds or any
// synchronisation between wavefronts -- a row's value replaces the low half of its packed word in place, as in
// k_probe_regions<RV_*>.  Rows the packed form cannot carry are listed (region, virtual row) and valued by k_rv_rest with the
// generic walk over the index in global memory.
#ifndef IVX_RV_WPS
#define IVX_RV_WPS 4
#endif
#ifndef IVX_RV_B
#define IVX_RV_B 8
#endif
template <int KIND, int B>
__global__ __launch_bounds__(RP_T, IVX_RV_WPS) void k_rv_fast(JoinIndexView ix, u64 *__restrict__ pool, const u32 *__restrict__ rcur, const u32 *__restrict__ cfirst,
                                                  PageTab pt, u32 rowbits, u64 *__restrict__ rest_rows, u32 *rest_n)
{
    constexpr u32 WB = IVX_WAVE * B, SUB = 8u / B;
    __shared__ unsigned short s_off[RP_CCAP];
    __shared__ u64 s_ent[RP_ECAP];
    __shared__ u32 s_cfirst[IVX_MAXREG_WIDE + 2];
    ProbeLds L{s_off, s_ent, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const u32 wv = __builtin_amdgcn_readfirstlane(threadIdx.x / IVX_WAVE), ln = lane_id();
    const u32 nreg = ix.hdr[HDR_NREG];
    for (u32 t = threadIdx.x; t <= nreg; t += RP_T) s_cfirst[t] = cfirst[t];
    __syncthreads();
    const u32 nchunk = s_cfirst[nreg];
    const u32 c_lo = (u32)((u64)nchunk * blockIdx.x / gridDim.x), c_hi = (u32)((u64)nchunk * (blockIdx.x + 1) / gridDim.x);
    if (c_lo >= c_hi) return;
    const u32 maxlen = pk_maxlen(rowbits);
    const u32 pmask = (1u << pt.lgpg) - 1u;
    Slice S;
    slice_init(ix, S, L);
    u32 r_next;
    { u32 a = 0, b = nreg; while (a < b) { const u32 m = (a + b + 1) >> 1; if (s_cfirst[m] <= c_lo) a = m; else b = m - 1; } r_next = a; }
    const u32 nbatch = (c_hi - c_lo) * SUB;
    u64 nx[B];
    u32 ncnt = 0, nfirst = 0; u64 *nsrc = pool;
    auto prefetch = [&](u32 i) {
        const u32 c = c_lo + i / SUB, sb = i % SUB;
        while (c >= s_cfirst[r_next + 1]) r_next++;
        nfirst = (c - s_cfirst[r_next]) * FP_CHUNK + wv * (8u * IVX_WAVE) + sb * WB;
        const u32 rows = rcur[r_next];
        ncnt = rows > nfirst ? (rows - nfirst < WB ? rows - nfirst : WB) : 0u;
        if (ncnt) {
            u32 pg = pt.ptab[(u64)r_next * pt.pstride + (nfirst >> pt.lgpg)];
            if (pg == 0u) pg = 1u;                                       // (never published: see pages_load; stay in bounds)
            nsrc = pool + (((u64)(pg - 1u) << pt.lgpg) + (nfirst & pmask));
#pragma unroll
            for (int q = 0; q < B; q++) nx[q] = nsrc[q * IVX_WAVE + ln];
        }
    };
    prefetch(0);
    u32 loaded_r = 0xFFFFFFFFu;
    for (u32 i = 0; i < nbatch; i++) {
        const u32 r = r_next, cnt = ncnt, first = nfirst;
        u64 *dst = nsrc;
        if (r != loaded_r) { slice_load(ix, S, L, r, true); loaded_r = r; }
        const bool full = cnt == WB;
        u32 rel[B], len[B], roww[B];
#pragma unroll
        for (int q = 0; q < B; q++) {
            const u32 lo32 = (u32)nx[q], hi32 = (u32)(nx[q] >> 32);
            rel[q] = lo32 & 0xFFFFFFu;
            len[q] = (lo32 >> 24) | ((rowbits < 32 ? (hi32 >> rowbits) & 0xFFu : 0u) << 8);
            roww[q] = hi32;                                              // (the un-permute masks the row id out of it)
        }
        if (i + 1 < nbatch) prefetch(i + 1);
        const u32 sh0 = S.sh0, off = S.off, cmax = S.cmax, ncm1 = S.ncm1;
        const i32 rbase = S.rbase;
        u32 ca[B], cb[B], slow = 0, okm = 0;
        auto cells = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int q = 0; q < B; q++) {
                const u32 t = ((rel[q] + 1u) >> sh0) + off;
                const u32 bl0 = (t > 1u ? t : 1u) - 1u;
                const u32 bh0 = ((rel[q] + len[q]) >> sh0) + off;
                const u32 bh = bh0 < cmax ? bh0 : cmax;
                bool bad = len[q] == maxlen || bh >= ncm1;
                const bool ok = FULL || (u32)q * IVX_WAVE + ln < cnt;
                if (ok && bad) slow |= 1u << q;
                if (ok && !bad) okm |= 1u << q;
                bad |= !ok;
                const u32 e1 = bad ? 0u : bh + 1u;
                const u32 bl = bl0 < e1 ? bl0 : e1;
                ca[q] = s_off[bl];
                cb[q] = s_off[e1];
            }
        };
        if (full) cells(std::true_type{}); else cells(std::false_type{});
#pragma unroll
        for (int q = 0; q < B; q++) {
            const i32 qs = (i32)((u32)rbase + rel[q]), qe = (i32)((u32)qs + len[q]);
            const i32 ca1 = rv_wadd(qe, 1), cb1 = rv_wsub(qs, 1);        // coverage: the closed query grown by one on either side
            u32 v = 0;
            for (u32 j = ca[q]; j < cb[q]; j++) {
                const u64 x = s_ent[j];
                const i32 xs = (i32)(u32)x, xe = (i32)(u32)(x >> 32);
                if (xs <= qe && xe >= qs) {
                    if (KIND == RV_COVERAGE) { const i32 d = rv_wsub(ca1 < xe ? ca1 : xe, cb1 > xs ? cb1 : xs); v = (u32)rv_wadd((i32)v, d > 1 ? d : 1); }
                    else v++;
                }
            }
            if ((okm >> q) & 1u) dst[q * IVX_WAVE + ln] = (u64)v | ((u64)roww[q] << 32);
        }
        if (__builtin_expect(slow != 0, 0)) {
#pragma unroll
            for (int q = 0; q < B; q++)
                if ((slow >> q) & 1u) rest_rows[atomicAdd(rest_n + 1, 1u)] = (u64)(first + (u32)q * IVX_WAVE + ln) | ((u64)r << 32);
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_rv_rest(JoinIndexView ix, u64 *__restrict__ pool, PageTab pt, const u64 *__restrict__ rest_rows,
                                                 const u32 *__restrict__ rest_n, const i32 *__restrict__ ps_in, const i32 *__restrict__ pe_in,
                                                 u32 rowbits, u32 adj)
{
    const u32 nrow = rest_n[1];
    const u32 rowmask = rowbits >= 32 ? 0xFFFFFFFFu : (1u << rowbits) - 1u;
    const u32 maxlen = pk_maxlen(rowbits);
    const u32 pmask = (1u << pt.lgpg) - 1u;
    const u32 sh0 = ix.hdr[HDR_SH0], nlev = ix.hdr[HDR_NLEV];
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += gridDim.x * blockDim.x) {
        const u64 ent = rest_rows[i];
        const u32 r = (u32)(ent >> 32), vr = (u32)ent;
        u32 pg = pt.ptab[(u64)r * pt.pstride + (vr >> pt.lgpg)];
        if (pg == 0u) pg = 1u;
        u64 *slot = pool + (((u64)(pg - 1u) << pt.lgpg) + (vr & pmask));
        const u64 x = *slot;
        const u32 lo32 = (u32)x, hi32 = (u32)(x >> 32);
        const u32 len = (lo32 >> 24) | ((rowbits < 32 ? (hi32 >> rowbits) & 0xFFu : 0u) << 8);
        const u32 row = hi32 & rowmask;
        i32 qs, qe;
        if (len == maxlen) { qs = (i32)((u32)ps_in[row] + adj); qe = (i32)((u32)pe_in[row] - adj); }
        else { qs = (i32)((u32)ix.rdesc[r].rbase + (lo32 & 0xFFFFFFu)); qe = (i32)((u32)qs + len); }
        u32 v = 0;
        if (KIND == RV_COVERAGE) {
            const i32 a = rv_wadd(qe, 1), b = rv_wsub(qs, 1);
            walk_ent(ix, sh0, 0, nlev, ix.rkey[r], qs, qe, [&](const ivx_ent &n) {
                const i32 d = rv_wsub(a < n.e ? a : n.e, b > n.s ? b : n.s);
                v = (u32)rv_wadd((i32)v, d > 1 ? d : 1);
            });
        } else if (KIND != RV_COUNT || !(qe < qs)) {
            walk_ent(ix, sh0, 0, nlev, ix.rkey[r], qs, qe, [&](const ivx_ent &) { v++; });
        }
        *slot = (u64)v | ((u64)hi32 << 32);
    }
}

// ------------------------------------------------------------------ match-dense fill: count, scan, write
// With several pairs per probe row the staging ring holds only one 64-row batch per wavefront and the 16
// wavefronts of a workgroup end up in lock step, round after round.  For such joins the pairs are written in two
// passes over the same decomposition instead, with no ring, no atomics and no synchronisation between wavefronts:
//   piece   = the rows of one 64-row granule [64g, 64g+64) of the partitioned order that belong to region r
//             (row shares start at multiples of 64, so a piece has ONE owner); slot(piece) = g + r, which grows
//             strictly along the row order
//   PASS 0  every piece's pair count goes to pcount[slot]; an exclusive scan turns it into output offsets
//   PASS 1  the piece is walked again in lock step (every lane steps through its candidates together); the lanes
//           that match in a step take consecutive positions after the piece's running offset by ballot rank, so
//           the stores of a step are contiguous.  Order inside a piece is arbitrary, like everywhere else.
template <int PASS, bool IDENT, bool PAGED = false, bool PK = false>
__global__ __launch_bounds__(RP_T) void k_probe_dense(JoinIndexView ix, const void *__restrict__ rows_a, const void *__restrict__ rows_b,
                                                      const u32 *__restrict__ offs, u32 nblk, u32 prow_stride, const u32 *unsorted,
                                                      u64 *__restrict__ pcount, u32 *__restrict__ ob, u32 *__restrict__ op, u64 cap,
                                                      PageTab pt = PageTab{nullptr, 0u, 0u}, const i32 *__restrict__ ps_in = nullptr, const i32 *__restrict__ pe_in = nullptr,
                                                      u32 rowbits = 32)
{
    const u32 rowmask = rowbits >= 32 ? 0xFFFFFFFFu : (1u << rowbits) - 1u;
    const u32 maxlen = pk_maxlen(rowbits);
    if ((unsorted != nullptr && *unsorted == 0) != IDENT) return;
    static_assert(!(IDENT && PAGED), "paged rows are never read in place");
    __shared__ u32 s_pg[RP_NPG];
    u32 pg_first = 0;
    // (the page id always comes from LDS: a choice between an LDS and a global pointer becomes a flat load with
    //  vmcnt(0) waits that serialise the row loads)
    auto row_at = [&](u64 i, u32, u64 rf) -> u64 {
        if (!PAGED) return i;
        const u32 x = (u32)(i - rf);
        const u32 pg = s_pg[(x >> pt.lgpg) - pg_first] - 1u;
        return ((u64)pg << pt.lgpg) + (x & ((1u << pt.lgpg) - 1u));
    };
    // (all threads; barriers inside) the page ids of routed rows [lo, c_hi) of region r; c_hi is cut back to what RP_NPG
    // pages hold (the caller then walks the rest of the region as another segment)
    auto pages_load = [&](u64 lo, u64 &c_hi, u32 r, u64 rf) {
        if (!PAGED) return;
        __syncthreads();
        pg_first = (u32)(lo - rf) >> pt.lgpg;
        const u64 lim = (rf + ((u64)(pg_first + RP_NPG) << pt.lgpg)) & ~63ull;   // (a cut falls between two 64-row granules)
        if (c_hi > lim) c_hi = lim;
        const u32 npg = ((u32)(c_hi - 1 - rf) >> pt.lgpg) - pg_first + 1u;
        if (threadIdx.x < npg) {
            u32 v = pt.ptab[(u64)r * pt.pstride + pg_first + threadIdx.x];
            // 0 = the partition never published this page (cannot happen once k_part_onepass has completed; it did in a
            // profiling build whose switch skipped the publication, and the page "0 - 1" then was a wild address: round 2's
            // fault).  Read page 0 instead and tell the host.
            if (v == 0u) v = 1u;                                    // (this kernel has no error word in reach: stay in bounds)
            s_pg[threadIdx.x] = v;
        }
        __syncthreads();
    };
    auto row_se = [&](u64 i) -> u64 {
        if (IDENT) return (u64)(u32)((const i32 *)rows_a)[i] | ((u64)(u32)((const i32 *)rows_b)[i] << 32);
        return ((const u64 *)rows_a)[i];
    };
    auto row_id = [&](u64 i, u64 at) -> u32 { return IDENT ? (u32)i : ((const u32 *)rows_b)[at * prow_stride]; };
    __shared__ unsigned short s_off[RP_CCAP];
    __shared__ u64 s_ent[RP_ECAP];
    __shared__ u32 s_row[RP_ECAP];
    __shared__ u32 s_rfirst[IVX_MAXREG_WIDE + 2];
    ProbeLds L{s_off, s_ent, s_row, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const u32 wv = threadIdx.x / IVX_WAVE, ln = lane_id();
    const u32 nreg = ix.hdr[HDR_NREG];
    const bool rf_lds = nreg <= IVX_MAXREG_WIDE;
    if (rf_lds) for (u32 t = threadIdx.x; t <= nreg; t += RP_T) s_rfirst[t] = offs[(u64)t * nblk];
    __syncthreads();
    auto rfirst = [&](u32 r) -> u32 { return rf_lds ? s_rfirst[r] : offs[(u64)r * nblk]; };
    const u64 total_rows = rfirst(nreg);
    const u32 nvb = gridDim.x, vb = blockIdx.x;
    u64 lo = (total_rows * vb / nvb) & ~63ull;
    const u64 hi = vb + 1 == nvb ? total_rows : ((total_rows * (vb + 1) / nvb) & ~63ull);
    if (lo >= hi) return;
    Slice S;
    slice_init(ix, S, L);
    u32 r;
    { u32 a = 0, b = nreg; while (a < b) { const u32 m = (a + b + 1) >> 1; if (rfirst(m) <= lo) a = m; else b = m - 1; } r = a; }
    for (; lo < hi; r++) {
        const u64 rend = rfirst(r + 1);
        u64 c_hi = hi < rend ? hi : rend;
        if (c_hi <= lo) continue;
        u64 g = (lo >> 6) + wv;
        // the first granule's rows are in flight while the slice loads
        u64 nx = 0; u32 nxr = 0;
        const u64 rf = PAGED ? rfirst(r) : 0;
        pages_load(lo, c_hi, r, rf);
        const u64 g1 = (c_hi + 63) >> 6;
        { const u64 i = g * 64 + ln; const bool ok = g < g1 && i >= lo && i < c_hi; const u64 at = ok ? row_at(i, r, rf) : 0; nx = ok ? row_se(at) : 0; nxr = (PASS == 1 && !PK && ok) ? row_id(i, at) : 0u; }
        slice_load(ix, S, L, r, true);
        for (; g < g1; g += RP_W) {
            const u64 i = g * 64 + ln;
            const bool ok = i >= lo && i < c_hi;
            i32 qs = (i32)(u32)nx, qe = (i32)(u32)(nx >> 32);
            u32 rowv = nxr;
            if (PK) {
                const u32 lo32 = (u32)nx, hi32 = (u32)(nx >> 32);
                const u32 len = (lo32 >> 24) | ((rowbits < 32 ? (hi32 >> rowbits) & 0xFFu : 0u) << 8);
                rowv = hi32 & rowmask;
                qs = (i32)((u32)S.rbase + (lo32 & 0xFFFFFFu)); qe = (i32)((u32)qs + len);
                if (ok && len == maxlen) { qs = ps_in[rowv]; qe = pe_in[rowv]; }
            }
            { const u64 g2 = g + RP_W; const u64 i2 = g2 * 64 + ln; const bool ok2 = g2 < g1 && i2 >= lo && i2 < c_hi; const u64 at = ok2 ? row_at(i2, r, rf) : 0; nx = ok2 ? row_se(at) : 0; nxr = (PASS == 1 && !PK && ok2) ? row_id(i2, at) : 0u; }
            const u64 slot = g + r;
            if (PASS == 0) {
                u32 c = 0;
                if (ok) probe_row(S, qs, qe, [&](u32, bool, i32, i32) { c++; });
                const u64 tot = wave_sum((u64)c);
                if (ln == 0) pcount[slot] = tot;
                continue;
            }
            // (scanned) the piece's pairs go to [base, base + its count); everything below is relative to it, in 32 bits
            const u64 base = __builtin_amdgcn_readfirstlane((u32)pcount[slot]) | ((u64)__builtin_amdgcn_readfirstlane((u32)(pcount[slot] >> 32)) << 32);
            u32 *const obp = ob + base, *const opp = op + base;
            const u64 room = cap > base ? cap - base : 0;
            const u32 lim = room > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)room;   // pairs of this piece the caller's buffers still hold
            u32 cnt = 0;                                                // pairs of the piece written so far (wavefront-uniform)
            // lock-step walk of one candidate list per lane: [ja, jb) of the staged slice or of the index in HBM
            // (four candidates per lane are fetched before any of them is tested: the steps are otherwise one LDS round
            // trip each, with nothing else in flight)
            auto walk = [&](u32 ja, u32 jb, bool lds) {
                constexpr int U = 4;
                while (__any(ja < jb)) {
                    i32 cs_[U], ce_[U]; u32 cr_[U]; bool v[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        v[u] = ja + u < jb;
                        cs_[u] = 0; ce_[u] = 0; cr_[u] = 0;
                        if (v[u]) {
                            if (lds) { const u64 x = S.s_ent[ja + u]; cs_[u] = (i32)(u32)x; ce_[u] = (i32)(u32)(x >> 32); cr_[u] = S.s_row[ja + u]; }
                            else { const ivx_ent x = ix.ent[ja + u]; cs_[u] = x.s; ce_[u] = x.e; cr_[u] = x.row; }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const bool hit = v[u] && cs_[u] <= qe && ce_[u] >= qs;
                        const u64 m = __ballot(hit);
                        const u32 pos = cnt + mask_rank(m);
                        if (hit && pos < lim) { obp[pos] = cr_[u]; opp[pos] = rowv; }
                        cnt += (u32)__popcll(m);
                    }
                    ja = jb - ja > U ? ja + U : jb;
                }
            };
            const i64 hi64 = (i64)qe - (i64)S.origin;
            const bool live = ok && hi64 >= 0;
            {   // level 0
                u32 ja = 0, jb = 0; bool lds = false;
                if (live && S.lev0) {
                    const u32 ncell = S.ncell0;
                    const i64 lo64 = (i64)qs - ((i64)1 << S.sh0) + 1 - (i64)S.origin;
                    const i64 bl = lo64 <= 0 ? 0 : (lo64 >> S.sh0);
                    if (bl < (i64)ncell) {
                        const u32 blo = (u32)bl;
                        const i64 bh = hi64 >> S.sh0;
                        const u32 bhi = bh >= (i64)ncell ? ncell - 1u : (u32)bh;
                        if (blo <= bhi) {
                            lds = S.inlds && blo >= S.slo && bhi < S.shi;
                            if (lds) { ja = S.s_off[blo - S.slo]; jb = S.s_off[bhi + 1 - S.slo]; }
                            else { ja = ix.binstart[S.lb + blo]; jb = ix.binstart[S.lb + bhi + 1]; }
                        }
                    }
                }
                // lanes of one wavefront may differ in where their list lives (a row reaching past the slice): two rounds
                const u32 la = lds ? ja : 0u, lb = lds ? jb : 0u, ga = lds ? 0u : ja, gb = lds ? 0u : jb;
                walk(la, lb, true);
                if (__any(ga < gb)) walk(ga, gb, false);
            }
            if (S.upper) {
                for (u32 l = 1; l < S.nlev; l++) {
                    if (ix.hdr[HDR_LEVCNT + l] == 0) continue;
                    const u32 sh = S.sh0 + IVX_LSTEP * l;
                    u32 ja = 0, jb = 0;
                    if (live) {
                        u32 blo = 0, bhi = 0; bool any = true;
                        if (sh < 32) {
                            const u32 ncell = (S.span >> sh) + 1u;
                            const i64 lo64 = (i64)qs - ((i64)1 << sh) + 1 - (i64)S.origin;
                            const i64 bl = lo64 <= 0 ? 0 : (lo64 >> sh);
                            const i64 bh = hi64 >> sh;
                            if (bl >= (i64)ncell) any = false;
                            else { blo = (u32)bl; bhi = bh >= (i64)ncell ? ncell - 1u : (u32)bh; if (blo > bhi) any = false; }
                        }
                        if (any) { const u32 lbase = ix.lbase[(u64)l * ix.nkeys + S.k]; ja = ix.binstart[lbase + blo]; jb = ix.binstart[lbase + bhi + 1]; }
                    }
                    if (__any(ja < jb)) walk(ja, jb, false);
                }
            }
        }
        lo = c_hi;
        if (PAGED && lo < hi && lo < rend) r--;                       // cut at the page window: same region again
    }
}

// fill pass: rows per lane and batch from the expected matches per row (cap / n)
static inline int fill_rows_per_lane(u64 cap, u64 n)
{
    if (const char *f = getenv("IVX_RP_ROWS")) return atoi(f);             // experiments
    const double per_row = (double)cap / (double)n;
    // (a round's pairs must fit half a staging ring: 64 * rows per lane * pairs per row <= ~205 of its 256, four sigma below it)
    return per_row <= 0.40 ? 8 : per_row <= 0.8 ? 4 : per_row <= 1.6 ? 2 : 1;
}

// rows one partition workgroup takes: 1, 2 or 4 tiles, so that mid-size batches still spread over all CUs
static inline u32 part_chunk(u64 n, u32 max_tiles = 4) { const u32 t = n >= (16u << 20) ? 4u : n >= (4u << 20) ? 2u : 1u; return (u32)PA_TILE * (t < max_tiles ? t : max_tiles); }

// ------------------------------------------------------------------ values back into input order
// The scatter wrote, for every (region, workgroup chunk), one contiguous run, and kept each row's index
// inside its chunk (one or two tiles).  So the values of one chunk are ~200 runs of the value stream: read them
// (each wavefront a contiguous 1/16 of the chunk's values, coalesced inside runs), drop them at their
// chunk-local index in LDS, write the chunk out in input order.  Rows that were never routed (unknown key,
// key without build rows) keep the zero LDS was cleared to -- the reference's answer for them.
constexpr u32 UP_CHUNK = 2u * PA_TILE;               // the per-row-output path keeps chunks at two tiles (values of a chunk sit in LDS; u16 chunk-local ids)

// OUT: UP_I64 zero-extended (count), UP_I64S sign-extended (coverage), UP_U32 (rle_right; the workgroup's sum
// goes to *total), UP_U8 (exists = value != 0)
enum { UP_I64 = 0, UP_I64S = 1, UP_U32 = 2, UP_U8 = 3 };
template <int OUT>
__device__ __forceinline__ void up_store(void *out, u64 i, u32 v)
{
    if (OUT == UP_I64) ((i64 *)out)[i] = (i64)v;
    else if (OUT == UP_I64S) ((i64 *)out)[i] = (i64)(i32)v;
    else if (OUT == UP_U32) ((u32 *)out)[i] = v;
    else ((u8 *)out)[i] = v != 0;
}

template <int OUT, int ND>
__global__ __launch_bounds__(PA_T) void k_unpermute(const u32 *__restrict__ val, const unsigned short *__restrict__ cidx,
                                                    const u32 *__restrict__ offs, u32 nblk, u32 chunk, u32 nreg, u64 n, void *__restrict__ out,
                                                    const u32 *unsorted, unsigned long long *total, int sorted_done = 0)
{
    __shared__ u64 s_sum[PA_T / IVX_WAVE];
    u64 mysum = 0;
    if (*unsorted == 0 && sorted_done) return;              // (routed callers: another kernel answered the unmoved rows in place)
    if (*unsorted == 0) {                                   // values already sit in input order
        const u64 lo0 = (u64)blockIdx.x * chunk;
        const u64 hi0 = lo0 + chunk < n ? lo0 + chunk : n;
        for (u64 t = lo0 + threadIdx.x; t < hi0; t += PA_T) { const u32 v = val[t]; up_store<OUT>(out, t, v); mysum += v; }
        if (OUT == UP_U32 && total) { const u64 b = block_sum<u64, PA_T>(mysum, s_sum); if (threadIdx.x == 0 && b) atomicAdd(total, (unsigned long long)b); }
        return;
    }
    __shared__ u32 s_val[UP_CHUNK];
    __shared__ u32 s_pre[ND + 1], s_g[ND];
    __shared__ u32 scan_lds[PA_T / IVX_WAVE + 1];
    const u32 tid = threadIdx.x, blk = blockIdx.x;
    const u64 lo = (u64)blk * chunk;
    const u32 len = (u32)(lo + chunk < n ? chunk : n - lo);
    u32 c = 0, g = 0;
    if (tid < nreg) { g = offs[(u64)tid * nblk + blk]; c = offs[(u64)tid * nblk + blk + 1] - g; }
    u32 tot;
    const u32 ex = block_excl_scan<u32, PA_T>(c, scan_lds, &tot);
    if (tid < ND) { s_pre[tid] = ex; s_g[tid] = g; }
    if (tid == 0) s_pre[ND] = tot;
    for (u32 t = tid; t < chunk; t += PA_T) s_val[t] = 0;
    __syncthreads();
    // wavefront w owns elements [w*per, (w+1)*per) of the chunk's region-major value list
    const u32 wv = tid / IVX_WAVE, ln = lane_id();
    const u32 per = (tot + PA_T / IVX_WAVE - 1) / (PA_T / IVX_WAVE);
    const u32 t_lo = wv * per, t_hi = t_lo + per < tot ? t_lo + per : tot;
    u32 r = 0;
    if (t_lo < t_hi) { u32 a = 0, b = ND; while (a < b) { const u32 m = (a + b + 1) >> 1; if (s_pre[m] <= t_lo + ln && m < ND) a = m; else b = m - 1; } r = a; }
    for (u32 t = t_lo + ln; t < t_hi; t += IVX_WAVE) {
        while (r + 1 < ND && s_pre[r + 1] <= t) r++;
        const u64 at = (u64)s_g[r] + (t - s_pre[r]);
        s_val[cidx[at]] = val[at];
    }
    __syncthreads();
    for (u32 t = tid; t < len; t += PA_T) { const u32 v = s_val[t]; up_store<OUT>(out, lo + t, v); mysum += v; }
    if (OUT == UP_U32 && total) { const u64 b = block_sum<u64, PA_T>(mysum, s_sum); if (tid == 0 && b) atomicAdd(total, (unsigned long long)b); }
}

// The same for rows routed by the one-pass partition (packed rows in region pages): the probe left every row's value in the
// low half of its packed word, the row id above it.  One workgroup per partition tile: vtab[tile][region] says where the
// tile's run of every region went (virtual start, rows); the words are read back run by run (each wavefront a contiguous
// share of the tile's words), dropped at row - tile start in LDS and written out in input order.
template <int OUT, int TILE>
__global__ __launch_bounds__(512) void k_unpermute_paged(const u64 *__restrict__ pool, const uint2 *__restrict__ vtab, PageTab pt, u32 nreg, u32 rowbits,
                                                         u64 n, void *__restrict__ out, unsigned long long *total)
{
    constexpr int T = 512, ND = 256;
    __shared__ u32 s_val[TILE];
    __shared__ u32 s_pre[ND + 1], s_v[ND], s_pg0[ND];
    __shared__ u32 scan_lds[T / IVX_WAVE + 1];
    __shared__ u64 s_sum[T / IVX_WAVE];
    const u32 tid = threadIdx.x;
    const u64 t0 = (u64)blockIdx.x * TILE;
    const u32 len = (u32)(t0 + TILE < n ? (u64)TILE : n - t0);
    const u32 rowmask = rowbits >= 32 ? 0xFFFFFFFFu : (1u << rowbits) - 1u;
    const u32 pmask = (1u << pt.lgpg) - 1u;
    u32 v = 0, c = 0;
    if (tid < nreg) { const uint2 vc = vtab[(u64)blockIdx.x * ND + tid]; v = vc.x; c = vc.y; }
    u32 tot;
    const u32 ex = block_excl_scan<u32, T>(c, scan_lds, &tot);
    if (tid < ND) { s_pre[tid] = ex; s_v[tid] = v; s_pg0[tid] = c ? pt.ptab[(u64)tid * pt.pstride + (v >> pt.lgpg)] - 1u : 0u; }
    if (tid == 0) s_pre[ND] = tot;
    for (u32 t = tid; t < TILE; t += T) s_val[t] = 0;           // rows that were never routed keep 0: the reference's answer for them
    __syncthreads();
    const u32 wv = tid / IVX_WAVE, ln = lane_id();
    const u32 per = (tot + T / IVX_WAVE - 1) / (T / IVX_WAVE);
    const u32 j_lo = wv * per, j_hi = j_lo + per < tot ? j_lo + per : tot;
    u32 r = 0;
    if (j_lo < j_hi) { u32 a = 0, b = ND; while (a < b) { const u32 m = (a + b + 1) >> 1; if (m < ND && s_pre[m] <= j_lo + ln) a = m; else b = m - 1; } r = a; }
    for (u32 j = j_lo + ln; j < j_hi; j += IVX_WAVE) {
        while (r + 1 < ND && s_pre[r + 1] <= j) r++;
        const u32 x = s_v[r] + (j - s_pre[r]);                  // virtual row number in region r
        u32 pg = s_pg0[r];
        if ((x >> pt.lgpg) != (s_v[r] >> pt.lgpg)) pg = pt.ptab[(u64)r * pt.pstride + (x >> pt.lgpg)] - 1u;   // the run's second page
        const u64 w = pool[((u64)pg << pt.lgpg) + (x & pmask)];
        const u32 slot = ((u32)(w >> 32) & rowmask) - (u32)t0;
        if (slot < (u32)TILE) s_val[slot] = (u32)w;
    }
    __syncthreads();
    u64 mysum = 0;
    for (u32 t = tid; t < len; t += T) { const u32 val = s_val[t]; up_store<OUT>(out, t0 + t, val); mysum += val; }
    if (OUT == UP_U32 && total) { const u64 b = block_sum<u64, T>(mysum, s_sum); if (tid == 0 && b) atomicAdd(total, (unsigned long long)b); }
}

}  // namespace

// One value per probe row, in input order, through the region partition.  kind: IVX_RV_COUNT (count_overlaps,
// jv over the build rows, i64 out), IVX_RV_COVERAGE (jv over the merged nodes, i64 out), IVX_RV_PER_ROW (the
// join's rle_right: u32 out, *d_total += all matches), IVX_RV_EXISTS (semi / anti join: u8 out).
ivx_status ivx_rowval_probe_regions(ivx_ctx *ctx, const JoinIndexView &jv, u32 nreg, int kind,
                                    const u32 *key, const i32 *s, const i32 *e, u64 n, int strict, void *out, u64 *d_total,
                                    bool has_filter, bool pk24, bool fast)
{
    if (n == 0) return IVX_OK;
    if (nreg == 0 || nreg > IVX_MAXREG_WIDE) return ctx->fail(IVX_ERR_INVALID, "per-row region probe: one partition pass only");
    hipStream_t st = ctx->stream;
    // Packed rows in region pages (the join's one-pass partition, two 512-thread workgroups per CU) when the index allows
    // them and one 256-digit pass routes the rows: no histogram pass, the probe leaves each value in the row's own word,
    // the un-permute reads the words back through vtab.  (IVX_PART=two / IVX_PACK=0: the two-pass form below.)
    const bool two_pass = (getenv("IVX_PART") && !strcmp(getenv("IVX_PART"), "two")) || (getenv("IVX_PACK") && !strcmp(getenv("IVX_PACK"), "0"));
    if (pk24 && nreg <= IVX_MAXREG && !two_pass) {
        constexpr u32 TILE = 512u * 16u;
        u32 rowbits = 1;
        while (rowbits < 32 && (n - 1) >> rowbits) rowbits++;
        const bool filter_off = getenv("IVX_FILTER") && !strcmp(getenv("IVX_FILTER"), "0");
        const bool use_filter = has_filter && !filter_off;
        if (use_filter) rowbits = 32;
        u32 lgpg = 14;
        while (lgpg < 31 && (n >> lgpg) > 4096) lgpg++;
        const u64 pstride = (n >> lgpg) + 2;
        const u64 npages = (n >> lgpg) + nreg + 1;
        const u64 ntiles = (n + TILE - 1) / TILE;
        u32 *ctl, *ptab; u64 *pool; uint2 *vtab;
        IVX_TRY(ctx->get_scratch(WS_SORTHIST, (1024 + 8 + 1032 + 1032) * sizeof(u32), (void **)&ctl));
        IVX_TRY(ctx->get_scratch(WS_T2, (size_t)nreg * pstride * sizeof(u32), (void **)&ptab));
        IVX_TRY(ctx->get_scratch(WS_T0, (size_t)(npages << lgpg) * sizeof(u64), (void **)&pool));
        IVX_TRY(ctx->get_scratch(WS_T1, (size_t)ntiles * 256 * sizeof(uint2), (void **)&vtab));
        IVX_HIP(ctx, hipMemsetAsync(ctl, 0, (1024 + 8) * sizeof(u32), st));
        IVX_HIP(ctx, hipMemsetAsync(ptab, 0, (size_t)nreg * pstride * sizeof(u32), st));
        const PageTab pt{ptab, (u32)pstride, lgpg};
        u32 *rcur = ctl, *pool_next = ctl + 1024, *rf = ctl + 1032;
        const bool vec = (((uintptr_t)key | (uintptr_t)s | (uintptr_t)e) & 15) == 0;
        const u32 adj = strict ? 1u : 0u;
        const u32 tiles = n >= (16u << 20) ? 4u : n >= (4u << 20) ? 2u : 1u;
        const u32 chunk1 = TILE * tiles;
        const u32 nblk1 = (u32)((n + chunk1 - 1) / chunk1);
#define IVX_RVPART(V_, K_, F_) hipLaunchKernelGGL((k_part_onepass<V_, 256, 16, K_, F_, true, 512>), dim3(nblk1), dim3(512), 0, st, jv, key, s, e, n, chunk1, rcur, pt, pool_next, pool, (u32 *)nullptr, rowbits, adj, vtab)
#define IVX_RVPART2(V_, K_) do { if (use_filter) IVX_RVPART(V_, K_, true); else IVX_RVPART(V_, K_, false); } while (0)
#define IVX_RVPART3(V_) do { if (jv.nkeys <= KT_MAX) IVX_RVPART2(V_, true); else IVX_RVPART2(V_, false); } while (0)
        if (vec) IVX_RVPART3(true); else IVX_RVPART3(false);
#undef IVX_RVPART3
#undef IVX_RVPART2
#undef IVX_RVPART
        // every region one LDS-resident level: the lean kernel values the rows, k_rv_rest the few the packed form cannot carry
        // (IVX_FILL=old: the general kernel)
        const bool lean = fast && lgpg >= 13 && !(getenv("IVX_FILL") && !strcmp(getenv("IVX_FILL"), "old"));
        if (lean) {
            u64 *rest_rows;
            IVX_TRY(ctx->get_scratch(WS_T3, (size_t)(n + 64) * sizeof(u64), (void **)&rest_rows));
            u32 *rest_n = ctl + 1024 + 4;
            hipLaunchKernelGGL(k_chunk_bounds, dim3(1), dim3(1024), 0, st, (const u32 *)rcur, nreg, rf, rf + 1032);
#define IVX_RVF(K_) do { \
            hipLaunchKernelGGL((k_rv_fast<K_, IVX_RV_B>), dim3(RP_GRID * (IVX_RV_WPS / 4)), dim3(RP_T), 0, st, jv, pool, (const u32 *)rcur, (const u32 *)(rf + 1032), pt, rowbits, rest_rows, rest_n); \
            hipLaunchKernelGGL((k_rv_rest<K_>), dim3(256), dim3(256), 0, st, jv, pool, pt, (const u64 *)rest_rows, (const u32 *)rest_n, s, e, rowbits, adj); } while (0)
            if (kind == IVX_RV_COVERAGE) IVX_RVF(RV_COVERAGE); else if (kind == IVX_RV_COUNT) IVX_RVF(RV_COUNT); else IVX_RVF(RV_MATCHES);
#undef IVX_RVF
        } else {
        hipLaunchKernelGGL(k_page_bounds, dim3(1), dim3(1024), 0, st, (const u32 *)rcur, nreg, rf);
#define IVX_RVP(M_) hipLaunchKernelGGL((k_probe_regions<M_, RP_B, false, true, true>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, (const void *)pool, (const void *)nullptr, (const u32 *)rf, 1u, 1u, (u32 *)pool, (u32 *)nullptr, (u64)0, (unsigned long long *)nullptr, 1u, adj, (const u32 *)nullptr, 0, pt, (const u32 *)nullptr, s, e, rowbits)
        if (kind == IVX_RV_COVERAGE) IVX_RVP(RV_COVERAGE); else if (kind == IVX_RV_COUNT) IVX_RVP(RV_COUNT); else IVX_RVP(RV_MATCHES);
#undef IVX_RVP
        }
#define IVX_UPP(O_) hipLaunchKernelGGL((k_unpermute_paged<O_, (int)TILE>), dim3((u32)ntiles), dim3(512), 0, st, (const u64 *)pool, (const uint2 *)vtab, pt, nreg, rowbits, n, out, (unsigned long long *)d_total)
        switch (kind) {
        case IVX_RV_COVERAGE: IVX_UPP(UP_I64S); break;
        case IVX_RV_COUNT: IVX_UPP(UP_I64); break;
        case IVX_RV_PER_ROW: IVX_UPP(UP_U32); break;
        default: IVX_UPP(UP_U8); break;
        }
#undef IVX_UPP
        IVX_HIP(ctx, hipGetLastError());
        return IVX_OK;
    }
    const bool wide = nreg > IVX_MAXREG;                                // 1024 digits instead of 256
    const u32 chunk = part_chunk(n, 2);
    const u32 nblk = (u32)((n + chunk - 1) / chunk);
    u32 *hist, *val; u64 *pse; unsigned short *cidx;
    const u64 nh = (u64)(wide ? 1024 : 256) * nblk + 1;
    IVX_TRY(ctx->get_scratch(WS_SORTHIST, nh * sizeof(u32), (void **)&hist));
    IVX_TRY(ctx->get_scratch(WS_T0, n * sizeof(u64), (void **)&pse));
    IVX_TRY(ctx->get_scratch(WS_T1, n * sizeof(unsigned short), (void **)&cidx));
    IVX_TRY(ctx->get_scratch(WS_T2, n * sizeof(u32), (void **)&val));
    IVX_HIP(ctx, hipMemsetAsync(hist + (nh - 1), 0, sizeof(u32), st));
    const u32 adj = strict ? 1u : 0u;
    const bool vec = (((uintptr_t)key | (uintptr_t)s | (uintptr_t)e) & 15) == 0;
    u32 *unsorted = (u32 *)(ctx->d_scalars + 10);                       // stays 0 if the rows already come in region order
    IVX_HIP(ctx, hipMemsetAsync(unsorted, 0, sizeof(u32), st));
#define IVX_PART(V_, ND_) do { \
    hipLaunchKernelGGL((k_part_hist<V_, ND_>), dim3(nblk), dim3(PA_T), 0, st, jv, key, s, n, nblk, chunk, hist, adj, unsorted); \
    IVX_TRY(ivx_scan_exclusive_u32(ctx, hist, nh)); \
    hipLaunchKernelGGL((k_part_scatter<V_, unsigned short, ND_>), dim3(nblk), dim3(PA_T), 0, st, jv, key, s, e, n, nblk, (const u32 *)hist, pse, cidx, chunk, adj, (const u32 *)unsorted, 0); } while (0)
    if (wide) { if (vec) IVX_PART(true, 1024); else IVX_PART(false, 1024); }
    else { if (vec) IVX_PART(true, 256); else IVX_PART(false, 256); }
#undef IVX_PART
#define IVX_RV(M_, ID_) hipLaunchKernelGGL((k_probe_regions<M_, RP_B, ID_>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, ID_ ? (const void *)s : (const void *)pse, ID_ ? (const void *)e : (const void *)nullptr, (const u32 *)hist, nblk, 1u, val, (u32 *)nullptr, (u64)0, (unsigned long long *)nullptr, 1u, adj, (const u32 *)unsorted, 0)
    if (kind == IVX_RV_COVERAGE) { IVX_RV(RV_COVERAGE, false); IVX_RV(RV_COVERAGE, true); }
    else if (kind == IVX_RV_COUNT) { IVX_RV(RV_COUNT, false); IVX_RV(RV_COUNT, true); }
    else { IVX_RV(RV_MATCHES, false); IVX_RV(RV_MATCHES, true); }
#undef IVX_RV
#define IVX_UP1(O_, ND_) hipLaunchKernelGGL((k_unpermute<O_, ND_>), dim3(nblk), dim3(PA_T), 0, st, (const u32 *)val, (const unsigned short *)cidx, (const u32 *)hist, nblk, chunk, (u32)ND_, n, out, (const u32 *)unsorted, (unsigned long long *)d_total)
#define IVX_UP(O_) do { if (wide) IVX_UP1(O_, 1024); else IVX_UP1(O_, 256); } while (0)
    switch (kind) {
    case IVX_RV_COVERAGE: IVX_UP(UP_I64S); break;
    case IVX_RV_COUNT: IVX_UP(UP_I64); break;
    case IVX_RV_PER_ROW: IVX_UP(UP_U32); break;
    default: IVX_UP(UP_U8); break;
    }
#undef IVX_UP
#undef IVX_UP1
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

// ------------------------------------------------------------------ routing for operators with their own probe (nearest)
namespace {

// two values per row back into input order: the un-permute above for (u32, i64) pairs with one-tile chunks; rows that
// were never routed get (IVX_NULL_IDX, dflt); op = the row's own index; vb / ob / op may be null (one i64 value per row)
__global__ __launch_bounds__(PA_T) void k_unpermute_pair(const u32 *__restrict__ vb, const i64 *__restrict__ vd, const unsigned short *__restrict__ cidx,
                                                         const u32 *__restrict__ offs, u32 nblk, u64 n, u32 *__restrict__ ob, u32 *__restrict__ op,
                                                         i64 *__restrict__ od, const u32 *unsorted, i64 dflt)
{
    constexpr int ND = 1024;
    __shared__ u32 s_b[PA_TILE];
    __shared__ i64 s_d[PA_TILE];
    __shared__ u32 s_pre[ND + 1], s_g[ND];
    __shared__ u32 scan_lds[PA_T / IVX_WAVE + 1];
    if (*unsorted == 0) return;                             // the probe wrote the outputs in place
    const u32 tid = threadIdx.x, blk = blockIdx.x;
    const u64 lo = (u64)blk * PA_TILE;
    const u32 len = (u32)(lo + PA_TILE < n ? PA_TILE : n - lo);
    const u32 g = offs[(u64)tid * nblk + blk];
    const u32 c = offs[(u64)tid * nblk + blk + 1] - g;
    u32 tot;
    const u32 ex = block_excl_scan<u32, PA_T>(c, scan_lds, &tot);
    s_pre[tid] = ex; s_g[tid] = g;
    if (tid == 0) s_pre[ND] = tot;
    for (u32 t = tid; t < PA_TILE; t += PA_T) { s_b[t] = IVX_NULL_IDX; s_d[t] = dflt; }
    __syncthreads();
    const u32 wv = tid / IVX_WAVE, ln = lane_id();
    const u32 per = (tot + PA_T / IVX_WAVE - 1) / (PA_T / IVX_WAVE);
    const u32 t_lo = wv * per, t_hi = t_lo + per < tot ? t_lo + per : tot;
    u32 r = 0;
    if (t_lo < t_hi) { u32 a = 0, b = ND; while (a < b) { const u32 m = (a + b + 1) >> 1; if (s_pre[m] <= t_lo + ln && m < ND) a = m; else b = m - 1; } r = a; }
    for (u32 t = t_lo + ln; t < t_hi; t += IVX_WAVE) {
        while (r + 1 < ND && s_pre[r + 1] <= t) r++;
        const u64 at = (u64)s_g[r] + (t - s_pre[r]);
        const u32 ci = cidx[at];
        if (vb) s_b[ci] = vb[at];
        if (vd) s_d[ci] = vd[at];
    }
    __syncthreads();
    for (u32 t = tid; t < len; t += PA_T) {
        if (ob) ob[lo + t] = s_b[t];
        if (op) op[lo + t] = (u32)(lo + t);
        if (od) od[lo + t] = vd ? s_d[t] : dflt;
    }
}

}  // namespace

ivx_status ivx_route_rows(ivx_ctx *ctx, const JoinIndexView &rv, const u32 *key, const i32 *s, const i32 *e, u64 n, u32 adj, ivx_routed *out)
{
    hipStream_t st = ctx->stream;
    const u32 chunk = PA_TILE;
    const u32 nblk = (u32)((n + chunk - 1) / chunk);
    const u64 nh = (u64)1024 * nblk + 1;
    u32 *hist; u64 *pse; unsigned short *cidx;
    IVX_TRY(ctx->get_scratch(WS_SORTHIST, nh * sizeof(u32), (void **)&hist));
    IVX_TRY(ctx->get_scratch(WS_T0, n * sizeof(u64), (void **)&pse));
    IVX_TRY(ctx->get_scratch(WS_T1, n * sizeof(unsigned short), (void **)&cidx));
    u32 *unsorted = (u32 *)(ctx->d_scalars + 10);
    IVX_HIP(ctx, hipMemsetAsync(hist + (nh - 1), 0, sizeof(u32), st));
    IVX_HIP(ctx, hipMemsetAsync(unsorted, 0, sizeof(u32), st));
    const bool vec = (((uintptr_t)key | (uintptr_t)s | (uintptr_t)e) & 15) == 0;
    if (vec) hipLaunchKernelGGL((k_part_hist<true, 1024>), dim3(nblk), dim3(PA_T), 0, st, rv, key, s, n, nblk, chunk, hist, adj, unsorted);
    else hipLaunchKernelGGL((k_part_hist<false, 1024>), dim3(nblk), dim3(PA_T), 0, st, rv, key, s, n, nblk, chunk, hist, adj, unsorted);
    IVX_TRY(ivx_scan_exclusive_u32(ctx, hist, nh));
    if (vec) hipLaunchKernelGGL((k_part_scatter<true, unsigned short, 1024>), dim3(nblk), dim3(PA_T), 0, st, rv, key, s, e, n, nblk, (const u32 *)hist, pse, cidx, chunk, adj, (const u32 *)unsorted, 0);
    else hipLaunchKernelGGL((k_part_scatter<false, unsigned short, 1024>), dim3(nblk), dim3(PA_T), 0, st, rv, key, s, e, n, nblk, (const u32 *)hist, pse, cidx, chunk, adj, (const u32 *)unsorted, 0);
    IVX_HIP(ctx, hipGetLastError());
    out->hist = hist; out->pse = pse; out->cidx = cidx; out->unsorted = unsorted; out->nblk = nblk; out->chunk = chunk;
    return IVX_OK;
}

// one u32 value per routed row back into input order, as u32 (out32) or as "value != 0" bytes (out8); rows that were
// never routed get 0
ivx_status ivx_unroute_u32(ivx_ctx *ctx, const ivx_routed &r, u64 n, const u32 *vb, u32 *out32, u8 *out8)
{
    hipStream_t st = ctx->stream;
    if (out32) hipLaunchKernelGGL((k_unpermute<UP_U32, 1024>), dim3(r.nblk), dim3(PA_T), 0, st, vb, r.cidx, r.hist, r.nblk, r.chunk, 1024u, n, (void *)out32, r.unsorted, (unsigned long long *)nullptr, 1);
    if (out8) hipLaunchKernelGGL((k_unpermute<UP_U8, 1024>), dim3(r.nblk), dim3(PA_T), 0, st, vb, r.cidx, r.hist, r.nblk, r.chunk, 1024u, n, (void *)out8, r.unsorted, (unsigned long long *)nullptr, 1);
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

ivx_status ivx_unroute_pair(ivx_ctx *ctx, const ivx_routed &r, u64 n, const u32 *vb, const i64 *vd, u32 *ob, u32 *op, i64 *od, i64 dflt)
{
    hipLaunchKernelGGL(k_unpermute_pair, dim3(r.nblk), dim3(PA_T), 0, ctx->stream, vb, vd, r.cidx, r.hist, r.nblk, n, ob, op, od, r.unsorted, dflt);
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

namespace {

constexpr int WR_T = 256;

// ------------------------------------------------------------------ second routing pass (more than IVX_MAXREG_WIDE regions)
// Pass A (k_part_* with SPLIT) grouped the probe rows by super-region = region / G and kept region % G ("sub") per
// row.  Pass B orders the rows of every super-region by sub.  Super-region s is cut into tiles of PA_TILE rows;
// the histogram is laid out as [s][sub][tile of s] -- exactly the output order, so ONE plain exclusive scan over it
// gives every (tile, sub) run its place, and (s, sub, tile 0) is where region s*G + sub starts.
constexpr u32 P2_SUBMAX = 64;                           // G <= 64: up to 1023 * 64 = 65 472 >= IVX_MAXREG2 regions

// tprefix[s] = tiles of the super-regions before s (one workgroup; nsuper <= 1023)
__global__ __launch_bounds__(1024) void k_p2_layout(const u32 *__restrict__ offs1, u32 nblk1, u32 nsuper, u32 *tprefix, const u32 *unsorted)
{
    __shared__ u32 red[1024 / IVX_WAVE + 1];
    if (*unsorted == 0) return;
    const u32 t = threadIdx.x;
    const u32 rows = t < nsuper ? offs1[(u64)(t + 1) * nblk1] - offs1[(u64)t * nblk1] : 0u;
    const u32 tiles = (rows + PA_TILE - 1) / PA_TILE;
    u32 tot;
    const u32 ex = block_excl_scan<u32, 1024>(tiles, red, &tot);
    if (t < nsuper) tprefix[t] = ex;
    if (t == 0) tprefix[nsuper] = tot;
}

struct P2Tile { u32 s, t, nt; u64 lo, hi; bool ok; };
__device__ __forceinline__ P2Tile p2_tile(const u32 *__restrict__ offs1, u32 nblk1, u32 nsuper, const u32 *__restrict__ tprefix, u32 *s_tp)
{
    for (u32 i = threadIdx.x; i <= nsuper; i += blockDim.x) s_tp[i] = tprefix[i];
    __syncthreads();
    P2Tile T; T.ok = blockIdx.x < s_tp[nsuper];
    if (!T.ok) return T;
    u32 a = 0, b = nsuper - 1;                                      // last s with tprefix[s] <= block
    while (a < b) { const u32 m = (a + b + 1) >> 1; if (s_tp[m] <= blockIdx.x) a = m; else b = m - 1; }
    T.s = a; T.t = blockIdx.x - s_tp[a]; T.nt = s_tp[a + 1] - s_tp[a];
    const u64 seg_lo = offs1[(u64)a * nblk1], seg_hi = offs1[(u64)(a + 1) * nblk1];
    T.lo = seg_lo + (u64)T.t * PA_TILE;
    T.hi = T.lo + PA_TILE < seg_hi ? T.lo + PA_TILE : seg_hi;
    return T;
}

__global__ __launch_bounds__(PA_T) void k_p2_hist(const unsigned char *__restrict__ sub1, const u32 *__restrict__ offs1, u32 nblk1, u32 nsuper,
                                                  const u32 *__restrict__ tprefix, u32 G, u32 *__restrict__ hist2, const u32 *unsorted)
{
    __shared__ u32 s_tp[IVX_MAXREG_WIDE + 2];
    __shared__ u32 cnt[P2_SUBMAX];
    if (*unsorted == 0) return;
    if (threadIdx.x < P2_SUBMAX) cnt[threadIdx.x] = 0;
    const P2Tile T = p2_tile(offs1, nblk1, nsuper, tprefix, s_tp);  // (barrier inside)
    if (!T.ok) return;
    for (u64 i0 = T.lo; i0 < T.hi; i0 += PA_T) {
        const u64 i = i0 + threadIdx.x;
        const bool ok = i < T.hi;
        lds_count_up(cnt, ok ? (u32)sub1[i] : 0u, ok);
    }
    __syncthreads();
    if (threadIdx.x < G) hist2[(u64)G * s_tp[T.s] + (u64)threadIdx.x * T.nt + T.t] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(PA_T) void k_p2_scatter(const unsigned char *__restrict__ sub1, const u64 *__restrict__ se1, const u32 *__restrict__ row1,
                                                     const u32 *__restrict__ offs1, u32 nblk1, u32 nsuper, const u32 *__restrict__ tprefix, u32 G,
                                                     const u32 *__restrict__ hist2, u64 *__restrict__ se2, u32 *__restrict__ row2, const u32 *unsorted)
{
    __shared__ u32 s_tp[IVX_MAXREG_WIDE + 2];
    __shared__ u32 cur[P2_SUBMAX];
    if (*unsorted == 0) return;
    const P2Tile T = p2_tile(offs1, nblk1, nsuper, tprefix, s_tp);
    if (!T.ok) return;
    if (threadIdx.x < G) cur[threadIdx.x] = hist2[(u64)G * s_tp[T.s] + (u64)threadIdx.x * T.nt + T.t];   // this tile's run of every sub
    __syncthreads();
    for (u64 i0 = T.lo; i0 < T.hi; i0 += PA_T) {
        const u64 i = i0 + threadIdx.x;
        const bool ok = i < T.hi;
        const u32 d = ok ? (u32)sub1[i] : 0u;
        const u32 pos = lds_count_up(cur, d, ok);                   // order inside a run is arbitrary
        if (ok) { se2[pos] = se1[i]; row2[pos] = row1[i]; }
    }
}

// rfirst[r] = first routed position of region r (r = 0 .. nreg), from the scanned second-pass histogram
__global__ __launch_bounds__(WR_T) void k_p2_bounds(u32 nreg, u32 G, u32 nsuper, const u32 *__restrict__ offs1, u32 nblk1, u64 nh1,
                                                    const u32 *__restrict__ tprefix, const u32 *__restrict__ hist2, u32 *__restrict__ rfirst,
                                                    const u32 *unsorted)
{
    if (*unsorted == 0) return;
    const u32 r = blockIdx.x * WR_T + threadIdx.x;
    if (r > nreg) return;
    if (r == nreg) { rfirst[r] = offs1[nh1 - 1]; return; }          // all routed rows
    const u32 sp = r / G, sub = r - sp * G;
    const u32 nt = tprefix[sp + 1] - tprefix[sp];
    rfirst[r] = nt ? hist2[(u64)G * tprefix[sp] + (u64)sub * nt] : offs1[(u64)sp * nblk1];
}

// the same table when the probe rows came in region order already (nothing was moved): binary searches in the input
__global__ __launch_bounds__(WR_T) void k_sorted_bounds(JoinIndexView ix, const u32 *__restrict__ pkey, const i32 *__restrict__ ps, u64 n,
                                                        u32 nreg, u32 *__restrict__ rfirst, const u32 *unsorted)
{
    __shared__ i32 s_origin[KT_MAX];
    __shared__ u32 s_last[KT_MAX], s_kreg[KT_MAX];
    if (*unsorted != 0) return;
    KeyTab kt;
    keytab_load(ix, s_origin, s_last, s_kreg, kt);
    __syncthreads();
    const u32 r = blockIdx.x * WR_T + threadIdx.x;
    if (r > nreg) return;
    u64 lo = 0, hi = n;                                             // first row whose region is >= r (every row is routable here)
    auto body = [&](auto klds_tag) {
        constexpr bool KLDS = decltype(klds_tag)::value;
        while (lo < hi) { const u64 mid = lo + ((hi - lo) >> 1); if (region_of<KLDS>(ix, kt, s_origin, s_last, s_kreg, pkey ? pkey[mid] : 0u, ps[mid]) < r) lo = mid + 1; else hi = mid; }
    };
    KEYTAB_DISPATCH(kt, body);
    rfirst[r] = (u32)lo;
}

// match-dense fill (k_probe_dense): count the pieces, scan, write.  rows/ident: the partitioned rows and row ids, and --
// when the partition may have found the input already in region order (`unsorted` given) -- the input start / end columns
static bool dense_fill_wanted(u64 cap, u64 n)
{
    if (const char *f = getenv("IVX_DENSE")) return atoi(f) != 0;          // tests / experiments
    return (double)cap / (double)n > 3.5;                                  // measured crossover (tools/probe_only.py IVX_DENSE=0/1): the ring wins below
}

ivx_status dense_fill(ivx_ctx *ctx, const JoinIndexView &jv, u32 nreg, const void *rows_se, const void *rows_id, u32 prow_stride,
                      const i32 *s, const i32 *e, const u32 *offs, u32 nblk, const u32 *unsorted, u64 n,
                      u32 *ob, u32 *op, u64 cap, u64 *d_cursor, const PageTab *pt = nullptr, bool packed = false, u32 rowbits = 32)
{
    hipStream_t st = ctx->stream;
    const u64 slots = (n >> 6) + nreg + 3;
    u64 *pcount;
    IVX_TRY(ctx->get_scratch(WS_T3, slots * sizeof(u64), (void **)&pcount));
    IVX_HIP(ctx, hipMemsetAsync(pcount, 0, slots * sizeof(u64), st));
#define IVX_DENSE_PASS(P_) do { \
        if (pt && packed) { hipLaunchKernelGGL((k_probe_dense<P_, false, true, true>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, rows_se, rows_id, offs, nblk, prow_stride, unsorted, pcount, ob, op, cap, *pt, s, e, rowbits); break; } \
        if (pt) { hipLaunchKernelGGL((k_probe_dense<P_, false, true>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, rows_se, rows_id, offs, nblk, prow_stride, unsorted, pcount, ob, op, cap, *pt); break; } \
        hipLaunchKernelGGL((k_probe_dense<P_, false>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, rows_se, rows_id, offs, nblk, prow_stride, unsorted, pcount, ob, op, cap); \
        if (unsorted) hipLaunchKernelGGL((k_probe_dense<P_, true>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, (const void *)s, (const void *)e, offs, nblk, 1u, unsorted, pcount, ob, op, cap); } while (0)
    IVX_DENSE_PASS(0);
    IVX_TRY(ivx_scan_exclusive_u64(ctx, pcount, slots));
    IVX_HIP(ctx, hipMemcpyAsync(d_cursor, pcount + (slots - 1), sizeof(u64), hipMemcpyDeviceToDevice, st));   // the pair total
    IVX_DENSE_PASS(1);
#undef IVX_DENSE_PASS
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

// more than IVX_MAXREG_WIDE regions: route the probe rows in two partition passes (super-region, then region inside
// it), or not at all if they already come in region order; then the same probe kernels with a global region table
ivx_status probe_two_level(ivx_ctx *ctx, const JoinIndexView &jv, int mode, u32 nreg, const u32 *key, const i32 *s, const i32 *e, u64 n,
                           u32 *ob, u32 *op, u64 cap, u64 *d_cursor, bool planned)
{
    hipStream_t st = ctx->stream;
    ivx_join_plan &pl = ctx->join_plan;
    u32 *unsorted = (u32 *)(ctx->d_scalars + 10);
    u32 *rfirst, *prow2; u64 *pse2;
    if (planned) {                                                      // routed by the count call that sized this fill call
        rfirst = const_cast<u32 *>(pl.hist); pse2 = const_cast<u64 *>(pl.pse); prow2 = const_cast<u32 *>(pl.prow);
        s = pl.ds; e = pl.de;
    } else {
    const u32 G = (nreg + IVX_MAXREG_WIDE - 1) / IVX_MAXREG_WIDE;       // regions per super-region
    if (G > P2_SUBMAX) return ctx->fail(IVX_ERR_INVALID, "overlap index with too many probe regions");
    const u32 nsuper = (nreg + G - 1) / G;
    const uint2 split = make_uint2(G, (u32)(((1ull << 32) + G - 1) / G));
    const u32 chunk = part_chunk(n);
    const u32 nblk1 = (u32)((n + chunk - 1) / chunk);
    const u64 nh1 = (u64)1024 * nblk1 + 1;
    const u32 grid2 = (u32)(n / PA_TILE) + nsuper + 1;                   // tiles of the second pass, at most
    const u64 nh2 = (u64)G * grid2 + 1;
    u32 *hist1, *hist2, *tprefix, *prow1; u64 *pse1; unsigned char *sub1;
    IVX_TRY(ctx->get_scratch(WS_SORTHIST, nh1 * sizeof(u32), (void **)&hist1));
    IVX_TRY(ctx->get_scratch(WS_T0, n * sizeof(u64), (void **)&pse1));
    IVX_TRY(ctx->get_scratch(WS_T1, n * sizeof(u32), (void **)&prow1));
    IVX_TRY(ctx->get_scratch(WS_T2, n, (void **)&sub1));
    IVX_TRY(ctx->get_scratch(WS_SA0, n * sizeof(u64), (void **)&pse2));
    IVX_TRY(ctx->get_scratch(WS_SA1, n * sizeof(u32), (void **)&prow2));
    IVX_TRY(ctx->get_scratch(WS_T4, ((size_t)nsuper + 1) * sizeof(u32), (void **)&tprefix));
    IVX_TRY(ctx->get_scratch(WS_T5, nh2 * sizeof(u32), (void **)&hist2));
    IVX_TRY(ctx->get_scratch(WS_T6, ((size_t)nreg + 1) * sizeof(u32), (void **)&rfirst));
    IVX_HIP(ctx, hipMemsetAsync(hist1 + (nh1 - 1), 0, sizeof(u32), st));
    IVX_HIP(ctx, hipMemsetAsync(hist2, 0, nh2 * sizeof(u32), st));
    IVX_HIP(ctx, hipMemsetAsync(unsorted, 0, sizeof(u32), st));
    const bool vec = (((uintptr_t)key | (uintptr_t)s | (uintptr_t)e) & 15) == 0;
    // pass A: by super-region (and the sortedness check on the regions themselves)
    if (vec) hipLaunchKernelGGL((k_part_hist<true, 1024, true>), dim3(nblk1), dim3(PA_T), 0, st, jv, key, s, n, nblk1, chunk, hist1, 0u, unsorted, split);
    else hipLaunchKernelGGL((k_part_hist<false, 1024, true>), dim3(nblk1), dim3(PA_T), 0, st, jv, key, s, n, nblk1, chunk, hist1, 0u, unsorted, split);
    IVX_TRY(ivx_scan_exclusive_u32(ctx, hist1, nh1));
    if (vec) hipLaunchKernelGGL((k_part_scatter<true, u32, 1024, true>), dim3(nblk1), dim3(PA_T), 0, st, jv, key, s, e, n, nblk1, (const u32 *)hist1, pse1, prow1, chunk, 0u, (const u32 *)unsorted, 0, split, sub1);
    else hipLaunchKernelGGL((k_part_scatter<false, u32, 1024, true>), dim3(nblk1), dim3(PA_T), 0, st, jv, key, s, e, n, nblk1, (const u32 *)hist1, pse1, prow1, chunk, 0u, (const u32 *)unsorted, 0, split, sub1);
    // pass B: inside every super-region by region
    hipLaunchKernelGGL(k_p2_layout, dim3(1), dim3(1024), 0, st, (const u32 *)hist1, nblk1, nsuper, tprefix, (const u32 *)unsorted);
    hipLaunchKernelGGL(k_p2_hist, dim3(grid2), dim3(PA_T), 0, st, (const unsigned char *)sub1, (const u32 *)hist1, nblk1, nsuper, (const u32 *)tprefix, G, hist2, (const u32 *)unsorted);
    IVX_TRY(ivx_scan_exclusive_u32(ctx, hist2, nh2));
    hipLaunchKernelGGL(k_p2_scatter, dim3(grid2), dim3(PA_T), 0, st, (const unsigned char *)sub1, (const u64 *)pse1, (const u32 *)prow1, (const u32 *)hist1, nblk1, nsuper,
                       (const u32 *)tprefix, G, (const u32 *)hist2, pse2, prow2, (const u32 *)unsorted);
    const u32 bgrid = (nreg + 1 + WR_T - 1) / WR_T;
    hipLaunchKernelGGL(k_p2_bounds, dim3(bgrid), dim3(WR_T), 0, st, nreg, G, nsuper, (const u32 *)hist1, nblk1, nh1, (const u32 *)tprefix, (const u32 *)hist2, rfirst, (const u32 *)unsorted);
    hipLaunchKernelGGL(k_sorted_bounds, dim3(bgrid), dim3(WR_T), 0, st, jv, key, s, n, nreg, rfirst, (const u32 *)unsorted);
    if (mode == JP_COUNT) {                                             // leave the routed rows for the fill call (ivx_capi.hip fills in whose they are)
        pl.hist = rfirst; pl.pse = pse2; pl.prow = prow2; pl.ds = s; pl.de = e; pl.chunk = 0; pl.nblk = 1; pl.paged = false;
        pl.slots = 0;
        for (int slot : {WS_SORTHIST, WS_T0, WS_T1, WS_T2, WS_SA0, WS_SA1, WS_T4, WS_T5, WS_T6, WS_IN_START, WS_IN_END}) pl.slots |= 1ull << slot;
        pl.valid = true;
    }
    }
    unsigned long long *cur = (unsigned long long *)d_cursor;
    const u64 hint = ctx->fill_hint ? ctx->fill_hint : (planned && pl.total < cap ? pl.total : cap);
    if (mode == JP_FILL && dense_fill_wanted(hint, n))
        return dense_fill(ctx, jv, nreg, (const void *)pse2, (const void *)prow2, 1u, s, e, (const u32 *)rfirst, 1u, (const u32 *)unsorted, n, ob, op, cap, d_cursor);
    if (mode == JP_FILL) {
        const int bsel = fill_rows_per_lane(hint, n);
#define IVX_FILL2(B_, ID_) hipLaunchKernelGGL((k_probe_regions<1, B_, ID_>), dim3(RP_GRID), dim3(RP_T), 0, st, jv, ID_ ? (const void *)s : (const void *)pse2, ID_ ? (const void *)e : (const void *)prow2, (const u32 *)rfirst, 1u, RP_VGRID / RP_GRID, ob, op, cap, cur, 1u, 0u, (const u32 *)unsorted, 0)
#define IVX_FILLW(B_) do { IVX_FILL2(B_, false); IVX_FILL2(B_, true); } while (0)
        switch (bsel) { case 1: IVX_FILLW(1); break; case 2: IVX_FILLW(2); break; case 4: IVX_FILLW(4); break; default: IVX_FILLW(8); }
#undef IVX_FILLW
#undef IVX_FILL2
    } else {
        hipLaunchKernelGGL((k_probe_regions<0, RP_B, false>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, (const void *)pse2, (const void *)prow2, (const u32 *)rfirst, 1u, 1u, ob, op, cap, cur, 1u, 0u, (const u32 *)unsorted, 0);
        hipLaunchKernelGGL((k_probe_regions<0, RP_B, true>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, (const void *)s, (const void *)e, (const u32 *)rfirst, 1u, 1u, ob, op, cap, cur, 1u, 0u, (const u32 *)unsorted, 0);
    }
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}

}  // namespace

ivx_status ivx_join_probe_regions(ivx_ctx *ctx, const JoinIndexView &jv, u32 nreg, int mode,
                                  const u32 *key, const i32 *s, const i32 *e, u64 n,
                                  u32 *ob, u32 *op, u64 cap, u64 *d_cursor, bool planned, bool has_filter, bool pk24, int fast, hipEvent_t ready)
{
    if (n == 0) { if (ready) IVX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ready, 0)); return IVX_OK; }
    ivx_join_plan &pl = ctx->join_plan;
    hipStream_t st = ctx->stream;
    // `ready` (an index whose build tail may still run on another stream): the routing pass reads only what was final
    // before that tail started; the probe kernels come behind the event
    auto wait_ready = [&]() -> ivx_status { if (ready) { IVX_HIP(ctx, hipStreamWaitEvent(st, ready, 0)); ready = nullptr; } return IVX_OK; };
    if (nreg > IVX_MAXREG_WIDE) { IVX_TRY(wait_ready()); return probe_two_level(ctx, jv, mode, nreg, key, s, e, n, ob, op, cap, d_cursor, planned); }
#ifdef IVX_ABLATE          // profiling builds only (tools/variant.sh <name> -DIVX_ABLATE; tools/ablate.sh): IVX_DBG bit switches
    const int dbg = getenv("IVX_DBG") ? atoi(getenv("IVX_DBG")) : 0;
#else
    const int dbg = 0;
#endif
    // one partition pass into region pages (k_part_onepass), unless the old two-pass partition is asked for
    // (IVX_PART=two: A/B measurements and the tests that pin both)
    const bool two_pass = getenv("IVX_PART") && !strcmp(getenv("IVX_PART"), "two");
    if ((planned && pl.paged) || (!planned && !two_pass)) {
        const bool wide = nreg > IVX_MAXREG;
        PageTab pt; const u32 *rfirst; const u64 *pool_se; const u32 *pool_row;
        u32 *ctl;                                                       // rcur[1024] | pool_next, rest_n | rfirst[<= 1025] | cfirst[<= 1025]
        void *rest_buf = nullptr;                                       // batches the lean fill kernel leaves to the generic walk (FpRest)
        bool all_routed = false;
        // 8-byte routed rows whenever a region's coordinates fit 24 bits (IVX_PACK=0: the 12-byte form, for A/B runs and tests)
        const bool pack_off = getenv("IVX_PACK") && !strcmp(getenv("IVX_PACK"), "0");
        bool packed = pk24 && !pack_off;
        u32 rowbits = 1;
        while (rowbits < 32 && (n - 1) >> rowbits) rowbits++;          // bits of the largest row id (the rest of the word's upper half extends the length)
        if (planned) {
            rowbits = pl.rowbits;
            pt = PageTab{const_cast<u32 *>(pl.ptab), pl.pstride, pl.lgpg};
            rfirst = pl.hist; pool_se = pl.pse; pool_row = pl.prow; packed = pl.packed; all_routed = pl.all_routed;
            ctl = const_cast<u32 *>(pl.hist) - 1032; rest_buf = pl.rest;
            s = pl.ds; e = pl.de;                                       // (the columns the count call read: packed rows refer to them)
        } else {
            u32 lgpg = 14;                                              // a page holds at least a tile; at most ~4096 pages per region
            while (lgpg < 31 && (n >> lgpg) > 4096) lgpg++;
            const u64 pstride = (n >> lgpg) + 2;
            const u64 npages = (n >> lgpg) + nreg + 1;
            u32 *ptab; u64 *pse; u32 *prow = nullptr;
            IVX_TRY(ctx->get_scratch(WS_SORTHIST, (1024 + 8 + 1032 + 1032) * sizeof(u32), (void **)&ctl));
            IVX_TRY(ctx->get_scratch(WS_T2, (size_t)nreg * pstride * sizeof(u32), (void **)&ptab));
            IVX_TRY(ctx->get_scratch(WS_T0, (size_t)(npages << lgpg) * sizeof(u64), (void **)&pse));
            if (!packed) IVX_TRY(ctx->get_scratch(WS_T1, (size_t)(npages << lgpg) * sizeof(u32), (void **)&prow));
            else if (fast) IVX_TRY(ctx->get_scratch(WS_T1, (size_t)fp_max_batches(n, nreg) * sizeof(FpRest) + (size_t)(n + 64) * sizeof(u64), &rest_buf));
            IVX_HIP(ctx, hipMemsetAsync(ctl, 0, (1024 + 8) * sizeof(u32), st));
            IVX_HIP(ctx, hipMemsetAsync(ptab, 0, (size_t)nreg * pstride * sizeof(u32), st));
            pt = PageTab{ptab, (u32)pstride, lgpg};
            u32 *rcur = ctl, *pool_next = ctl + 1024, *rf = ctl + 1032;
            const bool vec = (((uintptr_t)key | (uintptr_t)s | (uintptr_t)e) & 15) == 0;
            const bool half = packed && !wide;                          // two 512-thread workgroups per CU, 8192-row tiles
            const u32 tile = wide ? PA_T * 8u : half ? 512u * 16u : PA_T * 12u;
            const u32 tiles = n >= (16u << 20) ? 4u : n >= (4u << 20) ? 2u : 1u;
            const u32 chunk1 = tile * tiles;
            const u32 nblk1 = (u32)((n + chunk1 - 1) / chunk1);
#define IVX_ONEPASS4(V_, ND_, I_, K_, F_, P_) hipLaunchKernelGGL((k_part_onepass<V_, ND_, I_, K_, F_, P_>), dim3(nblk1), dim3(PA_T), 0, st, jv, key, s, e, n, chunk1, rcur, pt, pool_next, pse, prow, rowbits)
#define IVX_ONEPASS3(V_, ND_, I_, K_, F_) do { \
        if (half) hipLaunchKernelGGL((k_part_onepass<V_, 256, 16, K_, F_, true, 512>), dim3(nblk1), dim3(512), 0, st, jv, key, s, e, n, chunk1, rcur, pt, pool_next, pse, prow, rowbits); \
        else if (packed) IVX_ONEPASS4(V_, ND_, I_, K_, F_, true); else IVX_ONEPASS4(V_, ND_, I_, K_, F_, false); } while (0)
#define IVX_ONEPASS2(V_, ND_, I_, K_) do { if (use_filter) IVX_ONEPASS3(V_, ND_, I_, K_, true); else IVX_ONEPASS3(V_, ND_, I_, K_, false); } while (0)
#define IVX_ONEPASS(V_, ND_, I_) do { if (jv.nkeys <= KT_MAX) IVX_ONEPASS2(V_, ND_, I_, true); else IVX_ONEPASS2(V_, ND_, I_, false); } while (0)
            const bool filter_off = getenv("IVX_FILTER") && !strcmp(getenv("IVX_FILTER"), "0");   // experiments: IVX_FILTER=0 routes every row
            const bool use_filter = has_filter && !filter_off;
            if (use_filter) rowbits = 32;                               // (see pk_maxlen)
            all_routed = !use_filter;
            if (wide) { if (vec) IVX_ONEPASS(true, 1024, 8); else IVX_ONEPASS(false, 1024, 8); }
            else { if (vec) IVX_ONEPASS(true, 256, 12); else IVX_ONEPASS(false, 256, 12); }
#undef IVX_ONEPASS2
#undef IVX_ONEPASS3
#undef IVX_ONEPASS4
#undef IVX_ONEPASS
            hipLaunchKernelGGL(k_chunk_bounds, dim3(1), dim3(1024), 0, st, (const u32 *)rcur, nreg, rf, rf + 1032);
            rfirst = rf; pool_se = pse; pool_row = prow;
            IVX_TRY(wait_ready());
            if (mode == JP_COUNT) {
                pl.hist = rf; pl.pse = pse; pl.prow = prow; pl.ds = s; pl.de = e; pl.chunk = 0; pl.nblk = 1;
                pl.paged = true; pl.packed = packed; pl.rest = rest_buf; pl.rowbits = rowbits; pl.ptab = ptab; pl.pstride = (u32)pstride; pl.lgpg = lgpg; pl.all_routed = all_routed;
                pl.slots = (1ull << WS_SORTHIST) | (1ull << WS_T0) | (1ull << WS_T1) | (1ull << WS_T2) | (1ull << WS_IN_START) | (1ull << WS_IN_END);
                pl.valid = true;
            }
        }
        IVX_TRY(wait_ready());
        unsigned long long *cur = (unsigned long long *)d_cursor;
        // the rows the probe walks are the routed ones: the density hint is pairs per INPUT row, as the caller sized it
        const u64 hint = ctx->fill_hint ? ctx->fill_hint : (planned && pl.total < cap ? pl.total : cap);
        if (mode == JP_FILL && dense_fill_wanted(hint, n))
            return dense_fill(ctx, jv, nreg, (const void *)pool_se, (const void *)pool_row, 1u, s, e, rfirst, 1u, nullptr, n, ob, op, cap, d_cursor, &pt, packed, rowbits);
        const u32 *slow_gate = nullptr;
        if (mode == JP_FILL) {
            // rows per lane follow from the pairs per ROUTED row: known here when every row was routed; with the occupancy
            // bitmap in use the count sits on the device, k_pick_rows decides there and every variant is launched (three exit)
            u32 *bsel = all_routed ? nullptr : (u32 *)(ctx->d_scalars + 11);
            const u32 force = getenv("IVX_RP_ROWS") ? (u32)atoi(getenv("IVX_RP_ROWS")) : 0u;
            if (bsel) hipLaunchKernelGGL(k_pick_rows, dim3(1), dim3(1), 0, st, rfirst, nreg, hint, force, bsel);
            // packed rows over an index whose every region is one LDS-resident level: the lean kernel, then whatever batches it
            // left to the generic walk (IVX_FILL=old: the general kernel, for A/B runs and the tests that pin both)
            const bool lean_off = getenv("IVX_FILL") && !strcmp(getenv("IVX_FILL"), "old");
            if (packed && fast != 0 && !lean_off && pt.lgpg >= 13 && rest_buf != nullptr) {
                u32 *rest_n = ctl + 1024 + 4;                           // batches, rows
                FpRest *rest = (FpRest *)rest_buf;
                u64 *rest_rows = (u64 *)(rest + fp_max_batches(n, nreg));
                {
                    if (planned) IVX_HIP(ctx, hipMemsetAsync(rest_n, 0, 2 * sizeof(u32), st));   // (else: zeroed with the routing pass's counters just now)
#define IVX_FILLF(B_) hipLaunchKernelGGL((k_fill_fast<B_>), dim3(RP_GRID), dim3(RP_T), 0, st, jv, pool_se, (const u32 *)ctl, (const u32 *)(rfirst + 1032), pt, ob, op, cap, cur, (const u32 *)bsel, rowbits, rest, rest_rows, rest_n)
                    if (bsel) { IVX_FILLF(8); IVX_FILLF(4); IVX_FILLF(2); IVX_FILLF(1); }
                    else switch (fill_rows_per_lane(hint, n)) { case 1: IVX_FILLF(1); break; case 2: IVX_FILLF(2); break; case 4: IVX_FILLF(4); break; default: IVX_FILLF(8); }
#undef IVX_FILLF
                    hipLaunchKernelGGL(k_fill_rest, dim3(512), dim3(256), 0, st, jv, pool_se, pt, (const FpRest *)rest, (const u64 *)rest_rows, (const u32 *)rest_n, ob, op, cap, cur, s, e, rowbits);
                    IVX_HIP(ctx, hipGetLastError());
                    if (fast == 1) return IVX_OK;
                    // fast == 2: whether every region is one LDS-resident level is known on the device only (the index's build
                    // tail set hdr[HDR_SLOW] after the host had its copy): k_fill_fast has left at once if not, and the
                    // general kernel below leaves at once if so
                    slow_gate = jv.hdr + HDR_SLOW;
                }
            }
#define IVX_FILLP2(B_, P_) hipLaunchKernelGGL((k_probe_regions<1, B_, false, true, P_>), dim3(RP_GRID), dim3(RP_T), 0, st, jv, (const void *)pool_se, (const void *)pool_row, rfirst, 1u, RP_VGRID / RP_GRID, ob, op, cap, cur, 1u, 0u, (const u32 *)nullptr, dbg, pt, (const u32 *)bsel, s, e, rowbits, slow_gate)
#define IVX_FILLP(B_) do { if (packed) IVX_FILLP2(B_, true); else IVX_FILLP2(B_, false); } while (0)
            if (bsel) { IVX_FILLP(8); IVX_FILLP(4); IVX_FILLP(2); IVX_FILLP(1); }
            else switch (fill_rows_per_lane(hint, n)) { case 1: IVX_FILLP(1); break; case 2: IVX_FILLP(2); break; case 4: IVX_FILLP(4); break; default: IVX_FILLP(8); }
#undef IVX_FILLP
#undef IVX_FILLP2
        } else if (packed) {
            hipLaunchKernelGGL((k_probe_regions<0, RP_B, false, true, true>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, (const void *)pool_se, (const void *)pool_row, rfirst, 1u, 1u, ob, op, cap, cur, 1u, 0u, (const u32 *)nullptr, dbg, pt, (const u32 *)nullptr, s, e, rowbits);
        } else {
            hipLaunchKernelGGL((k_probe_regions<0, RP_B, false, true>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, (const void *)pool_se, (const void *)pool_row, rfirst, 1u, 1u, ob, op, cap, cur, 1u, 0u, (const u32 *)nullptr, dbg, pt);
        }
        IVX_HIP(ctx, hipGetLastError());
        return IVX_OK;
    }
    IVX_TRY(wait_ready());                                              // (the two-pass partition's histogram kernel may leave the rows in place: no overlap here)
    u32 *unsorted = (u32 *)(ctx->d_scalars + 10);                       // stays 0 if the rows already come in region order
    u32 chunk, nblk; u32 *hist; u64 *pse; u32 *prow;
    if (planned) {
        // the count call that sized this fill call routed the rows already (and left `unsorted` as it is)
        chunk = pl.chunk; nblk = pl.nblk;
        hist = const_cast<u32 *>(pl.hist); pse = const_cast<u64 *>(pl.pse); prow = const_cast<u32 *>(pl.prow);
        s = pl.ds; e = pl.de;
    } else {
        chunk = part_chunk(n);
        nblk = (u32)((n + chunk - 1) / chunk);
        const bool wide = nreg > IVX_MAXREG;                // 1024 digits instead of 256
        const u64 nh = (u64)(wide ? 1024 : 256) * nblk + 1;
        IVX_TRY(ctx->get_scratch(WS_SORTHIST, nh * sizeof(u32), (void **)&hist));
        IVX_TRY(ctx->get_scratch(WS_T0, n * sizeof(u64), (void **)&pse));
        IVX_TRY(ctx->get_scratch(WS_T1, n * sizeof(u32), (void **)&prow));
        IVX_HIP(ctx, hipMemsetAsync(hist + (nh - 1), 0, sizeof(u32), st));
        const bool vec = (((uintptr_t)key | (uintptr_t)s | (uintptr_t)e) & 15) == 0;
        IVX_HIP(ctx, hipMemsetAsync(unsorted, 0, sizeof(u32), st));
#define IVX_PART(V_, ND_) do { \
        hipLaunchKernelGGL((k_part_hist<V_, ND_>), dim3(nblk), dim3(PA_T), 0, st, jv, key, s, n, nblk, chunk, hist, 0u, unsorted); \
        IVX_TRY(ivx_scan_exclusive_u32(ctx, hist, nh)); \
        hipLaunchKernelGGL((k_part_scatter<V_, u32, ND_>), dim3(nblk), dim3(PA_T), 0, st, jv, key, s, e, n, nblk, (const u32 *)hist, pse, prow, chunk, 0u, (const u32 *)unsorted, dbg); } while (0)
        if (wide) { if (vec) IVX_PART(true, 1024); else IVX_PART(false, 1024); }
        else { if (vec) IVX_PART(true, 256); else IVX_PART(false, 256); }
#undef IVX_PART
        if (mode == JP_COUNT) {
            // leave the routed rows for the fill call (ivx_capi.hip fills in whose columns they are)
            pl.hist = hist; pl.pse = pse; pl.prow = prow; pl.ds = s; pl.de = e; pl.chunk = chunk; pl.nblk = nblk; pl.paged = false;
            pl.slots = (1ull << WS_SORTHIST) | (1ull << WS_T0) | (1ull << WS_T1) | (1ull << WS_IN_START) | (1ull << WS_IN_END);
            pl.valid = true;
        }
    }
    unsigned long long *cur = (unsigned long long *)d_cursor;
    // how dense the matches are: from the caller's capacity, or -- planned -- from what the count call found
    const u64 hint = ctx->fill_hint ? ctx->fill_hint : (planned && pl.total < cap ? pl.total : cap);
    if (mode == JP_FILL && dense_fill_wanted(hint, n)) {
        return dense_fill(ctx, jv, nreg, (const void *)pse, (const void *)prow, 1u, s, e, (const u32 *)hist, nblk, (const u32 *)unsorted, n, ob, op, cap, d_cursor);
    } else if (mode == JP_FILL) {   // single walk: pairs staged per wavefront, one output reservation per workgroup and round
        // rows per lane and batch by the expected matches per row (cap / n: callers size the output from the
        // count pass): two consecutive rounds of a wavefront must fit its 512-pair staging ring, else the
        // batch takes the slow direct path
        const int b = fill_rows_per_lane(hint, n);
#define IVX_FILL1(B_, ID_) hipLaunchKernelGGL((k_probe_regions<1, B_, ID_>), dim3(RP_GRID), dim3(RP_T), 0, st, jv, ID_ ? (const void *)s : (const void *)pse, ID_ ? (const void *)e : (const void *)prow, (const u32 *)hist, nblk, RP_VGRID / RP_GRID, ob, op, cap, cur, 1u, 0u, (const u32 *)unsorted, dbg)
#define IVX_FILL(B_) do { IVX_FILL1(B_, false); IVX_FILL1(B_, true); } while (0)
        switch (b) { case 1: IVX_FILL(1); break; case 2: IVX_FILL(2); break; case 4: IVX_FILL(4); break; default: IVX_FILL(8); }
#undef IVX_FILL
#undef IVX_FILL1
    } else {
        hipLaunchKernelGGL((k_probe_regions<0, RP_B, false>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, (const void *)pse, (const void *)prow, (const u32 *)hist, nblk, 1u, ob, op, cap, cur, 1u, 0u, (const u32 *)unsorted, dbg);
        hipLaunchKernelGGL((k_probe_regions<0, RP_B, true>), dim3(RP_VGRID), dim3(RP_T), 0, st, jv, (const void *)s, (const void *)e, (const u32 *)hist, nblk, 1u, ob, op, cap, cur, 1u, 0u, (const u32 *)unsorted, dbg);
    }
    IVX_HIP(ctx, hipGetLastError());
    return IVX_OK;
}
