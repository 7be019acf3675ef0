// ivx_internal.hpp -- shared host/device definitions of libivx_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <memory>
#include <string>
#include <vector>
#include "../../include/ivx.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef int32_t i32;
typedef uint64_t u64;
typedef int64_t i64;

#define IVX_WAVE 64

// ---------------------------------------------------------------- context
struct ivx_buf { void *p = nullptr; size_t cap = 0; };

enum { IVX_NSCRATCH = 56, IVX_NPIN = 4 };

// What a subtract sizing call leaves behind for the fill call that follows it: the sorted sides, the right
// side's running max / gap heads and the scanned per-row output counts all sit in scratch slots (`slots` =
// bit per slot); touching any of them drops the plan.
struct ivx_sub_plan {
    bool valid = false;
    u64 slots = 0;
    int mem = 0;
    const void *in[6] = {};             // the caller's six input columns
    u64 nl = 0, nr = 0, nh = 0, total = 0;
    u32 nkeys = 0;
    int strict = 0;
    hipStream_t stream = nullptr;
    const u32 *lk, *lrow, *rk, *hk, *hj;
    const i64 *lsv, *lev, *rsv, *hrs, *hpm;
    const void *sm;
    const u64 *offs;
    const u32 *plan_hlo; const i64 *plan_tail;      // per left row: its first gap head, where its tail fragment starts
};

// What a region-partitioned overlap COUNT call leaves behind for the fill call that follows it: the probe
// rows routed to their regions (tile histogram, partitioned (start,end) and row ids) -- the fill call for
// the same index, columns and stream goes straight to its probe kernel.
struct ivx_join_plan {
    bool valid = false;
    u64 slots = 0;
    int mem = 0;
    const void *in[3] = {};
    u64 n = 0;
    u64 total = 0;                      // the pairs the count call found: the fill call's density hint, whatever its cap
    const void *ix = nullptr; u64 ix_serial = 0;
    hipStream_t stream = nullptr;
    const u32 *hist = nullptr; const u64 *pse = nullptr; const u32 *prow = nullptr;
    const i32 *ds = nullptr, *de = nullptr;     // device copies of the start / end columns (read in place when the rows were in region order)
    u32 chunk = 0, nblk = 0;
    // one-pass routing (ivx_join_regions.hip): the rows sit in pages; hist = first routed row of every region,
    // ptab = [region][page slot] -> page + 1
    bool paged = false, packed = false;   // packed: 8-byte routed rows (start in region | length | row), else (start,end) + row id
    const u32 *ptab = nullptr; u32 pstride = 0, lgpg = 0, rowbits = 32;
    void *rest = nullptr;                 // the lean fill kernel's list of batches left to the generic walk (scratch, in `slots`)
    bool all_routed = false;              // no occupancy bitmap in use: every row was routed (the fill's rows-per-lane rule is then known on the host)
};

struct ivx_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = 0.0;
    std::string err;
    ivx_buf scratch[IVX_NSCRATCH];      // grow-only device scratch, one slot per use
    ivx_buf pinned[IVX_NPIN];           // grow-only pinned host staging
    u64 *d_scalars = nullptr;           // 64 device words for counters / totals
    u64 *h_scalars = nullptr;           // pinned mirror
    ivx_sub_plan sub_plan;
    ivx_join_plan join_plan;
    ivx_metrics metrics{};              // BuildProbeJoinMetrics, see ivx.h
    u64 mem_limit = 0;                  // device bytes of scratch + the index being built (0 = unlimited)
    u64 fill_hint = 0;                  // pairs a fill call should expect instead of its cap (a chunk of a host-resident batch writes into the whole batch's buffers)
    u64 scratch_bytes = 0, building_bytes = 0;
    // Build overlap (ivx_ctx_set_build_overlap): an overlap index build returns once the per-key tables and the region layout
    // are final -- all a probe's routing pass reads -- and leaves the rest (cell count, scan, scatter, region descriptors) running
    // on `aux`; `tail_ev` marks its end.  A probe's routing then runs beside it and only its probe kernel waits for the
    // index's `ready` event; anything else that could touch what the tail still uses waits first (get_scratch, join_tail).
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, tail_ev = nullptr;
    bool overlap = false, tail_pending = false;
    void join_tail();                   // order this context's stream behind a pending build tail
    // device bytes of the indexes this context built that are still alive (shared with those indexes: ivx_index_free
    // gives the bytes back whichever thread calls it, also after the context is gone)
    std::shared_ptr<std::atomic<u64>> live_index_bytes = std::make_shared<std::atomic<u64>>(0);
    u64 reserved() const { return scratch_bytes + building_bytes + live_index_bytes->load(std::memory_order_relaxed); }

    ivx_status fail(ivx_status st, const std::string &msg) { err = msg; return st; }
    ivx_status fail_hip(const char *what, hipError_t e)
    {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? IVX_ERR_OOM : IVX_ERR_HIP;
    }
    // device scratch of at least `bytes` in slot `slot` (contents undefined)
    ivx_status get_scratch(int slot, size_t bytes, void **out);
    ivx_status get_pinned(int slot, size_t bytes, void **out);
};

#define IVX_HIP(ctx, call)                                              \
    do {                                                                \
        hipError_t e__ = (call);                                        \
        if (e__ != hipSuccess) return (ctx)->fail_hip(#call, e__);      \
    } while (0)

#define IVX_TRY(call)                                                   \
    do {                                                                \
        ivx_status s__ = (call);                                        \
        if (s__ != IVX_OK) return s__;                                  \
    } while (0)

// scratch slot ids (one per concurrent use inside a call)
enum {
    WS_IN_KEY = 0, WS_IN_START, WS_IN_END, WS_IN2_KEY, WS_IN2_START, WS_IN2_END,   // staged host inputs
    WS_OUT_A, WS_OUT_B, WS_OUT_C, WS_OUT_D, WS_OUT_E, WS_OUT_F, WS_OUT_G, WS_OUT_H,     // staged host outputs
    WS_SCAN0, WS_SCAN1, WS_SCAN2,                                                   // scan partials
    WS_GRID0, WS_GRID1, WS_GRID2,                                                   // key stats + cell cursors
    WS_SORTHIST,                                                                    // radix-sort histograms
    WS_SA0, WS_SA1, WS_SA2, WS_SB0, WS_SB1, WS_SB2,                                 // sort ping/pong records
    WS_RA0, WS_RA1, WS_RA2, WS_RB0, WS_RB1, WS_RB2,                                 // second record set (subtract's right side)
    WS_T0, WS_T1, WS_T2, WS_T3, WS_T4, WS_T5, WS_T6, WS_T7, WS_T8, WS_T9            // per-op temporaries
};

// ---------------------------------------------------------------- indexes
#ifndef IVX_LSTEP
#define IVX_LSTEP 2       // bin width grows 4x per level: a level's cell is at most 4x wider than the rows it holds
#endif
#define IVX_MAXL (32 / IVX_LSTEP)   // length-class levels of the binned overlap index
#define IVX_SH_MIN 4

struct ivx_ent { i32 s, e; u32 row; };          // 12-byte AoS entry of the overlap index

// header words written by the layout kernel (device resident, read by probes)
enum { HDR_SH0 = 0, HDR_NLEV = 1, HDR_NBINS = 2, HDR_CS = 3 /* log2(cells per region) or ~0u */, HDR_NREG = 4,
       HDR_RCELLS = 5 /* cells per region */, HDR_RMUL_LO = 6, HDR_RMUL_HI = 7 /* ceil(2^40 / cells per region) */, HDR_LEVCNT = 8 /* .. +IVX_MAXL */,
       HDR_FG = 8 + IVX_MAXL /* log2(block width) of the occupancy bitmap, or ~0u: no bitmap */, HDR_FBITS = 9 + IVX_MAXL /* its size in bits */,
       HDR_PK24 = 10 + IVX_MAXL /* 1: a region spans at most 2^24 coordinates (routed rows pack into 8 bytes) */,
       HDR_SLOW = 11 + IVX_MAXL /* 0: every region is ONE LDS-resident level (all build rows in level 0, every slice fits): the lean fill probe applies */,
       HDR_WORDS = 12 + IVX_MAXL };
// Occupancy bitmap of the build side ("can a probe row match anything at all"): per key one bit per 2^g-wide block of
// [origin, origin + span] plus one overflow block behind it; a bit is set when some build row touches the block.  Sized to
// stay resident in an XCD's 4 MiB L2 next to the streamed probe rows.
#define IVX_FBITS_MAX (1u << 24)    // 2 MiB
#define IVX_MAXREG 255   // probe regions routed with ONE partition pass (one radix digit; 255 = rows that cannot match)
#define IVX_MAXREG_WIDE 1023   // ... still one pass, with 1024 digits (overlap join count / fill only)
#define IVX_MAXREG2 65025u   // most regions at all: beyond 255 the probe rows are routed by a two-digit stable sort
#define IVX_REG_CS_MAX 13   // a region spans at most 2^13 level-0 cells, what a workgroup can stage in LDS

// what a probe workgroup needs to stage one region's slice of level 0 (ivx_join_regions.hip), precomputed at
// build time so that staging starts with ONE load instead of a chain of four dependent ones
#ifndef IVX_RP_ECAP
#define IVX_RP_ECAP 6144          // level-0 entries a region's LDS slice holds (ivx_join_regions.hip)
#endif
#define IVX_RP_HALO 8u      // slice cells past the region's last cell
struct ivx_regdesc { u32 k; i32 origin; u32 span, lb, slo, shi, e0, ne; i32 rbase; /* coordinate of the region's first cell */ };

struct JoinIndexView {
    const i32 *origin;      // [nkeys] smallest start of the key
    const u32 *span;        // [nkeys] largest start - smallest start
    const u32 *kcnt;        // [nkeys] build rows of the key
    const u32 *lbase;       // [IVX_MAXL*nkeys] first bin of (level,key)
    const u32 *binstart;    // [nbins+1] CSR offsets into ent
    const ivx_ent *ent;     // [n] entries grouped by bin
    const u32 *hdr;         // [HDR_WORDS]
    const u32 *kreg;        // [nkeys+1] first probe region of the key (regions never straddle keys)
    const u32 *rkey;        // [IVX_MAXREG2+1] key of a region
    const ivx_regdesc *rdesc;   // [IVX_MAXREG2+1] slice window of a region
    const u32 *fbits;       // occupancy bitmap words (+2 words of padding), valid when hdr[HDR_FG] != ~0u
    const u32 *fbase;       // [nkeys] first bit of the key (a multiple of 32)
    u32 nkeys;
};

// sorted-rank grid over one int32 column grouped by key (count_overlaps,
// coverage, nearest): rank(x) = cum[bin(x)] + #{v in bin : v <= x}
struct RankGridView {
    const i32 *origin;      // [nkeys] smallest value of the key
    const u32 *span;        // [nkeys]
    const u32 *kcnt;        // [nkeys]
    const u32 *koff;        // [nkeys+1] rows before the key
    const u32 *kbase;       // [nkeys] first bin of the key
    const u32 *binstart;    // [nbins+1]
    const i32 *val;         // [n] values grouped by (key,bin); sorted if `sorted`
    const u32 *hdr;         // [0] = shift
    u32 nkeys;
};

struct CoverageView {
    RankGridView first, last;   // over merged nodes' first / last (both ascending per key)
    const i64 *pw;              // [m+1] prefix sums of max(1, last-first)
    const i32 *nfirst, *nlast;  // [m] merged nodes in order
};

// one build row of the nearest index, 16 bytes so that a candidate is ONE load: in by_start order
// a = start, b = end, pmax = running max of end within the key; in by_end order a = end, b = start
struct ivx_nrec { i32 a, b; u32 row; i32 pmax; };

struct NearestView {
    RankGridView by_start, by_end, pmax;   // rank grids over rs[].a / re[].a / rs[].pmax (value stride 4 words)
    const ivx_nrec *rs;                    // by_start order (start,end,row)
    const ivx_nrec *re;                    // by_end order (end,start,row)
    JoinIndexView ov;                               // overlap index for k>1 include_overlaps
};

struct ivx_index {
    u64 serial = 0;             // unique per built index (a freed index's address may come back)
    int kind = 0;
    int device = 0;
    u64 n = 0;
    u32 nkeys = 0;
    size_t bytes = 0;
    std::shared_ptr<std::atomic<u64>> owner_bytes;   // the building context's live_index_bytes (null until the build succeeded)
    std::vector<void *> allocs;
    std::vector<size_t> alloc_caps;
    JoinIndexView jv{};
    u32 jv_nreg = 0;            // >0: the region-partitioned probe is available
    bool jv_filter = false;     // jv carries an occupancy bitmap (hdr[HDR_FG] != ~0u)
    bool jv_pk24 = false;       // hdr[HDR_PK24]
    bool jv_fast = false;       // !hdr[HDR_SLOW]
    bool jv_fast_unknown = false;   // built with the tail overlapped: hdr[HDR_SLOW] is final only behind `ready` (kernels test it themselves)
    hipEvent_t ready = nullptr; // set when the build's tail ran on the aux stream: everything but the routing tables is valid behind it
    RankGridView gs{}, ge{};
    CoverageView cv{};
    NearestView nv{};
    JoinIndexView nroute{};     // nearest index: regions that only ROUTE big probe batches (origin/span/kcnt/kreg/rkey/hdr; no cells)
    u32 nroute_nreg = 0;
    int flags = 0;              // IVX_IXF_*
};
enum { IVX_IXF_REGION_ROWVAL = 1 };   // count/coverage index: jv is usable for the region-partitioned per-row probe

// probe rows routed to the regions of a routing view (ivx_route_rows): scanned [1024][nblk] histogram (region r starts at
// hist[r * nblk]), the rows' (start,end) in routed order, each row's index inside its one-tile chunk; *unsorted == 0:
// the rows came in region order already and nothing was moved
struct ivx_routed { const u32 *hist; const u64 *pse; const unsigned short *cidx; const u32 *unsorted; u32 nblk, chunk; };

// ---------------------------------------------------------------- internal API
// scan.hip
ivx_status ivx_scan_exclusive_u32(ivx_ctx *ctx, u32 *data, u64 n);      // in place, uses WS_SCAN*
ivx_status ivx_scan_exclusive_u64(ivx_ctx *ctx, u64 *data, u64 n);

// join.hip
ivx_status ivx_join_build(ivx_ctx *ctx, ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n, bool overlap = false);
ivx_status ivx_join_probe(ivx_ctx *ctx, const JoinIndexView &jv, int mode,
                          const u32 *key, const i32 *s, const i32 *e, u64 n,
                          u32 *per_row, u8 *exists, u32 *ob, u32 *op, u64 cap, u64 *d_cursor);
enum { JP_COUNT = 0, JP_PER_ROW = 1, JP_EXISTS = 2, JP_FILL = 3 };
// join_regions.hip: partition the probe rows by index region, probe each region from LDS
ivx_status ivx_join_probe_regions(ivx_ctx *ctx, const JoinIndexView &jv, u32 nreg, int mode,
                                  const u32 *key, const i32 *s, const i32 *e, u64 n,
                                  u32 *ob, u32 *op, u64 cap, u64 *d_cursor, bool planned = false, bool has_filter = false, bool pk24 = false,
                                  int fast = 0 /* 0 no, 1 yes, 2 the kernels test hdr[HDR_SLOW] themselves */, hipEvent_t ready = nullptr);

// per-row-output operators through the same partition (count_overlaps: jv over the build rows, no row with
// end < start; coverage: jv over the merged nodes)
enum { IVX_RV_COUNT = 0, IVX_RV_COVERAGE = 1, IVX_RV_PER_ROW = 2, IVX_RV_EXISTS = 3 };
ivx_status ivx_route_view_build(ivx_ctx *ctx, ivx_index *ix, const i32 *origin, const u32 *span, const u32 *kcnt);
void ivx_route_view_ready(ivx_ctx *ctx, ivx_index *ix);
ivx_status ivx_join_rowval_routed(ivx_ctx *ctx, const ivx_index *ix, int mode, const u32 *key, const i32 *s, const i32 *e, u64 n,
                                  u32 *per_row, u8 *exists, u64 *d_total);
ivx_status ivx_unroute_u32(ivx_ctx *ctx, const ivx_routed &r, u64 n, const u32 *vb, u32 *out32, u8 *out8);
ivx_status ivx_route_rows(ivx_ctx *ctx, const JoinIndexView &rv, const u32 *key, const i32 *s, const i32 *e, u64 n, u32 adj, ivx_routed *out);
ivx_status ivx_unroute_pair(ivx_ctx *ctx, const ivx_routed &r, u64 n, const u32 *vb, const i64 *vd, u32 *ob, u32 *op, i64 *od, i64 dflt);
ivx_status ivx_rowval_probe_regions(ivx_ctx *ctx, const JoinIndexView &jv, u32 nreg, int kind,
                                    const u32 *key, const i32 *s, const i32 *e, u64 n, int strict, void *out, u64 *d_total,
                                    bool has_filter = false, bool pk24 = false, bool fast = false);

ivx_status ivx_index_alloc(ivx_ctx *ctx, ivx_index *ix, size_t bytes, void **out);
