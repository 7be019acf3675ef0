// ivx_capi.hip -- extern "C" entry points of libivx_hip.so (see include/ivx.h).
// Host-side plumbing only: argument checks, staging of host buffers, stream and
// scratch management.  All arithmetic happens in the HIP kernels.
#include "ivx_internal.hpp"
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <condition_variable>
#include <deque>

// ------------------------------------------------------------- ctx helpers

void ivx_ctx::join_tail()
{
    if (!tail_pending) return;
    (void)hipStreamWaitEvent(stream, tail_ev, 0);
    tail_pending = false;
}

ivx_status ivx_ctx::get_scratch(int slot, size_t bytes, void **out)
{
    // a build tail still running on the aux stream uses the build's scratch (cell ids, ranks, scan partials): whoever asks
    // for scratch is ordered behind it first -- except for the slots a probe's routing pass takes, which the tail never
    // touches: that pass is what runs beside the tail
    if (tail_pending && slot != WS_SORTHIST && slot != WS_T0 && slot != WS_T1 && slot != WS_T2) join_tail();
    ivx_buf &b = scratch[slot];
    if (sub_plan.valid && ((sub_plan.slots >> slot) & 1)) sub_plan.valid = false;
    if (join_plan.valid && ((join_plan.slots >> slot) & 1)) join_plan.valid = false;
    if (bytes < 256) bytes = 256;
    if (b.cap < bytes) {
        size_t want = bytes + bytes / 8;
        if (mem_limit) {
            const u64 others = reserved() - b.cap;
            if (others + bytes > mem_limit)
                return fail(IVX_ERR_OOM, "Resources exhausted: failed to reserve " + std::to_string(bytes) + " bytes of device scratch (" +
                                         std::to_string(others) + " bytes reserved, limit " + std::to_string(mem_limit) + ")");
            if (others + want > mem_limit) want = bytes;
        }
        if (b.p) { (void)hipStreamSynchronize(stream); if (aux) (void)hipStreamSynchronize(aux); (void)hipFree(b.p); b.p = nullptr; scratch_bytes -= b.cap; b.cap = 0; }
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) { want = bytes; e = hipMalloc(&b.p, want); }
        if (e != hipSuccess) { b.p = nullptr; return fail_hip("hipMalloc(scratch)", e); }
        b.cap = want; scratch_bytes += want;
    }
    *out = b.p;
    return IVX_OK;
}

ivx_status ivx_ctx::get_pinned(int slot, size_t bytes, void **out)
{
    ivx_buf &b = pinned[slot];
    if (bytes < 256) bytes = 256;
    if (b.cap < bytes) {
        if (b.p) { (void)hipStreamSynchronize(stream); (void)hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
        hipError_t e = hipHostMalloc(&b.p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { b.p = nullptr; return fail_hip("hipHostMalloc", e); }
        b.cap = bytes;
    }
    *out = b.p;
    return IVX_OK;
}

// Index buffers are recycled through a small per-process pool: IntervalJoinExec builds one
// index per query/partition and drops it, and hipMalloc/hipFree cost more than the build kernels.
namespace {
struct PoolBuf { void *p; size_t cap; int device; };
std::mutex g_pool_mu;
std::vector<PoolBuf> g_pool;
constexpr size_t POOL_MAX_BUFS = 96;

void *pool_take(int device, size_t bytes, size_t *cap)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t best = g_pool.size();
    for (size_t i = 0; i < g_pool.size(); i++) {
        const PoolBuf &b = g_pool[i];
        if (b.device != device || b.cap < bytes || b.cap > 2 * bytes + 4096) continue;
        if (best == g_pool.size() || b.cap < g_pool[best].cap) best = i;
    }
    if (best == g_pool.size()) return nullptr;
    void *p = g_pool[best].p;
    *cap = g_pool[best].cap;
    g_pool.erase(g_pool.begin() + best);
    return p;
}

void pool_give(int device, void *p, size_t cap)
{
    void *drop = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_pool.push_back({p, cap, device});
        if (g_pool.size() > POOL_MAX_BUFS) { drop = g_pool.front().p; g_pool.erase(g_pool.begin()); }
    }
    if (drop) (void)hipFree(drop);
}

void pool_drop(int device)
{
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size();)
            if (g_pool[i].device == device) { drop.push_back(g_pool[i].p); g_pool.erase(g_pool.begin() + i); }
            else i++;
    }
    for (void *p : drop) (void)hipFree(p);
}
}  // namespace

ivx_status ivx_index_alloc(ivx_ctx *ctx, ivx_index *ix, size_t bytes, void **out)
{
    if (bytes < 256) bytes = 256;
    if (ctx->mem_limit && ctx->reserved() + bytes > ctx->mem_limit)
        return ctx->fail(IVX_ERR_OOM, "Resources exhausted: failed to reserve " + std::to_string(bytes) + " bytes for the index (" +
                                      std::to_string(ctx->reserved()) + " bytes reserved, limit " + std::to_string(ctx->mem_limit) + ")");
    ctx->building_bytes += bytes;
    size_t cap = bytes;
    void *p = pool_take(ctx->device, bytes, &cap);
    if (!p) {
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return ctx->fail_hip("hipMalloc(index)", e);
    }
    ix->allocs.push_back(p);
    ix->alloc_caps.push_back(cap);
    ix->bytes += bytes;
    *out = p;
    return IVX_OK;
}

namespace {

// stage one input column: host -> device scratch, or pass the device pointer through
template <typename T>
ivx_status stage_in(ivx_ctx *ctx, int mem, int slot, const T *src, u64 n, const T **dev)
{
    if (src == nullptr) { *dev = nullptr; return IVX_OK; }
    if (mem == IVX_MEM_DEVICE) { *dev = src; return IVX_OK; }
    T *d;
    IVX_TRY(ctx->get_scratch(slot, (size_t)n * sizeof(T), (void **)&d));
    if (n) IVX_HIP(ctx, hipMemcpyAsync(d, src, (size_t)n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    *dev = d;
    return IVX_OK;
}

// output column: device scratch (host mode) or the caller's device pointer
template <typename T>
ivx_status stage_out(ivx_ctx *ctx, int mem, int slot, T *dst, u64 n, T **dev)
{
    if (dst == nullptr) { *dev = nullptr; return IVX_OK; }
    if (mem == IVX_MEM_DEVICE) { *dev = dst; return IVX_OK; }
    T *d;
    IVX_TRY(ctx->get_scratch(slot, (size_t)n * sizeof(T), (void **)&d));
    *dev = d;
    return IVX_OK;
}

template <typename T>
ivx_status copy_out(ivx_ctx *ctx, int mem, T *dst, const T *dev, u64 n)
{
    if (mem == IVX_MEM_DEVICE || dst == nullptr || n == 0) return IVX_OK;
    IVX_HIP(ctx, hipMemcpyAsync(dst, dev, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    return IVX_OK;
}

ivx_status read_scalar(ivx_ctx *ctx, int word, u64 *out)
{
    IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + word, ctx->d_scalars + word, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = ctx->h_scalars[word];
    return IVX_OK;
}

ivx_status check_probe_args(ivx_ctx *ctx, const ivx_index *ix, int kind, int mem, const void *s, const void *e, u64 n, bool defer_ready = false)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (!ix) return ctx->fail(IVX_ERR_INVALID, "null index");
    if (ix->kind != kind) return ctx->fail(IVX_ERR_UNSUPPORTED, "index kind does not match this probe");
    if (ix->device != ctx->device) return ctx->fail(IVX_ERR_INVALID, "index lives on another device");
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if (n && (!s || !e)) return ctx->fail(IVX_ERR_INVALID, "null coordinate column");
    if (n > 0xFFFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "probe batch exceeds UInt32 index capacity");
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    // an index whose build tail ran beside other work is complete behind its `ready` event: this context's stream is
    // ordered behind it here, unless the caller takes care of that itself (the region path: after its routing pass)
    if (ix->ready != nullptr && !defer_ready) IVX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ix->ready, 0));
    return IVX_OK;
}

// BuildProbeJoinMetrics bookkeeping of one call (host wall time, as the reference's timers)
struct CallMetrics {
    ivx_ctx *c; bool build; std::chrono::steady_clock::time_point t0;
    CallMetrics(ivx_ctx *ctx, bool is_build, u64 rows) : c(ctx), build(is_build), t0(std::chrono::steady_clock::now())
    {
        if (build) { c->metrics.build_input_batches++; c->metrics.build_input_rows += rows; }
        else { c->metrics.input_batches++; c->metrics.input_rows += rows; }
    }
    void out(u64 rows) { if (rows) { c->metrics.output_batches++; c->metrics.output_rows += rows; } }
    ~CallMetrics()
    {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        (build ? c->metrics.build_time : c->metrics.join_time) += ms;
    }
};

struct KernelTimer {
    ivx_ctx *c;
    explicit KernelTimer(ivx_ctx *ctx) : c(ctx) { (void)hipEventRecord(c->ev0, c->stream); c->last_ms = -1.0; }
    ~KernelTimer() { (void)hipEventRecord(c->ev1, c->stream); }
};

}  // namespace

// ------------------------------------------------------------------ ctx

extern "C" const char *ivx_version(void) { return "ivx-hip 0.1 (gfx950)"; }

// the build tail's stream: highest priority, so that its short kernels take the workgroup slots a long routing kernel frees
// first and are done long before that kernel is
static bool create_aux_stream(hipStream_t *out)
{
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi) == hipSuccess) return true;
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking) == hipSuccess;
}

extern "C" ivx_status ivx_ctx_create(int device_ordinal, ivx_ctx **out)
{
    if (!out) return IVX_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return IVX_ERR_NO_DEVICE;   // no CPU fallback
    if (device_ordinal < 0 || device_ordinal >= ndev) return IVX_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return IVX_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return IVX_ERR_NO_DEVICE;       // code objects are gfx950 only
    ivx_ctx *c = new (std::nothrow) ivx_ctx();
    if (!c) return IVX_ERR_OOM;
    c->device = device_ordinal;
    bool ok = hipSetDevice(device_ordinal) == hipSuccess &&
              hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess &&
              create_aux_stream(&c->aux) &&
              hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->tail_ev, hipEventDisableTiming) == hipSuccess &&
              hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
              hipMalloc((void **)&c->d_scalars, 64 * sizeof(u64)) == hipSuccess &&
              hipHostMalloc((void **)&c->h_scalars, 64 * sizeof(u64), hipHostMallocDefault) == hipSuccess;
    if (!ok) { ivx_ctx_free(c); return IVX_ERR_HIP; }
    c->stream = c->own_stream;
    if (const char *e = getenv("IVX_BUILD_OVERLAP")) c->overlap = e[0] == '1';
    *out = c;
    return IVX_OK;
}

extern "C" void ivx_ctx_free(ivx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    for (auto &b : c->scratch) if (b.p) (void)hipFree(b.p);
    for (auto &b : c->pinned) if (b.p) (void)hipHostFree(b.p);
    if (c->d_scalars) (void)hipFree(c->d_scalars);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->tail_ev) (void)hipEventDestroy(c->tail_ev);
    delete c;
}

extern "C" const char *ivx_last_error(const ivx_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

extern "C" ivx_status ivx_ctx_set_stream(ivx_ctx *c, void *hip_stream)
{
    if (!c) return IVX_ERR_INVALID;
    if (c->tail_pending) { (void)hipStreamSynchronize(c->aux); c->tail_pending = false; }
    c->stream = (hipStream_t)hip_stream;
    return IVX_OK;
}

extern "C" ivx_status ivx_ctx_use_own_stream(ivx_ctx *c)
{
    if (!c) return IVX_ERR_INVALID;
    if (c->tail_pending) { (void)hipStreamSynchronize(c->aux); c->tail_pending = false; }
    c->stream = c->own_stream;
    return IVX_OK;
}

extern "C" ivx_status ivx_ctx_set_build_overlap(ivx_ctx *c, int on)
{
    if (!c) return IVX_ERR_INVALID;
    c->overlap = on != 0;
    return IVX_OK;
}

extern "C" ivx_status ivx_ctx_synchronize(ivx_ctx *c)
{
    if (!c) return IVX_ERR_INVALID;
    if (c->tail_pending) { IVX_HIP(c, hipStreamSynchronize(c->aux)); c->tail_pending = false; }
    IVX_HIP(c, hipStreamSynchronize(c->stream));
    return IVX_OK;
}

extern "C" ivx_status ivx_ctx_metrics(const ivx_ctx *c, ivx_metrics *out)
{
    if (!c || !out) return IVX_ERR_INVALID;
    *out = c->metrics;
    return IVX_OK;
}

extern "C" void ivx_ctx_reset_metrics(ivx_ctx *c) { if (c) c->metrics = ivx_metrics{}; }

extern "C" ivx_status ivx_ctx_set_memory_limit(ivx_ctx *c, uint64_t bytes)
{
    if (!c) return IVX_ERR_INVALID;
    c->mem_limit = bytes;
    return IVX_OK;
}

extern "C" uint64_t ivx_ctx_reserved_bytes(const ivx_ctx *c) { return c ? c->reserved() : 0; }

// Scratch back to the device: the reference's build-side reservation is returned when the stream ends
// (interval_join.rs:614-639); a context that served a 10^9-row call would otherwise keep ~100 GB for good.
extern "C" ivx_status ivx_ctx_trim(ivx_ctx *c, uint64_t keep_bytes)
{
    if (!c) return IVX_ERR_INVALID;
    IVX_HIP(c, hipSetDevice(c->device));
    IVX_HIP(c, hipStreamSynchronize(c->stream));
    if (c->aux) { IVX_HIP(c, hipStreamSynchronize(c->aux)); c->tail_pending = false; }
    c->sub_plan.valid = false; c->join_plan.valid = false;      // both live in scratch slots
    while (c->scratch_bytes > keep_bytes) {
        int big = -1;
        for (int i = 0; i < IVX_NSCRATCH; i++)
            if (c->scratch[i].p && (big < 0 || c->scratch[i].cap > c->scratch[big].cap)) big = i;
        if (big < 0) break;
        IVX_HIP(c, hipFree(c->scratch[big].p));
        c->scratch_bytes -= c->scratch[big].cap;
        c->scratch[big] = ivx_buf{};
    }
    pool_drop(c->device);                                       // recycled index buffers of freed indexes on this device, too
    return IVX_OK;
}

extern "C" double ivx_ctx_last_kernel_ms(const ivx_ctx *cc)
{
    ivx_ctx *c = const_cast<ivx_ctx *>(cc);
    if (!c) return -1.0;
    if (c->last_ms < 0.0) {
        float ms = 0.f;
        if (hipEventSynchronize(c->ev1) == hipSuccess && hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms;
    }
    return c->last_ms;
}

// ---------------------------------------------------------------- index

ivx_status ivx_count_build(ivx_ctx *ctx, ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n);
ivx_status ivx_coverage_build(ivx_ctx *ctx, ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n);
ivx_status ivx_nearest_build(ivx_ctx *ctx, ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n);

extern "C" ivx_status ivx_index_build(ivx_ctx *ctx, int kind, int mem, const uint32_t *key, const int32_t *start,
                                      const int32_t *end, uint64_t n, uint32_t n_keys, ivx_index **out)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (!out) return ctx->fail(IVX_ERR_INVALID, "null out");
    *out = nullptr;
    if (kind < IVX_KIND_OVERLAP || kind > IVX_KIND_NEAREST) return ctx->fail(IVX_ERR_INVALID, "bad index kind");
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if (n && (!start || !end)) return ctx->fail(IVX_ERR_INVALID, "null coordinate column");
    if (n > 0xFFFFFFFEull) return ctx->fail(IVX_ERR_INVALID, "build side exceeds UInt32 index capacity");   // interval_join.rs:759
    if (n_keys == 0) n_keys = 1;
    if (!key) n_keys = 1;
    if (kind == IVX_KIND_NEAREST && n >= 0x80000000ull) return ctx->fail(IVX_ERR_INVALID, "nearest index: more than 2^31-1 rows");
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    ivx_index *ix = new (std::nothrow) ivx_index();
    static std::atomic<u64> next_serial{1};
    if (ix) ix->serial = next_serial.fetch_add(1);
    if (!ix) return ctx->fail(IVX_ERR_OOM, "host allocation failed");
    ix->kind = kind; ix->device = ctx->device; ix->n = n; ix->nkeys = n_keys;
    CallMetrics cm(ctx, true, n);
    ctx->building_bytes = 0;
    const u32 *dk; const i32 *ds, *de;
    ivx_status st = stage_in(ctx, mem, WS_IN_KEY, key, n, &dk);
    if (st == IVX_OK) st = stage_in(ctx, mem, WS_IN_START, start, n, &ds);
    if (st == IVX_OK) st = stage_in(ctx, mem, WS_IN_END, end, n, &de);
    if (st == IVX_OK) {
        KernelTimer t(ctx);
        switch (kind) {
        // (overlapped tail: device-resident columns only -- staged host columns live in scratch the next call reuses)
        case IVX_KIND_OVERLAP: st = ivx_join_build(ctx, ix, dk, ds, de, n, mem == IVX_MEM_DEVICE); break;
        case IVX_KIND_COUNT: st = ivx_count_build(ctx, ix, dk, ds, de, n); break;
        case IVX_KIND_COVERAGE: st = ivx_coverage_build(ctx, ix, dk, ds, de, n); break;
        default: st = ivx_nearest_build(ctx, ix, dk, ds, de, n); break;
        }
    }
    if (st == IVX_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = ctx->fail(IVX_ERR_HIP, "index build failed on device");
    ctx->building_bytes = 0;
    if (st != IVX_OK) { ivx_index_free(ix); return st; }
    // the finished index stays reserved against the context's limit until ivx_index_free (the reference holds the
    // build side's MemoryReservation until the join stream ends, interval_join.rs:614-639)
    ix->owner_bytes = ctx->live_index_bytes;
    ix->owner_bytes->fetch_add(ix->bytes, std::memory_order_relaxed);
    ctx->metrics.build_mem_used += ix->bytes;
    *out = ix;
    return IVX_OK;
}

extern "C" void ivx_index_free(ivx_index *ix)
{
    if (!ix) return;
    // the caller guarantees no probe on this index is still running (same contract as dropping
    // Arc<JoinLeftData>); buffers go back to the pool, to be reused only by later builds
    if (ix->ready) { (void)hipEventSynchronize(ix->ready); (void)hipEventDestroy(ix->ready); }   // (its build tail may still be writing)
    for (size_t i = 0; i < ix->allocs.size(); i++) pool_give(ix->device, ix->allocs[i], ix->alloc_caps[i]);
    if (ix->owner_bytes) ix->owner_bytes->fetch_sub(ix->bytes, std::memory_order_relaxed);
    delete ix;
}

extern "C" uint64_t ivx_index_rows(const ivx_index *ix) { return ix ? ix->n : 0; }
extern "C" uint64_t ivx_index_device_bytes(const ivx_index *ix) { return ix ? ix->bytes : 0; }

// ---------------------------------------------------------------- a3 probes

static bool rowval_routed_ok(u64 n)
{
    if (const char *f = getenv("IVX_JOIN_PATH")) return !strcmp(f, "routed");      // tests: "direct" | "regions" | "routed"
    return n >= (1u << 21);
}

static ivx_status overlap_common(ivx_ctx *ctx, const ivx_index *ix, int mem, int mode,
                                 const u32 *key, const i32 *start, const i32 *end, u64 n,
                                 u32 *per_row, u8 *exists, u32 *bidx, u32 *pidx, u64 cap, u64 *total)
{
    IVX_TRY(check_probe_args(ctx, ix, IVX_KIND_OVERLAP, mem, start, end, n, true));
    CallMetrics cm(ctx, false, n);
    const u32 *dk = nullptr; const i32 *ds = nullptr, *de = nullptr;
    // large COUNT/FILL batches: partition the probe rows by index region and probe from LDS;
    // small batches and the per-row modes gather straight from the index
    bool regions = ix->jv_nreg > 0 && (mode == JP_COUNT || mode == JP_FILL) && n >= (1u << 21);   // measured crossover (tools/crossover.py)
    // the per-row modes (rle_right, semi / anti) of big batches: same partition, one value per row, un-permuted
    bool rowval = ix->jv_nreg > 0 && ix->jv_nreg <= IVX_MAXREG_WIDE && (mode == JP_PER_ROW || mode == JP_EXISTS) && n >= (1u << 21);
    if (const char *f = getenv("IVX_JOIN_PATH")) {
        if (!strcmp(f, "direct")) regions = rowval = false;
        else if (!strcmp(f, "regions")) {
            regions = ix->jv_nreg > 0 && (mode == JP_COUNT || mode == JP_FILL);
            rowval = ix->jv_nreg > 0 && ix->jv_nreg <= IVX_MAXREG_WIDE && (mode == JP_PER_ROW || mode == JP_EXISTS);
        }
    }
    // a fill call right after the count call that sized it: the routed probe rows are still in the context
    const void *in[3] = {key, start, end};
    ivx_join_plan &pl = ctx->join_plan;
    const bool planned = regions && mode == JP_FILL && pl.valid && !getenv("IVX_NO_PLAN") && pl.mem == mem && memcmp(pl.in, in, sizeof(in)) == 0 && pl.n == n &&
                         pl.ix == (const void *)ix && pl.ix_serial == ix->serial && pl.stream == ctx->stream;
    // (the region path orders itself behind the index's build tail after its routing pass; everything else here)
    if (ix->ready != nullptr && !(regions && !planned)) IVX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ix->ready, 0));
    if (!planned) {
        IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, key, n, &dk));
        IVX_TRY(stage_in(ctx, mem, WS_IN_START, start, n, &ds));
        IVX_TRY(stage_in(ctx, mem, WS_IN_END, end, n, &de));
    }
    u32 *d_row = nullptr, *d_b = nullptr, *d_p = nullptr; u8 *d_ex = nullptr;
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, per_row, n, &d_row));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_B, exists, n, &d_ex));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_C, bidx, cap, &d_b));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_D, pidx, cap, &d_p));
    IVX_HIP(ctx, hipMemsetAsync(ctx->d_scalars, 0, 2 * sizeof(u64), ctx->stream));   // pair cursor | probe fault flags
    {
        KernelTimer t(ctx);
        if (regions) {
            if (!planned) pl.valid = false;                 // whatever an earlier count call left is gone now
            const ivx_status st = ivx_join_probe_regions(ctx, ix->jv, ix->jv_nreg, mode, dk, ds, de, n, d_b, d_p, cap, ctx->d_scalars, planned, ix->jv_filter, ix->jv_pk24,
                                                         ix->jv_fast_unknown ? 2 : (ix->jv_fast ? 1 : 0), planned ? nullptr : ix->ready);
            if (st != IVX_OK) { pl.valid = false; return st; }
            if (mode == JP_COUNT && pl.valid) { memcpy(pl.in, in, sizeof(in)); pl.mem = mem; pl.n = n; pl.ix = ix; pl.ix_serial = ix->serial; pl.stream = ctx->stream; }
        }
        else if (rowval) IVX_TRY(ivx_rowval_probe_regions(ctx, ix->jv, ix->jv_nreg, mode == JP_PER_ROW ? IVX_RV_PER_ROW : IVX_RV_EXISTS, dk, ds, de, n, 0,
                                                          mode == JP_PER_ROW ? (void *)d_row : (void *)d_ex, ctx->d_scalars, ix->jv_filter, ix->jv_pk24, ix->jv_fast && !ix->jv_fast_unknown));
        else if (n && (mode == JP_PER_ROW || mode == JP_EXISTS) && ix->nroute_nreg > 0 && rowval_routed_ok(n))
            IVX_TRY(ivx_join_rowval_routed(ctx, ix, mode, dk, ds, de, n, d_row, d_ex, ctx->d_scalars));   // too many regions for the LDS slices: route, gather, put back
        else IVX_TRY(ivx_join_probe(ctx, ix->jv, mode, dk, ds, de, n, d_row, d_ex, d_b, d_p, cap, ctx->d_scalars));
    }
    u64 tot = 0;
    if (mode != JP_EXISTS) {
        IVX_HIP(ctx, hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 2 * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
        IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        tot = ctx->h_scalars[0];
        // a probe kernel met a routed-row page that the partition never published (it substituted a valid page, so nothing
        // was read out of bounds): an internal error, reported instead of returning wrong pairs
        if (ctx->h_scalars[1]) { ctx->join_plan.valid = false; return ctx->fail(IVX_ERR_HIP, "internal error: the probe met an unpublished page of routed rows"); }
    }
    if (mode == JP_COUNT && regions && pl.valid) pl.total = tot;
    if (total) *total = tot;
    if (mode == JP_FILL && tot > cap) return ctx->fail(IVX_ERR_CAPACITY, "pair buffers too small");   // (a plan survives this: the retry with bigger buffers reuses it)
    cm.out(mode == JP_FILL ? tot : (mode == JP_COUNT ? 0 : n));
    if (planned) pl.valid = false;                          // a plan serves ONE successful fill: a caller that refills the same buffers with its next batch must not get this batch's rows
    IVX_TRY(copy_out(ctx, mem, per_row, d_row, n));
    IVX_TRY(copy_out(ctx, mem, exists, d_ex, n));
    if (mode == JP_FILL) { IVX_TRY(copy_out(ctx, mem, bidx, d_b, tot)); IVX_TRY(copy_out(ctx, mem, pidx, d_p, tot)); }
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

extern "C" ivx_status ivx_probe_overlap_count(ivx_ctx *ctx, const ivx_index *ix, int mem, const uint32_t *key,
                                              const int32_t *start, const int32_t *end, uint64_t n,
                                              uint32_t *per_row, uint64_t *total)
{
    return overlap_common(ctx, ix, mem, per_row ? JP_PER_ROW : JP_COUNT, key, start, end, n, per_row, nullptr, nullptr, nullptr, 0, total);
}

// ---- host-resident fill of a big batch: chunks, so that the link runs in both directions at once
// The columns of 100 M probe rows cross the link in ~21 ms, the pairs come back in ~5 ms, the kernels take 1.3 ms: done one
// after the other that is what the call costs.  Cut into chunks, the pairs of chunk c go back to the host (a helper thread
// on a stream of its own: a copy from or to pageable memory occupies the thread that issues it) while the columns of chunk
// c + 1 come in.  Every chunk is a device-resident fill of its own that appends to the same pair buffers; probe row ids
// are shifted by the chunk's first row afterwards.  Order of the pairs is free, as everywhere.
__global__ void k_add_base(u32 *__restrict__ p, u64 n, u32 base)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] += base;
}

// device -> host copies on a thread and stream of their own, each after an event of the compute stream
struct HostCopier {
    struct Job { void *dst; const void *src; size_t bytes; hipEvent_t ev; };
    std::mutex mu; std::condition_variable cv, cv_done; std::deque<Job> jobs; bool closed = false; hipError_t herr = hipSuccess;
    u64 done = 0;                                           // copies finished so far (in the order they were pushed)
    hipStream_t cs = nullptr; std::thread th; std::vector<hipEvent_t> evs; bool started = false;
    hipError_t start(int device)
    {
        const hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        try {
        th = std::thread([this, device]() {
            (void)hipSetDevice(device);
            for (;;) {
                Job j;
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return closed || !jobs.empty(); }); if (jobs.empty()) return; j = jobs.front(); jobs.pop_front(); }
                hipError_t e2 = j.ev ? hipStreamWaitEvent(cs, j.ev, 0) : hipSuccess;
                if (e2 == hipSuccess && j.bytes) e2 = hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyDeviceToHost, cs);
                if (e2 == hipSuccess) e2 = hipStreamSynchronize(cs);
                { std::lock_guard<std::mutex> lk(mu); if (e2 != hipSuccess) herr = e2; done++; }
                cv_done.notify_all();
            }
        });
        } catch (...) { (void)hipStreamDestroy(cs); cs = nullptr; return hipErrorOutOfMemory; }     // (no thread to be had: the caller reports it)
        started = true;
        return hipSuccess;
    }
    // an event recorded on `st` now; the copies pushed with it run after everything enqueued on `st` so far
    hipError_t mark(hipStream_t st, hipEvent_t *ev)
    {
        hipError_t e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        evs.push_back(*ev);
        return hipEventRecord(*ev, st);
    }
    void push(void *dst, const void *src, size_t bytes, hipEvent_t ev)
    {
        { std::lock_guard<std::mutex> lk(mu); jobs.push_back(Job{dst, src, bytes, ev}); }
        cv.notify_all();
    }
    void wait_until(u64 k) { std::unique_lock<std::mutex> lk(mu); cv_done.wait(lk, [&] { return done >= k; }); }   // the first k copies are done
    hipError_t finish()                                     // waits for every copy
    {
        if (!started) return hipSuccess;
        { std::lock_guard<std::mutex> lk(mu); closed = true; }
        cv.notify_all();
        th.join();
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(cs);
        started = false;
        return herr;
    }
    ~HostCopier() { (void)finish(); }
};

static ivx_status overlap_fill_host_chunked(ivx_ctx *ctx, const ivx_index *ix, const u32 *key, const i32 *start, const i32 *end, u64 n,
                                            u32 *bidx, u32 *pidx, u64 cap, u64 *written, u32 nchunk)
{
    IVX_TRY(check_probe_args(ctx, ix, IVX_KIND_OVERLAP, IVX_MEM_HOST, start, end, n));
    hipStream_t st = ctx->stream;
    const u64 rows = ((n + nchunk - 1) / nchunk + 65535) & ~65535ull;           // rows per chunk
    u32 *d_b, *d_p;
    IVX_TRY(ctx->get_scratch(WS_OUT_C, (size_t)(cap ? cap : 1) * sizeof(u32), (void **)&d_b));
    IVX_TRY(ctx->get_scratch(WS_OUT_D, (size_t)(cap ? cap : 1) * sizeof(u32), (void **)&d_p));
    // two sets of staged columns: chunk c + 1 arrives while the kernels of chunk c may still read theirs
    u32 *dk[2] = {nullptr, nullptr}; i32 *ds[2], *de[2];
    for (int b = 0; b < 2; b++) {
        if (key) IVX_TRY(ctx->get_scratch(b ? WS_IN2_KEY : WS_IN_KEY, rows * sizeof(u32), (void **)&dk[b]));
        IVX_TRY(ctx->get_scratch(b ? WS_IN2_START : WS_IN_START, rows * sizeof(i32), (void **)&ds[b]));
        IVX_TRY(ctx->get_scratch(b ? WS_IN2_END : WS_IN_END, rows * sizeof(i32), (void **)&de[b]));
    }
    HostCopier hc;
    IVX_HIP(ctx, hc.start(ctx->device));
    const ivx_metrics m0 = ctx->metrics;                                        // (the chunks are one batch to the caller)
    u64 cum = 0, need = 0;
    ivx_status rc = IVX_OK;
    bool over = false;                                                          // the pair buffers ran out: only count from here on
    for (u32 c = 0; c < nchunk && rc == IVX_OK; c++) {
        const u64 c0 = (u64)c * rows, c1 = c0 + rows < n ? c0 + rows : n;
        if (c0 >= c1) break;
        const int b = (int)(c & 1u);
        const u64 nc = c1 - c0;
        hipError_t e = hipSuccess;
        if (key) e = hipMemcpyAsync(dk[b], key + c0, nc * sizeof(u32), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(ds[b], start + c0, nc * sizeof(i32), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(de[b], end + c0, nc * sizeof(i32), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { rc = ctx->fail_hip("hipMemcpyAsync(chunk)", e); break; }
        u64 tot = 0;
        if (!over) {
            ctx->fill_hint = (u64)((double)cap * (double)nc / (double)n) + 1;    // (the chunk's share of the pairs the caller expects: cap - cum would read as a dense join)
            rc = overlap_common(ctx, ix, IVX_MEM_DEVICE, JP_FILL, dk[b], ds[b], de[b], nc, nullptr, nullptr, d_b + cum, d_p + cum, cap - cum, &tot);
            ctx->fill_hint = 0;
            if (rc == IVX_ERR_CAPACITY) { over = true; rc = IVX_OK; need = cum + tot; continue; }
            if (rc != IVX_OK) break;
            if (tot && c0) hipLaunchKernelGGL(k_add_base, dim3(1024), dim3(256), 0, st, d_p + cum, tot, (u32)c0);
            hipEvent_t ev;
            e = hc.mark(st, &ev);
            if (e != hipSuccess) { rc = ctx->fail_hip("hipEventRecord(chunk)", e); break; }
            hc.push(bidx + cum, d_b + cum, tot * sizeof(u32), ev);
            hc.push(pidx + cum, d_p + cum, tot * sizeof(u32), ev);
            cum += tot;
        } else {
            rc = overlap_common(ctx, ix, IVX_MEM_DEVICE, JP_COUNT, dk[b], ds[b], de[b], nc, nullptr, nullptr, nullptr, nullptr, 0, &tot);
            need += tot;
        }
    }
    // (the staged columns of the last chunks must not be reused before their kernels are done: the stream is idle here)
    const hipError_t es = hipStreamSynchronize(st);
    const hipError_t herr = hc.finish();
    ctx->join_plan.valid = false;                                               // (the count calls above may have left one: it refers to a chunk)
    ctx->metrics.input_batches = m0.input_batches + 1;
    ctx->metrics.output_batches = m0.output_batches + ((rc == IVX_OK && !over && cum) ? 1 : 0);
    if (over) ctx->metrics.output_rows = m0.output_rows;
    if (rc != IVX_OK) return rc;
    if (es != hipSuccess) return ctx->fail_hip("hipStreamSynchronize", es);
    if (herr != hipSuccess) return ctx->fail_hip("hipMemcpyAsync(pairs)", herr);
    if (written) *written = over ? need : cum;
    if (over) return ctx->fail(IVX_ERR_CAPACITY, "pair buffers too small");
    return IVX_OK;
}

extern "C" ivx_status ivx_probe_overlap_fill(ivx_ctx *ctx, const ivx_index *ix, int mem, const uint32_t *key,
                                             const int32_t *start, const int32_t *end, uint64_t n,
                                             uint32_t *build_idx, uint32_t *probe_idx, uint64_t cap, uint64_t *written)
{
    if (ctx && cap && (!build_idx || !probe_idx)) return ctx->fail(IVX_ERR_INVALID, "null pair buffers");
    // host-resident columns of a big batch, and no count call left its routed rows (and device copies) behind: chunked
    // (IVX_HOST_CHUNKS = number of chunks, 1 switches it off; default 4 from 16 M rows)
    if (ctx && ix && mem == IVX_MEM_HOST && cap && start && end) {
        const char *ce = getenv("IVX_HOST_CHUNKS");
        const u32 nchunk = ce ? (u32)atoi(ce) : (n >= (16ull << 20) ? 4u : 1u);
        const void *in[3] = {key, start, end};
        const ivx_join_plan &pl = ctx->join_plan;
        const bool planned = pl.valid && pl.mem == mem && memcmp(pl.in, in, sizeof(in)) == 0 && pl.n == n && pl.ix == (const void *)ix && pl.ix_serial == ix->serial;
        if (nchunk > 1 && n >= (u64)nchunk * 65536 && !planned && ix->kind == IVX_KIND_OVERLAP)
            return overlap_fill_host_chunked(ctx, ix, key, start, end, n, build_idx, probe_idx, cap, written, nchunk);
    }
    return overlap_common(ctx, ix, mem, JP_FILL, key, start, end, n, nullptr, nullptr, build_idx, probe_idx, cap, written);
}

extern "C" ivx_status ivx_probe_exists(ivx_ctx *ctx, const ivx_index *ix, int mem, const uint32_t *key,
                                       const int32_t *start, const int32_t *end, uint64_t n, uint8_t *exists)
{
    if (ctx && n && !exists) return ctx->fail(IVX_ERR_INVALID, "null exists buffer");
    return overlap_common(ctx, ix, mem, JP_EXISTS, key, start, end, n, nullptr, exists, nullptr, nullptr, 0, nullptr);
}

// ---------------------------------------------------------------- a4 / a5 / a6 probes

ivx_status ivx_count_probe(ivx_ctx *ctx, const ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n, int strict, i64 *out);
ivx_status ivx_coverage_probe(ivx_ctx *ctx, const ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n, int strict, i64 *out);
ivx_status ivx_nearest_probe(ivx_ctx *ctx, const ivx_index *ix, const u32 *key, const i32 *s, const i32 *e, u64 n,
                             int strict, u32 k, int include_overlaps, u32 *ob, u32 *op, i64 *od, u64 cap, u64 *rows);
ivx_status ivx_merge_device(ivx_ctx *ctx, const u32 *key, const i64 *s, const i64 *e, u64 n, u32 nkeys,
                            i64 min_dist, int strict, u32 *ok, i64 *os, i64 *oe, i64 *on, u64 *m);
ivx_status ivx_subtract_device(ivx_ctx *ctx, const u32 *lkey, const i64 *ls, const i64 *le, u64 nl,
                               const u32 *rkey, const i64 *rs, const i64 *re, u64 nr, u32 nkeys, int strict,
                               u32 *ok, i64 *os, i64 *oe, u32 *orow, u64 cap, u64 *n_out);
ivx_status ivx_subtract_fill_planned(ivx_ctx *ctx, u32 *ok, i64 *os, i64 *oe, u32 *orow, u64 cap, u64 *n_out);
ivx_status ivx_cluster_device(ivx_ctx *ctx, const u32 *key, const i64 *s, const i64 *e, u64 n, u32 nkeys,
                              i64 min_dist, int strict, const i64 *key_base,
                              u32 *ok, i64 *os, i64 *oe, u32 *orow, i64 *oc, i64 *ocs, i64 *oce, u64 *key_clusters, u64 *m);
ivx_status ivx_complement_device(ivx_ctx *ctx, const u32 *key, const i64 *s, const i64 *e, u64 n,
                                 const u32 *vkey, const i64 *vs, const i64 *ve, u64 nv, u32 nkeys, int strict,
                                 u32 *ok, i64 *os, i64 *oe, u64 cap, u64 *n_out);
ivx_status ivx_take_fixed_device(ivx_ctx *ctx, const void *src, u32 width, u64 n_src, const u8 *src_valid,
                                 const u32 *idx, u64 n, void *out, u8 *out_valid);
ivx_status ivx_take_view_device(ivx_ctx *ctx, const void *views, const u8 *const *bufs, u64 n_src, const u8 *src_valid,
                                const u32 *idx, u64 n, void *out_views, u8 *out_data, u64 data_cap, u64 *data_bytes, u8 *out_valid);
ivx_status ivx_take_bits_device(ivx_ctx *ctx, const u8 *src_bits, u64 n_src, const u8 *src_valid, const u32 *idx, u64 n, u8 *out_bits, u8 *out_valid);
ivx_status ivx_take_utf8_device(ivx_ctx *ctx, int large, const void *offsets, const u8 *data, u64 n_src, const u8 *src_valid,
                                const u32 *idx, u64 n, void *out_offsets, u8 *out_data, u64 data_cap, u64 *data_bytes, u8 *out_valid);

static ivx_status per_row_i64(ivx_ctx *ctx, const ivx_index *ix, int kind, int mem, const u32 *key, const i32 *start,
                              const i32 *end, u64 n, int strict, i64 *out)
{
    IVX_TRY(check_probe_args(ctx, ix, kind, mem, start, end, n));
    if (n && !out) return ctx->fail(IVX_ERR_INVALID, "null output column");
    CallMetrics cm(ctx, false, n);
    cm.out(n);
    // host-resident columns of a big batch: in chunks, the values of chunk c go back (helper thread, own stream) while the
    // columns of chunk c + 1 come in -- see overlap_fill_host_chunked
    const char *ce = getenv("IVX_HOST_CHUNKS");
    const u32 nchunk = ce ? (u32)atoi(ce) : (n >= (16ull << 20) ? 4u : 1u);
    if (mem == IVX_MEM_HOST && nchunk > 1 && n >= (u64)nchunk * 65536) {
        hipStream_t st = ctx->stream;
        const u64 rows = ((n + nchunk - 1) / nchunk + 65535) & ~65535ull;
        u32 *dkk[2] = {nullptr, nullptr}; i32 *dss[2], *dee[2]; i64 *doo[2];
        for (int b = 0; b < 2; b++) {
            if (key) IVX_TRY(ctx->get_scratch(b ? WS_IN2_KEY : WS_IN_KEY, rows * sizeof(u32), (void **)&dkk[b]));
            IVX_TRY(ctx->get_scratch(b ? WS_IN2_START : WS_IN_START, rows * sizeof(i32), (void **)&dss[b]));
            IVX_TRY(ctx->get_scratch(b ? WS_IN2_END : WS_IN_END, rows * sizeof(i32), (void **)&dee[b]));
            IVX_TRY(ctx->get_scratch(b ? WS_OUT_B : WS_OUT_A, rows * sizeof(i64), (void **)&doo[b]));
        }
        HostCopier hc;
        IVX_HIP(ctx, hc.start(ctx->device));
        ivx_status rc = IVX_OK;
        for (u32 c = 0; c < nchunk && rc == IVX_OK; c++) {
            const u64 c0 = (u64)c * rows, c1 = c0 + rows < n ? c0 + rows : n;
            if (c0 >= c1) break;
            const int b = (int)(c & 1u);
            const u64 nc = c1 - c0;
            // (input set b was last read by the kernels of chunk c - 2, earlier on this stream; output set b by the COPY of
            //  chunk c - 2, on the helper's stream: that one is awaited before this chunk's kernels are enqueued)
            hipError_t e = hipSuccess;
            if (key) e = hipMemcpyAsync(dkk[b], key + c0, nc * sizeof(u32), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(dss[b], start + c0, nc * sizeof(i32), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(dee[b], end + c0, nc * sizeof(i32), hipMemcpyHostToDevice, st);
            if (e != hipSuccess) { rc = ctx->fail_hip("hipMemcpyAsync(chunk)", e); break; }
            if (c >= 2) hc.wait_until(c - 1);
            {
                KernelTimer t(ctx);
                rc = kind == IVX_KIND_COUNT ? ivx_count_probe(ctx, ix, dkk[b], dss[b], dee[b], nc, strict, doo[b])
                                            : ivx_coverage_probe(ctx, ix, dkk[b], dss[b], dee[b], nc, strict, doo[b]);
            }
            if (rc != IVX_OK) break;
            hipEvent_t ev;
            e = hc.mark(st, &ev);
            if (e != hipSuccess) { rc = ctx->fail_hip("hipEventRecord(chunk)", e); break; }
            hc.push(out + c0, doo[b], nc * sizeof(i64), ev);
        }
        const hipError_t es = hipStreamSynchronize(st);
        const hipError_t herr = hc.finish();
        if (rc != IVX_OK) return rc;
        if (es != hipSuccess) return ctx->fail_hip("hipStreamSynchronize", es);
        if (herr != hipSuccess) return ctx->fail_hip("hipMemcpyAsync(values)", herr);
        return IVX_OK;
    }
    const u32 *dk; const i32 *ds, *de; i64 *dout;
    IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, key, n, &dk));
    IVX_TRY(stage_in(ctx, mem, WS_IN_START, start, n, &ds));
    IVX_TRY(stage_in(ctx, mem, WS_IN_END, end, n, &de));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, out, n, &dout));
    {
        KernelTimer t(ctx);
        if (kind == IVX_KIND_COUNT) IVX_TRY(ivx_count_probe(ctx, ix, dk, ds, de, n, strict, dout));
        else IVX_TRY(ivx_coverage_probe(ctx, ix, dk, ds, de, n, strict, dout));
    }
    IVX_TRY(copy_out(ctx, mem, out, dout, n));
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

extern "C" ivx_status ivx_probe_count(ivx_ctx *ctx, const ivx_index *ix, int mem, const uint32_t *key, const int32_t *start,
                                      const int32_t *end, uint64_t n, int strict, int64_t *out)
{
    return per_row_i64(ctx, ix, IVX_KIND_COUNT, mem, key, start, end, n, strict, out);
}

extern "C" ivx_status ivx_probe_coverage(ivx_ctx *ctx, const ivx_index *ix, int mem, const uint32_t *key, const int32_t *start,
                                         const int32_t *end, uint64_t n, int strict, int64_t *out)
{
    return per_row_i64(ctx, ix, IVX_KIND_COVERAGE, mem, key, start, end, n, strict, out);
}

extern "C" ivx_status ivx_probe_nearest(ivx_ctx *ctx, const ivx_index *ix, int mem, const uint32_t *key, const int32_t *start,
                                        const int32_t *end, uint64_t n, int strict, uint32_t k, int include_overlaps,
                                        uint32_t *build_idx, uint32_t *probe_idx, int64_t *distance, uint64_t cap, uint64_t *rows)
{
    IVX_TRY(check_probe_args(ctx, ix, IVX_KIND_NEAREST, mem, start, end, n));
    if (!rows) return ctx->fail(IVX_ERR_INVALID, "null rows");
    if (n && (!build_idx || !probe_idx)) return ctx->fail(IVX_ERR_INVALID, "null output column");
    if ((u64)k * n > 0xFFFFFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "nearest: k * rows too large");
    CallMetrics cm(ctx, false, n);
    const u32 *dk; const i32 *ds, *de; u32 *db, *dp; i64 *dd;
    IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, key, n, &dk));
    IVX_TRY(stage_in(ctx, mem, WS_IN_START, start, n, &ds));
    IVX_TRY(stage_in(ctx, mem, WS_IN_END, end, n, &de));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, build_idx, cap, &db));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_B, probe_idx, cap, &dp));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_C, distance, cap, &dd));
    u64 r = 0;
    {
        KernelTimer t(ctx);
        ivx_status st = ivx_nearest_probe(ctx, ix, dk, ds, de, n, strict, k, include_overlaps, db, dp, dd, cap, &r);
        *rows = r;
        if (st != IVX_OK) return st;
    }
    cm.out(r);
    IVX_TRY(copy_out(ctx, mem, build_idx, db, r));
    IVX_TRY(copy_out(ctx, mem, probe_idx, dp, r));
    IVX_TRY(copy_out(ctx, mem, distance, dd, r));
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

// ---------------------------------------------------------------- a7..a9

extern "C" ivx_status ivx_merge(ivx_ctx *ctx, int mem, const uint32_t *key, const int64_t *start, const int64_t *end, uint64_t n,
                                uint32_t n_keys, int64_t min_dist, int strict,
                                uint32_t *out_key, int64_t *out_start, int64_t *out_end, int64_t *out_n,
                                uint64_t cap, uint64_t *n_out)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (!n_out) return ctx->fail(IVX_ERR_INVALID, "null n_out");
    *n_out = 0;
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if (n && (!start || !end)) return ctx->fail(IVX_ERR_INVALID, "null coordinate column");
    if (min_dist < 0) return ctx->fail(IVX_ERR_INVALID, "merge() min_dist must be >= 0, got " + std::to_string(min_dist));   // table_function.rs:237
    if (cap < n) return ctx->fail(IVX_ERR_CAPACITY, "merge: output buffers need capacity n");
    if (n >= 0xFFFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "merge: more than 2^32-1 rows in one call");
    if (!key || n_keys == 0) n_keys = 1;
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    const u32 *dk; const i64 *ds, *de; u32 *ok; i64 *os, *oe, *on;
    IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, key, n, &dk));
    IVX_TRY(stage_in(ctx, mem, WS_IN_START, start, n, &ds));
    IVX_TRY(stage_in(ctx, mem, WS_IN_END, end, n, &de));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, out_key, n, &ok));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_B, out_start, n, &os));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_C, out_end, n, &oe));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_D, out_n, n, &on));
    u64 m = 0;
    {
        KernelTimer t(ctx);
        IVX_TRY(ivx_merge_device(ctx, dk, ds, de, n, n_keys, min_dist, strict, ok, os, oe, on, &m));
    }
    *n_out = m;
    IVX_TRY(copy_out(ctx, mem, out_key, ok, m));
    IVX_TRY(copy_out(ctx, mem, out_start, os, m));
    IVX_TRY(copy_out(ctx, mem, out_end, oe, m));
    IVX_TRY(copy_out(ctx, mem, out_n, on, m));
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

extern "C" ivx_status ivx_subtract(ivx_ctx *ctx, int mem,
                                   const uint32_t *lkey, const int64_t *lstart, const int64_t *lend, uint64_t nl,
                                   const uint32_t *rkey, const int64_t *rstart, const int64_t *rend, uint64_t nr,
                                   uint32_t n_keys, int strict,
                                   uint32_t *out_key, int64_t *out_start, int64_t *out_end, uint32_t *out_row,
                                   uint64_t cap, uint64_t *n_out)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (!n_out) return ctx->fail(IVX_ERR_INVALID, "null n_out");
    *n_out = 0;
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if ((nl && (!lstart || !lend)) || (nr && (!rstart || !rend))) return ctx->fail(IVX_ERR_INVALID, "null coordinate column");
    if (nl >= 0xFFFFFFFFull || nr >= 0xFFFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "subtract: more than 2^32-1 rows in one call");
    if ((lkey == nullptr) != (rkey == nullptr) && nl && nr) return ctx->fail(IVX_ERR_INVALID, "subtract: key given for one side only");
    if (n_keys == 0 || (!lkey && !rkey)) n_keys = 1;
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    const u32 *dlk = nullptr, *drk = nullptr; const i64 *dls = nullptr, *dle = nullptr, *drs = nullptr, *dre = nullptr; u32 *ok, *orow; i64 *os, *oe;
    // a fill call right after the sizing call for the same arguments: the sorted sides, gap heads and output
    // offsets are still in the context (nothing else ran in between), only the output pass is left
    const void *in[6] = {lkey, lstart, lend, rkey, rstart, rend};
    ivx_sub_plan &pl = ctx->sub_plan;
    const bool sizing = cap == 0 && !out_key && !out_start && !out_end && !out_row;
    const bool planned = !sizing && nl && pl.valid && !getenv("IVX_NO_PLAN") && pl.mem == mem && memcmp(pl.in, in, sizeof(in)) == 0 && pl.nl == nl && pl.nr == nr &&
                         pl.nkeys == n_keys && pl.strict == strict && pl.stream == ctx->stream;
    if (!planned) {
        IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, lkey, nl, &dlk));
        IVX_TRY(stage_in(ctx, mem, WS_IN_START, lstart, nl, &dls));
        IVX_TRY(stage_in(ctx, mem, WS_IN_END, lend, nl, &dle));
        IVX_TRY(stage_in(ctx, mem, WS_IN2_KEY, rkey, nr, &drk));
        IVX_TRY(stage_in(ctx, mem, WS_IN2_START, rstart, nr, &drs));
        IVX_TRY(stage_in(ctx, mem, WS_IN2_END, rend, nr, &dre));
    }
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, out_key, cap, &ok));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_B, out_start, cap, &os));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_C, out_end, cap, &oe));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_D, out_row, cap, &orow));
    u64 m = 0;
    {
        KernelTimer t(ctx);
        if (!planned) pl.valid = false;                     // whatever an earlier sizing call left is gone now
        ivx_status st = planned ? ivx_subtract_fill_planned(ctx, ok, os, oe, orow, cap, &m)
                                : ivx_subtract_device(ctx, dlk, dls, dle, nl, drk, drs, dre, nr, n_keys, strict, ok, os, oe, orow, cap, &m);
        *n_out = m;
        if (st != IVX_OK) { if (st != IVX_ERR_CAPACITY) pl.valid = false; return st; }
        if (planned) pl.valid = false;                      // consumed: the next fill call sorts and counts again (its input columns may have been refilled in place)
        if (sizing && pl.valid) { memcpy(pl.in, in, sizeof(in)); pl.mem = mem; }
    }
    if (cap) {
        IVX_TRY(copy_out(ctx, mem, out_key, ok, m));
        IVX_TRY(copy_out(ctx, mem, out_start, os, m));
        IVX_TRY(copy_out(ctx, mem, out_end, oe, m));
        IVX_TRY(copy_out(ctx, mem, out_row, orow, m));
    }
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

extern "C" ivx_status ivx_cluster(ivx_ctx *ctx, int mem, const uint32_t *key, const int64_t *start, const int64_t *end, uint64_t n,
                                  uint32_t n_keys, int64_t min_dist, int strict, const int64_t *key_base,
                                  uint32_t *out_key, int64_t *out_start, int64_t *out_end, uint32_t *out_row,
                                  int64_t *out_cluster, int64_t *out_cluster_start, int64_t *out_cluster_end,
                                  uint64_t *key_clusters, uint64_t *n_clusters)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (!n_clusters) return ctx->fail(IVX_ERR_INVALID, "null n_clusters");
    *n_clusters = 0;
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if (n && (!start || !end)) return ctx->fail(IVX_ERR_INVALID, "null coordinate column");
    if (min_dist < 0) return ctx->fail(IVX_ERR_INVALID, "cluster() min_dist must be >= 0, got " + std::to_string(min_dist));   // table_function.rs:237
    if (n >= 0xFFFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "cluster: more than 2^32-1 rows in one call");
    if (!key || n_keys == 0) n_keys = 1;
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    const u32 *dk; const i64 *ds, *de, *dbase; u32 *ok, *orow; i64 *os, *oe, *oc, *ocs, *oce; u64 *okc;
    IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, key, n, &dk));
    IVX_TRY(stage_in(ctx, mem, WS_IN_START, start, n, &ds));
    IVX_TRY(stage_in(ctx, mem, WS_IN_END, end, n, &de));
    IVX_TRY(stage_in(ctx, mem, WS_IN2_KEY, key_base, (u64)n_keys, &dbase));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, out_key, n, &ok));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_B, out_start, n, &os));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_C, out_end, n, &oe));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_D, out_row, n, &orow));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_E, out_cluster, n, &oc));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_F, out_cluster_start, n, &ocs));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_G, out_cluster_end, n, &oce));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_H, key_clusters, (u64)n_keys, &okc));
    u64 m = 0;
    {
        KernelTimer t(ctx);
        IVX_TRY(ivx_cluster_device(ctx, dk, ds, de, n, n_keys, min_dist, strict, dbase, ok, os, oe, orow, oc, ocs, oce, okc, &m));
    }
    *n_clusters = m;
    IVX_TRY(copy_out(ctx, mem, out_key, ok, n));
    IVX_TRY(copy_out(ctx, mem, out_start, os, n));
    IVX_TRY(copy_out(ctx, mem, out_end, oe, n));
    IVX_TRY(copy_out(ctx, mem, out_row, orow, n));
    IVX_TRY(copy_out(ctx, mem, out_cluster, oc, n));
    IVX_TRY(copy_out(ctx, mem, out_cluster_start, ocs, n));
    IVX_TRY(copy_out(ctx, mem, out_cluster_end, oce, n));
    IVX_TRY(copy_out(ctx, mem, key_clusters, okc, (u64)n_keys));
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

extern "C" ivx_status ivx_complement(ivx_ctx *ctx, int mem, const uint32_t *key, const int64_t *start, const int64_t *end, uint64_t n,
                                     const uint32_t *vkey, const int64_t *vstart, const int64_t *vend, uint64_t nv,
                                     uint32_t n_keys, int strict, uint32_t *out_key, int64_t *out_start, int64_t *out_end,
                                     uint64_t cap, uint64_t *n_out)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (!n_out) return ctx->fail(IVX_ERR_INVALID, "null n_out");
    *n_out = 0;
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if ((n && (!start || !end)) || (nv && (!vstart || !vend))) return ctx->fail(IVX_ERR_INVALID, "null coordinate column");
    if (n >= 0xFFFFFFFFull || nv >= 0xFFFFFFFFull) return ctx->fail(IVX_ERR_INVALID, "complement: more than 2^32-1 rows in one call");
    if ((key == nullptr) != (vkey == nullptr) && n && nv) return ctx->fail(IVX_ERR_INVALID, "complement: key given for one side only");
    if (n_keys == 0 || (!key && !vkey)) n_keys = 1;
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    const u32 *dk, *dvk; const i64 *ds, *de, *dvs, *dve; u32 *ok; i64 *os, *oe;
    IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, key, n, &dk));
    IVX_TRY(stage_in(ctx, mem, WS_IN_START, start, n, &ds));
    IVX_TRY(stage_in(ctx, mem, WS_IN_END, end, n, &de));
    IVX_TRY(stage_in(ctx, mem, WS_IN2_KEY, vkey, nv, &dvk));
    IVX_TRY(stage_in(ctx, mem, WS_IN2_START, vstart, nv, &dvs));
    IVX_TRY(stage_in(ctx, mem, WS_IN2_END, vend, nv, &dve));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, out_key, cap, &ok));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_B, out_start, cap, &os));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_C, out_end, cap, &oe));
    u64 m = 0;
    {
        KernelTimer t(ctx);
        ivx_status st = ivx_complement_device(ctx, dk, ds, de, n, dvk, dvs, dve, nv, n_keys, strict, ok, os, oe, cap, &m);
        *n_out = m;
        if (st != IVX_OK) return st;
    }
    if (cap) {
        IVX_TRY(copy_out(ctx, mem, out_key, ok, m));
        IVX_TRY(copy_out(ctx, mem, out_start, os, m));
        IVX_TRY(copy_out(ctx, mem, out_end, oe, m));
    }
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

extern "C" ivx_status ivx_take_fixed(ivx_ctx *ctx, int mem, const void *src, uint32_t width, uint64_t n_src,
                                     const uint8_t *src_valid_bits, const uint32_t *idx, uint64_t n, void *out, uint8_t *out_valid)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if (n && (!idx || !out)) return ctx->fail(IVX_ERR_INVALID, "take: null idx or out");
    if (n_src && !src) return ctx->fail(IVX_ERR_INVALID, "take: null source column");
    if (width == 0 || width > 32 || (width & (width - 1))) return ctx->fail(IVX_ERR_UNSUPPORTED, "take: fixed width must be 1, 2, 4, 8, 16 or 32 bytes");
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    const u8 *dsrc, *dvalid; const u32 *didx; u8 *dout, *dov;
    IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, (const u8 *)src, n_src * width, &dsrc));
    IVX_TRY(stage_in(ctx, mem, WS_IN_START, src_valid_bits, (n_src + 7) / 8, &dvalid));
    IVX_TRY(stage_in(ctx, mem, WS_IN_END, idx, n, &didx));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, (u8 *)out, n * width, &dout));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_B, out_valid, n, &dov));
    {
        KernelTimer t(ctx);
        IVX_TRY(ivx_take_fixed_device(ctx, dsrc, width, n_src, dvalid, didx, n, dout, dov));
    }
    IVX_TRY(copy_out(ctx, mem, (u8 *)out, dout, n * width));
    IVX_TRY(copy_out(ctx, mem, out_valid, dov, n));
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

ivx_status ivx_scatter_fixed_device(ivx_ctx *ctx, const void *src, u32 width, const u32 *idx, u64 n, void *out, u64 n_out);

extern "C" ivx_status ivx_scatter_fixed(ivx_ctx *ctx, int mem, const void *src, uint32_t width, const uint32_t *idx, uint64_t n,
                                        void *out, uint64_t n_out)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if (n && (!idx || !src || !out)) return ctx->fail(IVX_ERR_INVALID, "scatter: null src, idx or out");
    if (width == 0 || width > 32 || (width & (width - 1))) return ctx->fail(IVX_ERR_UNSUPPORTED, "scatter: fixed width must be 1, 2, 4, 8, 16 or 32 bytes");
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    const u8 *dsrc; const u32 *didx; u8 *dout;
    IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, (const u8 *)src, n * width, &dsrc));
    IVX_TRY(stage_in(ctx, mem, WS_IN_END, idx, n, &didx));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, (u8 *)out, n_out * width, &dout));
    // host mode: rows that no index names keep what the caller's buffer holds
    if (mem == IVX_MEM_HOST && n_out) IVX_HIP(ctx, hipMemcpyAsync(dout, out, (size_t)n_out * width, hipMemcpyHostToDevice, ctx->stream));
    {
        KernelTimer t(ctx);
        IVX_TRY(ivx_scatter_fixed_device(ctx, dsrc, width, didx, n, dout, n_out));
    }
    IVX_TRY(copy_out(ctx, mem, (u8 *)out, dout, n_out * width));
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

extern "C" ivx_status ivx_take_utf8(ivx_ctx *ctx, int mem, int large, const void *offsets, const uint8_t *data, uint64_t n_src,
                                    uint64_t src_data_bytes, const uint8_t *src_valid_bits, const uint32_t *idx, uint64_t n,
                                    void *out_offsets, uint8_t *out_data, uint64_t data_cap, uint64_t *data_bytes, uint8_t *out_valid)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (!data_bytes) return ctx->fail(IVX_ERR_INVALID, "null data_bytes");
    *data_bytes = 0;
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if (n && !idx) return ctx->fail(IVX_ERR_INVALID, "take: null idx");
    if (!offsets || (src_data_bytes && !data)) return ctx->fail(IVX_ERR_INVALID, "take: null source column");
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ow = large ? 8 : 4;
    const u8 *doff, *ddata, *dvalid; const u32 *didx; u8 *dooff, *dodata, *dov;
    IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, (const u8 *)offsets, (n_src + 1) * ow, &doff));
    IVX_TRY(stage_in(ctx, mem, WS_IN_START, data, src_data_bytes, &ddata));
    IVX_TRY(stage_in(ctx, mem, WS_IN_END, src_valid_bits, (n_src + 7) / 8, &dvalid));
    IVX_TRY(stage_in(ctx, mem, WS_IN2_KEY, idx, n, &didx));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, (u8 *)out_offsets, (n + 1) * ow, &dooff));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_B, out_data, data_cap, &dodata));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_C, out_valid, n, &dov));
    u64 total = 0;
    {
        KernelTimer t(ctx);
        ivx_status st = ivx_take_utf8_device(ctx, large, doff, ddata, n_src, dvalid, didx, n, dooff, dodata, data_cap, &total, dov);
        *data_bytes = total;
        if (st != IVX_OK) return st;
    }
    IVX_TRY(copy_out(ctx, mem, (u8 *)out_offsets, dooff, (n + 1) * ow));
    if (out_data) IVX_TRY(copy_out(ctx, mem, out_data, dodata, total));
    IVX_TRY(copy_out(ctx, mem, out_valid, dov, n));
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

extern "C" ivx_status ivx_take_bits(ivx_ctx *ctx, int mem, const uint8_t *src_bits, uint64_t n_src, const uint8_t *src_valid_bits,
                                    const uint32_t *idx, uint64_t n, uint8_t *out_bits, uint8_t *out_valid)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if (n && (!idx || !out_bits)) return ctx->fail(IVX_ERR_INVALID, "take: null idx or out");
    if (n_src && !src_bits) return ctx->fail(IVX_ERR_INVALID, "take: null source column");
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    const u8 *dsrc, *dvalid; const u32 *didx; u8 *dout, *dov;
    IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, src_bits, (n_src + 7) / 8, &dsrc));
    IVX_TRY(stage_in(ctx, mem, WS_IN_START, src_valid_bits, (n_src + 7) / 8, &dvalid));
    IVX_TRY(stage_in(ctx, mem, WS_IN_END, idx, n, &didx));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, out_bits, (n + 7) / 8, &dout));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_B, out_valid, n, &dov));
    {
        KernelTimer t(ctx);
        IVX_TRY(ivx_take_bits_device(ctx, dsrc, n_src, dvalid, didx, n, dout, dov));
    }
    IVX_TRY(copy_out(ctx, mem, out_bits, dout, (n + 7) / 8));
    IVX_TRY(copy_out(ctx, mem, out_valid, dov, n));
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}

extern "C" ivx_status ivx_take_view(ivx_ctx *ctx, int mem, const void *views, const uint8_t *const *data_bufs, const uint64_t *data_buf_bytes,
                                    uint32_t n_bufs, uint64_t n_src, const uint8_t *src_valid_bits, const uint32_t *idx, uint64_t n,
                                    void *out_views, uint8_t *out_data, uint64_t data_cap, uint64_t *data_bytes, uint8_t *out_valid)
{
    if (!ctx) return IVX_ERR_INVALID;
    if (!data_bytes) return ctx->fail(IVX_ERR_INVALID, "null data_bytes");
    *data_bytes = 0;
    if (mem != IVX_MEM_HOST && mem != IVX_MEM_DEVICE) return ctx->fail(IVX_ERR_INVALID, "bad mem");
    if (n && !idx) return ctx->fail(IVX_ERR_INVALID, "take: null idx");
    if ((n_src && !views) || (n_bufs && (!data_bufs || !data_buf_bytes))) return ctx->fail(IVX_ERR_INVALID, "take: null source column");
    if (n_bufs > 4096) return ctx->fail(IVX_ERR_UNSUPPORTED, "take: more than 4096 view data buffers");
    IVX_HIP(ctx, hipSetDevice(ctx->device));
    const u8 *dviews, *dvalid; const u32 *didx; u8 *doviews, *dodata, *dov;
    IVX_TRY(stage_in(ctx, mem, WS_IN_KEY, (const u8 *)views, n_src * 16, &dviews));
    IVX_TRY(stage_in(ctx, mem, WS_IN_START, src_valid_bits, (n_src + 7) / 8, &dvalid));
    IVX_TRY(stage_in(ctx, mem, WS_IN_END, idx, n, &didx));
    // the variadic data buffers: device pointers in one small device table (host mode: one staging area for all)
    std::vector<const u8 *> ptrs(n_bufs ? n_bufs : 1, nullptr);
    if (mem == IVX_MEM_HOST) {
        u64 tot = 0;
        for (u32 b = 0; b < n_bufs; b++) tot += (data_buf_bytes[b] + 15) & ~15ull;
        u8 *area;
        IVX_TRY(ctx->get_scratch(WS_IN2_START, tot ? tot : 16, (void **)&area));
        u64 at = 0;
        for (u32 b = 0; b < n_bufs; b++) {
            if (data_buf_bytes[b]) IVX_HIP(ctx, hipMemcpyAsync(area + at, data_bufs[b], data_buf_bytes[b], hipMemcpyHostToDevice, ctx->stream));
            ptrs[b] = area + at;
            at += (data_buf_bytes[b] + 15) & ~15ull;
        }
    } else {
        for (u32 b = 0; b < n_bufs; b++) ptrs[b] = data_bufs[b];
    }
    const u8 **dtable;
    IVX_TRY(ctx->get_scratch(WS_IN2_KEY, ptrs.size() * sizeof(u8 *), (void **)&dtable));
    IVX_HIP(ctx, hipMemcpyAsync(dtable, ptrs.data(), ptrs.size() * sizeof(u8 *), hipMemcpyHostToDevice, ctx->stream));
    IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));                            // `ptrs` is pageable host memory
    IVX_TRY(stage_out(ctx, mem, WS_OUT_A, (u8 *)out_views, n * 16, &doviews));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_B, out_data, data_cap, &dodata));
    IVX_TRY(stage_out(ctx, mem, WS_OUT_C, out_valid, n, &dov));
    u64 total = 0;
    {
        KernelTimer t(ctx);
        ivx_status st = ivx_take_view_device(ctx, dviews, dtable, n_src, dvalid, didx, n, doviews, dodata, data_cap, &total, dov);
        *data_bytes = total;
        if (st != IVX_OK) return st;
    }
    if (out_views && (out_data || total == 0)) IVX_TRY(copy_out(ctx, mem, (u8 *)out_views, doviews, n * 16));
    if (out_data) IVX_TRY(copy_out(ctx, mem, out_data, dodata, total));
    IVX_TRY(copy_out(ctx, mem, out_valid, dov, n));
    if (mem == IVX_MEM_HOST) IVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IVX_OK;
}
