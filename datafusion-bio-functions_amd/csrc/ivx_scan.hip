// ivx_scan.hip -- sum scans used by the counting sorts (instances of ivx_scan.hpp).
#include "ivx_scan.hpp"

namespace {
template <typename V>
struct SumOp {
    using T = V;
    __host__ __device__ static T identity() { return 0; }
    __device__ static T combine(const T &a, const T &b) { return a + b; }
    __device__ static T shfl_up(const T &v, int d) { return __shfl_up(v, d, IVX_WAVE); }
};
}  // namespace

ivx_status ivx_scan_exclusive_u32(ivx_ctx *ctx, u32 *data, u64 n) { return ivxscan::exclusive<SumOp<u32>>(ctx, data, n); }
ivx_status ivx_scan_exclusive_u64(ivx_ctx *ctx, u64 *data, u64 n) { return ivxscan::exclusive<SumOp<unsigned long long>>(ctx, (unsigned long long *)data, n); }
