/* Plain C99 client of include/ivx.h: the 12 x 10 reads/targets join of the reference's
 * tests (R/tests/integration_test.rs:61-84 -> 16 pairs with the contig as equi-key), count_overlaps
 * on the ranges fixture (:641-657) and a merge -- linked against libivx_hip.so only.
 * Built and run by tests/test_gpu_c_client.py. */
#include <stdio.h>
#include <stdlib.h>
#include "ivx.h"

#define CHECK(call) do { ivx_status st_ = (call); if (st_ != IVX_OK) { \
    fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, ivx_last_error(ctx)); return 1; } } while (0)

int main(void)
{
    ivx_ctx *ctx = NULL;
    if (ivx_ctx_create(0, &ctx) != IVX_OK) { fprintf(stderr, "no gfx950 device\n"); return 2; }

    /* testing/data/interval: targets (build) and reads (probe); key 0 = chr1, 1 = chr2 */
    const uint32_t bk[10] = {0,0,0,0,0, 1,1,1,1,1};
    const int32_t  bs[10] = {100,200,400,10000,22100, 100,200,400,10000,22100};
    const int32_t  be[10] = {190,290,600,20000,22100, 190,290,600,20000,22100};
    const uint32_t pk[12] = {0,0,0,0,0,0, 1,1,1,1,1,1};
    const int32_t  ps[12] = {150,190,300,500,15000,22000, 150,190,300,500,15000,22000};
    const int32_t  pe[12] = {250,300,501,700,15000,22300, 250,300,500,700,15000,22300};

    ivx_index *ix = NULL;
    CHECK(ivx_index_build(ctx, IVX_KIND_OVERLAP, IVX_MEM_HOST, bk, bs, be, 10, 2, &ix));
    uint64_t total = 0;
    uint32_t per_row[12];
    CHECK(ivx_probe_overlap_count(ctx, ix, IVX_MEM_HOST, pk, ps, pe, 12, per_row, &total));
    uint32_t *bi = malloc(sizeof(uint32_t) * (total + 1)), *pi = malloc(sizeof(uint32_t) * (total + 1));
    uint64_t written = 0;
    CHECK(ivx_probe_overlap_fill(ctx, ix, IVX_MEM_HOST, pk, ps, pe, 12, bi, pi, total, &written));
    uint64_t rle = 0;
    for (int i = 0; i < 12; i++) rle += per_row[i];
    int ok = 1;
    for (uint64_t j = 0; j < written; j++)          /* every emitted pair really overlaps */
        ok &= bk[bi[j]] == pk[pi[j]] && bs[bi[j]] <= pe[pi[j]] && be[bi[j]] >= ps[pi[j]];
    printf("join pairs=%llu written=%llu rle_sum=%llu valid=%d\n", (unsigned long long)total, (unsigned long long)written,
           (unsigned long long)rle, ok);
    ivx_index_free(ix);

    /* merge: (a,100,200) (a,150,250) (a,300,400) -> (100,250,2) (300,400,1)   R/tests/integration_test.rs:2123-2149 */
    const int64_t ms[3] = {100, 150, 300}, me[3] = {200, 250, 400};
    uint32_t ok_[3]; int64_t os[3], oe[3], on[3]; uint64_t m = 0;
    CHECK(ivx_merge(ctx, IVX_MEM_HOST, NULL, ms, me, 3, 1, 0, 0, ok_, os, oe, on, 3, &m));
    printf("merge rows=%llu first=(%lld,%lld,%lld) second=(%lld,%lld,%lld)\n", (unsigned long long)m,
           (long long)os[0], (long long)oe[0], (long long)on[0], (long long)os[1], (long long)oe[1], (long long)on[1]);

    /* round-3 entry points: a sharded per-row column back in input order (ivx_scatter_fixed), what the context has reserved,
     * the memory limit, scratch back to the device (ivx_ctx_trim), the overlapped build switch */
    const int64_t vals[3] = {30, 10, 20};
    const uint32_t rows[3] = {3, 1, 2};
    int64_t col[5] = {-1, -1, -1, -1, -1};
    CHECK(ivx_scatter_fixed(ctx, IVX_MEM_HOST, vals, 8, rows, 3, col, 5));
    const int scat_ok = col[0] == -1 && col[1] == 10 && col[2] == 20 && col[3] == 30 && col[4] == -1;
    const uint32_t bad_rows[1] = {5};
    const int scat_bad = ivx_scatter_fixed(ctx, IVX_MEM_HOST, vals, 8, bad_rows, 1, col, 5) == IVX_ERR_INVALID;
    const uint64_t held = ivx_ctx_reserved_bytes(ctx);
    CHECK(ivx_ctx_set_memory_limit(ctx, 1));                      /* one byte: the next build cannot reserve its index */
    ivx_index *ix2 = NULL;
    const int oom = ivx_index_build(ctx, IVX_KIND_OVERLAP, IVX_MEM_HOST, bk, bs, be, 10, 2, &ix2) == IVX_ERR_OOM && ix2 == NULL;
    CHECK(ivx_ctx_set_memory_limit(ctx, 0));
    CHECK(ivx_ctx_trim(ctx, 0));
    const uint64_t after = ivx_ctx_reserved_bytes(ctx);
    CHECK(ivx_ctx_set_build_overlap(ctx, 1));
    CHECK(ivx_index_build(ctx, IVX_KIND_OVERLAP, IVX_MEM_HOST, bk, bs, be, 10, 2, &ix2));
    uint64_t total2 = 0;
    CHECK(ivx_probe_overlap_count(ctx, ix2, IVX_MEM_HOST, pk, ps, pe, 12, NULL, &total2));
    CHECK(ivx_ctx_synchronize(ctx));
    ivx_index_free(ix2);
    printf("scatter ok=%d bad_index_refused=%d reserved=%llu after_trim=%llu oom=%d total2=%llu\n", scat_ok, scat_bad,
           (unsigned long long)held, (unsigned long long)after, oom, (unsigned long long)total2);

    free(bi); free(pi);
    ivx_ctx_free(ctx);
    if (!(scat_ok && scat_bad && held > 0 && after == 0 && oom && total2 == 16)) return 1;
    return (total == 16 && written == 16 && rle == 16 && ok && m == 2 && os[0] == 100 && oe[0] == 250 && on[0] == 2 && os[1] == 300) ? 0 : 1;
}
