/*
 * ivx_oracle.c -- CPU restatement of the reference's interval algorithms.
 * TEST INFRASTRUCTURE ONLY (see ivx_oracle.h).  Plain C, gcc -O2 -fopenmp.
 *
 * R/ = /root/reference/datafusion/bio-function-ranges/
 */
#include "ivx_oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ utils */

static inline int32_t wrap_add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wrap_sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t min32(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t max32(int32_t a, int32_t b) { return a > b ? a : b; }

static inline int64_t sat_add64(int64_t a, int64_t b)
{
    int64_t r;
    if (__builtin_add_overflow(a, b, &r)) return b > 0 ? INT64_MAX : INT64_MIN;
    return r;
}

static uint32_t max_key(const uint32_t *k, uint64_t n)
{
    uint32_t m = 0;
    for (uint64_t i = 0; i < n; i++) if (k[i] > m) m = k[i];
    return m;
}

/* group rows by key with a counting sort: off[nkeys+1], rows[n] (stable, so
 * rows of one key stay in input order = the reference's `position` order,
 * interval_join.rs:922-928) */
typedef struct { uint32_t nkeys; uint64_t *off; uint64_t *rows; } groups_t;

static groups_t group_by_key(const uint32_t *key, uint64_t n)
{
    groups_t g;
    g.nkeys = n ? max_key(key, n) + 1 : 0;
    g.off = (uint64_t *)calloc((size_t)g.nkeys + 1, sizeof(uint64_t));
    g.rows = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) g.off[key[i] + 1]++;
    for (uint32_t k = 0; k < g.nkeys; k++) g.off[k + 1] += g.off[k];
    uint64_t *cur = (uint64_t *)malloc(((size_t)g.nkeys + 1) * sizeof(uint64_t));
    memcpy(cur, g.off, ((size_t)g.nkeys + 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) g.rows[cur[key[i]]++] = i;
    free(cur);
    return g;
}
static void groups_free(groups_t *g) { free(g->off); free(g->rows); }

/* -------------------------------------------------- a3: brute-force join */

uint64_t orc_join_brute(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                        const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                        uint32_t *out_build, uint32_t *out_probe, uint64_t cap)
{
    uint64_t n = 0;
    for (uint64_t i = 0; i < np; i++)
        for (uint64_t j = 0; j < nb; j++)
            if (bkey[j] == pkey[i] && bs[j] <= pe[i] && be[j] >= ps[i]) {
                if (n < cap) {
                    if (out_build) out_build[n] = (uint32_t)j;
                    if (out_probe) out_probe[n] = (uint32_t)i;
                }
                n++;
            }
    return n;
}

/* ------------------------------- sorted + max-end augmented implicit tree */

typedef struct { int32_t s, e; uint32_t row; } iv32_t;

static int cmp_iv32_start(const void *a, const void *b)
{
    const iv32_t *x = (const iv32_t *)a, *y = (const iv32_t *)b;
    if (x->s != y->s) return x->s < y->s ? -1 : 1;
    if (x->e != y->e) return x->e < y->e ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row ? 1 : 0);
}

/* maxe[mid of [lo,hi)] = max end over [lo,hi): a balanced BST laid over the
 * start-sorted array (COITree keeps the same augmentation, `subtree_last`,
 * in a van Emde Boas layout) */
static int32_t tree_augment(const iv32_t *a, int32_t *maxe, int64_t lo, int64_t hi)
{
    if (lo >= hi) return INT32_MIN;
    int64_t mid = lo + (hi - lo) / 2;
    int32_t m = a[mid].e;
    int32_t l = tree_augment(a, maxe, lo, mid);
    int32_t r = tree_augment(a, maxe, mid + 1, hi);
    if (l > m) m = l;
    if (r > m) m = r;
    maxe[mid] = m;
    return m;
}

typedef void (*visit_fn)(const iv32_t *iv, void *ctx);

static void tree_query(const iv32_t *a, const int32_t *maxe, int64_t lo, int64_t hi,
                       int32_t qs, int32_t qe, visit_fn f, void *ctx)
{
    while (lo < hi) {
        if (hi - lo <= 16) { /* small subtree: scan the sorted run (coitrees does the same) */
            for (int64_t i = lo; i < hi && a[i].s <= qe; i++)
                if (a[i].e >= qs) f(&a[i], ctx);
            return;
        }
        int64_t mid = lo + (hi - lo) / 2;
        if (maxe[mid] < qs) return;               /* nothing below reaches the query */
        tree_query(a, maxe, lo, mid, qs, qe, f, ctx);
        if (a[mid].s > qe) return;                /* mid and everything right start too late */
        if (a[mid].e >= qs) f(&a[mid], ctx);
        lo = mid + 1;
    }
}

typedef struct {
    groups_t g;
    iv32_t *iv;     /* per key segment sorted by (start,end,row) */
    int32_t *maxe;
} tree_index_t;

static tree_index_t tree_index_build(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb)
{
    tree_index_t t;
    t.g = group_by_key(bkey, nb);
    t.iv = (iv32_t *)malloc((nb ? nb : 1) * sizeof(iv32_t));
    t.maxe = (int32_t *)malloc((nb ? nb : 1) * sizeof(int32_t));
    for (uint64_t i = 0; i < nb; i++) {
        uint64_t r = t.g.rows[i];
        t.iv[i].s = bs[r]; t.iv[i].e = be[r]; t.iv[i].row = (uint32_t)r;
    }
    for (uint32_t k = 0; k < t.g.nkeys; k++) {
        uint64_t lo = t.g.off[k], hi = t.g.off[k + 1];
        qsort(t.iv + lo, hi - lo, sizeof(iv32_t), cmp_iv32_start);
        tree_augment(t.iv, t.maxe, (int64_t)lo, (int64_t)hi);
    }
    return t;
}
static void tree_index_free(tree_index_t *t) { groups_free(&t->g); free(t->iv); free(t->maxe); }

typedef struct { uint64_t n, cap, base; uint32_t *ob, *op; uint32_t probe; } emit_ctx_t;

static void emit_pair(const iv32_t *iv, void *c)
{
    emit_ctx_t *e = (emit_ctx_t *)c;
    uint64_t at = e->base + e->n;
    if (at < e->cap) {
        if (e->ob) e->ob[at] = iv->row;
        if (e->op) e->op[at] = e->probe;
    }
    e->n++;
}
static void count_only(const iv32_t *iv, void *c) { (void)iv; (*(uint64_t *)c)++; }

uint64_t orc_join_tree(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                       const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                       uint32_t *out_build, uint32_t *out_probe, uint64_t cap,
                       uint64_t *per_row, int threads)
{
    tree_index_t t = tree_index_build(bkey, bs, be, nb);
    if (threads < 1) threads = 1;
    /* pass 1: per-row counts (also what the reference's rle_right holds) */
    uint64_t *cnt = per_row ? per_row : (uint64_t *)malloc((np ? np : 1) * sizeof(uint64_t));
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)np; i++) {
        uint64_t c = 0;
        uint32_t k = pkey[i];
        if (k < t.g.nkeys)
            tree_query(t.iv, t.maxe, (int64_t)t.g.off[k], (int64_t)t.g.off[k + 1], ps[i], pe[i], count_only, &c);
        cnt[i] = c;
    }
    uint64_t total = 0;
    if (out_build || out_probe) {
        /* exclusive offsets, then pass 2 writes each row's matches in place */
        uint64_t *off = (uint64_t *)malloc((np + 1) * sizeof(uint64_t));
        off[0] = 0;
        for (uint64_t i = 0; i < np; i++) off[i + 1] = off[i] + cnt[i];
        total = off[np];
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int64_t i = 0; i < (int64_t)np; i++) {
            uint32_t k = pkey[i];
            if (k >= t.g.nkeys || cnt[i] == 0) continue;
            emit_ctx_t e = { 0, cap, off[i], out_build, out_probe, (uint32_t)i };
            tree_query(t.iv, t.maxe, (int64_t)t.g.off[k], (int64_t)t.g.off[k + 1], ps[i], pe[i], emit_pair, &e);
        }
        free(off);
    } else {
        for (uint64_t i = 0; i < np; i++) total += cnt[i];
    }
    if (!per_row) free(cnt);
    tree_index_free(&t);
    return total;
}

/* The same join in ONE walk per probe row, as the reference's probe loop does it
 * (interval_join.rs:1614-1653: every partition appends its matches to growable index builders and
 * emits them batch by batch): thread t walks the t-th contiguous share of the probe rows and appends
 * to its own buffers; the shares are concatenated in probe order afterwards.  This is the form timed as
 * bench.py's cpu_baseline (orc_join_tree above walks the tree twice: count, then emit). */
typedef struct { uint32_t *b, *p; uint64_t n, cap; uint32_t probe; } grow_t;
typedef struct { int threads; grow_t *part; uint64_t total; } join1_t;

static void emit_grow(const iv32_t *iv, void *c)
{
    grow_t *g = (grow_t *)c;
    if (g->n == g->cap) {
        g->cap = g->cap ? g->cap * 2 : 4096;
        g->b = (uint32_t *)realloc(g->b, g->cap * sizeof(uint32_t));
        g->p = (uint32_t *)realloc(g->p, g->cap * sizeof(uint32_t));
    }
    g->b[g->n] = iv->row; g->p[g->n] = g->probe; g->n++;
}

void *orc_join_single_run(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                          const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np, int threads)
{
    tree_index_t t = tree_index_build(bkey, bs, be, nb);
    if (threads < 1) threads = 1;
    join1_t *j = (join1_t *)calloc(1, sizeof(join1_t));
    j->threads = threads;
    j->part = (grow_t *)calloc((size_t)threads, sizeof(grow_t));
#pragma omp parallel for num_threads(threads) schedule(static, 1)
    for (int w = 0; w < threads; w++) {
        grow_t *g = &j->part[w];
        const uint64_t lo = np * (uint64_t)w / (uint64_t)threads, hi = np * (uint64_t)(w + 1) / (uint64_t)threads;
        for (uint64_t i = lo; i < hi; i++) {
            uint32_t k = pkey[i];
            if (k >= t.g.nkeys) continue;
            g->probe = (uint32_t)i;
            tree_query(t.iv, t.maxe, (int64_t)t.g.off[k], (int64_t)t.g.off[k + 1], ps[i], pe[i], emit_grow, g);
        }
    }
    for (int w = 0; w < threads; w++) j->total += j->part[w].n;
    tree_index_free(&t);
    return j;
}
uint64_t orc_join_single_total(const void *h) { return ((const join1_t *)h)->total; }
void orc_join_single_copy(const void *h, uint32_t *out_build, uint32_t *out_probe)
{
    const join1_t *j = (const join1_t *)h;
    uint64_t at = 0;
    for (int w = 0; w < j->threads; w++) {
        if (j->part[w].n) {
            memcpy(out_build + at, j->part[w].b, j->part[w].n * sizeof(uint32_t));
            memcpy(out_probe + at, j->part[w].p, j->part[w].n * sizeof(uint32_t));
        }
        at += j->part[w].n;
    }
}
void orc_join_single_free(void *h)
{
    join1_t *j = (join1_t *)h;
    for (int w = 0; w < j->threads; w++) { free(j->part[w].b); free(j->part[w].p); }
    free(j->part); free(j);
}

void orc_join_exists(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                     const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                     uint8_t *exists)
{
    tree_index_t t = tree_index_build(bkey, bs, be, nb);
    for (uint64_t i = 0; i < np; i++) {
        uint64_t c = 0;
        uint32_t k = pkey[i];
        if (k < t.g.nkeys)
            tree_query(t.iv, t.maxe, (int64_t)t.g.off[k], (int64_t)t.g.off[k + 1], ps[i], pe[i], count_only, &c);
        exists[i] = c != 0;
    }
    tree_index_free(&t);
}

/* ------------------------------------------------ a4: count_overlaps */

static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
/* slice::partition_point(|v| v <= x) */
static uint64_t pp_le(const int32_t *a, uint64_t n, int32_t x)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) { uint64_t m = lo + (hi - lo) / 2; if (a[m] <= x) lo = m + 1; else hi = m; }
    return lo;
}
/* slice::partition_point(|v| v < x) */
static uint64_t pp_lt(const int32_t *a, uint64_t n, int32_t x)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) { uint64_t m = lo + (hi - lo) / 2; if (a[m] < x) lo = m + 1; else hi = m; }
    return lo;
}

void orc_count_overlaps_mt(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                           const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                           int strict, int64_t *out, int threads)
{
    if (threads < 1) threads = 1;
    groups_t g = group_by_key(bkey, nb);
    int32_t *S = (int32_t *)malloc((nb ? nb : 1) * sizeof(int32_t));
    int32_t *E = (int32_t *)malloc((nb ? nb : 1) * sizeof(int32_t));
    for (uint64_t i = 0; i < nb; i++) { S[i] = bs[g.rows[i]]; E[i] = be[g.rows[i]]; }
    for (uint32_t k = 0; k < g.nkeys; k++) {           /* interval_tree.rs:35-36 */
        qsort(S + g.off[k], g.off[k + 1] - g.off[k], sizeof(int32_t), cmp_i32);
        qsort(E + g.off[k], g.off[k + 1] - g.off[k], sizeof(int32_t), cmp_i32);
    }
    /* probe rows are independent (one DataFusion partition per thread over a shared index) */
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)np; i++) {
        int32_t qs = ps[i], qe = pe[i];
        if (strict) { qs = wrap_add32(qs, 1); qe = wrap_sub32(qe, 1); }   /* :253-256 */
        uint32_t k = pkey[i];
        if (k >= g.nkeys || g.off[k] == g.off[k + 1] || qe < qs) { out[i] = 0; continue; }
        uint64_t lo = g.off[k], n = g.off[k + 1] - lo;
        uint64_t started = pp_le(S + lo, n, qe);                          /* :46 */
        uint64_t ended_before = pp_lt(E + lo, n, qs);                     /* :47 */
        out[i] = (int64_t)(started - ended_before);                       /* :48 */
    }
    free(S); free(E); groups_free(&g);
}

void orc_count_overlaps(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                        const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                        int strict, int64_t *out)
{
    orc_count_overlaps_mt(bkey, bs, be, nb, pkey, ps, pe, np, strict, out, 1);
}

/* ------------------------------------------------------ a5: coverage */

typedef struct { int32_t s, e; uint64_t ord; } ivo_t;
static int cmp_ivo_first_stable(const void *a, const void *b)
{
    const ivo_t *x = (const ivo_t *)a, *y = (const ivo_t *)b;
    if (x->s != y->s) return x->s < y->s ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord ? 1 : 0);   /* sort_by is stable (:57) */
}

uint64_t orc_merge_intervals_i32(int32_t *s, int32_t *e, uint64_t n)
{
    if (n == 0) return 0;
    ivo_t *v = (ivo_t *)malloc(n * sizeof(ivo_t));
    for (uint64_t i = 0; i < n; i++) { v[i].s = s[i]; v[i].e = e[i]; v[i].ord = i; }
    qsort(v, n, sizeof(ivo_t), cmp_ivo_first_stable);
    uint64_t m = 0;
    int32_t cs = v[0].s, ce = v[0].e;
    for (uint64_t i = 1; i < n; i++) {
        if (v[i].s <= ce) { if (v[i].e > ce) ce = v[i].e; }        /* :63-64 */
        else { s[m] = cs; e[m] = ce; m++; cs = v[i].s; ce = v[i].e; }
    }
    s[m] = cs; e[m] = ce; m++;
    free(v);
    return m;
}

typedef struct { int32_t qs, qe; int32_t cov; } cov_ctx_t;
static void cov_visit(const iv32_t *iv, void *c)
{
    cov_ctx_t *q = (cov_ctx_t *)c;
    /* interval_tree.rs:148: max(1, min(end + 1, node.last) - max(start - 1, node.first)), i32 */
    int32_t hi = min32(wrap_add32(q->qe, 1), iv->e);
    int32_t lo = max32(wrap_sub32(q->qs, 1), iv->s);
    int32_t ov = max32(1, wrap_sub32(hi, lo));
    q->cov = wrap_add32(q->cov, ov);
}

void orc_coverage_mt(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                     const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                     int strict, int64_t *out, int threads)
{
    if (threads < 1) threads = 1;
    groups_t g = group_by_key(bkey, nb);
    /* merged nodes per key, then the same augmented tree over them */
    iv32_t *iv = (iv32_t *)malloc((nb ? nb : 1) * sizeof(iv32_t));
    int32_t *maxe = (int32_t *)malloc((nb ? nb : 1) * sizeof(int32_t));
    uint64_t *moff = (uint64_t *)calloc((size_t)g.nkeys + 1, sizeof(uint64_t));
    uint64_t w = 0;
    for (uint32_t k = 0; k < g.nkeys; k++) {
        uint64_t lo = g.off[k], n = g.off[k + 1] - lo;
        moff[k] = w;
        if (n == 0) continue;
        int32_t *s = (int32_t *)malloc(n * sizeof(int32_t));
        int32_t *e = (int32_t *)malloc(n * sizeof(int32_t));
        for (uint64_t i = 0; i < n; i++) { s[i] = bs[g.rows[lo + i]]; e[i] = be[g.rows[lo + i]]; }
        uint64_t m = orc_merge_intervals_i32(s, e, n);
        for (uint64_t i = 0; i < m; i++) { iv[w + i].s = s[i]; iv[w + i].e = e[i]; iv[w + i].row = (uint32_t)i; }
        tree_augment(iv, maxe, (int64_t)w, (int64_t)(w + m));
        w += m;
        free(s); free(e);
    }
    moff[g.nkeys] = w;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)np; i++) {
        int32_t qs = ps[i], qe = pe[i];
        if (strict) { qs = wrap_add32(qs, 1); qe = wrap_sub32(qe, 1); }   /* :185-188 */
        uint32_t k = pkey[i];
        cov_ctx_t c = { qs, qe, 0 };
        if (k < g.nkeys && moff[k] != moff[k + 1])
            tree_query(iv, maxe, (int64_t)moff[k], (int64_t)moff[k + 1], qs, qe, cov_visit, &c);
        out[i] = (int64_t)c.cov;                                          /* :207 */
    }
    free(iv); free(maxe); free(moff); groups_free(&g);
}

void orc_coverage(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                  const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                  int strict, int64_t *out)
{
    orc_coverage_mt(bkey, bs, be, nb, pkey, ps, pe, np, strict, out, 1);
}

/* ------------------------------------------------------- a6: nearest */

static int cmp_iv32_end(const void *a, const void *b)
{   /* nearest_index.rs:77-82: (last, first, metadata) */
    const iv32_t *x = (const iv32_t *)a, *y = (const iv32_t *)b;
    if (x->e != y->e) return x->e < y->e ? -1 : 1;
    if (x->s != y->s) return x->s < y->s ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row ? 1 : 0);
}

static inline int64_t cand_dist(int32_t qs, int32_t qe, int32_t s, int32_t e)
{   /* nearest_index.rs:252-260 */
    if (qe < s) return (int64_t)s - (int64_t)qe;
    if (e < qs) return (int64_t)qs - (int64_t)e;
    return 0;
}
static inline int cmp_meta(const iv32_t *a, const iv32_t *b) { return cmp_iv32_start(a, b); } /* :245-250 */
static inline int cmp_cand(int32_t qs, int32_t qe, const iv32_t *a, const iv32_t *b)
{   /* :262-266 */
    int64_t ad = cand_dist(qs, qe, a->s, a->e), bd = cand_dist(qs, qe, b->s, b->e);
    if (ad != bd) return ad < bd ? -1 : 1;
    return cmp_meta(a, b);
}

typedef struct {
    const iv32_t *by_start; const iv32_t *by_end; const int32_t *pmax; const int32_t *maxe; uint64_t n;
} nidx_t;

static uint64_t pp_first_le(const iv32_t *a, uint64_t n, int32_t x)
{ uint64_t lo = 0, hi = n; while (lo < hi) { uint64_t m = lo + (hi - lo) / 2; if (a[m].s <= x) lo = m + 1; else hi = m; } return lo; }
static uint64_t pp_last_lt(const iv32_t *a, uint64_t n, int32_t x)
{ uint64_t lo = 0, hi = n; while (lo < hi) { uint64_t m = lo + (hi - lo) / 2; if (a[m].e < x) lo = m + 1; else hi = m; } return lo; }

static int first_overlap_by_start(const nidx_t *x, int32_t qs, int32_t qe, iv32_t *out)
{   /* :222-234 */
    if (qe < qs) return 0;
    uint64_t plen = pp_first_le(x->by_start, x->n, qe);
    if (plen == 0 || x->pmax[plen - 1] < qs) return 0;
    uint64_t idx = pp_lt(x->pmax, plen, qs);
    *out = x->by_start[idx];
    return 1;
}
static int nearest_non_overlap_one(const nidx_t *x, int32_t qs, int32_t qe, iv32_t *out)
{   /* :192-220 */
    uint64_t li = pp_last_lt(x->by_end, x->n, qs);
    uint64_t ri = pp_first_le(x->by_start, x->n, qe);
    int hl = li > 0, hr = ri < x->n;
    if (!hl && !hr) return 0;
    if (hl && !hr) { *out = x->by_end[li - 1]; return 1; }
    if (!hl && hr) { *out = x->by_start[ri]; return 1; }
    const iv32_t *l = &x->by_end[li - 1], *r = &x->by_start[ri];
    *out = cmp_cand(qs, qe, l, r) <= 0 ? *l : *r;
    return 1;
}
static int nearest_one(const nidx_t *x, int32_t qs, int32_t qe, int ovl, iv32_t *out)
{   /* :91-101 */
    if (x->n == 0) return 0;
    if (ovl && first_overlap_by_start(x, qs, qe, out)) return 1;
    return nearest_non_overlap_one(x, qs, qe, out);
}

typedef struct { iv32_t *v; uint64_t n, cap; } ivvec_t;
static void push_iv(const iv32_t *iv, void *c)
{
    ivvec_t *v = (ivvec_t *)c;
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->v = (iv32_t *)realloc(v->v, v->cap * sizeof(iv32_t)); }
    v->v[v->n++] = *iv;
}
static int seen_has(const uint32_t *seen, uint64_t n, uint32_t row)
{ for (uint64_t i = 0; i < n; i++) if (seen[i] == row) return 1; return 0; }

/* :103-190; out receives up to k build rows, returns how many */
static uint64_t nearest_k(const nidx_t *x, int32_t qs, int32_t qe, uint64_t k, int ovl, iv32_t *out, ivvec_t *scratch)
{
    if (k == 0 || x->n == 0) return 0;
    if (k == 1) return (uint64_t)nearest_one(x, qs, qe, ovl, out);
    uint64_t n = 0;
    uint32_t *seen = (uint32_t *)malloc(k * sizeof(uint32_t));
#define TAKE(iv) do { if (!seen_has(seen, n, (iv).row)) { seen[n] = (iv).row; out[n] = (iv); n++; } } while (0)
    if (ovl) {
        scratch->n = 0;
        tree_query(x->by_start, x->maxe, 0, (int64_t)x->n, qs, qe, push_iv, scratch);
        qsort(scratch->v, scratch->n, sizeof(iv32_t), cmp_iv32_start);
        for (uint64_t i = 0; i < scratch->n; i++) {
            if (n == k) { free(seen); return n; }
            TAKE(scratch->v[i]);
        }
    }
    if (n == k) { free(seen); return n; }
    uint64_t li = pp_last_lt(x->by_end, x->n, qs);
    uint64_t ri = pp_first_le(x->by_start, x->n, qe);
    while (n < k) {
        int hl = li > 0, hr = ri < x->n;
        iv32_t next;
        if (!hl && !hr) break;
        if (hl && !hr) { next = x->by_end[--li]; }
        else if (!hl && hr) { next = x->by_start[ri++]; }
        else {
            const iv32_t *l = &x->by_end[li - 1], *r = &x->by_start[ri];
            if (cmp_cand(qs, qe, l, r) <= 0) { next = *l; li--; } else { next = *r; ri++; }
        }
        if (!ovl && cand_dist(qs, qe, next.s, next.e) == 0) continue;
        TAKE(next);
    }
#undef TAKE
    free(seen);
    return n;
}

uint64_t orc_nearest(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                     const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                     int strict, uint32_t k, int include_overlaps,
                     uint32_t *out_build, uint32_t *out_probe, int64_t *out_dist, uint64_t cap)
{
    tree_index_t t = tree_index_build(bkey, bs, be, nb);   /* by_start order = (start,end,row), :50-55 */
    iv32_t *by_end = (iv32_t *)malloc((nb ? nb : 1) * sizeof(iv32_t));
    int32_t *pmax = (int32_t *)malloc((nb ? nb : 1) * sizeof(int32_t));
    memcpy(by_end, t.iv, nb * sizeof(iv32_t));
    for (uint32_t kk = 0; kk < t.g.nkeys; kk++) {
        uint64_t lo = t.g.off[kk], hi = t.g.off[kk + 1];
        qsort(by_end + lo, hi - lo, sizeof(iv32_t), cmp_iv32_end);
        int32_t m = INT32_MIN;                              /* :58-63 */
        for (uint64_t i = lo; i < hi; i++) { if (t.iv[i].e > m) m = t.iv[i].e; pmax[i] = m; }
    }
    iv32_t *buf = (iv32_t *)malloc((k ? k : 1) * sizeof(iv32_t));
    ivvec_t scratch = { 0, 0, 0 };
    uint64_t w = 0;
    for (uint64_t i = 0; i < np; i++) {
        int32_t qs = ps[i], qe = pe[i];
        if (strict) { qs = wrap_add32(qs, 1); qe = wrap_sub32(qe, 1); }   /* nearest.rs:341-344 */
        uint32_t kk = pkey[i];
        uint64_t found = 0;
        if (kk < t.g.nkeys && t.g.off[kk] != t.g.off[kk + 1]) {
            uint64_t lo = t.g.off[kk];
            nidx_t x = { t.iv + lo, by_end + lo, pmax + lo, t.maxe + lo, t.g.off[kk + 1] - lo };
            /* the tree over a sub-range must be queried with the same [lo,hi)
             * it was augmented with; maxe was built on absolute indices */
            if (k == 1) found = (uint64_t)nearest_one(&x, qs, qe, include_overlaps, buf);
            else {
                /* re-base: nearest_k's tree_query uses (by_start,maxe) from 0..n,
                 * which matches the absolute-index augmentation only when the
                 * recursion splits identically; it does, because tree_augment
                 * and tree_query both split [lo,hi) at lo+(hi-lo)/2 */
                found = nearest_k(&x, qs, qe, k, include_overlaps, buf, &scratch);
            }
        }
        if (found == 0) {                                   /* nearest.rs:377-384, :424-430 */
            if (w < cap) { out_build[w] = ORC_NULL_IDX; out_probe[w] = (uint32_t)i; if (out_dist) out_dist[w] = -1; }
            w++;
        } else {
            for (uint64_t j = 0; j < found; j++) {
                if (w < cap) {
                    out_build[w] = buf[j].row; out_probe[w] = (uint32_t)i;
                    if (out_dist) out_dist[w] = cand_dist(ps[i], pe[i], buf[j].s, buf[j].e);  /* raw coords :367-374 */
                }
                w++;
            }
        }
    }
    free(buf); free(scratch.v); free(by_end); free(pmax);
    tree_index_free(&t);
    return w;
}

/* k = 1 only, for full-size checks: one output row per probe row (out index = probe row), the per-key sorts and the
 * probe loop spread over `threads` host threads.  Same index, same nearest_one as above. */
uint64_t orc_nearest1_mt(const uint32_t *bkey, const int32_t *bs, const int32_t *be, uint64_t nb,
                         const uint32_t *pkey, const int32_t *ps, const int32_t *pe, uint64_t np,
                         int strict, int include_overlaps, uint32_t *out_build, int64_t *out_dist, int threads)
{
    if (threads < 1) threads = 1;
    tree_index_t t;
    t.g = group_by_key(bkey, nb);
    t.iv = (iv32_t *)malloc((nb ? nb : 1) * sizeof(iv32_t));
    t.maxe = (int32_t *)malloc((nb ? nb : 1) * sizeof(int32_t));
    iv32_t *by_end = (iv32_t *)malloc((nb ? nb : 1) * sizeof(iv32_t));
    int32_t *pmax = (int32_t *)malloc((nb ? nb : 1) * sizeof(int32_t));
#pragma omp parallel for num_threads(threads) schedule(static)
    for (uint64_t i = 0; i < nb; i++) {
        uint64_t r = t.g.rows[i];
        t.iv[i].s = bs[r]; t.iv[i].e = be[r]; t.iv[i].row = (uint32_t)r;
    }
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (uint32_t kk = 0; kk < t.g.nkeys; kk++) {
        uint64_t lo = t.g.off[kk], hi = t.g.off[kk + 1];
        qsort(t.iv + lo, hi - lo, sizeof(iv32_t), cmp_iv32_start);       /* by_start order = (start,end,row), :50-55 */
        memcpy(by_end + lo, t.iv + lo, (hi - lo) * sizeof(iv32_t));
        qsort(by_end + lo, hi - lo, sizeof(iv32_t), cmp_iv32_end);        /* :77-82 */
        int32_t m = INT32_MIN;                                            /* :58-63 */
        for (uint64_t i = lo; i < hi; i++) { if (t.iv[i].e > m) m = t.iv[i].e; pmax[i] = m; }
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (uint64_t i = 0; i < np; i++) {
        int32_t qs = ps[i], qe = pe[i];
        if (strict) { qs = wrap_add32(qs, 1); qe = wrap_sub32(qe, 1); }   /* nearest.rs:341-344 */
        uint32_t kk = pkey[i];
        iv32_t best;
        int found = 0;
        if (kk < t.g.nkeys && t.g.off[kk] != t.g.off[kk + 1]) {
            uint64_t lo = t.g.off[kk];
            nidx_t x = { t.iv + lo, by_end + lo, pmax + lo, t.maxe + lo, t.g.off[kk + 1] - lo };
            found = nearest_one(&x, qs, qe, include_overlaps, &best);
        }
        out_build[i] = found ? best.row : ORC_NULL_IDX;
        if (out_dist) out_dist[i] = found ? cand_dist(ps[i], pe[i], best.s, best.e) : -1;
    }
    free(by_end); free(pmax);
    tree_index_free(&t);
    return np;
}

/* ------------------------------------------------- a7+a8: merge sweep */

typedef struct { int64_t s, e; uint64_t row; } iv64_t;
static int cmp_iv64(const void *a, const void *b)
{   /* sort_unstable on (i64,i64[,usize]) tuples: lexicographic, grouped_stream.rs:93-95, :213-215 */
    const iv64_t *x = (const iv64_t *)a, *y = (const iv64_t *)b;
    if (x->s != y->s) return x->s < y->s ? -1 : 1;
    if (x->e != y->e) return x->e < y->e ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row ? 1 : 0);
}

static iv64_t *sorted_groups64(const uint32_t *key, const int64_t *s, const int64_t *e, uint64_t n, groups_t *g)
{
    *g = group_by_key(key, n);
    iv64_t *v = (iv64_t *)malloc((n ? n : 1) * sizeof(iv64_t));
    for (uint64_t i = 0; i < n; i++) { uint64_t r = g->rows[i]; v[i].s = s[r]; v[i].e = e[r]; v[i].row = r; }
    for (uint32_t k = 0; k < g->nkeys; k++)
        qsort(v + g->off[k], g->off[k + 1] - g->off[k], sizeof(iv64_t), cmp_iv64);
    return v;
}

uint64_t orc_merge(const uint32_t *key, const int64_t *s, const int64_t *e, uint64_t n,
                   int64_t min_dist, int strict,
                   uint32_t *out_key, int64_t *out_s, int64_t *out_e, int64_t *out_n, uint64_t cap)
{
    groups_t g;
    iv64_t *v = sorted_groups64(key, s, e, n, &g);
    uint64_t w = 0;
#define EMIT(k_, s_, e_, n_) do { if (w < cap) { if (out_key) out_key[w] = (k_); if (out_s) out_s[w] = (s_); \
        if (out_e) out_e[w] = (e_); if (out_n) out_n[w] = (n_); } w++; } while (0)
    for (uint32_t k = 0; k < g.nkeys; k++) {
        int has = 0; int64_t cs = 0, ce = 0, cn = 0;
        for (uint64_t i = g.off[k]; i < g.off[k + 1]; i++) {
            int64_t is = v[i].s, ie = v[i].e;
            if (has) {
                int64_t boundary = sat_add64(ce, min_dist);               /* merge.rs:291 */
                int mergec = strict ? (is < boundary) : (is <= boundary); /* :292-296 */
                if (mergec) { if (ie > ce) ce = ie; cn++; }
                else { EMIT(k, cs, ce, cn); cs = is; ce = ie; cn = 1; }
            } else { cs = is; ce = ie; cn = 1; has = 1; }
        }
        if (has) EMIT(k, cs, ce, cn);                                     /* :325-333 */
    }
#undef EMIT
    free(v); groups_free(&g);
    return w;
}

/* ---------------------------------------------- a7+a9: subtract sweep */

uint64_t orc_subtract(const uint32_t *lkey, const int64_t *ls, const int64_t *le, uint64_t nl,
                      const uint32_t *rkey, const int64_t *rs, const int64_t *re, uint64_t nr,
                      int strict,
                      uint32_t *out_key, int64_t *out_s, int64_t *out_e, uint32_t *out_row,
                      uint64_t cap)
{
    groups_t gl, gr;
    iv64_t *L = sorted_groups64(lkey, ls, le, nl, &gl);
    iv64_t *R = sorted_groups64(rkey, rs, re, nr, &gr);
    uint64_t w = 0;
#define EMIT(k_, s_, e_, r_) do { if (w < cap) { if (out_key) out_key[w] = (k_); if (out_s) out_s[w] = (s_); \
        if (out_e) out_e[w] = (e_); if (out_row) out_row[w] = (uint32_t)(r_); } w++; } while (0)
    for (uint32_t k = 0; k < gl.nkeys; k++) {
        const iv64_t *rv = 0; uint64_t rn = 0;
        if (k < gr.nkeys) { rv = R + gr.off[k]; rn = gr.off[k + 1] - gr.off[k]; }
        uint64_t rc = 0;                                                   /* right_cursor */
        for (uint64_t i = gl.off[k]; i < gl.off[k + 1]; i++) {
            int64_t s0 = L[i].s, e0 = L[i].e;
            while (rc < rn) {                                              /* subtract.rs:401-412 */
                int skip = strict ? (rv[rc].e <= s0) : (rv[rc].e < s0);
                if (skip) rc++; else break;
            }
            int64_t cursor = s0;
            for (uint64_t j = rc; j < rn; j++) {                           /* :416-433 */
                int64_t a = rv[j].s, b = rv[j].e;
                int no = strict ? (a >= e0) : (a > e0);
                if (no) break;
                if (a > cursor) EMIT(k, cursor, a, L[i].row);
                if (b > cursor) cursor = b;
            }
            if (cursor < e0) EMIT(k, cursor, e0, L[i].row);                /* :435-440 */
        }
    }
#undef EMIT
    free(L); free(R); groups_free(&gl); groups_free(&gr);
    return w;
}

/* ---------------------------------------------------- f1: cluster */

uint64_t orc_cluster(const uint32_t *key, const int64_t *s, const int64_t *e, uint64_t n, uint32_t nkeys,
                     int64_t min_dist, int strict, const int64_t *key_base,
                     uint32_t *out_key, int64_t *out_s, int64_t *out_e, uint32_t *out_row,
                     int64_t *out_cluster, int64_t *out_cs, int64_t *out_ce, uint64_t *key_clusters)
{
    groups_t g;
    iv64_t *v = sorted_groups64(key, s, e, n, &g);        /* FullBatchCollector order (start,end,row) */
    if (key_clusters) for (uint32_t k = 0; k < nkeys; k++) key_clusters[k] = 0;
    int64_t next_id = 0;                                   /* cluster.rs:398: contigs in name order */
    uint64_t total = 0;
    for (uint32_t k = 0; k < g.nkeys; k++) {
        uint64_t lo = g.off[k], hi = g.off[k + 1];
        if (lo == hi) continue;
        int64_t id = key_base ? key_base[k] : next_id;    /* set_group_cluster_base, :571-580 */
        uint64_t nclu = 0;
        uint64_t i = lo;
        while (i < hi) {                                   /* one pending cluster, :638-661 */
            int64_t cs = v[i].s, ce = v[i].e;
            uint64_t j = i + 1;
            for (; j < hi; j++) {
                int64_t boundary = sat_add64(ce, min_dist);
                int mergec = strict ? (v[j].s < boundary) : (v[j].s <= boundary);
                if (!mergec) break;
                if (v[j].e > ce) ce = v[j].e;
            }
            for (uint64_t r = i; r < j; r++) {             /* flush_pending_cluster, :554-569 */
                if (out_key) out_key[r] = k;
                if (out_s) out_s[r] = v[r].s;
                if (out_e) out_e[r] = v[r].e;
                if (out_row) out_row[r] = (uint32_t)v[r].row;
                if (out_cluster) out_cluster[r] = id;
                if (out_cs) out_cs[r] = cs;
                if (out_ce) out_ce[r] = ce;
            }
            id++; nclu++;
            i = j;
        }
        if (key_clusters && k < nkeys) key_clusters[k] = nclu;
        next_id += (int64_t)nclu;
        total += nclu;
    }
    free(v); groups_free(&g);
    return total;
}

/* ---------------------------------------------------- f2: complement */

uint64_t orc_complement(const uint32_t *key, const int64_t *s, const int64_t *e, uint64_t n,
                        const uint32_t *vkey, const int64_t *vs, const int64_t *ve, uint64_t nv,
                        int strict, uint32_t *out_key, int64_t *out_s, int64_t *out_e, uint64_t cap)
{
    groups_t g, gv;
    iv64_t *v = sorted_groups64(key, s, e, n, &g);
    iv64_t *w = sorted_groups64(vkey, vs, ve, nv, &gv);   /* view_bounds: per contig, sorted (start,end) */
    uint64_t o = 0;
#define EMIT(k_, s_, e_) do { if (o < cap) { if (out_key) out_key[o] = (k_); if (out_s) out_s[o] = (s_); \
        if (out_e) out_e[o] = (e_); } o++; } while (0)
    iv64_t *m = (iv64_t *)malloc((n ? n : 1) * sizeof(iv64_t));
    for (uint32_t k = 0; k < g.nkeys; k++) {
        uint64_t lo = g.off[k], hi = g.off[k + 1];
        if (lo == hi) continue;                            /* only contigs with input rows, :397 */
        uint64_t nm = 0;                                   /* merge_intervals, :297-317 */
        int64_t cs = v[lo].s, ce = v[lo].e;
        for (uint64_t i = lo + 1; i < hi; i++) {
            int mergec = strict ? (v[i].s < ce) : (v[i].s <= ce);
            if (mergec) { if (v[i].e > ce) ce = v[i].e; }
            else { m[nm].s = cs; m[nm].e = ce; nm++; cs = v[i].s; ce = v[i].e; }
        }
        m[nm].s = cs; m[nm].e = ce; nm++;
        iv64_t implicit = { 0, INT64_MAX, 0 };             /* :401-403 */
        const iv64_t *views = &implicit; uint64_t nviews = 1;
        if (k < gv.nkeys && gv.off[k + 1] > gv.off[k]) { views = w + gv.off[k]; nviews = gv.off[k + 1] - gv.off[k]; }
        for (uint64_t q = 0; q < nviews; q++) {            /* emit_contig_complement, :320-356 */
            int64_t view_start = views[q].s, view_end = views[q].e;
            int64_t cursor = view_start;
            for (uint64_t j = 0; j < nm; j++) {
                if (m[j].e <= view_start) continue;
                if (m[j].s >= view_end) break;
                int64_t is = m[j].s > view_start ? m[j].s : view_start;
                int64_t ie = m[j].e < view_end ? m[j].e : view_end;
                if (is > cursor) EMIT(k, cursor, is);
                cursor = ie;
            }
            if (cursor < view_end) EMIT(k, cursor, view_end);
        }
    }
    for (uint32_t k = 0; k < gv.nkeys; k++) {              /* EmitTrailingGaps, :431-456 */
        if (gv.off[k + 1] == gv.off[k]) continue;
        if (k < g.nkeys && g.off[k + 1] > g.off[k]) continue;
        for (uint64_t q = gv.off[k]; q < gv.off[k + 1]; q++) EMIT(k, w[q].s, w[q].e);
    }
#undef EMIT
    free(m); free(v); free(w); groups_free(&g); groups_free(&gv);
    return o;
}

int64_t orc_check_i32(const int64_t *v, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++)
        if (v[i] > INT32_MAX || v[i] < INT32_MIN) return (int64_t)i;
    return -1;
}
