// bio_ranges_host.cpp -- host-side operator mirror over the Arrow C Data Interface
// (see include/bio_ranges_host.h).  Plain C++17; the only dependency is libivx_hip.so.
#include "../../include/bio_ranges_host.h"
#include "../../include/ivx.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

struct brh_session {
    ivx_ctx *ctx = nullptr;
    std::string err;
    // NULL contigs: 0 = as the reference treats them (see get_contig), 1 = refuse the batch (opt-in; BIO_STRICT_NULL_CONTIGS=1
    // makes it the default of new sessions)
    int strict_null_contigs = [] { const char *e = std::getenv("BIO_STRICT_NULL_CONTIGS"); return e && *e && *e != '0' ? 1 : 0; }();
};

namespace {

int fail(brh_session *s, const std::string &m) { if (s) s->err = m; return 1; }

int fail_ivx(brh_session *s, ivx_status st)
{
    return fail(s, std::string(ivx_last_error(s->ctx)) + " (ivx status " + std::to_string(st) + ")");
}

std::string column_list(const ArrowSchema *sch)
{
    std::string o = "[";
    for (int64_t i = 0; i < sch->n_children; i++) {
        if (i) o += ", ";
        o += "\"" + std::string(sch->children[i]->name ? sch->children[i]->name : "") + "\"";
    }
    return o + "]";
}

int find_col(const ArrowSchema *sch, const char *name)
{
    for (int64_t i = 0; i < sch->n_children; i++)
        if (sch->children[i]->name && !std::strcmp(sch->children[i]->name, name)) return (int)i;
    return -1;
}

int64_t null_count(const ArrowArray *a)
{
    if (a->null_count >= 0) return a->null_count;
    if (a->n_buffers < 1 || !a->buffers[0]) return 0;
    const uint8_t *v = (const uint8_t *)a->buffers[0];
    int64_t n = 0;
    for (int64_t i = 0; i < a->length; i++) { int64_t j = i + a->offset; n += !((v[j >> 3] >> (j & 7)) & 1); }
    return n;
}

// ---- ContigArray (array_utils.rs:10-24): Utf8 / LargeUtf8 / Utf8View
struct StrCol {
    int kind = 0;               // 0 Utf8, 1 LargeUtf8, 2 Utf8View
    const ArrowArray *a = nullptr;
    bool has_nulls = false;
    bool null_at(int64_t i) const
    {
        if (!has_nulls) return false;
        const uint8_t *v = (const uint8_t *)a->buffers[0];
        const int64_t j = i + a->offset;
        return !((v[j >> 3] >> (j & 7)) & 1);
    }
    std::string_view at(int64_t i) const
    {
        const int64_t j = i + a->offset;
        if (kind == 0) { const int32_t *o = (const int32_t *)a->buffers[1]; return {(const char *)a->buffers[2] + o[j], (size_t)(o[j + 1] - o[j])}; }
        if (kind == 1) { const int64_t *o = (const int64_t *)a->buffers[1]; return {(const char *)a->buffers[2] + o[j], (size_t)(o[j + 1] - o[j])}; }
        const uint8_t *v = (const uint8_t *)a->buffers[1] + 16 * j;
        int32_t len; std::memcpy(&len, v, 4);
        if (len <= 12) return {(const char *)v + 4, (size_t)len};
        int32_t bi, off; std::memcpy(&bi, v + 8, 4); std::memcpy(&off, v + 12, 4);
        return {(const char *)a->buffers[2 + bi] + off, (size_t)len};
    }
};

int get_contig(brh_session *s, brh_batch t, const char *name, StrCol *out)
{
    const int c = find_col(t.schema, name);
    if (c < 0) return fail(s, "contig column '" + std::string(name) + "' not found in batch with columns: " + column_list(t.schema));
    const char *f = t.schema->children[c]->format;
    out->a = t.array->children[c];
    if (!std::strcmp(f, "u")) out->kind = 0;
    else if (!std::strcmp(f, "U")) out->kind = 1;
    else if (!std::strcmp(f, "vu")) out->kind = 2;
    else return fail(s, "unsupported data type " + std::string(f) + " for contig column '" + name + "'; expected Utf8, LargeUtf8, or Utf8View");
    // NULL contigs.  The reference never looks at the validity bitmap of a contig column: the table functions read
    // `value(i)` (whatever bytes the slot's offsets span: "" for arrays made by the Arrow builders), and the join hashes the
    // key columns with create_hashes, where a NULL is a key of its own that equals only other NULLs (the reference compares
    // hashes, never values: interval_join.rs:857, :922-928).  That is what happens here by default (the callers of
    // build_keys say which of the two); none of the reference's tests pins either, so a session can refuse such batches
    // instead (brh_session_set_strict_null_contigs / BIO_STRICT_NULL_CONTIGS=1).
    out->has_nulls = null_count(out->a) > 0 && out->a->n_buffers >= 1 && out->a->buffers[0];
    if (out->has_nulls && s && s->strict_null_contigs) {
        int64_t row = 0;
        for (int64_t i = 0; i < out->a->length; i++) if (out->null_at(i)) { row = i; break; }
        return fail(s, "contig column '" + std::string(name) + "' contains a NULL at row " + std::to_string(row) + "; NULL contigs are not supported");
    }
    return 0;
}

// ---- PosArray (array_utils.rs:26-172)
struct PosCol { char fmt = 0; const ArrowArray *a = nullptr; };

int get_pos(brh_session *s, brh_batch t, const char *name, const char *label, PosCol *out)
{
    const int c = find_col(t.schema, name);
    if (c < 0) return fail(s, std::string(label) + " column '" + name + "' not found in batch with columns: " + column_list(t.schema));
    const char *f = t.schema->children[c]->format;
    if (std::strlen(f) != 1 || !std::strchr("ilIL", f[0]))
        return fail(s, "unsupported data type " + std::string(f) + " for " + label + " column '" + name + "'; expected Int32, Int64, UInt32, or UInt64");
    out->fmt = f[0]; out->a = t.array->children[c];
    return 0;
}

// PosArray::resolve (array_utils.rs:66-135)
int resolve_i32(brh_session *s, const PosCol &p, std::vector<int32_t> *out)
{
    if (null_count(p.a) > 0) return fail(s, "coordinate column contains null values; nearest requires non-null coordinates");
    const int64_t n = p.a->length, o = p.a->offset;
    out->resize((size_t)n);
    char buf[160];
    for (int64_t i = 0; i < n; i++) {
        if (p.fmt == 'i') { (*out)[i] = ((const int32_t *)p.a->buffers[1])[o + i]; continue; }
        bool ok; long long sv = 0; unsigned long long uv = 0;
        if (p.fmt == 'l') { sv = ((const int64_t *)p.a->buffers[1])[o + i]; ok = sv >= INT32_MIN && sv <= INT32_MAX; }
        else if (p.fmt == 'I') { uv = ((const uint32_t *)p.a->buffers[1])[o + i]; ok = uv <= (unsigned long long)INT32_MAX; sv = (long long)uv; }
        else { uv = ((const uint64_t *)p.a->buffers[1])[o + i]; ok = uv <= (unsigned long long)INT32_MAX; sv = (long long)uv; }
        if (!ok) {
            if (p.fmt == 'l') std::snprintf(buf, sizeof buf, "coordinate value %lld at row %lld overflows i32 (max 2147483647)", sv, (long long)i);
            else std::snprintf(buf, sizeof buf, "coordinate value %llu at row %lld overflows i32 (max 2147483647)", uv, (long long)i);
            return fail(s, buf);
        }
        (*out)[i] = (int32_t)sv;
    }
    return 0;
}

// PosArray::resolve_i64 (array_utils.rs:137-172)
int resolve_i64(brh_session *s, const PosCol &p, std::vector<int64_t> *out)
{
    if (null_count(p.a) > 0) return fail(s, "coordinate column contains null values; requires non-null coordinates");
    const int64_t n = p.a->length, o = p.a->offset;
    out->resize((size_t)n);
    char buf[160];
    for (int64_t i = 0; i < n; i++) {
        switch (p.fmt) {
        case 'i': (*out)[i] = ((const int32_t *)p.a->buffers[1])[o + i]; break;
        case 'l': (*out)[i] = ((const int64_t *)p.a->buffers[1])[o + i]; break;
        case 'I': (*out)[i] = ((const uint32_t *)p.a->buffers[1])[o + i]; break;
        default: {
            const uint64_t v = ((const uint64_t *)p.a->buffers[1])[o + i];
            if (v > (uint64_t)INT64_MAX) {
                std::snprintf(buf, sizeof buf, "coordinate value %llu at row %lld overflows i64 (max 9223372036854775807)", (unsigned long long)v, (long long)i);
                return fail(s, buf);
            }
            (*out)[i] = (int64_t)v;
        }
        }
    }
    return 0;
}

// ---- key dictionary: ids in byte order of the (composite) key strings
struct KeyDict {
    std::vector<std::string> names;                 // id -> name
    std::vector<std::vector<uint32_t>> ids;         // per table: row -> id
};

// what a NULL slot of a key column contributes to the key in the JOIN (a key of its own): a byte string no contig name holds
const std::string_view NULL_KEY("\x1e\x00NULL", 6);

int build_keys(brh_session *s, const std::vector<std::pair<brh_batch, brh_columns>> &tables, KeyDict *kd, bool null_as_key = false)
{
    std::vector<std::vector<std::string>> composite(tables.size());
    std::vector<std::vector<StrCol>> cols(tables.size());
    for (size_t t = 0; t < tables.size(); t++) {
        const brh_columns &c = tables[t].second;
        if (c.n_keys < 1) return fail(s, "at least one key column is required");
        cols[t].resize(c.n_keys);
        for (int k = 0; k < c.n_keys; k++)
            if (get_contig(s, tables[t].first, c.keys[k], &cols[t][k])) return 1;
    }
    std::unordered_map<std::string_view, uint32_t> seen;
    std::vector<std::string_view> uniq;
    auto part = [&](size_t t, size_t k, int64_t i) -> std::string_view {
        return null_as_key && cols[t][k].null_at(i) ? NULL_KEY : cols[t][k].at(i);
    };
    auto key_of = [&](size_t t, int64_t i, std::string &scratch) -> std::string_view {
        if (cols[t].size() == 1) return part(t, 0, i);
        scratch.clear();
        for (size_t k = 0; k < cols[t].size(); k++) { if (k) scratch.push_back('\x1f'); scratch.append(part(t, k, i)); }
        return scratch;
    };
    // pass 1: unique keys (composite keys are materialised once per table).  Rows of one contig come in runs in
    // coordinate-sorted tables: a key equal to the previous row's is not looked up again.
    for (size_t t = 0; t < tables.size(); t++) {
        const int64_t n = tables[t].first.array->length;
        if (cols[t].size() > 1) {
            composite[t].resize((size_t)n);
            std::string scratch;
            for (int64_t i = 0; i < n; i++) composite[t][i] = std::string(key_of(t, i, scratch));
        }
        std::string_view last; bool have = false;
        for (int64_t i = 0; i < n; i++) {
            std::string_view k = cols[t].size() > 1 ? std::string_view(composite[t][i]) : part(t, 0, i);
            if (have && k == last) continue;
            last = k; have = true;
            if (seen.emplace(k, 0).second) uniq.push_back(k);
        }
    }
    std::sort(uniq.begin(), uniq.end());             // byte-lexicographic, as String::cmp (grouped_stream.rs:111)
    for (uint32_t i = 0; i < uniq.size(); i++) seen[uniq[i]] = i;
    kd->names.assign(uniq.begin(), uniq.end());
    kd->ids.resize(tables.size());
    // pass 2: ids.  Names of up to 7 bytes ("chr1" ... ) are compared and looked up as one packed 64-bit word.
    auto pack = [](std::string_view k) -> uint64_t { uint64_t w = 0; std::memcpy(&w, k.data(), k.size()); return w | ((uint64_t)(k.size() + 1) << 56); };
    std::unordered_map<uint64_t, uint32_t> short_ids;
    for (const auto &kv : seen) if (kv.first.size() <= 7) short_ids.emplace(pack(kv.first), kv.second);
    for (size_t t = 0; t < tables.size(); t++) {
        const int64_t n = tables[t].first.array->length;
        kd->ids[t].resize((size_t)n);
        std::string_view last; uint64_t last_packed = 0; uint32_t last_id = 0; bool have = false;
        for (int64_t i = 0; i < n; i++) {
            std::string_view k = cols[t].size() > 1 ? std::string_view(composite[t][i]) : part(t, 0, i);
            if (k.size() <= 7) {
                const uint64_t w = pack(k);
                if (w != last_packed) { last_id = short_ids.find(w)->second; last_packed = w; have = false; }
            } else if (!have || k != last) {
                last_id = seen[k]; last = k; have = true; last_packed = 0;
            }
            kd->ids[t][i] = last_id;
        }
    }
    return 0;
}

// ---- Arrow output helpers
struct OutPriv {
    std::vector<void *> bufs; const void *ptrs[4] = {nullptr, nullptr, nullptr, nullptr}; char *fmt = nullptr; ArrowArray *dict = nullptr;
    std::vector<ArrowArray *> children;                 // nested outputs (struct / list): owned, released with the parent
};

void release_array(ArrowArray *a)
{
    if (!a || !a->release) return;
    OutPriv *p = (OutPriv *)a->private_data;
    for (void *b : p->bufs) std::free(b);
    if (p->dict) { if (p->dict->release) p->dict->release(p->dict); std::free(p->dict); }
    for (ArrowArray *c : p->children) { if (c->release) c->release(c); std::free(c); }
    delete p;
    a->release = nullptr;
}
void release_schema(ArrowSchema *sc)
{
    if (!sc || !sc->release) return;
    std::free((void *)sc->format); std::free((void *)sc->name);
    if (sc->dictionary) { if (sc->dictionary->release) sc->dictionary->release(sc->dictionary); std::free(sc->dictionary); }
    for (int64_t c = 0; c < sc->n_children; c++) { if (sc->children[c]->release) sc->children[c]->release(sc->children[c]); std::free(sc->children[c]); }
    std::free(sc->children);
    sc->release = nullptr;
}

void make_schema(ArrowSchema *sc, const char *fmt, const char *name, bool nullable)
{
    std::memset(sc, 0, sizeof *sc);
    sc->format = strdup(fmt); sc->name = strdup(name); sc->flags = nullable ? 2 : 0;   // ARROW_FLAG_NULLABLE
    sc->release = release_schema;
}

// fixed-width column; validity (may be null) is one byte per row, converted to a bitmap
template <typename T>
void make_primitive(ArrowArray *a, const T *data, int64_t n, const uint8_t *valid)
{
    std::memset(a, 0, sizeof *a);
    OutPriv *p = new OutPriv();
    T *d = (T *)std::malloc((size_t)(n ? n : 1) * sizeof(T));
    if (n) std::memcpy(d, data, (size_t)n * sizeof(T));
    p->bufs.push_back(d);
    uint8_t *bm = nullptr; int64_t nulls = 0;
    if (valid) {
        bm = (uint8_t *)std::calloc((size_t)(n + 7) / 8 + 1, 1);
        for (int64_t i = 0; i < n; i++) { if (valid[i]) bm[i >> 3] |= (uint8_t)(1u << (i & 7)); else nulls++; }
        p->bufs.push_back(bm);
    }
    p->ptrs[0] = nulls ? bm : nullptr; p->ptrs[1] = d;
    a->length = n; a->null_count = nulls; a->n_buffers = 2; a->buffers = p->ptrs; a->private_data = p; a->release = release_array;
}

void make_utf8(ArrowArray *a, const std::vector<std::string> &names, const uint32_t *ids, int64_t n)
{
    std::memset(a, 0, sizeof *a);
    OutPriv *p = new OutPriv();
    int32_t *off = (int32_t *)std::malloc((size_t)(n + 1) * sizeof(int32_t));
    size_t total = 0;
    for (int64_t i = 0; i < n; i++) { off[i] = (int32_t)total; total += names[ids[i]].size(); }
    off[n] = (int32_t)total;
    char *data = (char *)std::malloc(total ? total : 1);
    for (int64_t i = 0; i < n; i++) std::memcpy(data + off[i], names[ids[i]].data(), names[ids[i]].size());
    p->bufs.push_back(off); p->bufs.push_back(data);
    p->ptrs[0] = nullptr; p->ptrs[1] = off; p->ptrs[2] = data;
    a->length = n; a->n_buffers = 3; a->buffers = p->ptrs; a->private_data = p; a->release = release_array;
}

struct Side32 { std::vector<int32_t> s, e; };
int load_side32(brh_session *s, brh_batch t, const brh_columns &c, Side32 *o)
{
    PosCol ps, pe;
    if (get_pos(s, t, c.start, "start", &ps) || get_pos(s, t, c.end, "end", &pe)) return 1;
    return resolve_i32(s, ps, &o->s) || resolve_i32(s, pe, &o->e);
}
struct Side64 { std::vector<int64_t> s, e; };
int load_side64(brh_session *s, brh_batch t, const brh_columns &c, Side64 *o)
{
    PosCol ps, pe;
    if (get_pos(s, t, c.start, "start", &ps) || get_pos(s, t, c.end, "end", &pe)) return 1;
    return resolve_i64(s, ps, &o->s) || resolve_i64(s, pe, &o->e);
}

struct IndexGuard { ivx_index *ix = nullptr; ~IndexGuard() { if (ix) ivx_index_free(ix); } };

}  // namespace

extern "C" int brh_session_create(int device_ordinal, brh_session **out)
{
    if (!out) return 1;
    brh_session *s = new brh_session();
    ivx_status st = ivx_ctx_create(device_ordinal, &s->ctx);
    if (st != IVX_OK) { delete s; *out = nullptr; return (int)st; }   // no CPU fallback
    *out = s;
    return 0;
}
extern "C" void brh_session_free(brh_session *s) { if (s) { ivx_ctx_free(s->ctx); delete s; } }
extern "C" const char *brh_last_error(const brh_session *s) { return s ? s->err.c_str() : "null session"; }

extern "C" int brh_check_position_column(brh_session *s, brh_batch table, const char *column, int as_i64, char *errbuf, int errbuf_len)
{
    brh_session tmp;
    brh_session *u = s ? s : &tmp;
    PosCol p;
    int rc = get_pos(u, table, column, "start", &p);
    if (!rc) {
        if (as_i64) { std::vector<int64_t> v; rc = resolve_i64(u, p, &v); }
        else { std::vector<int32_t> v; rc = resolve_i32(u, p, &v); }
    }
    if (rc && errbuf && errbuf_len > 0) std::snprintf(errbuf, (size_t)errbuf_len, "%s", u->err.c_str());
    return rc;
}

extern "C" int brh_check_contig_column(brh_session *s, brh_batch table, const char *column, char *errbuf, int errbuf_len)
{
    brh_session tmp;
    brh_session *u = s ? s : &tmp;
    StrCol c;
    const int rc = get_contig(u, table, column, &c);
    if (rc && errbuf && errbuf_len > 0) std::snprintf(errbuf, (size_t)errbuf_len, "%s", u->err.c_str());
    return rc;
}

extern "C" int brh_session_metrics(brh_session *s, ivx_metrics *out)
{
    if (!s || !out) return 1;
    return ivx_ctx_metrics(s->ctx, out) == IVX_OK ? 0 : 1;
}

extern "C" int brh_session_set_strict_null_contigs(brh_session *s, int on)
{
    if (!s) return 1;
    s->strict_null_contigs = on != 0;
    return 0;
}

extern "C" int brh_session_set_memory_limit(brh_session *s, uint64_t bytes)
{
    if (!s) return 1;
    return ivx_ctx_set_memory_limit(s->ctx, bytes) == IVX_OK ? 0 : 1;
}

extern "C" int brh_count_overlaps(brh_session *s, brh_batch left, brh_columns lcols, brh_batch right, brh_columns rcols,
                                  int filter_op, int coverage, ArrowArray *out, ArrowSchema *out_schema)
{
    if (!s) return 1;
    KeyDict kd; Side32 L, R;
    if (build_keys(s, {{left, lcols}, {right, rcols}}, &kd) || load_side32(s, left, lcols, &L) || load_side32(s, right, rcols, &R)) return 1;
    const uint32_t nk = (uint32_t)std::max<size_t>(kd.names.size(), 1);
    IndexGuard g;
    ivx_status st = ivx_index_build(s->ctx, coverage ? IVX_KIND_COVERAGE : IVX_KIND_COUNT, IVX_MEM_HOST, kd.ids[0].data(),
                                    L.s.data(), L.e.data(), L.s.size(), nk, &g.ix);
    if (st != IVX_OK) return fail_ivx(s, st);
    std::vector<int64_t> col(R.s.size());
    st = (coverage ? ivx_probe_coverage : ivx_probe_count)(s->ctx, g.ix, IVX_MEM_HOST, kd.ids[1].data(), R.s.data(), R.e.data(),
                                                           R.s.size(), filter_op == BRH_STRICT, col.data());
    if (st != IVX_OK) return fail_ivx(s, st);
    make_primitive<int64_t>(out, col.data(), (int64_t)col.size(), nullptr);
    make_schema(out_schema, "l", coverage ? "coverage" : "count", false);      // count_overlaps.rs:60-66
    return 0;
}

extern "C" int brh_nearest(brh_session *s, brh_batch left, brh_columns lcols, brh_batch right, brh_columns rcols,
                           int filter_op, uint32_t k, int include_overlaps, int compute_distance,
                           ArrowArray *left_idx, ArrowSchema *left_idx_schema, ArrowArray *right_idx, ArrowSchema *right_idx_schema,
                           ArrowArray *distance, ArrowSchema *distance_schema)
{
    if (!s) return 1;
    KeyDict kd; Side32 L, R;
    if (build_keys(s, {{left, lcols}, {right, rcols}}, &kd) || load_side32(s, left, lcols, &L) || load_side32(s, right, rcols, &R)) return 1;
    const uint32_t nk = (uint32_t)std::max<size_t>(kd.names.size(), 1);
    IndexGuard g;
    ivx_status st = ivx_index_build(s->ctx, IVX_KIND_NEAREST, IVX_MEM_HOST, kd.ids[0].data(), L.s.data(), L.e.data(), L.s.size(), nk, &g.ix);
    if (st != IVX_OK) return fail_ivx(s, st);
    const uint64_t cap = R.s.size() * (uint64_t)std::max<uint32_t>(k, 1);
    std::vector<uint32_t> bi(cap ? cap : 1), pi(cap ? cap : 1);
    std::vector<int64_t> di(compute_distance ? (cap ? cap : 1) : 0);
    uint64_t rows = 0;
    st = ivx_probe_nearest(s->ctx, g.ix, IVX_MEM_HOST, kd.ids[1].data(), R.s.data(), R.e.data(), R.s.size(), filter_op == BRH_STRICT, k,
                           include_overlaps, bi.data(), pi.data(), compute_distance ? di.data() : nullptr, cap, &rows);
    if (st != IVX_OK) return fail_ivx(s, st);
    std::vector<uint8_t> valid(rows ? rows : 1);
    for (uint64_t i = 0; i < rows; i++) { valid[i] = bi[i] != IVX_NULL_IDX; if (!valid[i]) bi[i] = 0; }   // nearest.rs:378-379
    make_primitive<uint32_t>(left_idx, bi.data(), (int64_t)rows, valid.data());
    make_schema(left_idx_schema, "I", "left_idx", true);
    make_primitive<uint32_t>(right_idx, pi.data(), (int64_t)rows, nullptr);
    make_schema(right_idx_schema, "I", "right_idx", false);
    if (compute_distance && distance) {
        make_primitive<int64_t>(distance, di.data(), (int64_t)rows, valid.data());
        make_schema(distance_schema, "l", "distance", true);
    }
    return 0;
}

extern "C" int brh_interval_join(brh_session *s, brh_batch build, brh_columns bcols, brh_batch probe, brh_columns pcols,
                                 int join_type, int strict_predicate, int nearest_algorithm,
                                 ArrowArray *build_idx, ArrowSchema *build_idx_schema, ArrowArray *probe_idx, ArrowSchema *probe_idx_schema)
{
    if (!s) return 1;
    KeyDict kd; Side32 B, P;
    if (build_keys(s, {{build, bcols}, {probe, pcols}}, &kd, true) || load_side32(s, build, bcols, &B) || load_side32(s, probe, pcols, &P)) return 1;
    if (strict_predicate) {                                       // `a.start < b.end AND a.end > b.start`: end - 1 on both sides
        for (auto &v : B.e) v = (int32_t)((uint32_t)v - 1u);
        for (auto &v : P.e) v = (int32_t)((uint32_t)v - 1u);
    }
    const uint32_t nk = (uint32_t)std::max<size_t>(kd.names.size(), 1);
    IndexGuard g;
    ivx_status st = ivx_index_build(s->ctx, nearest_algorithm ? IVX_KIND_NEAREST : IVX_KIND_OVERLAP, IVX_MEM_HOST, kd.ids[0].data(),
                                    B.s.data(), B.e.data(), B.s.size(), nk, &g.ix);
    if (st != IVX_OK) return fail_ivx(s, st);
    const uint64_t np = P.s.size();
    if (nearest_algorithm) {                                      // interval_join.rs:864-870, :1628-1635
        std::vector<uint32_t> bi(np ? np : 1), pi(np ? np : 1);
        uint64_t rows = 0;
        st = ivx_probe_nearest(s->ctx, g.ix, IVX_MEM_HOST, kd.ids[1].data(), P.s.data(), P.e.data(), np, 0, 1, 1, bi.data(), pi.data(), nullptr, np, &rows);
        if (st != IVX_OK) return fail_ivx(s, st);
        std::vector<uint8_t> valid(rows ? rows : 1);
        for (uint64_t i = 0; i < rows; i++) { valid[i] = bi[i] != IVX_NULL_IDX; if (!valid[i]) bi[i] = 0; }
        make_primitive<uint32_t>(build_idx, bi.data(), (int64_t)rows, valid.data()); make_schema(build_idx_schema, "I", "build_idx", true);
        make_primitive<uint32_t>(probe_idx, pi.data(), (int64_t)rows, nullptr); make_schema(probe_idx_schema, "I", "probe_idx", false);
        return 0;
    }
    if (join_type == BRH_JOIN_RIGHT_SEMI || join_type == BRH_JOIN_RIGHT_ANTI) {      // :1014-1024, :1433-1447
        std::vector<uint8_t> ex(np ? np : 1);
        st = ivx_probe_exists(s->ctx, g.ix, IVX_MEM_HOST, kd.ids[1].data(), P.s.data(), P.e.data(), np, ex.data());
        if (st != IVX_OK) return fail_ivx(s, st);
        std::vector<uint32_t> pi;
        const bool want = join_type == BRH_JOIN_RIGHT_SEMI;
        for (uint64_t i = 0; i < np; i++) if ((ex[i] != 0) == want) pi.push_back((uint32_t)i);
        make_primitive<uint32_t>(build_idx, nullptr, 0, nullptr); make_schema(build_idx_schema, "I", "build_idx", false);
        make_primitive<uint32_t>(probe_idx, pi.data(), (int64_t)pi.size(), nullptr); make_schema(probe_idx_schema, "I", "probe_idx", false);
        return 0;
    }
    uint64_t total = 0;
    st = ivx_probe_overlap_count(s->ctx, g.ix, IVX_MEM_HOST, kd.ids[1].data(), P.s.data(), P.e.data(), np, nullptr, &total);
    if (st != IVX_OK) return fail_ivx(s, st);
    std::vector<uint32_t> bi(total ? total : 1), pi(total ? total : 1);
    uint64_t written = 0;
    st = ivx_probe_overlap_fill(s->ctx, g.ix, IVX_MEM_HOST, kd.ids[1].data(), P.s.data(), P.e.data(), np, bi.data(), pi.data(), total, &written);
    if (st != IVX_OK) return fail_ivx(s, st);
    make_primitive<uint32_t>(build_idx, bi.data(), (int64_t)written, nullptr); make_schema(build_idx_schema, "I", "build_idx", false);
    make_primitive<uint32_t>(probe_idx, pi.data(), (int64_t)written, nullptr); make_schema(probe_idx_schema, "I", "probe_idx", false);
    return 0;
}

extern "C" int brh_merge(brh_session *s, brh_batch table, brh_columns cols, int64_t min_dist, int filter_op,
                         ArrowArray *contig, ArrowSchema *contig_schema, ArrowArray *start, ArrowSchema *start_schema,
                         ArrowArray *end, ArrowSchema *end_schema, ArrowArray *n_intervals, ArrowSchema *n_schema)
{
    if (!s) return 1;
    KeyDict kd; Side64 T;
    if (build_keys(s, {{table, cols}}, &kd) || load_side64(s, table, cols, &T)) return 1;
    const uint64_t n = T.s.size();
    std::vector<uint32_t> ok(n ? n : 1); std::vector<int64_t> os(n ? n : 1), oe(n ? n : 1), on(n ? n : 1);
    uint64_t m = 0;
    ivx_status st = ivx_merge(s->ctx, IVX_MEM_HOST, kd.ids[0].data(), T.s.data(), T.e.data(), n, (uint32_t)std::max<size_t>(kd.names.size(), 1),
                              min_dist, filter_op == BRH_STRICT, ok.data(), os.data(), oe.data(), on.data(), n, &m);
    if (st != IVX_OK) return fail_ivx(s, st);
    make_utf8(contig, kd.names, ok.data(), (int64_t)m); make_schema(contig_schema, "u", cols.keys[0], false);
    make_primitive<int64_t>(start, os.data(), (int64_t)m, nullptr); make_schema(start_schema, "l", cols.start, false);   // merge.rs:43-48
    make_primitive<int64_t>(end, oe.data(), (int64_t)m, nullptr); make_schema(end_schema, "l", cols.end, false);
    make_primitive<int64_t>(n_intervals, on.data(), (int64_t)m, nullptr); make_schema(n_schema, "l", "n_intervals", false);
    return 0;
}

extern "C" int brh_subtract(brh_session *s, brh_batch left, brh_columns lcols, brh_batch right, brh_columns rcols, int filter_op,
                            ArrowArray *contig, ArrowSchema *contig_schema, ArrowArray *start, ArrowSchema *start_schema,
                            ArrowArray *end, ArrowSchema *end_schema, ArrowArray *left_row, ArrowSchema *left_row_schema)
{
    if (!s) return 1;
    KeyDict kd; Side64 L, R;
    if (build_keys(s, {{left, lcols}, {right, rcols}}, &kd) || load_side64(s, left, lcols, &L) || load_side64(s, right, rcols, &R)) return 1;
    const uint32_t nk = (uint32_t)std::max<size_t>(kd.names.size(), 1);
    uint64_t m = 0, m2 = 0;                              // sizing call, then the fill call
    ivx_status st = ivx_subtract(s->ctx, IVX_MEM_HOST, kd.ids[0].data(), L.s.data(), L.e.data(), L.s.size(), kd.ids[1].data(), R.s.data(), R.e.data(),
                                 R.s.size(), nk, filter_op == BRH_STRICT, nullptr, nullptr, nullptr, nullptr, 0, &m);
    if (st != IVX_OK) return fail_ivx(s, st);
    std::vector<uint32_t> ok(m ? m : 1), orow(m ? m : 1); std::vector<int64_t> os(m ? m : 1), oe(m ? m : 1);
    st = ivx_subtract(s->ctx, IVX_MEM_HOST, kd.ids[0].data(), L.s.data(), L.e.data(), L.s.size(), kd.ids[1].data(), R.s.data(), R.e.data(),
                      R.s.size(), nk, filter_op == BRH_STRICT, ok.data(), os.data(), oe.data(), orow.data(), m, &m2);
    if (st != IVX_OK) return fail_ivx(s, st);
    make_utf8(contig, kd.names, ok.data(), (int64_t)m2); make_schema(contig_schema, "u", lcols.keys[0], false);
    make_primitive<int64_t>(start, os.data(), (int64_t)m2, nullptr); make_schema(start_schema, "l", lcols.start, false);   // subtract.rs:57-72
    make_primitive<int64_t>(end, oe.data(), (int64_t)m2, nullptr); make_schema(end_schema, "l", lcols.end, false);
    make_primitive<uint32_t>(left_row, orow.data(), (int64_t)m2, nullptr); make_schema(left_row_schema, "I", "left_row", false);
    return 0;
}

extern "C" int brh_cluster(brh_session *s, brh_batch table, brh_columns cols, int64_t min_dist, int filter_op,
                           ArrowArray *contig, ArrowSchema *contig_schema, ArrowArray *start, ArrowSchema *start_schema,
                           ArrowArray *end, ArrowSchema *end_schema, ArrowArray *row, ArrowSchema *row_schema,
                           ArrowArray *cluster, ArrowSchema *cluster_schema, ArrowArray *cluster_start, ArrowSchema *cluster_start_schema,
                           ArrowArray *cluster_end, ArrowSchema *cluster_end_schema)
{
    if (!s) return 1;
    KeyDict kd; Side64 T;
    if (build_keys(s, {{table, cols}}, &kd) || load_side64(s, table, cols, &T)) return 1;
    const uint64_t n = T.s.size(), na = n ? n : 1;
    std::vector<uint32_t> ok(na), orow(na); std::vector<int64_t> os(na), oe(na), oc(na), ocs(na), oce(na);
    uint64_t m = 0;
    ivx_status st = ivx_cluster(s->ctx, IVX_MEM_HOST, kd.ids[0].data(), T.s.data(), T.e.data(), n, (uint32_t)std::max<size_t>(kd.names.size(), 1),
                                min_dist, filter_op == BRH_STRICT, nullptr, ok.data(), os.data(), oe.data(), orow.data(),
                                oc.data(), ocs.data(), oce.data(), nullptr, &m);
    if (st != IVX_OK) return fail_ivx(s, st);
    make_utf8(contig, kd.names, ok.data(), (int64_t)n); make_schema(contig_schema, "u", cols.keys[0], false);
    make_primitive<int64_t>(start, os.data(), (int64_t)n, nullptr); make_schema(start_schema, "l", cols.start, false);   // cluster.rs:55-67
    make_primitive<int64_t>(end, oe.data(), (int64_t)n, nullptr); make_schema(end_schema, "l", cols.end, false);
    make_primitive<uint32_t>(row, orow.data(), (int64_t)n, nullptr); make_schema(row_schema, "I", "row", false);
    make_primitive<int64_t>(cluster, oc.data(), (int64_t)n, nullptr); make_schema(cluster_schema, "l", "cluster", false);
    make_primitive<int64_t>(cluster_start, ocs.data(), (int64_t)n, nullptr); make_schema(cluster_start_schema, "l", "cluster_start", false);
    make_primitive<int64_t>(cluster_end, oce.data(), (int64_t)n, nullptr); make_schema(cluster_end_schema, "l", "cluster_end", false);
    return 0;
}

extern "C" int brh_complement(brh_session *s, brh_batch table, brh_columns cols, brh_batch view, brh_columns view_cols, int filter_op,
                              ArrowArray *contig, ArrowSchema *contig_schema, ArrowArray *start, ArrowSchema *start_schema,
                              ArrowArray *end, ArrowSchema *end_schema)
{
    if (!s) return 1;
    const bool has_view = view.array != nullptr;
    KeyDict kd; Side64 T, V;
    if (has_view) {
        if (build_keys(s, {{table, cols}, {view, view_cols}}, &kd) || load_side64(s, table, cols, &T) || load_side64(s, view, view_cols, &V)) return 1;
    } else {
        if (build_keys(s, {{table, cols}}, &kd) || load_side64(s, table, cols, &T)) return 1;
        kd.ids.emplace_back();
    }
    const uint32_t nk = (uint32_t)std::max<size_t>(kd.names.size(), 1);
    const bool strict = filter_op == BRH_STRICT;
    uint64_t m = T.s.size() + 2 * V.s.size() + 2 * (uint64_t)nk + 16, m2 = 0;
    std::vector<uint32_t> ok(m); std::vector<int64_t> os(m), oe(m);
    ivx_status st = ivx_complement(s->ctx, IVX_MEM_HOST, kd.ids[0].data(), T.s.data(), T.e.data(), T.s.size(),
                                   kd.ids[1].data(), V.s.data(), V.e.data(), V.s.size(), nk, strict, ok.data(), os.data(), oe.data(), m, &m2);
    if (st == IVX_ERR_CAPACITY) {                        // heavily overlapping views: repeat with the exact size
        m = m2 ? m2 : 1;
        ok.resize(m); os.resize(m); oe.resize(m);
        st = ivx_complement(s->ctx, IVX_MEM_HOST, kd.ids[0].data(), T.s.data(), T.e.data(), T.s.size(),
                            kd.ids[1].data(), V.s.data(), V.e.data(), V.s.size(), nk, strict, ok.data(), os.data(), oe.data(), m, &m2);
    }
    if (st != IVX_OK) return fail_ivx(s, st);
    make_utf8(contig, kd.names, ok.data(), (int64_t)m2); make_schema(contig_schema, "u", cols.keys[0], false);           // complement.rs:52-58
    make_primitive<int64_t>(start, os.data(), (int64_t)m2, nullptr); make_schema(start_schema, "l", cols.start, false);
    make_primitive<int64_t>(end, oe.data(), (int64_t)m2, nullptr); make_schema(end_schema, "l", cols.end, false);
    return 0;
}

// ---- f3: payload gather
namespace {

// bytes per element of a fixed-width Arrow format string, 0 if it is not one we gather
uint32_t fixed_width(const char *f)
{
    if (std::strlen(f) == 1) {
        switch (f[0]) {
        case 'c': case 'C': return 1;
        case 's': case 'S': case 'e': return 2;
        case 'i': case 'I': case 'f': return 4;
        case 'l': case 'L': case 'g': return 8;
        default: return 0;
        }
    }
    if (!std::strncmp(f, "tdD", 3) || !std::strncmp(f, "tts", 3) || !std::strncmp(f, "ttm", 3)) return 4;      // date32, time32
    if (!std::strncmp(f, "tdm", 3) || !std::strncmp(f, "ttu", 3) || !std::strncmp(f, "ttn", 3)) return 8;      // date64, time64
    if (!std::strncmp(f, "ts", 2) || !std::strncmp(f, "tD", 2)) return 8;                                       // timestamp, duration
    if (!std::strncmp(f, "d:", 2)) {                                                                            // decimal128 / decimal256
        int commas = 0; const char *last = nullptr;
        for (const char *q = f; *q; q++) if (*q == ',') { commas++; last = q; }
        if (commas < 2) return 16;
        const int bits = std::atoi(last + 1);
        return bits == 128 ? 16 : bits == 256 ? 32 : 0;
    }
    if (!std::strncmp(f, "w:", 2)) { const int w = std::atoi(f + 2); return (w == 1 || w == 2 || w == 4 || w == 8 || w == 16 || w == 32) ? (uint32_t)w : 0; }
    return 0;
}

// validity bitmap of `a` re-based to bit 0 (nullptr when there are no nulls)
const uint8_t *rebased_validity(const ArrowArray *a, std::vector<uint8_t> *store)
{
    if (a->n_buffers < 1 || !a->buffers[0] || null_count(a) == 0) return nullptr;
    const uint8_t *v = (const uint8_t *)a->buffers[0];
    if (a->offset % 8 == 0) return v + a->offset / 8;
    store->assign((size_t)(a->length + 7) / 8, 0);
    for (int64_t i = 0; i < a->length; i++) { const int64_t j = i + a->offset; if ((v[j >> 3] >> (j & 7)) & 1) (*store)[i >> 3] |= (uint8_t)(1u << (i & 7)); }
    return store->data();
}

// an owned copy (offset 0) of a flat array: fixed-width primitives, Boolean, Utf8 / LargeUtf8 / Binary / LargeBinary --
// the value types a dictionary-encoded payload column keeps its values in; false for anything else
bool copy_flat_array(const ArrowArray *src, const char *f, ArrowArray *dst)
{
    std::memset(dst, 0, sizeof *dst);
    const int64_t n = src->length, o = src->offset;
    OutPriv *p = new OutPriv();
    int nb = 2;
    const bool s32 = !std::strcmp(f, "u") || !std::strcmp(f, "z"), s64 = !std::strcmp(f, "U") || !std::strcmp(f, "Z");
    if (s32 || s64) {
        const size_t ow = s64 ? 8 : 4;
        const uint8_t *off = (const uint8_t *)src->buffers[1] + (size_t)o * ow;
        const int64_t base = s64 ? ((const int64_t *)off)[0] : (int64_t)((const int32_t *)off)[0];
        const int64_t end = s64 ? ((const int64_t *)off)[n] : (int64_t)((const int32_t *)off)[n];
        void *oo = std::malloc((size_t)(n + 1) * ow);
        for (int64_t i = 0; i <= n; i++) {
            if (s64) ((int64_t *)oo)[i] = ((const int64_t *)off)[i] - base; else ((int32_t *)oo)[i] = ((const int32_t *)off)[i] - (int32_t)base;
        }
        void *od = std::malloc((size_t)(end - base > 0 ? end - base : 1));
        if (end > base) std::memcpy(od, (const uint8_t *)src->buffers[2] + base, (size_t)(end - base));
        p->bufs.push_back(oo); p->bufs.push_back(od); p->ptrs[1] = oo; p->ptrs[2] = od; nb = 3;
    } else if (!std::strcmp(f, "b")) {
        uint8_t *ob = (uint8_t *)std::calloc((size_t)(n + 7) / 8 + 1, 1);
        const uint8_t *v = (const uint8_t *)src->buffers[1];
        for (int64_t i = 0; i < n; i++) { const int64_t j = i + o; if ((v[j >> 3] >> (j & 7)) & 1) ob[i >> 3] |= (uint8_t)(1u << (i & 7)); }
        p->bufs.push_back(ob); p->ptrs[1] = ob;
    } else {
        const uint32_t w = fixed_width(f);
        if (!w) { delete p; return false; }
        void *ov = std::malloc((size_t)(n ? n : 1) * w);
        if (n) std::memcpy(ov, (const uint8_t *)src->buffers[1] + (size_t)o * w, (size_t)n * w);
        p->bufs.push_back(ov); p->ptrs[1] = ov;
    }
    int64_t nulls = 0;
    if (src->n_buffers > 0 && src->buffers[0] && null_count(src) > 0) {
        uint8_t *bm = (uint8_t *)std::calloc((size_t)(n + 7) / 8 + 1, 1);
        const uint8_t *v = (const uint8_t *)src->buffers[0];
        for (int64_t i = 0; i < n; i++) { const int64_t j = i + o; if ((v[j >> 3] >> (j & 7)) & 1) bm[i >> 3] |= (uint8_t)(1u << (i & 7)); else nulls++; }
        p->bufs.push_back(bm); p->ptrs[0] = bm;
    }
    dst->length = n; dst->null_count = nulls; dst->n_buffers = nb; dst->buffers = p->ptrs; dst->private_data = p; dst->release = release_array;
    return true;
}

void finish_array(ArrowArray *a, OutPriv *p, int64_t n, const uint8_t *valid_bytes, int n_buffers)
{
    uint8_t *bm = (uint8_t *)std::calloc((size_t)(n + 7) / 8 + 1, 1);
    int64_t nulls = 0;
    for (int64_t i = 0; i < n; i++) { if (valid_bytes[i]) bm[i >> 3] |= (uint8_t)(1u << (i & 7)); else nulls++; }
    p->bufs.push_back(bm);
    p->ptrs[0] = nulls ? bm : nullptr;
    a->length = n; a->null_count = nulls; a->n_buffers = n_buffers; a->buffers = p->ptrs; a->private_data = p; a->release = release_array;
}

}  // namespace

extern "C" int brh_take(brh_session *s, const ArrowArray *column, const ArrowSchema *column_schema,
                        const ArrowArray *idx, const ArrowSchema *idx_schema, ArrowArray *out, ArrowSchema *out_schema)
{
    if (!s) return 1;
    if (!column || !column_schema || !idx || !idx_schema || !out || !out_schema) return fail(s, "take: null argument");
    if (std::strcmp(idx_schema->format, "I")) return fail(s, "take: the index array must be UInt32, got " + std::string(idx_schema->format));
    const char *f = column_schema->format;
    const int64_t n = idx->length;
    // indices: nulls become IVX_NULL_IDX (nearest.rs:462: left index with a null buffer)
    std::vector<uint32_t> ix((size_t)(n ? n : 1));
    {
        const uint32_t *iv = (const uint32_t *)idx->buffers[1] + idx->offset;
        const uint8_t *vb = (idx->n_buffers > 0 && idx->buffers[0] && null_count(idx) > 0) ? (const uint8_t *)idx->buffers[0] : nullptr;
        for (int64_t i = 0; i < n; i++) {
            const int64_t j = i + idx->offset;
            ix[i] = (vb && !((vb[j >> 3] >> (j & 7)) & 1)) ? IVX_NULL_IDX : iv[i];
        }
    }
    std::vector<uint8_t> vstore, valid((size_t)(n ? n : 1));
    const uint8_t *svb = rebased_validity(column, &vstore);
    const uint64_t n_src = (uint64_t)column->length;
    std::memset(out, 0, sizeof *out);
    const bool str32 = !std::strcmp(f, "u") || !std::strcmp(f, "z"), str64 = !std::strcmp(f, "U") || !std::strcmp(f, "Z");
    const bool is_struct = !std::strcmp(f, "+s"), is_list = !std::strcmp(f, "+l") || !std::strcmp(f, "+m"), is_llist = !std::strcmp(f, "+L"),
               is_fsl = !std::strncmp(f, "+w:", 3);
    if (is_struct || is_list || is_llist || is_fsl) {
        // nested columns (arrow's take handles any type, interval_join.rs:1655-1667): the row selection of a struct is the same
        // take on every child; a list's rows become ranges of child elements, i.e. one more index array for a take on the
        // child.  The leaves end in the device gathers below; offsets and validity of the nesting levels are host work.
        for (int64_t i = 0; i < n; i++) {
            if (ix[i] != IVX_NULL_IDX && ix[i] >= n_src) return fail(s, "take: index " + std::to_string(ix[i]) + " out of range (column has " + std::to_string(n_src) + " rows)");
            valid[i] = ix[i] != IVX_NULL_IDX && (!svb || ((svb[ix[i] >> 3] >> (ix[i] & 7)) & 1));
        }
        const int64_t nchild = column->n_children;
        if (nchild != column_schema->n_children || (!is_struct && nchild != 1)) return fail(s, "take: malformed nested column");
        OutPriv *p = new OutPriv();
        ArrowSchema **cschemas = (ArrowSchema **)std::calloc((size_t)(nchild ? nchild : 1), sizeof(ArrowSchema *));
        auto cleanup = [&](int64_t upto) {
            for (int64_t c = 0; c < upto; c++) { if (cschemas[c]->release) cschemas[c]->release(cschemas[c]); std::free(cschemas[c]); }
            std::free(cschemas);
            for (ArrowArray *c : p->children) { if (c->release) c->release(c); std::free(c); }
            for (void *b : p->bufs) std::free(b);
            delete p;
        };
        // the child rows to take, as a UInt32 index array with nulls
        std::vector<uint32_t> cidx; std::vector<uint8_t> cvalid_bits;
        int n_buffers = 1;
        if (is_struct) {
            cidx = ix;                                              // (null indices stay null; rows under a null struct are taken as they are)
        } else if (is_fsl) {
            const int64_t w = std::atoll(f + 3);
            if (w < 0 || (uint64_t)(column->offset + (int64_t)n_src) * (uint64_t)w >= 0xFFFFFFFFull) { cleanup(0); return fail(s, "take: fixed-size list child too large for 32-bit indices"); }
            cidx.resize((size_t)((n * w) > 0 ? n * w : 1));
            for (int64_t i = 0; i < n; i++)
                for (int64_t j = 0; j < w; j++) cidx[(size_t)(i * w + j)] = ix[i] == IVX_NULL_IDX ? IVX_NULL_IDX : (uint32_t)((column->offset + (int64_t)ix[i]) * w + j);
        } else {
            const size_t ow = is_llist ? 8 : 4;
            const uint8_t *off = (const uint8_t *)column->buffers[1] + (size_t)column->offset * ow;
            auto at = [&](uint64_t r) -> int64_t { return is_llist ? ((const int64_t *)off)[r] : (int64_t)((const int32_t *)off)[r]; };
            void *noff = std::malloc((size_t)(n + 1) * ow);
            int64_t run = 0;
            for (int64_t i = 0; i <= n; i++) {
                if (is_llist) ((int64_t *)noff)[i] = run; else ((int32_t *)noff)[i] = (int32_t)run;
                if (i < n && valid[i]) {
                    const int64_t a = at(ix[i]), b = at((uint64_t)ix[i] + 1);
                    for (int64_t e = a; e < b; e++) cidx.push_back((uint32_t)e);
                    if ((uint64_t)b >= 0xFFFFFFFFull) { std::free(noff); cleanup(0); return fail(s, "take: list child too large for 32-bit indices"); }
                    run += b - a;
                }
            }
            if (!is_llist && run > INT32_MAX) { std::free(noff); cleanup(0); return fail(s, "take: the gathered lists overflow 32-bit offsets"); }
            p->bufs.push_back(noff); p->ptrs[1] = noff; n_buffers = 2;
            if (cidx.empty()) cidx.push_back(0);
        }
        const int64_t n_cidx = is_struct ? n : is_fsl ? n * std::atoll(f + 3) : ((!is_llist) ? (int64_t)((const int32_t *)p->ptrs[1])[n] : ((const int64_t *)p->ptrs[1])[n]);
        // a UInt32 Arrow array over cidx (nulls where IVX_NULL_IDX)
        cvalid_bits.assign((size_t)(n_cidx + 7) / 8 + 1, 0);
        int64_t cnulls = 0;
        for (int64_t i = 0; i < n_cidx; i++) { if (cidx[(size_t)i] != IVX_NULL_IDX) cvalid_bits[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7)); else cnulls++; }
        const void *cbufs[2] = {cnulls ? (const void *)cvalid_bits.data() : nullptr, cidx.data()};
        ArrowArray cia; std::memset(&cia, 0, sizeof cia);
        cia.length = n_cidx; cia.null_count = cnulls; cia.n_buffers = 2; cia.buffers = cbufs;
        for (int64_t c = 0; c < nchild; c++) {
            ArrowArray cc = *column->children[c];                   // (a shallow view: nothing of it is released here)
            if (is_struct) { cc.offset += column->offset; cc.length = column->length; cc.null_count = -1; }
            ArrowArray *co = (ArrowArray *)std::malloc(sizeof(ArrowArray));
            cschemas[c] = (ArrowSchema *)std::malloc(sizeof(ArrowSchema));
            if (brh_take(s, &cc, column_schema->children[c], &cia, idx_schema, co, cschemas[c])) { std::free(co); std::free(cschemas[c]); cleanup(c); return 1; }
            // the child's field keeps its own name and nullability
            cschemas[c]->flags = column_schema->children[c]->flags;
            p->children.push_back(co);
        }
        finish_array(out, p, n, valid.data(), n_buffers);
        out->n_children = nchild;
        out->children = p->children.data();
        make_schema(out_schema, f, column_schema->name ? column_schema->name : "", true);
        out_schema->flags |= column_schema->flags & 4;              // ARROW_FLAG_MAP_KEYS_SORTED
        out_schema->n_children = nchild; out_schema->children = cschemas;
        return 0;
    }
    if (column_schema->dictionary) {
        // dictionary-encoded column (arrow's take gathers the keys and keeps the dictionary): the keys are a fixed-width
        // gather on the device, the output carries an owned copy of the dictionary values
        const uint32_t w = fixed_width(f);
        if (!w || !column->dictionary) return fail(s, "take: malformed dictionary column");
        ArrowArray *dict = (ArrowArray *)std::malloc(sizeof(ArrowArray));
        if (!copy_flat_array(column->dictionary, column_schema->dictionary->format, dict)) {
            std::free(dict);
            return fail(s, "take: unsupported dictionary value type " + std::string(column_schema->dictionary->format) + " (flat value types only)");
        }
        const uint8_t *src = (const uint8_t *)column->buffers[1] + (size_t)column->offset * w;
        void *o = std::malloc((size_t)(n ? n : 1) * w);
        const ivx_status st = ivx_take_fixed(s->ctx, IVX_MEM_HOST, src, w, n_src, svb, ix.data(), (uint64_t)n, o, valid.data());
        if (st != IVX_OK) { std::free(o); dict->release(dict); std::free(dict); return fail_ivx(s, st); }
        OutPriv *p = new OutPriv();
        p->bufs.push_back(o);
        p->ptrs[1] = o;
        p->dict = dict;
        finish_array(out, p, n, valid.data(), 2);
        out->dictionary = dict;
        make_schema(out_schema, f, column_schema->name ? column_schema->name : "", true);
        out_schema->flags |= column_schema->flags & 1;              // ARROW_FLAG_DICTIONARY_ORDERED
        out_schema->dictionary = (ArrowSchema *)std::malloc(sizeof(ArrowSchema));
        make_schema(out_schema->dictionary, column_schema->dictionary->format, column_schema->dictionary->name ? column_schema->dictionary->name : "",
                    (column_schema->dictionary->flags & 2) != 0);
        return 0;
    }
    if (str32 || str64) {
        const size_t ow = str64 ? 8 : 4;
        const uint8_t *off = (const uint8_t *)column->buffers[1] + (size_t)column->offset * ow;
        // the data buffer is addressed by absolute offsets, so it goes over whole: up to the last offset of the slice
        const uint64_t nbytes = str64 ? (uint64_t)((const int64_t *)off)[n_src] : (uint64_t)((const int32_t *)off)[n_src];
        void *ooff = std::malloc((size_t)(n + 1) * ow);
        uint64_t need = 0;
        ivx_status st = ivx_take_utf8(s->ctx, IVX_MEM_HOST, str64, off, (const uint8_t *)column->buffers[2], n_src, nbytes, svb, ix.data(), (uint64_t)n,
                                      ooff, nullptr, 0, &need, valid.data());
        uint8_t *odata = (uint8_t *)std::malloc((size_t)(need ? need : 1));
        if (st == IVX_OK)
            st = ivx_take_utf8(s->ctx, IVX_MEM_HOST, str64, off, (const uint8_t *)column->buffers[2], n_src, nbytes, svb, ix.data(), (uint64_t)n,
                               ooff, odata, need ? need : 1, &need, valid.data());
        if (st != IVX_OK) { std::free(ooff); std::free(odata); return fail_ivx(s, st); }
        OutPriv *p = new OutPriv();
        p->bufs.push_back(ooff); p->bufs.push_back(odata);
        p->ptrs[1] = ooff; p->ptrs[2] = odata;
        finish_array(out, p, n, valid.data(), 3);
    } else if (!std::strcmp(f, "vu") || !std::strcmp(f, "vz")) {   // Utf8View / BinaryView (DataFusion's default string type for Parquet)
        if (column->n_buffers < 3) return fail(s, "take: malformed view array");
        const uint32_t nb = (uint32_t)(column->n_buffers - 3);
        const int64_t *sizes = (const int64_t *)column->buffers[column->n_buffers - 1];
        std::vector<const uint8_t *> bufs(nb ? nb : 1, nullptr); std::vector<uint64_t> bytes(nb ? nb : 1, 0);
        for (uint32_t b = 0; b < nb; b++) { bufs[b] = (const uint8_t *)column->buffers[2 + b]; bytes[b] = (uint64_t)sizes[b]; }
        const uint8_t *views = (const uint8_t *)column->buffers[1] + (size_t)column->offset * 16;
        void *oviews = std::malloc((size_t)(n ? n : 1) * 16);
        uint64_t need = 0;
        ivx_status st = ivx_take_view(s->ctx, IVX_MEM_HOST, views, bufs.data(), bytes.data(), nb, n_src, svb, ix.data(), (uint64_t)n,
                                      nullptr, nullptr, 0, &need, valid.data());
        uint8_t *odata = (uint8_t *)std::malloc((size_t)(need ? need : 1));
        if (st == IVX_OK)
            st = ivx_take_view(s->ctx, IVX_MEM_HOST, views, bufs.data(), bytes.data(), nb, n_src, svb, ix.data(), (uint64_t)n,
                               oviews, odata, need ? need : 1, &need, valid.data());
        if (st != IVX_OK) { std::free(oviews); std::free(odata); return fail_ivx(s, st); }
        int64_t *osz = (int64_t *)std::malloc(sizeof(int64_t));
        osz[0] = (int64_t)need;
        OutPriv *p = new OutPriv();
        p->bufs.push_back(oviews); p->bufs.push_back(odata); p->bufs.push_back(osz);
        p->ptrs[1] = oviews; p->ptrs[2] = odata; p->ptrs[3] = osz;
        finish_array(out, p, n, valid.data(), 4);
    } else if (!std::strcmp(f, "b")) {                       // Boolean: bit-packed values
        const uint8_t *src = (const uint8_t *)column->buffers[1];
        std::vector<uint8_t> sstore;
        if (column->offset % 8) {                             // re-base the value bitmap like the validity bitmap
            sstore.assign((size_t)(column->length + 7) / 8, 0);
            for (int64_t i = 0; i < column->length; i++) { const int64_t j = i + column->offset; if ((src[j >> 3] >> (j & 7)) & 1) sstore[i >> 3] |= (uint8_t)(1u << (i & 7)); }
            src = sstore.data();
        } else src += column->offset / 8;
        uint8_t *o = (uint8_t *)std::calloc((size_t)(n + 7) / 8 + 1, 1);
        const ivx_status st = ivx_take_bits(s->ctx, IVX_MEM_HOST, src, n_src, svb, ix.data(), (uint64_t)n, o, valid.data());
        if (st != IVX_OK) { std::free(o); return fail_ivx(s, st); }
        OutPriv *p = new OutPriv();
        p->bufs.push_back(o);
        p->ptrs[1] = o;
        finish_array(out, p, n, valid.data(), 2);
    } else {
        const uint32_t w = fixed_width(f);
        if (!w) return fail(s, "take: unsupported column type " + std::string(f) + " (fixed-width primitives, Boolean, Utf8/LargeUtf8/Binary/LargeBinary and their views, dictionaries and struct / list nestings of those only)");
        const uint8_t *src = (const uint8_t *)column->buffers[1] + (size_t)column->offset * w;
        void *o = std::malloc((size_t)(n ? n : 1) * w);
        const ivx_status st = ivx_take_fixed(s->ctx, IVX_MEM_HOST, src, w, n_src, svb, ix.data(), (uint64_t)n, o, valid.data());
        if (st != IVX_OK) { std::free(o); return fail_ivx(s, st); }
        OutPriv *p = new OutPriv();
        p->bufs.push_back(o);
        p->ptrs[1] = o;
        finish_array(out, p, n, valid.data(), 2);
    }
    make_schema(out_schema, f, column_schema->name ? column_schema->name : "", true);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// IntervalJoinStream as a push interface (interval_join.rs:934-1140, :1418-1677): the build side is indexed
// once (WaitBuildSide / collect_left_input :584-700), probe RecordBatches arrive one by one (FetchProbeBatch)
// and each is joined against the index (ProcessProbeBatch).  A GPU call per 8192-row batch would be all launch
// latency, so the probe batches are COALESCED here -- what DataFusion's CoalesceBatchesExec does in front of
// an operator -- into groups of `coalesce_rows` rows; a group is probed in one count + fill pair (its columns
// cross PCIe once, ivx.h "two-call protocol") and yields ONE result: pairs (build_idx, probe_idx) with
// probe_idx counted over the group's concatenated rows, and the row offset of every batch in the group.
struct brh_join_stream {
    brh_session *s = nullptr;
    ivx_index *ix = nullptr;
    std::vector<std::string> pkeys; std::string pstart, pend;      // the probe side's column names (owned)
    std::unordered_map<std::string, uint32_t> dict;                 // build-side key -> id; unknown keys get id nk (no build rows)
    uint32_t nk = 0;
    uint64_t short_keys[1024] = {}; uint32_t short_ids[1024] = {}; uint32_t n_short = 0;   // packed short names -> id
    bool strict = false;
    uint64_t coalesce_rows = 0;
    // the open group
    std::vector<uint32_t> gk; std::vector<int32_t> gs, ge; std::vector<int64_t> goff;
    uint64_t first_batch = 0, n_pushed = 0;
    int join_type = BRH_JOIN_INNER;                                 // BRH_JOIN_* or BRH_JOIN_NEAREST (Algorithm::CoitreesNearest)
    uint64_t max_rows = 0;                                          // output rows per result (0 = a group's rows in one result)
    struct Result { uint64_t first_batch, n_batches; std::vector<uint32_t> bi, pi; std::vector<uint8_t> bvalid; std::vector<int64_t> off; bool last; };
    std::deque<Result> ready;
    ~brh_join_stream() { if (ix) ivx_index_free(ix); }
};

namespace {

// One group of coalesced probe batches -> one result, or -- with an output budget (the reference's low-memory
// stream, interval_join.rs:1153-1299; BIO_MAX_OUTPUT_BATCH_SIZE, :543-548) -- several: a result holds whole probe rows
// and ends after the row at which its running output-row count reaches the budget (:1199-1216, :1256-1273), so it
// stays below budget + the matches of its last row.  probe_idx always counts over the group's concatenated rows.
int stream_flush(brh_join_stream *js)
{
    brh_session *s = js->s;
    if (js->goff.empty()) return 0;
    const uint64_t np = js->gs.size();
    std::vector<int64_t> off = js->goff; off.push_back((int64_t)np);
    const uint64_t first = js->first_batch, nb = js->goff.size();
    auto emit = [&](std::vector<uint32_t> &&bi, std::vector<uint32_t> &&pi, std::vector<uint8_t> &&bv, bool last) {
        brh_join_stream::Result r;
        r.first_batch = first; r.n_batches = nb; r.off = off; r.last = last;
        r.bi = std::move(bi); r.pi = std::move(pi); r.bvalid = std::move(bv);
        js->ready.push_back(std::move(r));
    };
    const uint32_t *gk = js->gk.data(); const int32_t *gs = js->gs.data(), *ge = js->ge.data();
    ivx_status st;
    if (js->join_type == BRH_JOIN_RIGHT_SEMI || js->join_type == BRH_JOIN_RIGHT_ANTI) {          // :1014-1024, :1167-1202, :1433-1447
        std::vector<uint8_t> ex(np ? np : 1);
        st = ivx_probe_exists(s->ctx, js->ix, IVX_MEM_HOST, gk, gs, ge, np, ex.data());
        if (st != IVX_OK) return fail_ivx(s, st);
        const bool want = js->join_type == BRH_JOIN_RIGHT_SEMI;
        std::vector<uint32_t> pi;
        for (uint64_t i = 0; i < np; i++) {
            if ((ex[i] != 0) == want) pi.push_back((uint32_t)i);
            if (js->max_rows && pi.size() >= js->max_rows && i + 1 < np) { emit({}, std::move(pi), {}, false); pi.clear(); }
        }
        emit({}, std::move(pi), {}, true);
    } else if (js->join_type == BRH_JOIN_NEAREST) {                                              // :864-870, :1226-1238: one row per probe row
        std::vector<uint32_t> bi(np ? np : 1), pi(np ? np : 1);
        uint64_t rows = 0;
        st = ivx_probe_nearest(s->ctx, js->ix, IVX_MEM_HOST, gk, gs, ge, np, 0, 1, 1, bi.data(), pi.data(), nullptr, np, &rows);
        if (st != IVX_OK) return fail_ivx(s, st);
        const uint64_t step = js->max_rows ? js->max_rows : (rows ? rows : 1);
        for (uint64_t a = 0; a < rows || a == 0; a += step) {
            const uint64_t b = std::min(rows, a + step);
            std::vector<uint32_t> sb(bi.begin() + a, bi.begin() + b), sp(pi.begin() + a, pi.begin() + b);
            std::vector<uint8_t> bv(b - a);
            for (uint64_t i = 0; i < b - a; i++) { bv[i] = sb[i] != IVX_NULL_IDX; if (!bv[i]) sb[i] = 0; }   // NULL build row (u32::MAX marker, :1233)
            emit(std::move(sb), std::move(sp), std::move(bv), b >= rows);
            if (b >= rows) break;
        }
    } else if (!js->max_rows) {
        uint64_t total = 0, written = 0;
        st = ivx_probe_overlap_count(s->ctx, js->ix, IVX_MEM_HOST, gk, gs, ge, np, nullptr, &total);
        if (st != IVX_OK) return fail_ivx(s, st);
        std::vector<uint32_t> bi(total ? total : 1), pi(total ? total : 1);
        st = ivx_probe_overlap_fill(s->ctx, js->ix, IVX_MEM_HOST, gk, gs, ge, np, bi.data(), pi.data(), total, &written);
        if (st != IVX_OK) return fail_ivx(s, st);
        bi.resize(written); pi.resize(written);
        emit(std::move(bi), std::move(pi), {}, true);
    } else {
        // rle_right of the whole group first (ONE device call), then the pairs slice by slice of probe rows
        std::vector<uint32_t> rle(np ? np : 1);
        uint64_t total = 0;
        st = ivx_probe_overlap_count(s->ctx, js->ix, IVX_MEM_HOST, gk, gs, ge, np, rle.data(), &total);
        if (st != IVX_OK) return fail_ivx(s, st);
        uint64_t r0 = 0;
        while (r0 < np || r0 == 0) {
            uint64_t r1 = r0, sum = 0;
            while (r1 < np && sum < js->max_rows) sum += rle[r1++];       // ends after the row that reaches the budget
            if (total - sum == 0 || r1 >= np) {                            // nothing but matchless rows behind it: this is the last result
                for (; r1 < np; r1++) sum += rle[r1];
            }
            std::vector<uint32_t> bi(sum ? sum : 1), pi(sum ? sum : 1);
            uint64_t written = 0;
            if (sum) {
                st = ivx_probe_overlap_fill(s->ctx, js->ix, IVX_MEM_HOST, gk + r0, gs + r0, ge + r0, r1 - r0, bi.data(), pi.data(), sum, &written);
                if (st != IVX_OK) return fail_ivx(s, st);
                if (written != sum) return fail(s, "join stream: a slice's pair count changed between the count and the fill call");
                for (uint64_t i = 0; i < written; i++) pi[i] += (uint32_t)r0;
            }
            bi.resize(written); pi.resize(written);
            total -= sum;
            emit(std::move(bi), std::move(pi), {}, r1 >= np);
            r0 = r1;
            if (r0 >= np) break;
        }
    }
    js->first_batch = js->n_pushed;
    js->gk.clear(); js->gs.clear(); js->ge.clear(); js->goff.clear();
    return 0;
}

}  // namespace

extern "C" int brh_join_stream_open(brh_session *s, brh_batch build, brh_columns bcols, brh_columns pcols, int strict_predicate,
                                    uint64_t coalesce_rows, int join_type, uint64_t max_output_rows, brh_join_stream **out)
{
    if (!s || !out) return 1;
    *out = nullptr;
    if (join_type < BRH_JOIN_INNER || join_type > BRH_JOIN_NEAREST) return fail(s, "join stream: unsupported join type");
    if (max_output_rows == BRH_MAX_OUTPUT_ENV) {                  // the reference's low-memory default (interval_join.rs:543-548)
        // usize::from_str: an optional '+', then digits only, no overflow; anything else is the default.  A parsed 0 is kept by
        // the reference (it then emits after every probe row); here 0 already means "regular mode", so 0 becomes 1: one
        // output row per batch is the smallest budget there is.
        const char *e = std::getenv("BIO_MAX_OUTPUT_BATCH_SIZE");
        max_output_rows = 100000;
        if (e) {
            const char *p = e + (*e == '+');
            unsigned long long v = 0; bool ok = *p != 0;
            for (; ok && *p; p++) {
                if (*p < '0' || *p > '9' || v > (~0ull - (unsigned)(*p - '0')) / 10) ok = false;
                else v = v * 10 + (unsigned)(*p - '0');
            }
            if (ok) max_output_rows = v ? v : 1;
        }
    }
    if (pcols.n_keys != bcols.n_keys || pcols.n_keys < 1) return fail(s, "both sides need the same number (>= 1) of key columns");
    KeyDict kd; Side32 B;
    if (build_keys(s, {{build, bcols}}, &kd, true) || load_side32(s, build, bcols, &B)) return 1;
    if (strict_predicate) for (auto &v : B.e) v = (int32_t)((uint32_t)v - 1u);          // intervals.rs:85-115
    std::unique_ptr<brh_join_stream> js(new brh_join_stream());
    js->s = s; js->strict = strict_predicate != 0;
    js->join_type = join_type; js->max_rows = max_output_rows;
    js->coalesce_rows = coalesce_rows ? coalesce_rows : (4u << 20);
    js->nk = (uint32_t)kd.names.size();
    for (uint32_t i = 0; i < js->nk; i++) js->dict.emplace(kd.names[i], i);
    for (int k = 0; k < pcols.n_keys; k++) js->pkeys.emplace_back(pcols.keys[k]);
    js->pstart = pcols.start; js->pend = pcols.end;
    // one key id more than the build side has: "contig the build side does not know", never matches
    ivx_status st = ivx_index_build(s->ctx, join_type == BRH_JOIN_NEAREST ? IVX_KIND_NEAREST : IVX_KIND_OVERLAP, IVX_MEM_HOST, kd.ids[0].data(),
                                    B.s.data(), B.e.data(), B.s.size(), js->nk + 1, &js->ix);
    if (st != IVX_OK) return fail_ivx(s, st);
    *out = js.release();
    return 0;
}

extern "C" int brh_join_stream_push(brh_join_stream *js, brh_batch probe, int *n_ready)
{
    if (!js) return 1;
    brh_session *s = js->s;
    std::vector<const char *> keyp;
    for (auto &k : js->pkeys) keyp.push_back(k.c_str());
    const brh_columns pc{keyp.data(), (int)keyp.size(), js->pstart.c_str(), js->pend.c_str()};
    std::vector<StrCol> cols(pc.n_keys);
    for (int k = 0; k < pc.n_keys; k++) if (get_contig(s, probe, pc.keys[k], &cols[k])) return 1;
    Side32 P;
    if (load_side32(s, probe, pc, &P)) return 1;
    const int64_t n = probe.array->length;
    const size_t base = js->gs.size();
    js->goff.push_back((int64_t)base);
    js->gk.resize(base + (size_t)n); js->gs.resize(base + (size_t)n); js->ge.resize(base + (size_t)n);
    // key ids.  Rows of one contig come in runs (coordinate-sorted files), so the previous row's answer is tried
    // first; names of up to 7 bytes ("chr1" ... "chrUn") are looked up as one packed 64-bit word in a small
    // open-addressing table (short_ids, filled as names are met), anything longer in the string dictionary
    std::string scratch, last; uint32_t last_id = 0; bool have_last = false; uint64_t last_packed = 0;
    for (int64_t i = 0; i < n; i++) {
        std::string_view k;
        if (cols.size() == 1) k = cols[0].null_at(i) ? NULL_KEY : cols[0].at(i);
        else { scratch.clear(); for (size_t c = 0; c < cols.size(); c++) { if (c) scratch.push_back('\x1f'); scratch.append(cols[c].null_at(i) ? NULL_KEY : cols[c].at(i)); } k = scratch; }
        if (k.size() <= 7) {
            uint64_t packed = 0;
            std::memcpy(&packed, k.data(), k.size());
            packed |= (uint64_t)(k.size() + 1) << 56;                    // never 0: 0 marks an empty slot
            if (packed != last_packed) {
                bool hit = false;
                uint32_t h = (uint32_t)((packed * 0x9E3779B97F4A7C15ull) >> 54);
                for (;; h = (h + 1) & 1023u) {
                    if (js->short_keys[h] == packed) { last_id = js->short_ids[h]; hit = true; break; }
                    if (js->short_keys[h] == 0) break;
                }
                if (!hit) {
                    auto it = js->dict.find(std::string(k));
                    last_id = it == js->dict.end() ? js->nk : it->second;
                    if (js->n_short < 512) { js->short_keys[h] = packed; js->short_ids[h] = last_id; js->n_short++; }   // h = the empty slot found
                }
                last_packed = packed; have_last = false;
            }
        } else if (!have_last || k != std::string_view(last)) {
            last.assign(k.data(), k.size());
            auto it = js->dict.find(last);
            last_id = it == js->dict.end() ? js->nk : it->second;
            have_last = true; last_packed = 0;
        }
        js->gk[base + (size_t)i] = last_id;
        js->gs[base + (size_t)i] = P.s[(size_t)i];
        js->ge[base + (size_t)i] = js->strict ? (int32_t)((uint32_t)P.e[(size_t)i] - 1u) : P.e[(size_t)i];
    }
    js->n_pushed++;
    if (js->gs.size() >= js->coalesce_rows && stream_flush(js)) return 1;
    if (n_ready) *n_ready = (int)js->ready.size();
    return 0;
}

extern "C" int brh_join_stream_finish(brh_join_stream *js, int *n_ready)
{
    if (!js) return 1;
    if (stream_flush(js)) return 1;
    if (n_ready) *n_ready = (int)js->ready.size();
    return 0;
}

extern "C" int brh_join_stream_next(brh_join_stream *js, uint64_t *first_batch, uint64_t *n_batches, int *group_done,
                                    ArrowArray *build_idx, ArrowSchema *build_idx_schema, ArrowArray *probe_idx, ArrowSchema *probe_idx_schema,
                                    ArrowArray *batch_offsets, ArrowSchema *batch_offsets_schema)
{
    if (!js) return 1;
    if (js->ready.empty()) return fail(js->s, "join stream: no result is ready");
    brh_join_stream::Result &r = js->ready.front();
    if (first_batch) *first_batch = r.first_batch;
    if (n_batches) *n_batches = r.n_batches;
    if (group_done) *group_done = r.last ? 1 : 0;
    const bool nullable = js->join_type == BRH_JOIN_NEAREST;
    make_primitive<uint32_t>(build_idx, r.bi.data(), (int64_t)r.bi.size(), nullable ? r.bvalid.data() : nullptr); make_schema(build_idx_schema, "I", "build_idx", nullable);
    make_primitive<uint32_t>(probe_idx, r.pi.data(), (int64_t)r.pi.size(), nullptr); make_schema(probe_idx_schema, "I", "probe_idx", false);
    make_primitive<int64_t>(batch_offsets, r.off.data(), (int64_t)r.off.size(), nullptr); make_schema(batch_offsets_schema, "l", "batch_offsets", false);
    js->ready.pop_front();
    return 0;
}

extern "C" void brh_join_stream_close(brh_join_stream *js) { delete js; }
