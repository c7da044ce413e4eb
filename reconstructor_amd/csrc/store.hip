// store.hip -- what leaves the matcher: host materialisation of a device match table and the
// on-disk store of features + matches.  C ABI: include/rcn.h.
//
//   rcn_match_compact_*  the per-pair (query feature, train feature) lists the reference's pair loop
//                        keeps in featureMatches (SequentialReconstructor.cpp:260-267, :272-275),
//                        compacted on the GPU and copied to (pinned) host memory
//   rcn_store_*          versioned binary file of descriptors (+ keypoint coordinates) and match lists:
//                        the "features / matches cache" the reference lists as a TODO (README.md:39),
//                        i.e. the resume point between the matching stage and the reconstruction
#include "rcn_internal.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>

namespace {

// ---- compaction ------------------------------------------------------------------------------
// exclusive scan of the per-pair counts (int32) into int64 offsets; one workgroup, n up to millions
__global__ __launch_bounds__(1024) void k_cmp_scan(const int32_t *__restrict__ counts, int n, long long *__restrict__ off)
{
    __shared__ long long sh[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    sh[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const long long v = t >= o ? sh[t - o] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    long long run = sh[t] - s;
    for (int i = lo; i < hi; ++i) { off[i] = run; run += counts[i]; }
    if (t == 1023) off[n] = sh[1023];
}

// one workgroup per pair: ordered compaction of the row (ascending query index) by ballots
__global__ __launch_bounds__(256) void k_cmp_fill(const int32_t *__restrict__ table, int64_t stride,
                                                  const int32_t *__restrict__ counts, const long long *__restrict__ off,
                                                  int2 *__restrict__ qt)
{
    __shared__ int wsum[4];
    const int pair = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int total = counts[pair];
    if (total <= 0) return;
    const int32_t *row = table + (size_t)pair * stride;
    int2 *dst = qt + off[pair];
    int pos = 0;
    for (int64_t q0 = 0; q0 < stride && pos < total; q0 += 256) {
        const int64_t q = q0 + t;
        const int tr = q < stride ? row[q] : -1;
        const unsigned long long m = __ballot(tr >= 0);
        if (lane == 0) wsum[w] = (int)__popcll(m);
        __syncthreads();
        int base = pos;
        for (int i = 0; i < w; ++i) base += wsum[i];
        // never past the pair's range: a table whose non-negative entries outnumber counts[pair] (not yet through the
        // uniqueness pass, stale counts) loses its surplus instead of overwriting the next pair's list
        const int at = base + (int)__popcll(m & ((1ull << lane) - 1ull));
        if (tr >= 0 && at < total) dst[at] = make_int2((int)q, tr);
        pos += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int rcn_host_alloc(void **out, size_t bytes)
{
    if (!out) return RCN_ERR_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return RCN_ERR_HIP; }
    return RCN_OK;
}

void rcn_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int rcn_match_compact_begin(rcn_ctx *ctx, const int32_t *table_dev, int64_t stride, const int32_t *counts_dev,
                            int32_t n_pairs, int64_t *offsets_host, int32_t *qt_host, int64_t capacity, int64_t *total_out)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return rcn_int_compact_begin(ctx, table_dev, stride, counts_dev, n_pairs, offsets_host, qt_host, capacity, total_out);
}

int rcn_match_compact_wait(rcn_ctx *ctx)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return rcn_int_compact_wait(ctx);
}

}  // extern "C"

// shard.hip (rcn_shard_gather_lists): the ordered compaction alone, no copy to the host
void rcn_int_launch_cmp_fill(hipStream_t st, const int32_t *table_dev, int64_t stride, const int32_t *counts_dev, const long long *off, int32_t n_pairs, int2 *qt)
{
    k_cmp_fill<<<n_pairs, 256, 0, st>>>(table_dev, stride, counts_dev, off, qt);
}

// the two above with ctx->mu held by the caller (rcn_shard_lists keeps it across table lookup, compaction and wait)
int rcn_int_compact_begin(rcn_ctx *ctx, const int32_t *table_dev, int64_t stride, const int32_t *counts_dev,
                          int32_t n_pairs, int64_t *offsets_host, int32_t *qt_host, int64_t capacity, int64_t *total_out)
{
    if (n_pairs < 0 || stride < 0 || capacity < 0 || !offsets_host || !total_out ||
        (n_pairs > 0 && (!table_dev || !counts_dev)) || (capacity > 0 && !qt_host)) {
        ctx->set_error("rcn_match_compact_begin: bad argument");
        return RCN_ERR_ARG;
    }
    *total_out = 0;
    offsets_host[0] = 0;
    if (n_pairs == 0) return RCN_OK;
    RCN_HIP(hipSetDevice(ctx->device));
    if (!ctx->copy_stream) {
        RCN_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (auto &e : ctx->cmp_ev) RCN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        RCN_HIP(hipEventCreateWithFlags(&ctx->cmp_filled, hipEventDisableTiming));
    }
    hipStream_t st = ctx->stream;
    RCN_HIP(ctx->cmp_off.reserve(sizeof(long long) * ((size_t)n_pairs + 1)));
    long long *off = ctx->cmp_off.as<long long>();
    k_cmp_scan<<<1, 1024, 0, st>>>(counts_dev, n_pairs, off);
    RCN_HIP(hipGetLastError());
    static_assert(sizeof(long long) == sizeof(int64_t), "offset type");
    RCN_HIP(hipMemcpyAsync(offsets_host, off, sizeof(int64_t) * ((size_t)n_pairs + 1), hipMemcpyDeviceToHost, st));
    RCN_HIP(hipStreamSynchronize(st));             // the one wait: the total decides the copy size
    const int64_t total = offsets_host[n_pairs];
    *total_out = total;
    if (total > capacity) {
        ctx->set_error("rcn_match_compact_begin: qt_host holds fewer entries than the table has matches");
        return RCN_ERR_ARG;
    }
    if (total == 0) return RCN_OK;
    // ONE staging buffer in HBM (round 3 alternated two: 2 x 4.1 GB at cfg 3).  The compaction of step k+1 waits for the copy
    // of step k on the device (an event, not the host) -- a copy that is issued a whole grid call earlier and has long finished.
    const int b = 0;
    // the staging buffer was last read by the copy of the previous call
    if (ctx->cmp_busy[b]) RCN_HIP(hipStreamWaitEvent(st, ctx->cmp_ev[b], 0));
    if ((size_t)total * sizeof(int2) > ctx->cmp_qt[b].cap) {
        if (ctx->cmp_busy[b]) RCN_HIP(hipEventSynchronize(ctx->cmp_ev[b]));
        RCN_HIP(ctx->cmp_qt[b].reserve((size_t)total * sizeof(int2)));
    }
    k_cmp_fill<<<n_pairs, 256, 0, st>>>(table_dev, stride, counts_dev, off, ctx->cmp_qt[b].as<int2>());
    RCN_HIP(hipGetLastError());
    RCN_HIP(hipEventRecord(ctx->cmp_filled, st));
    RCN_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->cmp_filled, 0));
    RCN_HIP(hipMemcpyAsync(qt_host, ctx->cmp_qt[b].p, (size_t)total * sizeof(int2), hipMemcpyDeviceToHost, ctx->copy_stream));
    RCN_HIP(hipEventRecord(ctx->cmp_ev[b], ctx->copy_stream));
    ctx->cmp_busy[b] = true;
    ctx->cmp_last = b;
    return RCN_OK;
}

int rcn_int_compact_wait(rcn_ctx *ctx)
{
    if (ctx->cmp_last >= 0 && ctx->cmp_busy[ctx->cmp_last]) RCN_HIP(hipEventSynchronize(ctx->cmp_ev[ctx->cmp_last]));
    return RCN_OK;
}

extern "C" {

// ---- store file --------------------------------------------------------------------------------
// layout (little endian, version 1):
//   header   64 B: "RCNSTORE", u32 version, u32 0x01020304, i32 n_images, i32 D, i32 has_coords, i32 n_pairs,
//                  i64 total_matches, 24 B zero
//   images   n_images x { i32 id, i32 K, f32 desc[K][D], (has_coords) i32 xy[K][2] }
//   matches  i32 pairs[n_pairs][2], i64 offsets[n_pairs + 1], i32 qt[total_matches][2]
//   trailer  u64 FNV-1a over every preceding byte
struct StoreHeader {
    char magic[8];
    uint32_t version, endian;
    int32_t n_images, D, has_coords, n_pairs;
    int64_t total_matches;
    char zero[24];
};
static_assert(sizeof(StoreHeader) == 64, "store header is 64 bytes");

}  // extern "C"

struct rcn_store {
    std::vector<char> bytes;              // the whole file
    StoreHeader h;
    std::vector<int32_t> ids, Ks;
    std::vector<const float *> desc;
    std::vector<const int32_t *> coords;
    const int32_t *pairs = nullptr;
    const int64_t *offsets = nullptr;
    const int32_t *qt = nullptr;
};

namespace {

struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void add(const void *p, size_t n)
    {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    }
};

struct Writer {
    FILE *f;
    Fnv sum;
    bool ok = true;
    void put(const void *p, size_t n)
    {
        if (!ok || n == 0) return;
        sum.add(p, n);
        if (fwrite(p, 1, n, f) != n) ok = false;
    }
};

}  // namespace

extern "C" {

int rcn_store_save(const char *path, const rcn_store_contents *c)
{
    if (!path || !c || c->n_images < 0 || c->n_pairs < 0 || c->D < 0 || (c->n_images > 0 && (!c->img_ids || !c->img_K || !c->desc)) ||
        (c->has_coords && c->n_images > 0 && !c->coords) || (c->n_pairs > 0 && (!c->pairs || !c->offsets)))
        return RCN_ERR_ARG;
    const int64_t total = c->n_pairs > 0 ? c->offsets[c->n_pairs] : 0;
    if (total < 0 || (total > 0 && !c->qt) || (c->n_pairs > 0 && c->offsets[0] != 0)) return RCN_ERR_ARG;
    for (int p = 0; p < c->n_pairs; ++p)
        if (c->offsets[p + 1] < c->offsets[p]) return RCN_ERR_ARG;
    for (int i = 0; i < c->n_images; ++i)
        if (c->img_K[i] < 0 || (c->img_K[i] > 0 && (!c->desc[i] || c->D <= 0 || (c->has_coords && !c->coords[i])))) return RCN_ERR_ARG;
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return RCN_ERR_IO;
    Writer w;
    w.f = f;
    StoreHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "RCNSTORE", 8);
    h.version = 1; h.endian = 0x01020304u;
    h.n_images = c->n_images; h.D = c->D; h.has_coords = c->has_coords ? 1 : 0; h.n_pairs = c->n_pairs;
    h.total_matches = total;
    w.put(&h, sizeof(h));
    for (int i = 0; i < c->n_images; ++i) {
        const int32_t rec[2] = {c->img_ids[i], c->img_K[i]};
        w.put(rec, sizeof(rec));
        w.put(c->desc[i], sizeof(float) * (size_t)c->img_K[i] * c->D);
        if (c->has_coords) w.put(c->coords[i], sizeof(int32_t) * 2 * (size_t)c->img_K[i]);
    }
    if (c->n_pairs > 0) {
        w.put(c->pairs, sizeof(int32_t) * 2 * (size_t)c->n_pairs);
        w.put(c->offsets, sizeof(int64_t) * ((size_t)c->n_pairs + 1));
        w.put(c->qt, sizeof(int32_t) * 2 * (size_t)total);
    }
    const uint64_t sum = w.sum.h;
    if (w.ok && fwrite(&sum, 1, 8, f) != 8) w.ok = false;
    if (fclose(f) != 0) w.ok = false;
    if (!w.ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return RCN_ERR_IO; }
    return RCN_OK;
}

int rcn_store_open(const char *path, rcn_store **out)
{
    if (!path || !out) return RCN_ERR_ARG;
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return RCN_ERR_IO;
    rcn_store *s = new rcn_store();
    bool ok = fseek(f, 0, SEEK_END) == 0;
    const long size = ok ? ftell(f) : -1;
    ok = ok && size >= (long)(sizeof(StoreHeader) + 8) && fseek(f, 0, SEEK_SET) == 0;
    if (ok) {
        // 8-byte aligned backing store: every section below starts at a multiple of 4 and the int64
        // offsets are copied out, so plain pointers into the buffer are safe
        s->bytes.resize((size_t)size);
        ok = fread(s->bytes.data(), 1, (size_t)size, f) == (size_t)size;
    }
    fclose(f);
    if (ok) {
        memcpy(&s->h, s->bytes.data(), sizeof(StoreHeader));
        ok = memcmp(s->h.magic, "RCNSTORE", 8) == 0 && s->h.version == 1 && s->h.endian == 0x01020304u &&
             s->h.n_images >= 0 && s->h.n_pairs >= 0 && s->h.D >= 0 && s->h.total_matches >= 0;
    }
    if (ok) {
        Fnv sum;
        sum.add(s->bytes.data(), (size_t)size - 8);
        uint64_t want;
        memcpy(&want, s->bytes.data() + size - 8, 8);
        ok = want == sum.h;
    }
    if (ok) {
        const char *p = s->bytes.data() + sizeof(StoreHeader), *end = s->bytes.data() + size - 8;
        for (int i = 0; ok && i < s->h.n_images; ++i) {
            int32_t rec[2];
            if (end - p < 8) { ok = false; break; }
            memcpy(rec, p, 8); p += 8;
            // sizes from two int32 header fields: checked by division, so a crafted K x D cannot wrap to a small byte count
            const size_t left = (size_t)(end - p), K = (size_t)std::max(rec[1], 0), rowb = sizeof(float) * (size_t)s->h.D + (s->h.has_coords ? 8 : 0);
            if (rec[1] < 0 || (rowb > 0 && K > left / rowb)) { ok = false; break; }
            const size_t db = sizeof(float) * K * s->h.D, cb = s->h.has_coords ? 8 * K : 0;
            s->ids.push_back(rec[0]); s->Ks.push_back(rec[1]);
            s->desc.push_back(reinterpret_cast<const float *>(p)); p += db;
            s->coords.push_back(s->h.has_coords ? reinterpret_cast<const int32_t *>(p) : nullptr); p += cb;
        }
        if (ok && s->h.n_pairs > 0) {
            const size_t pb = 8 * (size_t)s->h.n_pairs, ob = 8 * ((size_t)s->h.n_pairs + 1), qb = 8 * (size_t)s->h.total_matches;
            if ((uint64_t)s->h.total_matches > (uint64_t)(end - p) / 8 || (size_t)(end - p) != pb + ob + qb) ok = false;
            else {
                s->pairs = reinterpret_cast<const int32_t *>(p); p += pb;
                // the offsets may sit at an address that is only 4-byte aligned: keep an aligned copy
                int64_t *oc = static_cast<int64_t *>(malloc(ob));
                memcpy(oc, p, ob); p += ob;
                s->offsets = oc;
                s->qt = reinterpret_cast<const int32_t *>(p);
                ok = oc[0] == 0 && oc[s->h.n_pairs] == s->h.total_matches;
                for (int pi = 0; ok && pi < s->h.n_pairs; ++pi) ok = oc[pi + 1] >= oc[pi];     // as rcn_store_save demands: consumers slice qt by these
            }
        } else if (ok && p != end) ok = false;
    }
    if (!ok) { free(const_cast<int64_t *>(s->offsets)); delete s; return RCN_ERR_IO; }
    *out = s;
    return RCN_OK;
}

int rcn_store_contents_of(const rcn_store *s, rcn_store_contents *out)
{
    if (!s || !out) return RCN_ERR_ARG;
    memset(out, 0, sizeof(*out));
    out->n_images = s->h.n_images; out->D = s->h.D; out->has_coords = s->h.has_coords; out->n_pairs = s->h.n_pairs;
    out->img_ids = s->ids.data(); out->img_K = s->Ks.data();
    out->desc = s->desc.data(); out->coords = s->coords.data();
    out->pairs = s->pairs; out->offsets = s->offsets; out->qt = s->qt;
    return RCN_OK;
}

void rcn_store_close(rcn_store *s)
{
    if (!s) return;
    free(const_cast<int64_t *>(s->offsets));
    delete s;
}

int rcn_store_upload(rcn_ctx *ctx, const rcn_store *s)
{
    if (!ctx || !s) return RCN_ERR_ARG;
    for (int i = 0; i < s->h.n_images; ++i) {
        int rc = rcn_desc_upload(ctx, s->ids[i], s->Ks[i] ? s->desc[i] : nullptr, s->Ks[i], s->h.D);
        if (rc) return rc;
        if (s->h.has_coords) {
            rc = rcn_coords_upload(ctx, s->ids[i], s->Ks[i] ? s->coords[i] : nullptr, s->Ks[i]);
            if (rc) return rc;
        }
    }
    return RCN_OK;
}

}  // extern "C"
