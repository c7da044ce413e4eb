// shard.hip -- the image-pair grid sharded over the GPUs of one node: one rcn_shard per GPU,
// RCCL collectives over xGMI called directly (no torch, no MPI).  C ABI: include/rcn.h.
//
// Replaces the pair loop of SequentialReconstructor::matchFeatures (SequentialReconstructor.cpp:202-279)
// when its N x N iterations -- independent units, the reference already runs them under OpenMP
// collapse(2) -- are spread over several GPUs:
//   * every rank owns an equal contiguous block of images ("detected locally");
//   * one exchange replicates the descriptors: row statistics of the local block, an 16-byte
//     ncclAllReduce(max) that fixes the global fp16 scale, fp16 conversion of the LOCAL block only,
//     then in-place ncclAllGather of the converted payload (fp16 rows + half-norms + fp64 norms: what
//     the MFMA coarse pass and its error bound read -- half the bytes of the fp32 rows) on the ctx
//     stream, and an ncclAllGather of the fp32 rows on a side stream: only the exact re-rank of the few
//     uncertified rows reads those, so that transfer hides behind the coarse kernel;
//   * the canonical i < j pair list is dealt round-robin to the ranks; match tables stay on the rank
//     that computed them.  No other exchange.
// Every rank converts with the same global scale, so the tables are bit-identical to a one-GPU run.
#include "rcn_internal.h"

#include <rccl/rccl.h>

#include <algorithm>

static_assert(sizeof(ncclUniqueId) == RCN_SHARD_ID_BYTES, "rcn.h: RCN_SHARD_ID_BYTES must be sizeof(ncclUniqueId)");

struct rcn_shard {
    rcn_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;         // ctx stream: scale statistics + fp16 payload (the critical path)
    ncclComm_t comm32 = nullptr;       // side stream: fp32 rows.  A communicator of its own -- RCCL serialises the
                                       // operations of ONE communicator in issue order whatever stream they are on
    int rank = 0, world = 1;
    hipStream_t side = nullptr;        // carries the fp32 all-gather beside the coarse kernel
    hipEvent_t ev_local = nullptr;     // ctx stream: the local fp32 block is in place / previous readers are queued
    hipEvent_t ev_f32 = nullptr;       // side stream: fp32 rows of every rank have landed
    DevBuf landing;                    // [world * per][K][D] fp32: one K-row slot per image
    DevBuf counts;                     // [world * per] int32: rows in use per slot (all-gathered every exchange)
    std::vector<int32_t> local_K;      // this rank's block of `counts` (rcn_shard_put_image / exchange argument)
    std::vector<int32_t> all_K;        // host copy of `counts` after the gather
    int32_t n_images = 0, per = 0, K = 0, D = 0;
    int slab = -1;
    bool own_table = false;            // the last rcn_shard_match wrote into the ctx's own tables
    bool f32_queued = false;           // ev_f32 has been recorded: a later writer of the landing buffer waits for it
    std::vector<int32_t> pairs;        // this rank's share of the canonical grid
    int64_t bytes_f16 = 0, bytes_f32 = 0;   // payload sizes of the last exchange (whole gather, all ranks)
};

#define RCN_NCCL(call)                                                                   \
    do {                                                                                 \
        ncclResult_t r_ = (call);                                                        \
        if (r_ != ncclSuccess) {                                                         \
            ctx->set_error(std::string(#call) + ": " + ncclGetErrorString(r_));          \
            return RCN_ERR_COMM;                                                         \
        }                                                                                \
    } while (0)

// ---- partition: pure functions -------------------------------------------------------------
static int32_t per_rank(int32_t n_images, int32_t world) { return (n_images + world - 1) / world; }

// rank r's pairs are numbers r, r + world, r + 2 world, ... of the canonical row-major i < j list
static int64_t pairs_of(int32_t n, int32_t world, int32_t rank)
{
    const int64_t total = (int64_t)n * (n - 1) / 2;
    return total > rank ? (total - rank + world - 1) / world : 0;
}

// rows K_i .. K-1 of every local slot <- 0 (ragged images: the tail takes part in the row statistics)
__global__ void k_zero_tails(float *__restrict__ block, const int32_t *__restrict__ Ks, int K, int D)
{
    const int img = blockIdx.y, k = Ks[img];
    float *base = block + ((size_t)img * K + k) * D;
    const size_t n = (size_t)(K - k) * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) base[i] = 0.f;
}

extern "C" {

int rcn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int rcn_shard_owned_images(int32_t n_images, int32_t world, int32_t rank, int32_t *first, int32_t *count)
{
    if (n_images < 0 || world < 1 || rank < 0 || rank >= world || !first || !count) return RCN_ERR_ARG;
    const int32_t per = per_rank(n_images, world);
    const int32_t lo = std::min<int64_t>(n_images, (int64_t)rank * per);
    *first = lo;
    *count = std::min(n_images, lo + per) - lo;
    return RCN_OK;
}

int64_t rcn_shard_pair_count(int32_t n_images, int32_t world, int32_t rank)
{
    if (n_images < 0 || world < 1 || rank < 0 || rank >= world) return RCN_ERR_ARG;
    return pairs_of(n_images, world, rank);
}

int rcn_shard_pairs(int32_t n_images, int32_t world, int32_t rank, int32_t *pairs_out)
{
    if (n_images < 0 || world < 1 || rank < 0 || rank >= world || (!pairs_out && pairs_of(n_images, world, rank) > 0)) return RCN_ERR_ARG;
    int64_t p = 0, k = 0;
    for (int32_t i = 0; i < n_images; ++i) {
        // pairs (i, i+1 .. n-1) are numbers p .. p + (n-1-i) - 1; the first one of this rank is the
        // smallest number >= p congruent to rank
        const int64_t row = n_images - 1 - i;
        int64_t q = p + ((rank - p) % world + world) % world;
        for (; q < p + row; q += world) {
            pairs_out[2 * k] = i;
            pairs_out[2 * k + 1] = (int32_t)(i + 1 + (q - p));
            ++k;
        }
        p += row;
    }
    return RCN_OK;
}

int rcn_shard_unique_id(uint8_t id[RCN_SHARD_ID_BYTES])
{
    if (!id) return RCN_ERR_ARG;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return RCN_ERR_COMM;
    memcpy(id, &u, RCN_SHARD_ID_BYTES);
    return RCN_OK;
}

int rcn_shard_create(rcn_ctx *ctx, int32_t rank, int32_t world, const uint8_t id[RCN_SHARD_ID_BYTES], rcn_shard **out)
{
    if (!ctx || !out) return RCN_ERR_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (world < 1 || rank < 0 || rank >= world || !id) { ctx->set_error("rcn_shard_create: bad argument"); return RCN_ERR_ARG; }
    RCN_HIP(hipSetDevice(ctx->device));
    rcn_shard *sh = new rcn_shard();
    sh->ctx = ctx; sh->rank = rank; sh->world = world;
    ncclUniqueId u;
    memcpy(&u, id, RCN_SHARD_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&sh->comm, world, u, rank);
    if (r != ncclSuccess) {
        ctx->set_error(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
        delete sh;
        return RCN_ERR_COMM;
    }
    r = ncclCommSplit(sh->comm, 0, rank, &sh->comm32, nullptr);
    if (r != ncclSuccess) {
        ctx->set_error(std::string("ncclCommSplit: ") + ncclGetErrorString(r));
        (void)ncclCommDestroy(sh->comm);
        delete sh;
        return RCN_ERR_COMM;
    }
    hipError_t e = hipStreamCreateWithFlags(&sh->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->ev_local, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->ev_f32, hipEventDisableTiming);
    if (e != hipSuccess) {
        ctx->set_error(std::string("rcn_shard_create: ") + hipGetErrorString(e));
        (void)ncclCommDestroy(sh->comm32);
        (void)ncclCommDestroy(sh->comm);
        if (sh->side) (void)hipStreamDestroy(sh->side);
        if (sh->ev_local) (void)hipEventDestroy(sh->ev_local);
        delete sh;
        return RCN_ERR_HIP;
    }
    *out = sh;
    return RCN_OK;
}

void rcn_shard_destroy(rcn_shard *sh)
{
    if (!sh) return;
    rcn_ctx *ctx = sh->ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(sh->side);
        if (ctx->f32_ready == sh->ev_f32) ctx->f32_ready = nullptr;
        // the images of the landing buffer are borrowed views: they must not outlive it
        if (sh->slab >= 0) rcn_match_release(ctx);
        (void)ncclCommDestroy(sh->comm32);
        (void)ncclCommDestroy(sh->comm);
        (void)hipStreamDestroy(sh->side);
        (void)hipEventDestroy(sh->ev_local);
        (void)hipEventDestroy(sh->ev_f32);
        sh->landing.release();
        sh->counts.release();
    }
    delete sh;
}

rcn_ctx *rcn_shard_ctx(rcn_shard *sh) { return sh ? sh->ctx : nullptr; }

int rcn_shard_reserve(rcn_shard *sh, int32_t n_images, int32_t K, int32_t D, float **local_slot_dev)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (n_images < 1 || K < 1 || D < 1 || (int64_t)per_rank(n_images, sh->world) * sh->world * K > 0x7fffffffLL) {
        ctx->set_error("rcn_shard_reserve: bad shape");
        return RCN_ERR_ARG;
    }
    RCN_HIP(hipSetDevice(ctx->device));
    const int32_t per = per_rank(n_images, sh->world);
    if (n_images != sh->n_images || K != sh->K || D != sh->D) {
        // another shape: nothing may still read or fill the old buffer
        RCN_HIP(hipStreamSynchronize(ctx->stream));
        RCN_HIP(hipStreamSynchronize(sh->side));
        if (sh->slab >= 0) { rcn_match_release(ctx); sh->slab = -1; }
        RCN_HIP(sh->landing.reserve((size_t)sh->world * per * K * D * sizeof(float)));
        RCN_HIP(sh->counts.reserve((size_t)sh->world * per * sizeof(int32_t)));
        sh->local_K.assign((size_t)per, 0);
        sh->all_K.assign((size_t)sh->world * per, 0);
        {
            int32_t lo = 0, cnt = 0;
            rcn_shard_owned_images(n_images, sh->world, sh->rank, &lo, &cnt);
            for (int i = 0; i < cnt; ++i) sh->local_K[i] = K;        // full slots until told otherwise
        }
        sh->n_images = n_images; sh->per = per; sh->K = K; sh->D = D;
        sh->pairs.assign(2 * (size_t)pairs_of(n_images, sh->world, sh->rank), 0);
        rcn_shard_pairs(n_images, sh->world, sh->rank, sh->pairs.data());
    }
    if (local_slot_dev) *local_slot_dev = sh->landing.as<float>() + (size_t)sh->rank * per * K * D;
    return RCN_OK;
}

int rcn_shard_put_image(rcn_shard *sh, int32_t img_id, const float *desc_host, int32_t K_img)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    int32_t lo = 0, cnt = 0;
    rcn_shard_owned_images(sh->n_images, sh->world, sh->rank, &lo, &cnt);
    if (sh->n_images < 1 || img_id < lo || img_id >= lo + cnt || K_img < 0 || K_img > sh->K || (K_img > 0 && !desc_host)) {
        ctx->set_error("rcn_shard_put_image: the image is not owned by this rank, or its rows do not fit the reserved slot");
        return RCN_ERR_ARG;
    }
    RCN_HIP(hipSetDevice(ctx->device));
    float *slot = sh->landing.as<float>() + ((size_t)sh->rank * sh->per + (img_id - lo)) * sh->K * sh->D;
    // the fp32 all-gather of the previous exchange (side stream) still reads this rank's block
    if (sh->f32_queued) RCN_HIP(hipStreamWaitEvent(ctx->stream, sh->ev_f32, 0));
    if (K_img > 0)
        RCN_HIP(hipMemcpyAsync(slot, desc_host, (size_t)K_img * sh->D * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (K_img < sh->K)
        RCN_HIP(hipMemsetAsync(slot + (size_t)K_img * sh->D, 0, (size_t)(sh->K - K_img) * sh->D * sizeof(float), ctx->stream));
    RCN_HIP(hipStreamSynchronize(ctx->stream));                 // the host rows are borrowed
    sh->local_K[img_id - lo] = K_img;
    return RCN_OK;
}

int rcn_shard_exchange(rcn_shard *sh, const float *local_desc_dev, const int32_t *local_K)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (sh->n_images < 1) { ctx->set_error("rcn_shard_exchange: call rcn_shard_reserve first"); return RCN_ERR_ARG; }
    RCN_HIP(hipSetDevice(ctx->device));
    const int32_t per = sh->per, K = sh->K, D = sh->D, world = sh->world;
    int32_t lo = 0, cnt = 0;
    rcn_shard_owned_images(sh->n_images, world, sh->rank, &lo, &cnt);
    float *landing = sh->landing.as<float>();
    float *mine = landing + (size_t)sh->rank * per * K * D;
    hipStream_t st = ctx->stream;
    if (local_K)
        for (int i = 0; i < cnt; ++i) {
            if (local_K[i] < 0 || local_K[i] > K) { ctx->set_error("rcn_shard_exchange: a row count exceeds the reserved slot"); return RCN_ERR_ARG; }
            sh->local_K[i] = local_K[i];
        }
    // the fp32 all-gather of the previous exchange (side stream) reads `mine` and fills the rest of the landing
    // buffer: nothing of this exchange may overtake it (a no-op when a grid call with exact stages ran in between)
    if (sh->f32_queued) RCN_HIP(hipStreamWaitEvent(st, sh->ev_f32, 0));
    if (local_desc_dev && local_desc_dev != mine && cnt > 0)
        RCN_HIP(hipMemcpyAsync(mine, local_desc_dev, (size_t)cnt * K * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    // rows in use per slot: every rank's block of counts, all-gathered (always: whether the images are
    // ragged must not be a per-rank decision), then read back -- the image table needs them on the host
    int32_t *counts = sh->counts.as<int32_t>();
    RCN_HIP(hipMemcpyAsync(counts + (size_t)sh->rank * per, sh->local_K.data(), (size_t)per * sizeof(int32_t), hipMemcpyHostToDevice, st));
    RCN_NCCL(ncclAllGather(counts + (size_t)sh->rank * per, counts, (size_t)per, ncclInt32, sh->comm, st));
    RCN_HIP(hipMemcpyAsync(sh->all_K.data(), counts, (size_t)world * per * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    bool ragged_local = false;
    for (int i = 0; i < cnt; ++i) ragged_local |= sh->local_K[i] < K;
    if (ragged_local && local_desc_dev)       // rcn_shard_put_image zero-fills on its own
        k_zero_tails<<<dim3(64, (unsigned)cnt), 256, 0, st>>>(mine, counts + (size_t)sh->rank * per, K, D);
    RCN_HIP(hipGetLastError());
    RCN_HIP(hipStreamSynchronize(st));
    bool ragged = false;
    for (int i = 0; i < sh->n_images; ++i) ragged |= sh->all_K[i] != K;

    // every image of the grid becomes a view into the landing buffer; this rank converts its own block
    int rc = rcn_int_slab_attach(ctx, 0, sh->n_images, world * per, landing, K, D, sh->rank * per, cnt, &sh->slab,
                                 ragged ? sh->all_K.data() : nullptr, ragged ? counts : nullptr);
    if (rc) return rc;
    rc = rcn_int_slab_rowstats(ctx, sh->slab, sh->rank * per, cnt);
    if (rc) return rc;
    // global scale statistics: max |x| (fp32 bits) and max |x|^2 (fp64 bits); non-negative floats order like
    // their bit patterns, so an unsigned max is the floating-point max
    unsigned *cw = ctx->counters.as<unsigned>();
    RCN_NCCL(ncclGroupStart());
    RCN_NCCL(ncclAllReduce(cw, cw, 1, ncclUint32, ncclMax, sh->comm, st));
    RCN_NCCL(ncclAllReduce(cw + 2, cw + 2, 1, ncclUint64, ncclMax, sh->comm, st));
    RCN_NCCL(ncclGroupEnd());
    rc = rcn_int_prepare_all(ctx);          // reads the statistics, converts the local block, builds the image table
    if (rc) return rc;
    const Slab &sl = ctx->slabs[sh->slab];
    const int DPa = ctx->DP ? ctx->DP : 32;
    const size_t blk16 = (size_t)per * sl.Kp * DPa * sizeof(_Float16), blkhn = (size_t)per * sl.Kp * sizeof(float),
                 blkn2 = (size_t)per * K * sizeof(double);
    char *f16 = reinterpret_cast<char *>(sl.f16), *hn = reinterpret_cast<char *>(sl.hn), *n2 = reinterpret_cast<char *>(sl.nrm2);
    RCN_NCCL(ncclGroupStart());
    RCN_NCCL(ncclAllGather(f16 + sh->rank * blk16, f16, blk16, ncclChar, sh->comm, st));
    RCN_NCCL(ncclAllGather(hn + sh->rank * blkhn, hn, blkhn, ncclChar, sh->comm, st));
    RCN_NCCL(ncclAllGather(n2 + sh->rank * blkn2, n2, blkn2, ncclChar, sh->comm, st));
    RCN_NCCL(ncclGroupEnd());
    // fp32 rows: on the side stream with their own communicator, behind the fp16 payload (so the two do
    // not share the links while the coarse kernel is waiting) and hence behind (a) the local block being
    // in place and (b) every reader of the previous batch -- all earlier work of the ctx stream.  The
    // exact stages of the next grid call wait for ev_f32 (ctx->f32_ready).
    const size_t blk32 = (size_t)per * K * D * sizeof(float);
    RCN_HIP(hipEventRecord(sh->ev_local, st));
    RCN_HIP(hipStreamWaitEvent(sh->side, sh->ev_local, 0));
    RCN_NCCL(ncclAllGather(mine, landing, blk32, ncclChar, sh->comm32, sh->side));
    RCN_HIP(hipEventRecord(sh->ev_f32, sh->side));
    ctx->f32_ready = sh->ev_f32;
    sh->f32_queued = true;
    sh->bytes_f16 = (int64_t)world * (blk16 + blkhn + blkn2);
    sh->bytes_f32 = (int64_t)world * blk32;
    return RCN_OK;
}

int rcn_shard_match(rcn_shard *sh, float ratio, int32_t *out_dev, int64_t out_stride, int32_t *counts_dev)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (sh->slab < 0) { ctx->set_error("rcn_shard_match: call rcn_shard_exchange first"); return RCN_ERR_ARG; }
    const int32_t P = (int32_t)(sh->pairs.size() / 2);
    if (!out_dev && !counts_dev) {
        // tables owned by the ctx (host callers that only want the lists: rcn_shard_lists)
        RCN_HIP(hipSetDevice(ctx->device));
        out_stride = sh->K;
        RCN_HIP(ctx->out_tmp.reserve(std::max<size_t>(1, (size_t)P) * out_stride * sizeof(int32_t)));
        RCN_HIP(ctx->cnt_tmp.reserve(std::max<size_t>(1, (size_t)P) * sizeof(int32_t)));
        out_dev = ctx->out_tmp.as<int32_t>();
        counts_dev = ctx->cnt_tmp.as<int32_t>();
        sh->own_table = true;
    } else sh->own_table = false;
    return rcn_int_match_grid(ctx, sh->pairs.data(), P, ratio, out_dev, out_stride, counts_dev);
}

int rcn_shard_lists(rcn_shard *sh, int64_t *offsets_host, int32_t *qt_host, int64_t capacity, int64_t *total_out)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!sh->own_table) { ctx->set_error("rcn_shard_lists: call rcn_shard_match with NULL tables first"); return RCN_ERR_ARG; }
    }
    int rc = rcn_match_compact_begin(ctx, ctx->out_tmp.as<int32_t>(), sh->K, ctx->cnt_tmp.as<int32_t>(), (int32_t)(sh->pairs.size() / 2),
                                     offsets_host, qt_host, capacity, total_out);
    if (rc) return rc;
    return rcn_match_compact_wait(ctx);
}

int rcn_shard_info(const rcn_shard *sh, rcn_shard_stats *out)
{
    if (!sh || !out) return RCN_ERR_ARG;
    memset(out, 0, sizeof(*out));
    out->rank = sh->rank; out->world = sh->world;
    out->n_images = sh->n_images; out->images_per_rank = sh->per;
    out->n_pairs = (int64_t)(sh->pairs.size() / 2);
    out->exchange_bytes_f16 = sh->bytes_f16;
    out->exchange_bytes_f32 = sh->bytes_f32;
    return RCN_OK;
}

}  // extern "C"
