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
    bool exchanged = false;            // the last rcn_shard_exchange went through on every rank
    // Status vote.  A rank-local failure (a reserve that could not allocate, a put_image that did not fit, whatever
    // the host driver reports through rcn_shard_fail) must not leave the peers inside a collective this rank never
    // enters: it is remembered here and travels in the FIRST all-gather of the next exchange, next to the row counts,
    // in front of the one host synchronisation of that call -- every rank then leaves with an error, together.
    int32_t local_status = 0;
    DevBuf verdict;                    // rcn_shard_filter: the filter's verdict per pair of this rank
    DevBuf vote;                       // [world] int32, allocated with the communicators
    std::vector<int32_t> vote_host;
    // optional per-phase timing (rcn_shard_profile): HIP events on the streams the work runs on
    bool profile = false, prof_made = false;
    hipEvent_t pev[64][6];             // [step % 64]: exchange begin / end (ctx stream), fp32 gather begin / end (side), match begin / end
    int prof_x = 0, prof_m = 0;        // exchanges / matches recorded since the last read
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
    if (e == hipSuccess) e = sh->vote.reserve((size_t)world * sizeof(int32_t));
    sh->vote_host.assign((size_t)world, 0);
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
        if (sh->prof_made)
            for (auto &row : sh->pev)
                for (auto &ev : row) (void)hipEventDestroy(ev);
        sh->landing.release();
        sh->counts.release();
        sh->vote.release();
        sh->verdict.release();
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
        // the same arguments fail the same way on every rank: nobody goes on to the exchange
        ctx->set_error("rcn_shard_reserve: bad shape");
        return RCN_ERR_ARG;
    }
    // from here on a failure is local (HIP, memory): it is remembered and voted on by the next exchange
    auto local_fail = [&](hipError_t e, const char *what) {
        ctx->set_error(std::string(what) + ": " + hipGetErrorString(e));
        sh->local_status = RCN_ERR_HIP;
        return RCN_ERR_HIP;
    };
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return local_fail(e, "rcn_shard_reserve: hipSetDevice");
    const int32_t per = per_rank(n_images, sh->world);
    if (n_images != sh->n_images || K != sh->K || D != sh->D) {
        // another shape: nothing may still read or fill the old buffer
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(sh->side);
        if (sh->slab >= 0) { rcn_match_release(ctx); sh->slab = -1; }
        sh->exchanged = false;
        // the shape and the (tiny) count block first: with them even a rank whose landing buffer cannot be had joins the vote
        sh->n_images = n_images; sh->per = per; sh->K = K; sh->D = D;
        sh->local_K.assign((size_t)per, 0);
        sh->all_K.assign((size_t)sh->world * per, 0);
        {
            int32_t lo = 0, cnt = 0;
            rcn_shard_owned_images(n_images, sh->world, sh->rank, &lo, &cnt);
            for (int i = 0; i < cnt; ++i) sh->local_K[i] = K;        // full slots until told otherwise
        }
        sh->pairs.assign(2 * (size_t)pairs_of(n_images, sh->world, sh->rank), 0);
        rcn_shard_pairs(n_images, sh->world, sh->rank, sh->pairs.data());
        e = sh->counts.reserve((size_t)sh->world * per * sizeof(int32_t));
        if (e != hipSuccess) return local_fail(e, "rcn_shard_reserve: count block");
        e = sh->landing.reserve((size_t)sh->world * per * K * D * sizeof(float));
        if (e != hipSuccess) { (void)hipGetLastError(); sh->landing.release(); return local_fail(e, "rcn_shard_reserve: landing buffer"); }
    } else if (!sh->landing.p) {
        e = sh->landing.reserve((size_t)sh->world * per * K * D * sizeof(float));      // a retry after a failed allocation
        if (e != hipSuccess) { (void)hipGetLastError(); sh->landing.release(); return local_fail(e, "rcn_shard_reserve: landing buffer"); }
    }
    if (local_slot_dev) *local_slot_dev = sh->landing.as<float>() + (size_t)sh->rank * per * K * D;
    return RCN_OK;
}

int rcn_shard_fail(rcn_shard *sh, int32_t code)
{
    if (!sh) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(sh->ctx->mu);
    sh->local_status = code < 0 ? code : RCN_ERR_ARG;
    return RCN_OK;
}

int rcn_shard_profile(rcn_shard *sh, int enable)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_HIP(hipSetDevice(ctx->device));
    if (enable && !sh->prof_made) {
        for (auto &row : sh->pev)
            for (auto &ev : row) RCN_HIP(hipEventCreate(&ev));
        sh->prof_made = true;
    }
    sh->profile = enable != 0;
    sh->prof_x = sh->prof_m = 0;
    return RCN_OK;
}

int rcn_shard_put_image(rcn_shard *sh, int32_t img_id, const float *desc_host, int32_t K_img)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    int32_t lo = 0, cnt = 0;
    rcn_shard_owned_images(sh->n_images, sh->world, sh->rank, &lo, &cnt);
    if (sh->n_images < 1 || !sh->landing.p || img_id < lo || img_id >= lo + cnt || K_img < 0 || K_img > sh->K || (K_img > 0 && !desc_host)) {
        ctx->set_error("rcn_shard_put_image: the image is not owned by this rank, or its rows do not fit the reserved slot");
        sh->local_status = RCN_ERR_ARG;        // the peers learn of it in the vote of the next exchange
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
    // no shape at all: a caller bug that is the same on every rank (nobody enters a collective)
    if (sh->n_images < 1 || !sh->counts.p) { ctx->set_error("rcn_shard_exchange: call rcn_shard_reserve first"); return RCN_ERR_ARG; }
    RCN_HIP(hipSetDevice(ctx->device));
    const int32_t per = sh->per, K = sh->K, D = sh->D, world = sh->world;
    int32_t lo = 0, cnt = 0;
    rcn_shard_owned_images(sh->n_images, world, sh->rank, &lo, &cnt);
    float *landing = sh->landing.as<float>();
    float *mine = landing ? landing + (size_t)sh->rank * per * K * D : nullptr;
    hipStream_t st = ctx->stream;
    sh->exchanged = false;
    sh->own_table = false;            // whatever the ctx's own tables hold belongs to the previous exchange
    const int pslot = sh->prof_x % 64;
    const bool prof = sh->profile && sh->prof_made;
    if (prof) RCN_HIP(hipEventRecord(sh->pev[pslot][0], st));

    // ---- phase 1, local: everything that can fail on this rank alone happens here, and only sets the status
    int32_t status = sh->local_status;
    std::string why = status ? ctx->err : std::string();
    if (!status && !landing) { status = RCN_ERR_HIP; why = "rcn_shard_exchange: no landing buffer"; }
    if (!status && local_K)
        for (int i = 0; i < cnt; ++i) {
            if (local_K[i] < 0 || local_K[i] > K) { status = RCN_ERR_ARG; why = "rcn_shard_exchange: a row count exceeds the reserved slot"; break; }
            sh->local_K[i] = local_K[i];
        }
    if (!status) {
        // every image of the grid becomes a view into the landing buffer (full slots for now: the row counts of the
        // other ranks arrive with the gather below); the allocations of the fp16 side happen here, in front of the vote
        int rc = rcn_int_slab_attach(ctx, 0, sh->n_images, world * per, landing, K, D, sh->rank * per, cnt, &sh->slab, nullptr, nullptr);
        if (rc) { status = rc; why = ctx->err; }
    }
    bool ragged_local = false;
    for (int i = 0; i < cnt; ++i) ragged_local |= sh->local_K[i] < K;
    int32_t *counts = sh->counts.as<int32_t>();
    if (!status) {
        // the fp32 all-gather of the previous exchange (side stream) reads `mine` and fills the rest of the landing
        // buffer: nothing of this exchange may overtake it (a no-op when a grid call with exact stages ran in between)
        hipError_t e = sh->f32_queued ? hipStreamWaitEvent(st, sh->ev_f32, 0) : hipSuccess;
        if (e == hipSuccess && local_desc_dev && local_desc_dev != mine && cnt > 0)
            e = hipMemcpyAsync(mine, local_desc_dev, (size_t)cnt * K * D * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { status = RCN_ERR_HIP; why = std::string("rcn_shard_exchange: ") + hipGetErrorString(e); }
    }

    // ---- the vote and the row counts: ONE group of two small all-gathers, then the one host wait of this call.
    // Rows in use per slot are gathered always (whether the images are ragged must not be a per-rank decision) and
    // read back: the image table needs them on the host.
    sh->vote_host[(size_t)sh->rank] = status;
    int32_t *vote = sh->vote.as<int32_t>();
    RCN_HIP(hipMemcpyAsync(vote + sh->rank, &sh->vote_host[(size_t)sh->rank], sizeof(int32_t), hipMemcpyHostToDevice, st));
    RCN_HIP(hipMemcpyAsync(counts + (size_t)sh->rank * per, sh->local_K.data(), (size_t)per * sizeof(int32_t), hipMemcpyHostToDevice, st));
    RCN_NCCL(ncclGroupStart());
    RCN_NCCL(ncclAllGather(vote + sh->rank, vote, 1, ncclInt32, sh->comm, st));
    RCN_NCCL(ncclAllGather(counts + (size_t)sh->rank * per, counts, (size_t)per, ncclInt32, sh->comm, st));
    RCN_NCCL(ncclGroupEnd());
    RCN_HIP(hipMemcpyAsync(sh->vote_host.data(), vote, (size_t)world * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    RCN_HIP(hipMemcpyAsync(sh->all_K.data(), counts, (size_t)world * per * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (!status && ragged_local && local_desc_dev)       // rcn_shard_put_image zero-fills on its own
        k_zero_tails<<<dim3(64, (unsigned)cnt), 256, 0, st>>>(mine, counts + (size_t)sh->rank * per, K, D);
    RCN_HIP(hipGetLastError());
    RCN_HIP(hipStreamSynchronize(st));
    sh->local_status = 0;                              // voted: a later exchange starts clean
    for (int r = 0; r < world; ++r)
        if (sh->vote_host[(size_t)r] != 0) {
            // every rank sees the same votes and leaves here: no rank is left inside a later collective
            if (status) ctx->set_error(why);
            else ctx->set_error("rcn_shard_exchange: rank " + std::to_string(r) + " reported a failure (status " + std::to_string(sh->vote_host[(size_t)r]) +
                                "); the exchange was abandoned on every rank");
            return status ? status : RCN_ERR_COMM;
        }
    bool ragged = false;
    for (int i = 0; i < sh->n_images; ++i) ragged |= sh->all_K[i] != K;

    // ---- phase 2: launches and collectives only (nothing below allocates)
    int rc = RCN_OK;
    if (ragged)      // now with every image's own row count
        rc = rcn_int_slab_attach(ctx, 0, sh->n_images, world * per, landing, K, D, sh->rank * per, cnt, &sh->slab, sh->all_K.data(), counts);
    if (rc) return rc;
    // the histogram of row norms describes THIS exchange's rows (the maxima are running ones): cleared, filled by the local
    // statistics, summed over the ranks
    unsigned *cw = ctx->counters.as<unsigned>();
    RCN_HIP(hipMemsetAsync(cw + RCN_HIST_WORD, 0, RCN_HIST_BINS * sizeof(unsigned), st));
    rc = rcn_int_slab_rowstats(ctx, sh->slab, sh->rank * per, cnt);
    if (rc) return rc;
    // global scale statistics: max |x| (fp32 bits) and max |x|^2 (fp64 bits); non-negative floats order like
    // their bit patterns, so an unsigned max is the floating-point max
    RCN_NCCL(ncclGroupStart());
    RCN_NCCL(ncclAllReduce(cw, cw, 1, ncclUint32, ncclMax, sh->comm, st));
    RCN_NCCL(ncclAllReduce(cw + 2, cw + 2, 1, ncclUint64, ncclMax, sh->comm, st));
    RCN_NCCL(ncclAllReduce(cw + RCN_HIST_WORD, cw + RCN_HIST_WORD, RCN_HIST_BINS, ncclUint32, ncclSum, sh->comm, st));
    RCN_NCCL(ncclGroupEnd());
    // the scale is fixed ON THE DEVICE behind the all-reduce (k_fix_scale): the host neither reads the statistics nor
    // waits; converts the local block, builds the image table (uploaded only when it changed)
    ctx->want_dev_scale = true;
    rc = rcn_int_prepare_all(ctx);
    ctx->want_dev_scale = false;
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
    RCN_NCCL(ncclAllGather(sl.bigmin + (size_t)sh->rank * per, sl.bigmin, (size_t)per, ncclUint64, sh->comm, st));     // smallest BIG-row norm per image
    RCN_NCCL(ncclGroupEnd());
    // fp32 rows: on the side stream with their own communicator, behind the fp16 payload (so the two do
    // not share the links while the coarse kernel is waiting) and hence behind (a) the local block being
    // in place and (b) every reader of the previous batch -- all earlier work of the ctx stream.  The
    // exact stages of the next grid call wait for ev_f32 (ctx->f32_ready).
    const size_t blk32 = (size_t)per * K * D * sizeof(float);
    RCN_HIP(hipEventRecord(sh->ev_local, st));
    if (prof) RCN_HIP(hipEventRecord(sh->pev[pslot][1], st));
    RCN_HIP(hipStreamWaitEvent(sh->side, sh->ev_local, 0));
    if (prof) RCN_HIP(hipEventRecord(sh->pev[pslot][2], sh->side));
    RCN_NCCL(ncclAllGather(mine, landing, blk32, ncclChar, sh->comm32, sh->side));
    if (prof) RCN_HIP(hipEventRecord(sh->pev[pslot][3], sh->side));
    RCN_HIP(hipEventRecord(sh->ev_f32, sh->side));
    ctx->f32_ready = sh->ev_f32;
    sh->f32_queued = true;
    sh->bytes_f16 = (int64_t)world * (blk16 + blkhn + blkn2);
    sh->bytes_f32 = (int64_t)world * blk32;
    sh->exchanged = true;
    if (prof) sh->prof_x++;
    return RCN_OK;
}

int rcn_shard_match(rcn_shard *sh, float ratio, int32_t *out_dev, int64_t out_stride, int32_t *counts_dev)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (sh->slab < 0 || !sh->exchanged) { ctx->set_error("rcn_shard_match: no successful rcn_shard_exchange to match on"); return RCN_ERR_ARG; }
    const int32_t P = (int32_t)(sh->pairs.size() / 2);
    const bool prof = sh->profile && sh->prof_made;
    const int mslot = sh->prof_m % 64;
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
    if (prof) RCN_HIP(hipEventRecord(sh->pev[mslot][4], ctx->stream));
    const int rc = rcn_int_match_grid(ctx, sh->pairs.data(), P, ratio, out_dev, out_stride, counts_dev);
    if (rc) { sh->own_table = false; return rc; }
    if (prof) { RCN_HIP(hipEventRecord(sh->pev[mslot][5], ctx->stream)); sh->prof_m++; }
    return RCN_OK;
}

// The epipolar filter of the pair loop (SequentialReconstructor.cpp:237-269) on the tables the last
// rcn_shard_match(sh, ratio, NULL, 0, NULL) left in the ctx: in place, before rcn_shard_lists.
int rcn_shard_filter(rcn_shard *sh, int32_t *status_host)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!sh->own_table) { ctx->set_error("rcn_shard_filter: call rcn_shard_match with NULL tables first"); return RCN_ERR_ARG; }
    const int32_t P = (int32_t)(sh->pairs.size() / 2);
    if (P == 0) return RCN_OK;
    RCN_HIP(hipSetDevice(ctx->device));
    RCN_HIP(sh->verdict.reserve((size_t)P * sizeof(int32_t)));
    int rc = rcn_int_table_filter(ctx, sh->pairs.data(), P, ctx->out_tmp.as<int32_t>(), sh->K, ctx->cnt_tmp.as<int32_t>(), sh->verdict.as<int32_t>());
    if (rc) return rc;
    if (status_host) {
        RCN_HIP(hipMemcpyAsync(status_host, sh->verdict.p, (size_t)P * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        RCN_HIP(hipStreamSynchronize(ctx->stream));
    }
    return RCN_OK;
}

int rcn_shard_lists(rcn_shard *sh, int64_t *offsets_host, int32_t *qt_host, int64_t capacity, int64_t *total_out)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    // the lock is held from the look at the ctx's tables to the end of the copy: another thread's grid call on this
    // ctx would otherwise be free to resize them in between
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!sh->own_table) { ctx->set_error("rcn_shard_lists: call rcn_shard_match with NULL tables first"); return RCN_ERR_ARG; }
    int rc = rcn_int_compact_begin(ctx, ctx->out_tmp.as<int32_t>(), sh->K, ctx->cnt_tmp.as<int32_t>(), (int32_t)(sh->pairs.size() / 2),
                                   offsets_host, qt_host, capacity, total_out);
    if (rc) return rc;
    return rcn_int_compact_wait(ctx);
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
    int nr = 0;
    if (sh->comm && ncclCommCount(sh->comm, &nr) == ncclSuccess) out->comm_ranks = nr;
    return RCN_OK;
}

// Sums of the phase times recorded since the last call (rcn_shard_profile), then cleared.  Waits for both streams.
int rcn_shard_profile_read(rcn_shard *sh, rcn_shard_times *out)
{
    if (!sh || !out) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    memset(out, 0, sizeof(*out));
    if (!sh->prof_made) return RCN_OK;
    RCN_HIP(hipSetDevice(ctx->device));
    RCN_HIP(hipStreamSynchronize(ctx->stream));
    RCN_HIP(hipStreamSynchronize(sh->side));
    const int nx = std::min(sh->prof_x, 64), nm = std::min(sh->prof_m, 64);
    for (int i = 0; i < nx; ++i) {
        float ms = 0.f;
        RCN_HIP(hipEventElapsedTime(&ms, sh->pev[i][0], sh->pev[i][1])); out->exchange_ms += ms;
        RCN_HIP(hipEventElapsedTime(&ms, sh->pev[i][2], sh->pev[i][3])); out->f32_gather_ms += ms;
    }
    for (int i = 0; i < nm; ++i) {
        float ms = 0.f;
        RCN_HIP(hipEventElapsedTime(&ms, sh->pev[i][4], sh->pev[i][5])); out->match_ms += ms;
    }
    out->exchanges = nx; out->matches = nm;
    sh->prof_x = sh->prof_m = 0;
    return RCN_OK;
}

}  // extern "C"
