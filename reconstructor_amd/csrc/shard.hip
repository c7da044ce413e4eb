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

// Host words a read-back in front of a BOUNDED wait lands in.  Pinned on purpose: a device-to-host copy into pageable
// memory is staged by the runtime and may wait, inside hipMemcpyAsync, for everything queued before it -- the collective a dead
// peer never joins included -- and the polling loop with its time limit would never be reached.
template <class T> struct PinVec {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    hipError_t assign(size_t count, T v)
    {
        if (count > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault);
            if (e != hipSuccess) { p = nullptr; n = 0; return e; }
            cap = std::max<size_t>(count, 1);
        }
        n = count;
        for (size_t i = 0; i < count; ++i) p[i] = v;
        return hipSuccess;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = cap = 0; }
};

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
    PinVec<int32_t> all_K;             // host copy of `counts` after the gather (pinned: read back in front of a bounded wait)
    int32_t n_images = 0, per = 0, K = 0, D = 0;
    int slab = -1;
    bool own_table = false;            // the last rcn_shard_match wrote into the shard's own tables (tab / cnt below)
    DevBuf tab, cnt;                   // [pairs][K] int32 / [pairs] int32: tables of rcn_shard_match(sh, ratio, NULL, 0, NULL).  The shard's own --
                                       // the ctx's scratch tables are rewritten by every rcn_match_grid* / rcn_match_pair call
    // A collective cannot be left half-way, so the host never waits for one without a limit: every wait of this file polls an
    // event, watches ncclCommGetAsyncError and gives up after timeout_s.  Giving up means ncclCommAbort on both communicators
    // (the only way to get a stuck RCCL kernel off the streams) and a dead shard: every later call returns RCN_ERR_COMM.
    bool dead = false;
    double timeout_s = 600.0;
    hipEvent_t ev_wait = nullptr;
    int fault = 0;                     // diagnostic build: rcn_diag_shard_fault
    // gather of the lists to one rank (rcn_shard_gather_lists)
    DevBuf g_tot, g_cnt, g_recv, g_loff, g_goff, g_out, g_ccnt;
    PinVec<int64_t> g_tot_host;
    PinVec<int64_t> words;             // [8] pinned staging words of rcn_shard_gather_lists (status marker, capacity verdict, buffer vote)
    bool f32_queued = false;           // ev_f32 has been recorded: a later writer of the landing buffer waits for it
    bool exchanged = false;            // the last rcn_shard_exchange went through on every rank
    // Status vote.  A rank-local failure (a reserve that could not allocate, a put_image that did not fit, whatever
    // the host driver reports through rcn_shard_fail) must not leave the peers inside a collective this rank never
    // enters: it is remembered here and travels in the FIRST all-gather of the next exchange, next to the row counts,
    // in front of the one host synchronisation of that call -- every rank then leaves with an error, together.
    int32_t local_status = 0;
    DevBuf verdict;                    // rcn_shard_filter: the filter's verdict per pair of this rank
    DevBuf vote;                       // [world][VOTE_WORDS] int32, allocated with the communicators
    PinVec<int32_t> vote_host;         // status, n_images, K, D, "my block is ragged", 3 spare
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

#define VOTE_WORDS 8

#include <chrono>
#include <thread>
static double shard_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Off the streams, for good: the communicators are aborted (not destroyed: destroy waits for the operations in flight).
static void shard_abort(rcn_shard *sh)
{
    if (sh->comm32) (void)ncclCommAbort(sh->comm32);
    if (sh->comm) (void)ncclCommAbort(sh->comm);
    sh->comm = sh->comm32 = nullptr;
    sh->dead = true;
    sh->exchanged = false;
    sh->own_table = false;
}

// Bounded host wait for everything queued on `st` so far.  RCN_OK, or the shard is dead and the error text says why.
static int shard_wait(rcn_shard *sh, hipStream_t st, const char *what)
{
    rcn_ctx *ctx = sh->ctx;
    hipError_t e = hipEventRecord(sh->ev_wait, st);
    const double t0 = shard_now();
    for (unsigned spins = 0; e == hipSuccess; ++spins) {
        e = hipEventQuery(sh->ev_wait);
        if (e == hipSuccess) return RCN_OK;
        if (e != hipErrorNotReady) break;
        (void)hipGetLastError();
        e = hipSuccess;
        if ((spins & 255u) == 255u) {
            ncclResult_t ar = ncclSuccess;
            for (ncclComm_t c : {sh->comm, sh->comm32})
                if (c && ncclCommGetAsyncError(c, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                    ctx->set_error(std::string(what) + ": RCCL reported an asynchronous error (" + ncclGetErrorString(ar) + "); communicators aborted");
                    shard_abort(sh);
                    return RCN_ERR_COMM;
                }
            if (shard_now() - t0 > sh->timeout_s) {
                ctx->set_error(std::string(what) + ": no progress for " + std::to_string((int)sh->timeout_s) +
                               " s (a peer left the collective phase?); communicators aborted");
                shard_abort(sh);
                return RCN_ERR_COMM;
            }
        }
        if (spins > 20000u) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    ctx->set_error(std::string(what) + ": " + hipGetErrorString(e));
    shard_abort(sh);
    return RCN_ERR_HIP;
}
static int shard_wait_hook(void *arg, hipStream_t st) { return shard_wait(static_cast<rcn_shard *>(arg), st, "rcn_shard_exchange"); }

#define RCN_SHARD_ALIVE(sh)                                                                            \
    do {                                                                                               \
        if ((sh)->dead) {                                                                              \
            (sh)->ctx->set_error("rcn_shard: the communicators of this shard were aborted (an earlier collective failed or timed out)"); \
            return RCN_ERR_COMM;                                                                       \
        }                                                                                              \
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
    if (e == hipSuccess) e = sh->vote.reserve((size_t)world * VOTE_WORDS * sizeof(int32_t));
    if (e == hipSuccess) e = sh->vote_host.assign((size_t)world * VOTE_WORDS, 0);
    if (e == hipSuccess) e = sh->g_tot.reserve((size_t)world * sizeof(int64_t));
    if (e == hipSuccess) e = sh->g_tot_host.assign((size_t)world, 0);
    if (e == hipSuccess) e = sh->words.assign(8, 0);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->ev_wait, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->ev_local, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->ev_f32, hipEventDisableTiming);
    if (e != hipSuccess) {
        ctx->set_error(std::string("rcn_shard_create: ") + hipGetErrorString(e));
        (void)ncclCommDestroy(sh->comm32);
        (void)ncclCommDestroy(sh->comm);
        if (sh->side) (void)hipStreamDestroy(sh->side);
        if (sh->ev_local) (void)hipEventDestroy(sh->ev_local);
        if (sh->ev_wait) (void)hipEventDestroy(sh->ev_wait);
        sh->vote.release(); sh->g_tot.release();
        sh->vote_host.release(); sh->g_tot_host.release(); sh->words.release();
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
        if (!sh->dead) {      // a live shard's streams hold complete collectives only; a dead one's were cleared by the abort
            (void)shard_wait(sh, ctx->stream, "rcn_shard_destroy");
            if (!sh->dead) (void)shard_wait(sh, sh->side, "rcn_shard_destroy");
        }
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(sh->side);
        if (ctx->f32_ready == sh->ev_f32) ctx->f32_ready = nullptr;
        // the images of the landing buffer are borrowed views: they must not outlive it
        if (sh->slab >= 0) rcn_match_release(ctx);
        if (sh->comm32) (void)ncclCommDestroy(sh->comm32);
        if (sh->comm) (void)ncclCommDestroy(sh->comm);
        (void)hipStreamDestroy(sh->side);
        (void)hipEventDestroy(sh->ev_local);
        (void)hipEventDestroy(sh->ev_f32);
        (void)hipEventDestroy(sh->ev_wait);
        sh->tab.release(); sh->cnt.release();
        sh->g_tot.release(); sh->g_cnt.release(); sh->g_recv.release(); sh->g_loff.release(); sh->g_goff.release(); sh->g_out.release(); sh->g_ccnt.release();
        if (sh->prof_made)
            for (auto &row : sh->pev)
                for (auto &ev : row) (void)hipEventDestroy(ev);
        sh->landing.release();
        sh->counts.release();
        sh->vote.release();
        sh->verdict.release();
        sh->vote_host.release(); sh->g_tot_host.release(); sh->words.release(); sh->all_K.release();
    }
    delete sh;
}

rcn_ctx *rcn_shard_ctx(rcn_shard *sh) { return sh ? sh->ctx : nullptr; }

int rcn_shard_reserve(rcn_shard *sh, int32_t n_images, int32_t K, int32_t D, float **local_slot_dev)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_SHARD_ALIVE(sh);
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
        e = sh->all_K.assign((size_t)sh->world * per, 0);
        if (e != hipSuccess) return local_fail(e, "rcn_shard_reserve: host copy of the row counts");
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
    RCN_SHARD_ALIVE(sh);
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
    RCN_SHARD_ALIVE(sh);
    // no shape at all: a caller bug that is the same on every rank (nobody enters a collective)
    if (sh->n_images < 1) { ctx->set_error("rcn_shard_exchange: call rcn_shard_reserve first"); return RCN_ERR_ARG; }
    const int32_t per = sh->per, K = sh->K, D = sh->D, world = sh->world;
    int32_t lo = 0, cnt = 0;
    rcn_shard_owned_images(sh->n_images, world, sh->rank, &lo, &cnt);
    float *landing = sh->landing.as<float>();
    float *mine = landing ? landing + (size_t)sh->rank * per * K * D : nullptr;
    hipStream_t st = ctx->stream;
    sh->exchanged = false;
    sh->own_table = false;            // whatever the shard's own tables hold belongs to the previous exchange
    const int pslot = sh->prof_x % 64;
    const bool prof = sh->profile && sh->prof_made;

    // ---- phase 1, local: everything that can fail on this rank alone happens here -- every allocation of the whole call
    // included -- and only sets the status.  NOTHING returns between here and the vote.
    int32_t status = sh->local_status;
    std::string why = status ? ctx->err : std::string();
    auto local = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !status) { status = RCN_ERR_HIP; why = std::string("rcn_shard_exchange: ") + what + ": " + hipGetErrorString(e); (void)hipGetLastError(); }
    };
    local(hipSetDevice(ctx->device), "hipSetDevice");
    if (prof) local(hipEventRecord(sh->pev[pslot][0], st), "hipEventRecord");
    if (!status && (!landing || !sh->counts.p)) { status = RCN_ERR_HIP; why = "rcn_shard_exchange: no landing buffer (rcn_shard_reserve failed on this rank)"; }
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
    // what rcn_int_prepare_all would otherwise allocate BEHIND the vote: the scale record and the image table
    local(ctx->scale_dev.reserve(sizeof(ScaleDev)), "scale record");
    local(ctx->img_table.reserve((size_t)std::max(1, sh->n_images) * sizeof(ImgDev)), "image table");
    bool ragged_local = false;
    for (int i = 0; i < cnt; ++i) ragged_local |= sh->local_K[i] < K;
    int32_t *counts = sh->counts.as<int32_t>();
    if (!status) {
        // the fp32 all-gather of the previous exchange (side stream) reads `mine` and fills the rest of the landing
        // buffer: nothing of this exchange may overtake it (a no-op when a grid call with exact stages ran in between)
        if (sh->f32_queued) local(hipStreamWaitEvent(st, sh->ev_f32, 0), "hipStreamWaitEvent");
        if (local_desc_dev && local_desc_dev != mine && cnt > 0)
            local(hipMemcpyAsync(mine, local_desc_dev, (size_t)cnt * K * D * sizeof(float), hipMemcpyDeviceToDevice, st), "copy of the local block");
    }
#ifdef RCN_DIAG
    if (sh->fault == 3 && !status) { status = RCN_ERR_HIP; why = "rcn_shard_exchange: injected local failure (diagnostic build)"; sh->fault = 0; }
#endif

    // ---- the vote: ONE small all-gather -- status, shape and "my block is ragged" per rank -- then the one host wait of an
    // exchange of full slots.  A failure to even enqueue the vote cannot be reported to anybody: the communicators are
    // aborted and the peers' bounded waits end the collective phase on their side.
    int32_t *vh = &sh->vote_host[(size_t)sh->rank * VOTE_WORDS];
    vh[0] = status; vh[1] = sh->n_images; vh[2] = K; vh[3] = D; vh[4] = ragged_local ? 1 : 0; vh[5] = vh[6] = vh[7] = 0;
    int32_t *vote = sh->vote.as<int32_t>();
    {
        hipError_t e = hipMemcpyAsync(vote + (size_t)sh->rank * VOTE_WORDS, vh, VOTE_WORDS * sizeof(int32_t), hipMemcpyHostToDevice, st);
        ncclResult_t r = e == hipSuccess ? ncclAllGather(vote + (size_t)sh->rank * VOTE_WORDS, vote, VOTE_WORDS, ncclInt32, sh->comm, st) : ncclSuccess;
        if (e == hipSuccess && r == ncclSuccess)
            e = hipMemcpyAsync(sh->vote_host.data(), vote, (size_t)world * VOTE_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess || r != ncclSuccess) {
            ctx->set_error(std::string("rcn_shard_exchange: the status vote could not be queued (") + (e != hipSuccess ? hipGetErrorString(e) : ncclGetErrorString(r)) +
                           "); communicators aborted");
            shard_abort(sh);
            return e != hipSuccess ? RCN_ERR_HIP : RCN_ERR_COMM;
        }
    }
    { int rcw = shard_wait(sh, st, "rcn_shard_exchange (status vote)"); if (rcw) return rcw; }
    sh->local_status = 0;                              // voted: a later exchange starts clean
    bool any_ragged = false;
    for (int r = 0; r < world; ++r) {
        const int32_t *v = &sh->vote_host[(size_t)r * VOTE_WORDS];
        if (v[0] != 0) {
            // every rank sees the same votes and leaves here: no rank is left inside a later collective
            if (status) ctx->set_error(why);
            else ctx->set_error("rcn_shard_exchange: rank " + std::to_string(r) + " reported a failure (status " + std::to_string(v[0]) +
                                "); the exchange was abandoned on every rank");
            return status ? status : RCN_ERR_COMM;
        }
    }
    for (int r = 0; r < world; ++r) {
        const int32_t *v = &sh->vote_host[(size_t)r * VOTE_WORDS];
        if (v[1] != sh->n_images || v[2] != K || v[3] != D) {
            ctx->set_error("rcn_shard_exchange: rank " + std::to_string(r) + " reserved another shape (" + std::to_string(v[1]) + " images x " + std::to_string(v[2]) +
                           " x " + std::to_string(v[3]) + "); the exchange was abandoned on every rank");
            return RCN_ERR_ARG;
        }
        any_ragged |= v[4] != 0;
    }

    // ---- phase 2: launches and collectives only.  Nothing below allocates, and nothing returns before the last collective
    // has been entered: an error is recorded, the remaining collectives are still queued (their payload is wasted), and the
    // call reports at the end -- to this caller at once, to the peers through the vote of the next exchange.  Only an RCCL
    // call that refuses to queue leaves the peers without a partner: then the communicators are aborted.
    int p2 = RCN_OK;
    bool comm_broken = false;
    std::string p2why;
    auto soft = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !p2) { p2 = RCN_ERR_HIP; p2why = std::string("rcn_shard_exchange: ") + what + ": " + hipGetErrorString(e); (void)hipGetLastError(); }
    };
    auto softn = [&](ncclResult_t r, const char *what) {
        if (r != ncclSuccess) { comm_broken = true; if (!p2) { p2 = RCN_ERR_COMM; p2why = std::string("rcn_shard_exchange: ") + what + ": " + ncclGetErrorString(r); } }
    };
    auto softrc = [&](int rc) { if (rc && !p2) { p2 = rc; p2why = ctx->err; } };
#ifdef RCN_DIAG
    if (sh->fault == 1) { soft(hipErrorUnknown, "injected failure behind the vote (diagnostic build)"); sh->fault = 0; }
    if (sh->fault == 2) { softn(ncclInternalError, "injected RCCL failure behind the vote (diagnostic build)"); sh->fault = 0; }
#endif
    // Row counts per slot: gathered only when somebody's images are ragged (the vote says so on every rank alike), with a
    // second bounded wait -- the image table needs them on the host.
    if (any_ragged) {
        soft(hipMemcpyAsync(counts + (size_t)sh->rank * per, sh->local_K.data(), (size_t)per * sizeof(int32_t), hipMemcpyHostToDevice, st), "row counts");
        softn(ncclAllGather(counts + (size_t)sh->rank * per, counts, (size_t)per, ncclInt32, sh->comm, st), "ncclAllGather(row counts)");
        soft(hipMemcpyAsync(sh->all_K.data(), counts, (size_t)world * per * sizeof(int32_t), hipMemcpyDeviceToHost, st), "row counts");
        if (ragged_local && local_desc_dev && !p2)       // rcn_shard_put_image zero-fills on its own
            k_zero_tails<<<dim3(64, (unsigned)cnt), 256, 0, st>>>(mine, counts + (size_t)sh->rank * per, K, D);
        soft(hipGetLastError(), "k_zero_tails");
        if (!comm_broken) { int rcw = shard_wait(sh, st, "rcn_shard_exchange (row counts)"); if (rcw) return rcw; }
        // now with every image's own row count
        if (!p2) softrc(rcn_int_slab_attach(ctx, 0, sh->n_images, world * per, landing, K, D, sh->rank * per, cnt, &sh->slab, sh->all_K.data(), counts));
    } else {
        for (int i = 0; i < world * per; ++i) sh->all_K[(size_t)i] = i < sh->n_images ? K : 0;
    }
    // the histogram of row norms describes THIS exchange's rows (the maxima are running ones): cleared, filled by the local
    // statistics, summed over the ranks
    unsigned *cw = ctx->counters.as<unsigned>();
    soft(hipMemsetAsync(cw + RCN_HIST_WORD, 0, RCN_HIST_BINS * sizeof(unsigned), st), "histogram");
    if (!p2) softrc(rcn_int_slab_rowstats(ctx, sh->slab, sh->rank * per, cnt));
    // global scale statistics: max |x| (fp32 bits) and max |x|^2 (fp64 bits); non-negative floats order like
    // their bit patterns, so an unsigned max is the floating-point max
    softn(ncclGroupStart(), "ncclGroupStart");
    softn(ncclAllReduce(cw, cw, 1, ncclUint32, ncclMax, sh->comm, st), "ncclAllReduce");
    softn(ncclAllReduce(cw + 2, cw + 2, 1, ncclUint64, ncclMax, sh->comm, st), "ncclAllReduce");
    softn(ncclAllReduce(cw + RCN_HIST_WORD, cw + RCN_HIST_WORD, RCN_HIST_BINS, ncclUint32, ncclSum, sh->comm, st), "ncclAllReduce");
    softn(ncclGroupEnd(), "ncclGroupEnd");
    // the scale is fixed ON THE DEVICE behind the all-reduce (k_fix_scale): the host neither reads the statistics nor
    // waits; converts the local block, builds the image table (uploaded only when it changed; a wait in there is bounded)
    if (!p2) {
        ctx->want_dev_scale = true;
        ctx->wait_hook = shard_wait_hook; ctx->wait_arg = sh;
        softrc(rcn_int_prepare_all(ctx));
        ctx->wait_hook = nullptr; ctx->wait_arg = nullptr;
        ctx->want_dev_scale = false;
        if (sh->dead) return RCN_ERR_COMM;              // the bounded wait inside gave up
    }
    const Slab &sl = ctx->slabs[sh->slab];
    const int DPa = ctx->DP ? ctx->DP : 32;
    const size_t blk16 = (size_t)per * sl.Kp * DPa * sizeof(_Float16), blkhn = (size_t)per * sl.Kp * sizeof(float),
                 blkn2 = (size_t)per * K * sizeof(double);
    char *f16 = reinterpret_cast<char *>(sl.f16), *hn = reinterpret_cast<char *>(sl.hn), *n2 = reinterpret_cast<char *>(sl.nrm2);
    softn(ncclGroupStart(), "ncclGroupStart");
    softn(ncclAllGather(f16 + sh->rank * blk16, f16, blk16, ncclChar, sh->comm, st), "ncclAllGather(fp16 rows)");
    softn(ncclAllGather(hn + sh->rank * blkhn, hn, blkhn, ncclChar, sh->comm, st), "ncclAllGather(half-norms)");
    softn(ncclAllGather(n2 + sh->rank * blkn2, n2, blkn2, ncclChar, sh->comm, st), "ncclAllGather(norms)");
    softn(ncclAllGather(sl.bigmin + (size_t)sh->rank * per, sl.bigmin, (size_t)per, ncclUint64, sh->comm, st), "ncclAllGather(bigmin)");     // smallest BIG-row norm per image
    softn(ncclGroupEnd(), "ncclGroupEnd");
    // fp32 rows: on the side stream with their own communicator, behind the fp16 payload (so the two do
    // not share the links while the coarse kernel is waiting) and hence behind (a) the local block being
    // in place and (b) every reader of the previous batch -- all earlier work of the ctx stream.  The
    // exact stages of the next grid call wait for ev_f32 (ctx->f32_ready).
    const size_t blk32 = (size_t)per * K * D * sizeof(float);
    soft(hipEventRecord(sh->ev_local, st), "hipEventRecord");
    if (prof) soft(hipEventRecord(sh->pev[pslot][1], st), "hipEventRecord");
    soft(hipStreamWaitEvent(sh->side, sh->ev_local, 0), "hipStreamWaitEvent");
    if (prof) soft(hipEventRecord(sh->pev[pslot][2], sh->side), "hipEventRecord");
    softn(ncclAllGather(mine, landing, blk32, ncclChar, sh->comm32, sh->side), "ncclAllGather(fp32 rows)");
    if (prof) soft(hipEventRecord(sh->pev[pslot][3], sh->side), "hipEventRecord");
    soft(hipEventRecord(sh->ev_f32, sh->side), "hipEventRecord");
    if (p2) {
        if (comm_broken) {
            shard_abort(sh);
            ctx->set_error(p2why + " -- behind the vote: the communicators were aborted, the peers' bounded waits end the exchange on their side");
        } else {
            sh->local_status = p2;      // every collective was entered: the peers learn of it in the next vote
            ctx->set_error(p2why + " -- behind the vote: every collective was still entered; the peers hear of it in the vote of the next exchange");
        }
        return p2;
    }
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
    RCN_SHARD_ALIVE(sh);
    if (sh->slab < 0 || !sh->exchanged) { ctx->set_error("rcn_shard_match: no successful rcn_shard_exchange to match on"); return RCN_ERR_ARG; }
    const int32_t P = (int32_t)(sh->pairs.size() / 2);
    const bool prof = sh->profile && sh->prof_made;
    const int mslot = sh->prof_m % 64;
    if (!out_dev && !counts_dev) {
        // tables owned by the shard (host callers that only want the lists: rcn_shard_lists / rcn_shard_gather_lists)
        RCN_HIP(hipSetDevice(ctx->device));
        out_stride = sh->K;
        RCN_HIP(sh->tab.reserve(std::max<size_t>(1, (size_t)P) * out_stride * sizeof(int32_t)));
        RCN_HIP(sh->cnt.reserve(std::max<size_t>(1, (size_t)P) * sizeof(int32_t)));
        out_dev = sh->tab.as<int32_t>();
        counts_dev = sh->cnt.as<int32_t>();
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
    RCN_SHARD_ALIVE(sh);
    if (!sh->own_table) { ctx->set_error("rcn_shard_filter: call rcn_shard_match with NULL tables first"); return RCN_ERR_ARG; }
    const int32_t P = (int32_t)(sh->pairs.size() / 2);
    if (P == 0) return RCN_OK;
    RCN_HIP(hipSetDevice(ctx->device));
    RCN_HIP(sh->verdict.reserve((size_t)P * sizeof(int32_t)));
    int rc = rcn_int_table_filter(ctx, sh->pairs.data(), P, sh->tab.as<int32_t>(), sh->K, sh->cnt.as<int32_t>(), sh->verdict.as<int32_t>());
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
    RCN_SHARD_ALIVE(sh);
    if (!sh->own_table) { ctx->set_error("rcn_shard_lists: call rcn_shard_match with NULL tables first"); return RCN_ERR_ARG; }
    int rc = rcn_int_compact_begin(ctx, sh->tab.as<int32_t>(), sh->K, sh->cnt.as<int32_t>(), (int32_t)(sh->pairs.size() / 2),
                                   offsets_host, qt_host, capacity, total_out);
    if (rc) return rc;
    return rcn_int_compact_wait(ctx);
}

// ---- the lists of every rank on ONE rank: the single featureMatches map of the reference (SequentialReconstructor.cpp:224,264,274)
// Host half, pure: the canonical i < j order out of per-rank lists (pair number p sits on rank p % world at position p / world).
int rcn_shard_merge_lists(int32_t n_images, int32_t world, const int64_t *const *offsets, const int32_t *const *qt,
                          int64_t *offsets_out, int32_t *qt_out, int64_t capacity, int64_t *total_out)
{
    if (n_images < 0 || world < 1 || !offsets || !offsets_out || !total_out) return RCN_ERR_ARG;
    const int64_t P = (int64_t)n_images * (n_images - 1) / 2;
    int64_t run = 0;
    offsets_out[0] = 0;
    for (int64_t p = 0; p < P; ++p) {
        const int r = (int)(p % world);
        const int64_t l = p / world;
        if (!offsets[r]) return RCN_ERR_ARG;
        run += offsets[r][l + 1] - offsets[r][l];
        offsets_out[p + 1] = run;
    }
    *total_out = run;
    if (run > capacity) return RCN_ERR_ARG;
    if (run > 0 && (!qt || !qt_out)) return RCN_ERR_ARG;
    for (int64_t p = 0; p < P; ++p) {
        const int r = (int)(p % world);
        const int64_t l = p / world, n = offsets[r][l + 1] - offsets[r][l];
        if (n > 0) memcpy(qt_out + 2 * offsets_out[p], qt[r] + 2 * offsets[r][l], (size_t)n * 2 * sizeof(int32_t));
    }
    return RCN_OK;
}

}  // extern "C"

namespace {
// counts of the canonical pairs out of the per-rank blocks [world][pmax] (rank-local order)
__global__ void k_g_canon_counts(const int32_t *__restrict__ rcnt, int world, long pmax, long P, int32_t *__restrict__ ccnt)
{
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x)
        ccnt[p] = rcnt[(p % world) * pmax + p / world];
}
// 1024-thread exclusive scan, one block per segment (segment s: counts + s * stride, n entries -> off + s * (stride + 1))
__global__ __launch_bounds__(1024) void k_g_scan(const int32_t *__restrict__ counts, long stride, long n, long long *__restrict__ off)
{
    __shared__ long long sh[1024];
    counts += (size_t)blockIdx.x * stride;
    off += (size_t)blockIdx.x * (stride + 1);
    const int t = threadIdx.x;
    const long per = (n + 1023) / 1024;
    const long lo = min(n, t * per), hi = min(n, lo + per);
    long long s = 0;
    for (long i = lo; i < hi; ++i) s += counts[i];
    sh[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const long long v = t >= o ? sh[t - o] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    long long run = sh[t] - s;
    for (long i = lo; i < hi; ++i) { off[i] = run; run += counts[i]; }
    if (t == 1023) off[n] = sh[1023];
}
// one wave per canonical pair: its entries from the sender's segment to their place in the merged list
__global__ __launch_bounds__(256) void k_g_interleave(const int2 *__restrict__ recv, const long long *__restrict__ disp, const long long *__restrict__ loff,
                                                      const long long *__restrict__ goff, int world, long pmax, long P, int2 *__restrict__ out)
{
    const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int lane = threadIdx.x & 63, r = (int)(p % world);
    const long l = p / world;
    const long long src = disp[r] + loff[(size_t)r * (pmax + 1) + l], dst = goff[p], n = goff[p + 1] - dst;
    for (long long i = lane; i < n; i += 64) out[dst + i] = recv[src + i];
}
}  // namespace

extern "C" {

// Device half, collective: every rank compacts its tables, the lists travel to `root` over RCCL (device to device), the root
// puts them in canonical pair order on the device and copies offsets and lists to the host ONCE.
int rcn_shard_gather_lists(rcn_shard *sh, int32_t root, const int32_t *table_dev, int64_t stride, const int32_t *counts_dev,
                           int64_t *offsets_host, int32_t *qt_host, int64_t capacity, int64_t *total_out)
{
    if (!sh) return RCN_ERR_ARG;
    rcn_ctx *ctx = sh->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_SHARD_ALIVE(sh);
    const int world = sh->world, rank = sh->rank;
    const bool is_root = rank == root;
    // only what EVERY rank sees alike may return before the collectives: the root argument and the shape all ranks reserved
    if (root < 0 || root >= world || sh->n_images < 1) { ctx->set_error("rcn_shard_gather_lists: bad root, or nothing reserved"); return RCN_ERR_ARG; }
    int32_t status = 0;
    std::string why;
    // Rank-local: "my last exchange did not go through" -- after an error behind the vote only the failing rank knows, its
    // peers hear of it in the NEXT vote -- and a failure that is still waiting for that vote.  Such a rank must not walk away: its peers
    // are about to enter the totals all-gather.  It joins it with the failure marker, and every rank leaves with an error, together.
    if (!sh->exchanged) { status = RCN_ERR_ARG; why = "rcn_shard_gather_lists: this rank's last exchange did not go through (no exchange to gather from)"; }
    else if (sh->local_status) { status = sh->local_status; why = "rcn_shard_gather_lists: a local failure is waiting for the next exchange's vote"; }
    if (!table_dev && !counts_dev) {
        if (!sh->own_table) { status = RCN_ERR_ARG; why = "rcn_shard_gather_lists: call rcn_shard_match with NULL tables first (or pass the tables)"; }
        table_dev = sh->tab.as<int32_t>(); counts_dev = sh->cnt.as<int32_t>(); stride = sh->K;
    } else if (!table_dev || !counts_dev || stride < sh->K) { status = RCN_ERR_ARG; why = "rcn_shard_gather_lists: bad table arguments"; }
    if (is_root && (!offsets_host || !total_out || capacity < 0 || (capacity > 0 && !qt_host))) { status = RCN_ERR_ARG; why = "rcn_shard_gather_lists: bad host arguments on the root"; }
    if (total_out) *total_out = 0;
    const long P = (long)(sh->pairs.size() / 2), Ptot = (long)sh->n_images * (sh->n_images - 1) / 2, pmax = std::max<long>(1, (long)pairs_of(sh->n_images, world, 0));
    hipStream_t st = ctx->stream;
    auto local = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !status) { status = RCN_ERR_HIP; why = std::string("rcn_shard_gather_lists: ") + what + ": " + hipGetErrorString(e); (void)hipGetLastError(); }
    };
    // ---- local: every allocation whose size is known, the scan of this rank's counts
    local(hipSetDevice(ctx->device), "hipSetDevice");
    local(sh->g_cnt.reserve((size_t)world * pmax * sizeof(int32_t)), "count blocks");
    local(ctx->cmp_off.reserve(sizeof(long long) * ((size_t)P + 2)), "offsets");
    if (is_root) {
        local(sh->g_loff.reserve((size_t)world * (pmax + 1) * sizeof(long long) + (size_t)world * sizeof(long long)), "offsets of the segments");
        local(sh->g_goff.reserve(((size_t)Ptot + 2) * sizeof(long long)), "merged offsets");
        local(sh->g_ccnt.reserve((size_t)std::max<long>(1, Ptot) * sizeof(int32_t)), "merged counts");
    }
    int32_t *gcnt = sh->g_cnt.as<int32_t>();
    long long *off = ctx->cmp_off.as<long long>();
    int64_t *gtot = sh->g_tot.as<int64_t>();
    if (!status) {
        local(hipMemsetAsync(gcnt + (size_t)rank * pmax, 0, (size_t)pmax * sizeof(int32_t), st), "count block");
        if (P > 0) local(hipMemcpyAsync(gcnt + (size_t)rank * pmax, counts_dev, (size_t)P * sizeof(int32_t), hipMemcpyDeviceToDevice, st), "count block");
        k_g_scan<<<1, 1024, 0, st>>>(gcnt + (size_t)rank * pmax, pmax, P, off);
        local(hipGetLastError(), "scan");
        local(hipMemcpyAsync(gtot + rank, off + P, sizeof(int64_t), hipMemcpyDeviceToDevice, st), "total");
    }
    // ---- vote + totals: status rides in the high word of nothing -- a total of -1 - code marks a failed rank
    if (status) {
        sh->words[0] = -1 - (int64_t)(-status);
        (void)hipMemcpyAsync(gtot + rank, sh->words.data(), sizeof(int64_t), hipMemcpyHostToDevice, st);
    }
    {
        ncclResult_t r = ncclGroupStart();
        if (r == ncclSuccess) r = ncclAllGather(gtot + rank, gtot, 1, ncclInt64, sh->comm, st);
        if (r == ncclSuccess) r = ncclAllGather(gcnt + (size_t)rank * pmax, gcnt, (size_t)pmax, ncclInt32, sh->comm, st);
        ncclResult_t r2 = ncclGroupEnd();
        hipError_t e = hipMemcpyAsync(sh->g_tot_host.data(), gtot, (size_t)world * sizeof(int64_t), hipMemcpyDeviceToHost, st);
        if (r != ncclSuccess || r2 != ncclSuccess || e != hipSuccess) {
            ctx->set_error("rcn_shard_gather_lists: the totals could not be gathered; communicators aborted");
            shard_abort(sh);
            return RCN_ERR_COMM;
        }
    }
    { int rcw = shard_wait(sh, st, "rcn_shard_gather_lists (totals)"); if (rcw) return rcw; }
    int64_t total = 0;
    for (int r = 0; r < world; ++r) {
        if (sh->g_tot_host[(size_t)r] < 0) {
            if (status) ctx->set_error(why);
            else ctx->set_error("rcn_shard_gather_lists: rank " + std::to_string(r) + " reported a failure; the gather was abandoned on every rank");
            return status ? status : RCN_ERR_COMM;
        }
        total += sh->g_tot_host[(size_t)r];
    }
    if (total_out) *total_out = total;
    // the root's capacity is the root's business: it says so to everybody before any payload moves
    int64_t *fits = gtot;      // reuse word `root` of the totals block as the broadcast word
    {
        sh->words[1] = !is_root || total <= capacity ? 1 : 0;      // (pinned words: the copies below are asynchronous for real)
        sh->words[2] = 0;
        hipError_t e = hipSuccess;
        if (is_root) e = hipMemcpyAsync(fits + root, sh->words.data() + 1, sizeof(int64_t), hipMemcpyHostToDevice, st);
        ncclResult_t r = ncclBroadcast(fits + root, fits + root, 1, ncclInt64, root, sh->comm, st);
        if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(sh->words.data() + 2, fits + root, sizeof(int64_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess || r != ncclSuccess) { ctx->set_error("rcn_shard_gather_lists: broadcast failed; communicators aborted"); shard_abort(sh); return RCN_ERR_COMM; }
        { int rcw = shard_wait(sh, st, "rcn_shard_gather_lists (capacity)"); if (rcw) return rcw; }
        const int64_t got = sh->words[2];
        if (!got) {
            ctx->set_error(is_root ? "rcn_shard_gather_lists: qt_host holds fewer entries than the grid has matches (*total_out says how many)"
                                   : "rcn_shard_gather_lists: the root's buffer is too small; the gather was abandoned on every rank");
            return RCN_ERR_ARG;
        }
    }
    // ---- payload.  Allocation failures from here on cannot be voted on any more without another round trip: the sizes are
    // known now, so they are reserved and voted on in one more tiny all-gather only when something has to grow.
    const int64_t mine = sh->g_tot_host[(size_t)rank];
    int need_grow = 0;
    const int b = ctx->cmp_next;
    if ((size_t)std::max<int64_t>(mine, 1) * sizeof(int2) > ctx->cmp_qt[b].cap) need_grow = 1;
    if (is_root && ((size_t)std::max<int64_t>(total, 1) * sizeof(int2) > sh->g_recv.cap || (size_t)std::max<int64_t>(total, 1) * sizeof(int2) > sh->g_out.cap)) need_grow = 1;
    int32_t gstat = 0;
    if (need_grow) {
        if (ctx->cmp_busy[b]) (void)hipEventSynchronize(ctx->cmp_ev[b]);
        hipError_t e = ctx->cmp_qt[b].reserve((size_t)std::max<int64_t>(mine, 1) * sizeof(int2));
        if (e == hipSuccess && is_root) e = sh->g_recv.reserve((size_t)std::max<int64_t>(total, 1) * sizeof(int2));
        if (e == hipSuccess && is_root) e = sh->g_out.reserve((size_t)std::max<int64_t>(total, 1) * sizeof(int2));
        if (e != hipSuccess) { gstat = RCN_ERR_HIP; why = std::string("rcn_shard_gather_lists: list buffers: ") + hipGetErrorString(e); (void)hipGetLastError(); }
    }
    {   // one word per rank: "I could not get my buffers" (always entered: whether anybody had to grow is not known to the others)
        int32_t *vote = sh->vote.as<int32_t>();
        sh->words[3] = 0;
        *reinterpret_cast<int32_t *>(sh->words.data() + 3) = gstat;
        hipError_t e = hipMemcpyAsync(vote + (size_t)rank * VOTE_WORDS, sh->words.data() + 3, sizeof(int32_t), hipMemcpyHostToDevice, st);
        ncclResult_t r = ncclAllGather(vote + (size_t)rank * VOTE_WORDS, vote, VOTE_WORDS, ncclInt32, sh->comm, st);
        if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(sh->vote_host.data(), vote, (size_t)world * VOTE_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess || r != ncclSuccess) { ctx->set_error("rcn_shard_gather_lists: vote failed; communicators aborted"); shard_abort(sh); return RCN_ERR_COMM; }
        { int rcw = shard_wait(sh, st, "rcn_shard_gather_lists (buffers)"); if (rcw) return rcw; }
        for (int r = 0; r < world; ++r)
            if (sh->vote_host[(size_t)r * VOTE_WORDS] != 0) {
                ctx->set_error(gstat ? why : "rcn_shard_gather_lists: rank " + std::to_string(r) + " could not allocate its list buffers; abandoned on every rank");
                return gstat ? gstat : RCN_ERR_COMM;
            }
    }
    int p2 = RCN_OK;
    bool comm_broken = false;
    std::string p2why;
    auto soft = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !p2) { p2 = RCN_ERR_HIP; p2why = std::string("rcn_shard_gather_lists: ") + what + ": " + hipGetErrorString(e); (void)hipGetLastError(); }
    };
    auto softn = [&](ncclResult_t r, const char *what) {
        if (r != ncclSuccess) { comm_broken = true; if (!p2) { p2 = RCN_ERR_COMM; p2why = std::string("rcn_shard_gather_lists: ") + what + ": " + ncclGetErrorString(r); } }
    };
    int2 *myqt = ctx->cmp_qt[b].as<int2>();
    if (ctx->cmp_busy[b]) soft(hipStreamWaitEvent(st, ctx->cmp_ev[b], 0), "hipStreamWaitEvent");
    if (P > 0 && mine > 0) {
        // k_cmp_fill of store.hip through its host entry would copy to the host; the compaction kernel itself is all that is wanted
        rcn_int_launch_cmp_fill(st, table_dev, stride, counts_dev, off, (int32_t)P, myqt);
        soft(hipGetLastError(), "compaction");
    }
    std::vector<long long> disp((size_t)world + 1, 0);
    for (int r = 0; r < world; ++r) disp[(size_t)r + 1] = disp[(size_t)r] + sh->g_tot_host[(size_t)r];
    int2 *recv = sh->g_recv.as<int2>();
    softn(ncclGroupStart(), "ncclGroupStart");
    if (!is_root) { if (mine > 0) softn(ncclSend(myqt, (size_t)mine * 2, ncclInt32, root, sh->comm, st), "ncclSend"); }
    else
        for (int r = 0; r < world; ++r)
            if (r != root && sh->g_tot_host[(size_t)r] > 0) softn(ncclRecv(recv + disp[(size_t)r], (size_t)sh->g_tot_host[(size_t)r] * 2, ncclInt32, r, sh->comm, st), "ncclRecv");
    softn(ncclGroupEnd(), "ncclGroupEnd");
    if (is_root && !p2) {
        if (mine > 0) soft(hipMemcpyAsync(recv + disp[(size_t)root], myqt, (size_t)mine * sizeof(int2), hipMemcpyDeviceToDevice, st), "own segment");
        long long *loff = sh->g_loff.as<long long>(), *ddisp = loff + (size_t)world * (pmax + 1), *goff = sh->g_goff.as<long long>();
        int32_t *ccnt = sh->g_ccnt.as<int32_t>();
        soft(hipMemcpyAsync(ddisp, disp.data(), (size_t)world * sizeof(long long), hipMemcpyHostToDevice, st), "displacements");
        k_g_scan<<<world, 1024, 0, st>>>(gcnt, pmax, pmax, loff);
        if (Ptot > 0) k_g_canon_counts<<<(unsigned)std::min<long>(4096, (Ptot + 255) / 256), 256, 0, st>>>(gcnt, world, pmax, Ptot, ccnt);
        k_g_scan<<<1, 1024, 0, st>>>(ccnt, Ptot, Ptot, goff);
        if (Ptot > 0 && total > 0) k_g_interleave<<<(unsigned)((Ptot + 3) / 4), 256, 0, st>>>(recv, ddisp, loff, goff, world, pmax, Ptot, sh->g_out.as<int2>());
        soft(hipGetLastError(), "merge kernels");
        static_assert(sizeof(long long) == sizeof(int64_t), "offset type");
        soft(hipMemcpyAsync(offsets_host, goff, ((size_t)Ptot + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st), "offsets to the host");
        if (total > 0) soft(hipMemcpyAsync(qt_host, sh->g_out.p, (size_t)total * sizeof(int2), hipMemcpyDeviceToHost, st), "lists to the host");
    }
    if (comm_broken) { shard_abort(sh); ctx->set_error(p2why + "; communicators aborted"); return p2; }
    { int rcw = shard_wait(sh, st, "rcn_shard_gather_lists (payload)"); if (rcw) return rcw; }      // `disp` is on this frame
    if (p2) { ctx->set_error(p2why); return p2; }
    return RCN_OK;
}

int rcn_shard_set_timeout(rcn_shard *sh, double seconds)
{
    if (!sh || !(seconds > 0.0)) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(sh->ctx->mu);
    sh->timeout_s = seconds;
    return RCN_OK;
}

#ifdef RCN_DIAG
// diagnostic build: the next exchange meets a failure of the named kind -- 1: a HIP error behind the vote, 2: an RCCL call that
// refuses to queue behind the vote, 3: a local failure in front of the vote
int rcn_diag_shard_fault(rcn_shard *sh, int kind)
{
    if (!sh) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(sh->ctx->mu);
    sh->fault = kind;
    return RCN_OK;
}
#endif

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
