// rcn_internal.h -- shared host-side state of the C ABI (include/rcn.h).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rcn.h"
#include "chol_plan.h"

#define RCN_HIP(call)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ctx->set_error(std::string(#call) + ": " + hipGetErrorString(e_));           \
            return RCN_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

// One growable device buffer; never shrinks; no allocation once large enough.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Per-image device record (mirrored in a device table for the kernels).
struct ImgDev {
    const float *f32;      // [K][D]      original rows (exact re-rank reads these)
    const _Float16 *f16;   // [Kp][DP]    scaled by the global power of two, 16-B chunks XOR-swizzled
    const float *hn;       // [Kp]        0.5*s^2*|t|^2 + BIAS ; padded rows = huge
    const double *nrm2;    // [K]         |x|^2 in fp64
    const unsigned long long *bigmin;   // bits of the smallest |x|^2 among this image's BIG rows (+infinity: it has none)
    int32_t K, Kp;
};

struct ImgHost {
    float *f32 = nullptr;
    _Float16 *f16 = nullptr;
    float *hn = nullptr;
    double *nrm2 = nullptr;
    unsigned long long *bigmin = nullptr;
    int32_t K = 0, Kp = 0;
    int32_t slot = -1;  // row in the device image table
    int32_t slab = -1;  // >= 0: buffers are views into ctx->slabs[slab] (batch upload)
    int32_t cap_rows = 0, cap_D = 0;   // capacity of the owned allocations
    bool dirty = true;  // fp16 copy / half-norms not yet built for the current scale
};

// One batch of equally shaped images (rcn_desc_upload_batch_device): one allocation per array.
struct Slab {
    int32_t first_id = 0, n = 0, K = 0, Kp = 0, D = 0;   // n = slots of the block (>= n_images)
    int32_t n_images = 0;                 // ids first_id .. first_id + n_images - 1 live in slots 0 .. n_images - 1
    int32_t conv_first = 0, conv_n = 0;   // slots this ctx converts to fp16 itself (the rest arrives by all-gather)
    const float *f32 = nullptr;   // borrowed from the caller
    const int32_t *Ks_dev = nullptr;   // borrowed: rows in use per slot when the images are ragged (NULL = K everywhere)
    float *own_f32 = nullptr;          // rcn_desc_upload_batch: the block of fp32 rows belongs to the slab (f32 points into it)
    int32_t *own_Ks = nullptr;         // ... and so do the per-slot row counts (Ks_dev points at them)
    _Float16 *f16 = nullptr;
    float *hn = nullptr;
    double *nrm2 = nullptr;
    unsigned long long *bigmin = nullptr;   // [n] one word per slot
    bool live = false;
};

// ctx->counters (words): [0] max |x| bits, [2..3] max |x|^2 bits, [8..15] per-chunk list counts, then the histogram of rows per
// octave of |x|^2 (match.hip, k_rowstats / fix_scale)
#define RCN_HIST_BINS 512
#define RCN_HIST_WORD 64
#define RCN_COUNTER_BYTES ((RCN_HIST_WORD + RCN_HIST_BINS) * 4)

// The global fp16 scale and what follows from it, kept in HBM: fixed either by the host
// (rcn_int_prepare_all, after reading the row statistics) or by k_fix_scale on the device (the sharded exchange,
// which must not wait for the host between its collectives).  Every kernel that needs a constant reads it here.
struct ScaleDev {
    double s;           // power of two
    double s2;          // s^2
    double hs2;         // s^2 / 2
    double bias;        // BIAS, accumulator units
    double c_in;        // (2u+u^2) s^2            x |q| Nmax
    double c_sub;       // 2^-14 sqrt(DP) s        x (|q| + Nmax)
    double c_acc;       // (DP+8) 2^-23
    double hn_max;      // s^2 Nmax^2 / 2 + BIAS
    double n_max;       // Nmax
    double rel_slack;   // relative slack for the fp64 evaluation of the bound itself
    double thr2;        // rows with |x|^2 >= thr2 are BIG rows (fix_scale, match.hip); +infinity: none
    float  sf;          // (float)s
    float  pad;
};

// fused table filter (fmat.hip): per-pair keypoint coordinate lists
struct PairXY { const int32_t *q, *t; int32_t Kq, pad; };

struct rcn_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err = "";
    hipDeviceProp_t prop;

    // ---- matcher state
    int32_t D = 0, DP = 0;
    std::map<int32_t, ImgHost> images;
    std::vector<Slab> slabs;
    std::vector<int2> groups_host, groups_arranged;   // kept alive: uploaded asynchronously
    std::vector<int32_t> slots_host;                  // pair list as table slots (same reason)
    std::vector<int32_t> all_pairs_host;              // the canonical i < j grid when the caller passes pairs == NULL
    bool prepared = false;
    double scale = 0.0;      // s, power of two (0 = nothing prepared yet)
    double bias = 0.0;       // BIAS in accumulator units
    double max_norm = 0.0;   // max |x| over resident rows
    long hist_rows = 0;      // rows counted into the norm histogram since it was last cleared / rebuilt
    double thr2 = 0.0;       // BIG-row threshold in force (part of the scale: a change reconverts everything)
    DevBuf scale_dev;        // one ScaleDev: what the kernels read
    DevBuf desc_bad;         // desc.hip: keypoints outside their descriptor map since the last rcn_desc_sample_errors
    ScaleDev scale_host;     // staging of the host-fixed scale (uploaded asynchronously)
    bool scale_on_device = false;   // the last scale was fixed by k_fix_scale: scale / bias / max_norm above are stale until resolved
    bool want_dev_scale = false;    // shard.hip: fix the next scale on the device (no host read of the statistics)
    std::vector<ImgDev> table_host; // image table as last uploaded (unchanged tables are not uploaded again)
    DevBuf img_table, pairs_dev, groups_dev, cand, owner, fb_list, sv_list, counters, out_tmp, cnt_tmp;
    hipEvent_t f32_ready = nullptr;   // shard.hip: set while an all-gather of fp32 rows may be in flight on a side stream
    // host materialisation (store.hip): offsets, two alternating staging buffers, copy stream
    DevBuf cmp_off, cmp_qt[2];
    hipStream_t copy_stream = nullptr;
    hipEvent_t cmp_ev[2] = {nullptr, nullptr}, cmp_filled = nullptr;
    bool cmp_busy[2] = {false, false};
    int cmp_next = 0, cmp_last = -1;
    int32_t last_kq_stride = 0, last_n_pairs = 0;   // shape of the candidate table of the last grid call
    uint32_t last_idx_mask = 0;
    rcn_match_stats last_stats;
    bool profile = false;
#define RCN_EV_CHUNKS 64
    hipEvent_t ev_c[64][RCN_EV_CHUNKS][4];   // per kept call and pipeline chunk: before coarse / after coarse / after the exact stages / after uniqueness
    hipEvent_t ev_tail[64][3];               // per kept call: before the deferred pass of the middle tier / behind it / behind the uniqueness of all pairs
    bool ev_tail_on[64] = {false};
    int ev_chunks[64] = {0};                 // chunks of that call that carry events (further chunks are not timed)
    int last_chunks = 1;                     // pipeline chunks of the last grid call
    int64_t chunk_rows = 1ll << 27;          // query-row slots of the candidate table per pipeline chunk (diagnostic build: RCN_CHUNK_ROWS)
    int64_t mid_rows = 1ll << 21;            // rows per chunk the middle tier takes (diagnostic build: RCN_MID_ROWS)
    DevBuf mid_ws;                           // middle tier of the exact stages (match.hip): binned rows, thresholds, candidate lists, bins
    int mid_bins_slots = -1;                 // image slots the zeroed bins of mid_ws were laid out for
    bool ev_made = false;
    int ev_n = 0;          // recorded calls since the last stats read (<= 64)
    // Ablations / alternative device paths exist only in the diagnostic build (-DRCN_DIAG,
    // tools/librcn_diag.so), where rcn_create reads them from the environment; in the shipping
    // library they are compile-time constants and the alternative code is dead.
#ifdef RCN_DIAG
    bool coarse_w4 = false;    // RCN_COARSE_W4=1: the one-wave-per-SIMD form of K1 (k_coarse_w4, coarse_w4.h) instead of k_coarse_top2
    int coarse_shape = -1;     // RCN_COARSE_S16=0/1: force k_coarse_top2's MFMA shape (0: 32x32x16, 1: 16x16x32) at every D; -1: the shipping choice
    int ablate = 0;            // RCN_COARSE_ABL
    bool force_exact = false;  // RCN_FORCE_EXACT=1: skip the MFMA coarse pass
    bool no_item_order = false;   // RCN_MATCH_NO_ORDER=1: work items in pair order instead of heaviest-first per XCD
    bool ba_atomics = false;   // RCN_BA_SCHUR_ATOMICS=1: atomic Schur accumulation instead of the gather form
    bool ba_trsv_fwd = false;  // RCN_BA_TRSV_FWD=1: separate forward substitution instead of the rhs row inside the factorisation
    bool ba_pair_small = true; // RCN_PAIR_SMALL=0: the pair lists of the smallest graphs by the six general launches too
#else
    static constexpr bool coarse_w4 = false;
    static constexpr int coarse_shape = -1;
    static constexpr int ablate = 0;
    static constexpr bool force_exact = false, no_item_order = false, ba_atomics = false, ba_trsv_fwd = false, ba_pair_small = true;
#endif

    // ---- BA state (ba.hip)
    DevBuf ba_ws[40];
    DevBuf lm_ws;              // landmark validity sweep (validity.hip)
    DevBuf fm_ws, fm_state;    // epipolar filter (fmat.hip): host-API staging, per-pair RANSAC state
    std::vector<PairXY> fm_pairs_host;   // staging of fm_pairs (uploaded asynchronously)
    DevBuf fm_csr, fm_pairs;   // fused table filter: CSR of the matched points, per-pair coordinate pointers
    std::map<int32_t, std::pair<DevBuf, int32_t>> coords;   // image id -> (K x 2 int32 pixel coordinates in HBM, K)
    std::vector<int> ba_graph_cam, ba_graph_pt;     // observation graph of the last plain rcn_ba_solve (host copy): an identical graph reuses the pair lists
    int ba_graph_nc = 0, ba_graph_np = 0;
    uint64_t ba_graph_serial = 0;
    uint64_t ba_pair_token = 0;         // whose pair lists the Schur-build workspace holds (0 = nobody's)
    std::vector<int> ba_pair_camdim;    // ... and the per-camera tangent dimensions they were built for (a camera without free parameters has no pairs)
    hipStream_t aux_stream = nullptr;   // lookahead stream of the Cholesky: bulk trailing updates (CU mask leaves one CU per XCD to the diagonal kernel)
    hipStream_t chain_stream = nullptr;  // diagnostic build (RCN_CHOL_CHAIN_STREAM=1): the factorisation's chain on a highest-priority stream of the library's own
    int chol_chain_stream = 0;
    hipStream_t diag_stream = nullptr;   // the resident workgroup that factors the diagonal blocks of a factorisation (k_chol_diag_server)
    hipStream_t panel2_stream = nullptr; // two-level regime of the Cholesky, plans with pg_stream only (not what ships): the panel product for the rows below the head (same CU mask); made on demand
    std::vector<uint32_t> bulk_cu_mask;  // the bulk stream's CU mask
    hipStream_t panel_stream = nullptr; // second chain stream of the Cholesky: panels and first trailing columns behind the critical tile (same CU mask)
    // the factorisation's schedule (chol_plan.h: operations, streams, waits, tile maps) for the last shape solved; its maps in HBM
    DevBuf bulk_map;
    chol::Plan chol_plan;
    DevBuf diag_items;                               // the resident diagonal workgroup's work list of that plan
    bool chol_plan_valid = false;
    int chol_tl_g = 4;                               // two-level regime: panels per super-step (K = 128 g per bulk update); 0 = right-looking steps only (diagnostic build: RCN_CHOL_TL)
    int chol_diag_server = 0;                        // diagnostic build (RCN_CHOL_DIAG_SERVER=1): the diagonal blocks in one resident workgroup instead of a launch per block on the chain's stream
    int chol_bulk_behind = 0;                        // right-looking regime: a bulk update starts when the next diagonal block's kernel has started (chol_plan.h; RCN_CHOL_BULK_BEHIND=1 in the diagnostic build: measured, not shipped)
    int chol_carve_rows = 0;                         // right-looking regime: the chain's next tiles as latency-kernel operations of their own from this many rows on (chol_plan.h; RCN_CHOL_CARVE)
    int chol_window = 0;                             // two-level regime: 2 g-row window of the chain's latency kernels (RCN_CHOL_WINDOW)
    int chol_tl_serial = 0;                          // two-level regime: super-blocks with fewer tile rows below them run their small operations on the chain's stream (RCN_CHOL_TL_SERIAL)
    int chol_head_small = 1;                         // two-level regime: head rows' product + next super-diagonal block's update through k_gemm_qm (RCN_CHOL_HEAD_SMALL)
    int chol_fuse_tail = 1;                          // two-level regime: the panel product below the head rows as the tail of the previous bulk launch (RCN_CHOL_FUSE_TAIL)
    int chol_pg_stream = 0;                          // two-level regime: the panel product below the head rows on a stream of its own (RCN_CHOL_PGSTREAM)
    bool chol_pg_prio = true;                        // diagnostic build (RCN_CHOL_PG_PRIO=0): no raised wave priority for the panel product below the head rows
    int chol_gate_in_kernel = 0;                     // diagnostic build (RCN_CHOL_GATE_IN_KERNEL=1): waits of the small kernels off the chain inside them, not in a gate kernel in front; -1: in front on the chain too
    bool chol_host_time = false;                     // diagnostic build (RCN_CHOL_HOSTTIME=1): print the host time of every factorisation's enqueue
    int chol_tl_min = 40;                            // ... while at least this many tile rows remain below the super-block (RCN_CHOL_TL_MIN)
    int chol_group = 2;                              // right-looking regime: panels per bulk update while many tile rows remain, 2 (K = 256) or 1
    int chol_pipe_min = 32;                          // panel / column kernels go through the pipelined kernel from this many tiles on
    bool trsv_chain = true;                          // backward substitution as one launch (k_trsv_bwd_chain); off after a flag timeout
    int chol_break = 0;                              // diagnostic build: 1 = break one cross-stream hand-off (forces the one-stream fallback); 2 = and put a NaN pivot behind it
    int chol_pair_min = 24;                          // two-panel bulk updates while at least this many tile rows remain below the pair
    bool chol_safe = false;             // a device-counter hand-off timed out once: factorise on one stream, in plain order, from then on
    double *ba_host_scal = nullptr;   // pinned mirror of the LM step's scalars (16 doubles; [15]: sequence number), written by the step's last kernel
    unsigned long long ba_host_seq = 0;
    hipEvent_t ba_ev[9];             // [0]: fork of the factorisation's streams, [1..4]: their joins, [5]: the chain's own stream back to the caller's, [7]: pair lists
    hipEvent_t ba_tev[6];            // phase timing of rcn_ba_solve ([4], [5]: around k_ba_eval<true>)
    bool ba_ev_made = false;

    // Host waits inside rcn_int_prepare_all.  A sharded exchange (shard.hip) installs a BOUNDED wait for the duration of
    // its call -- the ctx stream then carries collectives, and a peer that died would otherwise keep this host thread in
    // hipStreamSynchronize for ever; NULL: plain hipStreamSynchronize.
    int (*wait_hook)(void *arg, hipStream_t st) = nullptr;
    void *wait_arg = nullptr;
    // match tables owned by the ctx (out_tmp / cnt_tmp): bumped by every entry point that writes them, so that a shard can tell
    // whether the tables its rcn_shard_match left there are still the ones it is about to filter / compact
    uint64_t own_table_gen = 0;

    void set_error(const std::string &s) { err = s; }
};
// RCN_HIP-compatible wait for everything queued on ctx->stream (the hook, or hipStreamSynchronize)
inline hipError_t rcn_int_stream_wait(rcn_ctx *ctx)
{
    if (ctx->wait_hook) return ctx->wait_hook(ctx->wait_arg, ctx->stream) == RCN_OK ? hipSuccess : hipErrorUnknown;
    return hipStreamSynchronize(ctx->stream);
}

// Device-resident part of a bundle-adjustment problem (rcn_ba_session, ba_session.hip -> rcn_int_ba_solve, ba.hip)
struct BaResident {
    double *pts;                  // n_points x 3, in and out
    const double *uv;             // n_obs x 2
    const int *ocam, *opt;        // n_obs each, landmark-major
    uint64_t pair_token;          // identifies (session, graph version): pair lists built under the same token are reused
};
int rcn_int_ba_solve(rcn_ctx *ctx, const rcn_ba_problem *pb, const rcn_ba_options *opt, rcn_ba_summary *sum, const BaResident *res);
int rcn_match_release(rcn_ctx *ctx);
// store.hip: rcn_match_compact_begin / _wait with ctx->mu already held
int rcn_int_compact_begin(rcn_ctx *ctx, const int32_t *table_dev, int64_t stride, const int32_t *counts_dev,
                          int32_t n_pairs, int64_t *offsets_host, int32_t *qt_host, int64_t capacity, int64_t *total_out);
int rcn_int_compact_wait(rcn_ctx *ctx);
// store.hip: the ordered compaction kernel alone (rcn_shard_gather_lists: the lists stay in HBM)
void rcn_int_launch_cmp_fill(hipStream_t st, const int32_t *table_dev, int64_t stride, const int32_t *counts_dev, const long long *off, int32_t n_pairs, int2 *qt);
// fmat.hip: rcn_match_table_filter_device with ctx->mu already held
int rcn_int_table_filter(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, int32_t *table_dev,
                         int64_t stride, int32_t *counts_dev, int32_t *out_status_dev);
// match.hip internals shared with shard.hip (all expect ctx->mu held)
int rcn_int_slab_attach(rcn_ctx *ctx, int32_t first_id, int32_t n_images, int32_t n_slots, const float *src,
                        int32_t K, int32_t D, int32_t conv_first, int32_t conv_n, int *slab_out,
                        const int32_t *Ks_host, const int32_t *Ks_dev);
int rcn_int_slab_rowstats(rcn_ctx *ctx, int slab, int32_t first, int32_t n);
int rcn_int_prepare_all(rcn_ctx *ctx);
int rcn_int_resolve_scale(rcn_ctx *ctx);     // host copies of scale / bias / max_norm are current afterwards (may synchronise)
int rcn_int_match_grid(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, float ratio,
                       int32_t *out_dev, int64_t out_stride, int32_t *counts_dev);
