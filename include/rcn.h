/*
 * rcn.h -- C ABI of the MI355X (gfx950) matching + bundle-adjustment core.
 *
 * Drop-in boundary for ONE hot path of smileyenot983/reconstructor:
 *   - all-pairs descriptor matching  (replaces FlannMatcher::matchFeatures' call into
 *     cv::DescriptorMatcher::knnMatch, FeatureMatcher.cpp:32-65, and the pair loop of
 *     SequentialReconstructor::matchFeatures, SequentialReconstructor.cpp:199-279)
 *   - bundle adjustment              (replaces BundleAdjuster::adjust's call into
 *     ceres::Solve, BundleAdjuster.cpp:72-146)
 *
 * Plain pointers and sizes only; no C++ or torch types; never throws.  Every entry point
 * returns an int status (RCN_OK == 0, negative = error; rcn_last_error() has the text).
 * Host arrays are borrowed for the duration of the call; device buffers belong to the ctx.
 * One ctx per GPU.  A ctx is safe to call from several host threads (internal mutex): the
 * reference calls its matcher from 4 OpenMP threads (SequentialReconstructor.cpp:202).
 *
 * There is NO CPU fallback behind this ABI: without a usable HIP device rcn_create fails.
 */
#ifndef RCN_H
#define RCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCN_OK               0
#define RCN_ERR_ARG         -1   /* bad argument (null pointer, negative size, mismatched D, ...) */
#define RCN_ERR_HIP         -2   /* HIP runtime error; text in rcn_last_error */
#define RCN_ERR_NO_DEVICE   -3   /* no gfx950 device visible */
#define RCN_ERR_UNSUPPORTED -4   /* shape outside what the kernels cover */
#define RCN_ERR_NOT_FOUND   -5   /* image id not resident */
#define RCN_ERR_NUMERIC     -6   /* BA: non-finite cost / Cholesky breakdown that LM could not recover */
#define RCN_ERR_COMM        -7   /* RCCL error (sharded grid); text in rcn_last_error */
#define RCN_ERR_IO          -8   /* store file: cannot open / short read / bad magic, version or checksum */

typedef struct rcn_ctx rcn_ctx;

/* ---- context ------------------------------------------------------------------------- */
int         rcn_device_count(void);                  /* HIP devices visible to this process (0 without a GPU) */
int         rcn_create(int device_id, rcn_ctx **out);
void        rcn_destroy(rcn_ctx *ctx);
const char *rcn_last_error(const rcn_ctx *ctx);      /* never NULL */
/* ABI revision of this header.  Bumped whenever a struct the library writes through a caller's pointer grows or an entry point changes
 * its arguments: a caller built against an older header must not be linked against a newer library (rcn_match_last_stats copies the whole
 * rcn_match_stats; revision 2 -> 3 added rows_brute_force, chunks, coarse_launches to it and rcn_ba_factor_plan to the library).
 * rcn_version() names the library build ("reconstructor_amd 0.<revision> (gfx950)"); compare the two at start-up. */
#define RCN_ABI_REVISION 3
const char *rcn_version(void);
/* Run all work of this ctx on an existing HIP stream (e.g. torch's current stream, passed as
 * the raw hipStream_t).  NULL = the ctx's own stream.  */
int         rcn_set_stream(rcn_ctx *ctx, void *hip_stream);
int         rcn_synchronize(rcn_ctx *ctx);

/* ---- descriptors -----------------------------------------------------------------------
 * One dense row-major K x D fp32 matrix per image: exactly what featDescToCV builds per call
 * (FeatureMatcher.cpp:11-25), built once per image here instead of once per pair.
 * All resident images must share D.  Re-uploading an id replaces it.  K may be 0. */
int rcn_desc_upload(rcn_ctx *ctx, int32_t img_id, const float *desc_host, int32_t K, int32_t D);
/* Same, from a DEVICE pointer (e.g. the landing buffer of an RCCL all-gather); copied. */
int rcn_desc_upload_device(rcn_ctx *ctx, int32_t img_id, const float *desc_dev, int32_t K, int32_t D);
/* n_images equally shaped images [n][K][D] in one DEVICE buffer, ids first_img_id.. ; the
 * buffer is BORROWED (zero copy) until rcn_desc_clear / re-upload of those ids: the caller
 * keeps it alive and unchanged.  One stats launch + one conversion launch for the whole
 * batch; calling it again with the same shape reuses every allocation (per-step ingest of
 * an all-gather landing buffer).  When D % 4 == 0 the block must be 16-byte aligned
 * (RCN_ERR_ARG otherwise; hipMalloc / torch allocations are). */
int rcn_desc_upload_batch_device(rcn_ctx *ctx, int32_t first_img_id, int32_t n_images,
                                 const float *desc_dev, int32_t K, int32_t D);
/* Producer contract.  The layout above -- [n][K][D] fp32 row-major in HBM, D = 256 unit-norm rows for SuperPoint
 * (FeatureSuperPoint.cpp:183-211), 128 for SIFT, 32 for ORB-as-float -- is what a detector running on the GPU
 * writes straight into: either its own buffer handed to rcn_desc_upload_batch_device (borrowed, zero copy), or
 * the slot rcn_shard_reserve returns.  Rows past an image's keypoint count must be zero (rcn_shard_exchange's
 * local_K) or the images are uploaded one by one.  rcn_desc_sample_device is the last step of such a detector:
 * processDescriptors of the reference (cell = keypoint / 8 in integers, the first D <= 256 channels of that cell
 * of the dense descriptor map, divided by FeatDesc::norm(): fp32 squares summed in fp64 in ascending order),
 * bit-exact, writing K rows of D floats to out_rows_dev.  The map is addressed by element strides: the network's
 * own [C][Hc][Wc] output is (Hc*Wc, Wc, 1), a channel-last copy is (1, Wc*C, C).  Asynchronous on the ctx stream. */
int rcn_desc_sample_device(rcn_ctx *ctx, const float *desc_map_dev, int64_t stride_c, int64_t stride_y, int64_t stride_x,
                           int32_t Hc, int32_t Wc, const int32_t *kp_xy_dev, int32_t K, int32_t D, float *out_rows_dev);
/* A keypoint outside [0, 8*Wc) x [0, 8*Hc) reads nothing (the reference's tensor indexing throws there): its row is
 * written as zeros and counted.  rcn_desc_sample_errors waits for the ctx stream and returns RCN_ERR_ARG (count in
 * *n_out_of_range, may be NULL) when any keypoint of the calls since the last read was outside; the count is cleared. */
int rcn_desc_sample_errors(rcn_ctx *ctx, int32_t *n_out_of_range);
/* Host-side batch ingest: n_images images with their own row counts K[i] >= 0, each a dense row-major K[i] x D
 * fp32 matrix in HOST memory (rows[i]; what featDescToCV packs per call, FeatureMatcher.cpp:11-25 -- here for every
 * image of the loop at once), become ids first_img_id .. first_img_id + n_images - 1.  One device block of
 * [n][max K][D] owned by the ctx (tails zeroed), one asynchronous copy per image straight out of the caller's rows
 * (pinned rows -- rcn_host_alloc -- travel at the link rate; pageable rows are staged by the runtime), ONE stats
 * launch, ONE conversion launch at the next grid call, ONE host synchronisation before the call returns: the host
 * rows are borrowed for the call only.  Calling it again with the same (first id, n, max K, D) reuses every
 * allocation.  Replaces n synchronous rcn_desc_upload calls (each of which waits for its own copy). */
int rcn_desc_upload_batch(rcn_ctx *ctx, int32_t first_img_id, int32_t n_images, const float *const *rows_host,
                          const int32_t *K, int32_t D);
/* Forget one image (no-op when the id is not resident).  Other resident images, and their D, stay. */
int rcn_desc_remove(rcn_ctx *ctx, int32_t img_id);
int rcn_desc_clear(rcn_ctx *ctx);
int rcn_desc_count(const rcn_ctx *ctx);

/* ---- matching --------------------------------------------------------------------------
 * Result of one (query image, train image) pair: out[i] = train row matched to query row i,
 * or -1.  This is the std::map<int,int> FlannMatcher::matchFeatures fills
 * (FeatureMatcher.cpp:53-64) in dense form: exact 2-NN under L2, ratio test
 * dist0 < ratio * dist1 in fp32, then the lowest query index keeps a contested train row.
 * K2 < 2 yields no matches (the reference reads knn[i][1] unconditionally there).          */

/* One pair straight from host rows (the per-call shape of FeatureMatcher::matchFeatures). */
int rcn_match_pair(rcn_ctx *ctx, const float *q_host, int32_t K1,
                   const float *t_host, int32_t K2, int32_t D, float ratio,
                   int32_t *out_train_for_query /* K1 */, int32_t *out_count);

/* Pair grid over resident images (SequentialReconstructor.cpp:199-279).  pairs = n_pairs x
 * (query image id, train image id) on the HOST.  out = n_pairs rows of out_stride int32 on
 * the HOST (out_stride >= K of every query image; tail of each row is set to -1),
 * counts[p] = matches of pair p.  pairs_host == NULL: the reference's canonical grid, every
 * i < j over the resident image ids in ascending order; n_pairs must then be n (n - 1) / 2.  */
int rcn_match_grid(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, float ratio,
                   int32_t *out_host, int64_t out_stride, int32_t *counts_host);

/* Same, results left in HBM: out_dev / counts_dev are DEVICE pointers; asynchronous on the
 * ctx stream (no host synchronisation inside once the workspace has reached its size).     */
int rcn_match_grid_device(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, float ratio,
                          int32_t *out_dev, int64_t out_stride, int32_t *counts_dev);

/* ---- host materialisation of a match table ------------------------------------------------
 * The (query feature, train feature) lists the pair loop keeps in featureMatches
 * (SequentialReconstructor.cpp:260-267, :272-275), for a table rcn_match_grid_device (or
 * rcn_shard_match, rcn_match_table_filter_device) left in HBM: compacted on the GPU, then copied to
 * host memory.  Pair p owns entries offsets[p] .. offsets[p+1]-1 of qt, each two int32 (query row,
 * train row), ascending query row -- the iteration order of the reference's std::map<int,int>.
 *
 * rcn_match_compact_begin: compacts on the ctx stream, waits until the device knows the total,
 * fills offsets_host (n_pairs + 1 entries) and *total_out, starts the copy of the lists into qt_host
 * on the ctx's copy stream and returns; the copy overlaps whatever is enqueued on the ctx stream
 * next (two device staging buffers alternate).  qt_host should be pinned (rcn_host_alloc) for a true
 * DMA; capacity = entries qt_host can hold (RCN_ERR_ARG with *total_out set when it is too small).
 * rcn_match_compact_wait: the lists of the last begin have landed.                              */
int  rcn_host_alloc(void **out, size_t bytes);      /* pinned host memory; usable without a ctx once a device exists */
void rcn_host_free(void *p);
int  rcn_match_compact_begin(rcn_ctx *ctx, const int32_t *table_dev, int64_t stride, const int32_t *counts_dev,
                             int32_t n_pairs, int64_t *offsets_host, int32_t *qt_host, int64_t capacity,
                             int64_t *total_out);
int  rcn_match_compact_wait(rcn_ctx *ctx);

/* Statistics of the last grid call (diagnostics; rows_total = sum of K1 over pairs). */
typedef struct {
    int64_t rows_total;
    int64_t rows_reranked;        /* query rows whose two candidates were re-computed exactly (fp64) */
    int64_t rows_exact_fallback;  /* query rows the coarse pass and the re-rank could not certify: middle tier (fp32 sweep shared by the rows
                                     of a train image -> a handful of candidates -> fp64 chain), K2b (every train row in fp64) behind it */
    int64_t pair_distances;       /* sum of K1*K2 */
    double  err_bound_d2;         /* largest certified bound on |coarse - exact| squared distance */
    int32_t used_mfma_path;       /* 1 = fp16 MFMA coarse pass + exact re-rank, 0 = exact kernel only */
    int32_t profiled_calls;       /* grid calls summed into the *_ms fields (rcn_match_profile) */
    double  coarse_ms;            /* HIP-event time of k_coarse_top2 launches, summed */
    double  rerank_ms;            /* k_rerank + k_exact_rows */
    double  unique_ms;            /* k_unique_claim + k_unique_emit */
    int64_t rows_brute_force;     /* of rows_exact_fallback: rows that went through K2b after all (candidate list overflowed, budget exceeded, D % 4 != 0) */
    int32_t chunks;               /* pipeline chunks of the last grid call (candidate table / row lists are sized per chunk) */
    int32_t coarse_launches;      /* coarse-kernel launches summed into coarse_ms (chunks x profiled calls) */
} rcn_match_stats;
int rcn_match_last_stats(const rcn_ctx *ctx, rcn_match_stats *out);
/* enable != 0: bracket the kernels of every following grid call with HIP events on the ctx
 * stream (up to 64 calls are kept); rcn_match_last_stats sums and clears them. */
int rcn_match_profile(rcn_ctx *ctx, int enable);
/* Workspace budget of the grid calls: query-row slots of the candidate table per pipeline chunk (8 bytes per slot, plus two row lists of
 * at most as many entries).  Default (rows = 0): 2^27 slots = 1 GiB of candidates -- cfg 3 (2.05e9 query rows) then runs in sixteen chunks
 * that reuse the workspace, at +0.2 % of K1's time against one 16-GiB chunk (same-box A/B in the bench line, `roofline.chunk_ab`). */
int rcn_match_set_workspace_rows(rcn_ctx *ctx, int64_t rows);

/* ---- pair grid sharded over the GPUs of one node -----------------------------------------
 * The N x N loop of SequentialReconstructor::matchFeatures (SequentialReconstructor.cpp:202-279) runs
 * its pairs as independent units (OpenMP collapse(2)); here they are spread over several GPUs, one
 * rcn_shard (= one rcn_ctx + RCCL communicators, called directly over xGMI) per GPU, in one process
 * per GPU or in one process with a host thread per GPU (reconstructor_amd/host/HipPairGridDriver.h).
 *   images  equal contiguous blocks: rank r owns ids [r*per, min(n, (r+1)*per)), per = ceil(n / world)
 *   pairs   the canonical i < j list (row-major) dealt round-robin: pair number p belongs to rank p % world
 *   exchange  local row statistics -> ncclAllReduce(max) of the two scale statistics -> fp16 conversion of
 *             the LOCAL block -> in-place ncclAllGather of fp16 rows + half-norms + norms (ctx stream);
 *             ncclAllGather of the fp32 rows on a side stream (read only by the exact re-rank stages)
 * Every rank converts with the same global scale: tables are bit-identical to a one-GPU run.     */
typedef struct rcn_shard rcn_shard;
#define RCN_SHARD_ID_BYTES 128            /* sizeof(ncclUniqueId) */

/* Partition: pure host functions (no GPU, no communicator). */
int     rcn_shard_owned_images(int32_t n_images, int32_t world, int32_t rank, int32_t *first, int32_t *count);
int64_t rcn_shard_pair_count(int32_t n_images, int32_t world, int32_t rank);
int     rcn_shard_pairs(int32_t n_images, int32_t world, int32_t rank, int32_t *pairs_out /* 2 x count */);

/* Rendezvous: ONE rank (or the single process) draws the id, the host hands it to every rank
 * (environment, file, MPI, torch store ...); rcn_shard_create is collective over the world. */
int      rcn_shard_unique_id(uint8_t id[RCN_SHARD_ID_BYTES]);
int      rcn_shard_create(rcn_ctx *ctx, int32_t rank, int32_t world, const uint8_t id[RCN_SHARD_ID_BYTES],
                          rcn_shard **out);
void     rcn_shard_destroy(rcn_shard *sh);   /* also drops the ctx's resident descriptors (views into the landing buffer) */
rcn_ctx *rcn_shard_ctx(rcn_shard *sh);

/* Shape of the next exchanges: n_images over all ranks, each K x D fp32.  *local_slot_dev (may be NULL)
 * receives this rank's block of the landing buffer, [count][K][D] fp32 in HBM: a detector can write its
 * rows straight there (the producer contract, see rcn_desc_upload_batch_device) and pass NULL to
 * rcn_shard_exchange.  The pointer stays valid until a reserve with another shape. */
int rcn_shard_reserve(rcn_shard *sh, int32_t n_images, int32_t K, int32_t D, float **local_slot_dev);
/* Host rows of ONE owned image into its slot (featDescToCV's gather, FeatureMatcher.cpp:11-25, done once per
 * image): K_img <= K rows are copied, the rest of the slot is zero-filled, K_img is remembered for the exchange. */
int rcn_shard_put_image(rcn_shard *sh, int32_t img_id, const float *desc_host, int32_t K_img);
/* Collective.  local_desc_dev: this rank's [count][K][D] fp32 block in HBM (copied), or NULL when the rows are
 * already in the slot.  local_K: rows in use per owned image (count entries, each <= K; tails are zeroed), or
 * NULL = what rcn_shard_put_image recorded, K for untouched slots.  Afterwards ids 0 .. n_images-1 are resident
 * in the shard's ctx, image i with its own row count.  One host wait when every slot of every rank is full, two
 * when some rank's images are ragged (the row counts are then gathered too). */
int rcn_shard_exchange(rcn_shard *sh, const float *local_desc_dev, const int32_t *local_K);
/* This rank's share of the canonical grid (rcn_shard_pairs order); out_dev / counts_dev as rcn_match_grid_device
 * (out_stride >= K).  Both NULL: the tables stay in buffers owned by the ctx, for rcn_shard_lists. */
int rcn_shard_match(rcn_shard *sh, float ratio, int32_t *out_dev, int64_t out_stride, int32_t *counts_dev);
/* Host lists of the last rcn_shard_match(sh, ratio, NULL, 0, NULL): rcn_match_compact_begin + _wait on the shard's
 * own tables (same arguments and the same behaviour when capacity is too small: *total_out tells how many). */
int rcn_shard_lists(rcn_shard *sh, int64_t *offsets_host, int32_t *qt_host, int64_t capacity, int64_t *total_out);

/* The lists of EVERY rank on one rank: the single featureMatches map the reference keeps
 * (SequentialReconstructor.cpp:224,264,274).  Collective.  table_dev / stride / counts_dev: this rank's tables of the last
 * rcn_shard_match (all three NULL / 0: the shard's own tables, after rcn_shard_match(sh, ratio, NULL, 0, NULL), filtered
 * or not).  Every rank compacts its tables on the GPU; the totals and the per-pair counts are all-gathered (one bounded host
 * wait); the lists travel to `root` device to device (ncclSend / ncclRecv over xGMI), are put in canonical pair order on
 * the root's GPU (pair number p of the row-major i < j list: offsets_host[p] .. offsets_host[p + 1]) and reach the root's
 * host memory in ONE copy.  offsets_host (n(n-1)/2 + 1 entries), qt_host, capacity and *total_out are read on the root only
 * (other ranks may pass NULL / 0); a capacity that is too small fails on every rank together with *total_out set on the
 * root.  rcn_shard_merge_lists is the same merge for lists that are already in host memory (per-rank results of
 * rcn_shard_lists, gathered by whatever the host has): pure host code, no GPU, no communicator. */
int rcn_shard_gather_lists(rcn_shard *sh, int32_t root, const int32_t *table_dev, int64_t stride, const int32_t *counts_dev,
                           int64_t *offsets_host, int32_t *qt_host, int64_t capacity, int64_t *total_out);
int rcn_shard_merge_lists(int32_t n_images, int32_t world, const int64_t *const *offsets_per_rank, const int32_t *const *qt_per_rank,
                          int64_t *offsets_out, int32_t *qt_out, int64_t capacity, int64_t *total_out);

typedef struct {
    int32_t rank, world, n_images, images_per_rank;
    int64_t n_pairs;              /* this rank's share */
    int64_t exchange_bytes_f16;   /* whole all-gather (all ranks' blocks): fp16 rows + half-norms + norms */
    int64_t exchange_bytes_f32;   /* whole all-gather of the fp32 rows (side stream) */
    int32_t comm_ranks;           /* ncclCommCount of the communicator the collectives run on */
    int32_t reserved;
} rcn_shard_stats;
int rcn_shard_info(const rcn_shard *sh, rcn_shard_stats *out);

/* Failure handling.  The collectives after rcn_shard_create are rcn_shard_exchange and rcn_shard_gather_lists, and both open
 * with a status vote: an all-gather of a few words per rank (status, shape, "my block is ragged") in front of the first
 * host wait.  Every allocation of the call happens BEFORE the vote.  A failure that only THIS rank saw -- rcn_shard_reserve
 * could not allocate, rcn_shard_put_image was refused, or the host driver reports one of its own through rcn_shard_fail --
 * is remembered in the shard; the rank must still call rcn_shard_exchange, which then returns an error on EVERY rank (the
 * failing rank its own code, the others RCN_ERR_COMM naming it) before any further collective is entered, so no peer is
 * left waiting.  Ranks that reserved different shapes fail the same way.  rcn_shard_match refuses to run until an exchange
 * has gone through again.
 *   Behind the vote nothing allocates and nothing returns early: a HIP error there is recorded, the remaining collectives
 * are still entered, the call returns the error to this caller and the peers hear of it in the NEXT vote.  An RCCL call that
 * refuses to queue leaves the peers without a partner; the communicators are then aborted (ncclCommAbort).
 *   The host never waits for a collective without a limit: every wait of the shard API polls, watches
 * ncclCommGetAsyncError and gives up after rcn_shard_set_timeout seconds (default 600) -- it then aborts both communicators,
 * and the shard is dead: every later call returns RCN_ERR_COMM at once (destroy it and create a new one).  A peer that
 * crashed or walked away therefore costs a timeout, never a hang. */
int rcn_shard_fail(rcn_shard *sh, int32_t code /* negative RCN_ERR_* */);
int rcn_shard_set_timeout(rcn_shard *sh, double seconds);

/* Phase times of the sharded step, HIP events on the streams the work runs on: enable, run steps (at most 64 are
 * kept), read the sums (waits for the shard's streams; clears them). */
typedef struct {
    int32_t exchanges, matches;   /* calls summed below */
    double  exchange_ms;          /* rcn_shard_exchange on the ctx stream: vote + counts, statistics, all-reduce, fp16 conversion, fp16 all-gather */
    double  f32_gather_ms;        /* all-gather of the fp32 rows on the side stream (beside the coarse kernel) */
    double  match_ms;             /* rcn_shard_match: this rank's share of the grid */
} rcn_shard_times;
int rcn_shard_profile(rcn_shard *sh, int enable);
int rcn_shard_profile_read(rcn_shard *sh, rcn_shard_times *out);

/* ---- bundle adjustment -----------------------------------------------------------------
 * Flat form of what BundleAdjuster::adjust packs (BundleAdjuster.cpp:17-97):
 *   poses      n_cams x 6   angle-axis * angle, translation   (world -> camera, :46-59)
 *   intrinsics n_cams x 6   fx fy cx cy k1 k2                 (:37-42)
 *   points     n_points x 3                                   (:65-70)
 *   observations landmark-major (:74-97): obs_pt non-decreasing.
 * Camera index = position in imgIdxOrder.  All three parameter arrays are updated in place. */
typedef struct {
    int32_t        n_cams, n_points, n_obs, reserved;
    double        *poses;
    double        *intrinsics;
    double        *points;
    const double  *obs_uv;    /* n_obs x 2 (integer pixel coordinates cast to double, :83-84) */
    const int32_t *obs_cam;   /* n_obs */
    const int32_t *obs_pt;    /* n_obs, non-decreasing */
} rcn_ba_problem;

typedef struct {
    int32_t max_iterations;        /* 150 if n_cams < 10 else 50           (BundleAdjuster.cpp:135-142) */
    int32_t intrinsics_mode;       /* 0: all intrinsics constant (n_cams<10, :112-115);
                                      1: cx,cy constant, fx,fy upper-bounded (:117-121) */
    int32_t fix_cam0_pose;         /* 1                                     (:100-101) */
    int32_t fix_cam1_translation;  /* 1                                     (:104-105) */
    double  focal_upper_bound;     /* 1000                                  (:120-121) */
    /* Ceres 2.x trust-region defaults (solver.h), none overridden by the reference */
    double  initial_trust_region_radius;   /* 1e4  */
    double  max_trust_region_radius;       /* 1e16 */
    double  min_trust_region_radius;       /* 1e-32 */
    double  min_relative_decrease;         /* 1e-3 */
    double  min_lm_diagonal;               /* 1e-6 */
    double  max_lm_diagonal;               /* 1e32 */
    double  function_tolerance;            /* 1e-6 */
    double  gradient_tolerance;            /* 1e-10 */
    double  parameter_tolerance;           /* 1e-8 */
    int32_t max_consecutive_invalid_steps; /* 5 */
    int32_t jacobi_scaling;                /* 1 */
} rcn_ba_options;

/* Fills the options exactly as BundleAdjuster::adjust + Ceres defaults would for n_cams. */
void rcn_ba_default_options(int32_t n_cams, rcn_ba_options *out);

#define RCN_BA_CONVERGENCE_FUNCTION   1
#define RCN_BA_CONVERGENCE_GRADIENT   2
#define RCN_BA_CONVERGENCE_PARAMETER  3
#define RCN_BA_CONVERGENCE_RADIUS     4
#define RCN_BA_NO_CONVERGENCE         5   /* max_iterations reached */
#define RCN_BA_FAILURE                6   /* too many consecutive invalid steps */

typedef struct {
    double  initial_cost;       /* 1/2 sum r^2 at the input */
    double  final_cost;
    double  initial_rms_px;     /* sqrt(sum r^2 / n_obs) */
    double  final_rms_px;
    int32_t iterations;         /* LM iterations attempted (successful + unsuccessful) */
    int32_t successful_steps;
    int32_t unsuccessful_steps;
    int32_t invalid_steps;
    int32_t termination;        /* RCN_BA_* */
    int32_t line_search_backtracks; /* backtracks of the projected Armijo search (bounds present: >= 10 cameras) */
    int32_t bound_projections;  /* focal lengths clamped to their upper bound by a Plus */
    int32_t reduced_dim;        /* rows of the reduced camera system */
    int32_t pair_lists_reused;  /* 1: the Schur build's observation-pair lists were still valid (rcn_ba_session_solve on an unchanged graph;
                                   rcn_ba_solve called again with the observation arrays of its last call, compared element by element) */
    int32_t reserved;
    double  solve_seconds;      /* wall time of the solve, inputs resident: pair lists of the Schur build + the LM loop */
    double  schur_seconds;      /* HIP-event time, summed over iterations: point blocks + Schur build */
    double  cholesky_seconds;   /* dense factorisation of the reduced system */
    double  trisolve_seconds;   /* triangular solves + back-substitution + model/candidate evaluation */
    double  cost_trace[160];    /* cost after each iteration, [0] = initial */
    double  jacobian_seconds;   /* HIP-event time of the residual + Jacobian kernel (k_ba_eval<true>: 224 B written per observation), summed */
    int32_t jacobian_evals;     /* launches summed into jacobian_seconds */
    int32_t factor_schedule;    /* 0: the factorisations ran on three streams; 1: on one stream in program order (a cross-stream
                                   wait gave up -- a runtime that does not run the streams side by side -- and the context
                                   latched the one-stream schedule; same bits either way) */
} rcn_ba_summary;

int rcn_ba_solve(rcn_ctx *ctx, const rcn_ba_problem *problem, const rcn_ba_options *options,
                 rcn_ba_summary *summary);

/* Introspection, pure host code (no device, no ctx): the SCHEDULE of the dense factorisation inside rcn_ba_solve for a reduced system of
 * n_blocks 128-row blocks (csrc/chol_plan.h) -- the list of tile operations in an order that is a correct sequential algorithm, each
 * with its stream and the device counters it waits for, and the tile maps of the pipelined launches.  tests/test_chol_plan.py executes
 * it in numpy (list order; random orders that respect only the waits) and checks that the waits order every pair of operations that
 * touch a common tile.  params: {panels per super-step, min rows for a super-step, pairs (0/1), min rows for a pair, min tiles for the
 * pipelined panel kernels, own stream for the two-level panel product (0/1), that product as the tail of the previous bulk launch (0/1), the head rows'
 * product and the next super-diagonal block's update through the latency kernel (0/1), rows below which a super-block's small operations
 * run on the chain's own stream, 2 g-row window of the chain's latency kernels (0/1), the diagonal
 * blocks in one resident workgroup (0/1), bulk updates of the right-looking regime behind the next diagonal block (0/1), rows from which on the chain's next tiles are
 * carved out of the right-looking regime's updates (0: never)},
 * NULL = what the library uses.  An operation is
 * RCN_PLAN_OP_WORDS int32: kind, stream, ticket, kb, first, m, dj, nst, map_off, map_n, g, pos, n_waits, 6 x (counter, value), timeline
 * slot, awaited, index of the bulk update whose launch carries this operation's tiles as its tail (-1: none), 1 = latency-kernel form.  Counters 0 .. 4 are the streams' progress counters (4: the resident workgroup that factors the diagonal blocks), 5 and 6 count the two
 * classes of leading tiles of the bulk updates.  Returns RCN_ERR_ARG when a buffer is too
 * small (the needed sizes are still written). */
#define RCN_PLAN_OP_WORDS 29
int rcn_ba_factor_plan(int32_t n_blocks, const int32_t *params, int32_t *ops, int64_t ops_cap, uint32_t *maps, int64_t maps_cap,
                       int64_t *n_ops, int64_t *n_maps);

/* ---- device-resident bundle adjustment across the incremental loop ----------------------------
 * SequentialReconstructor::reconstruct (SequentialReconstructor.cpp:1040-1094) runs, after every registered view,
 * checkLandmarkValidity -> BundleAdjuster().adjust(all views so far) -> checkLandmarkValidity -> removeOutlierLandmarks:
 * N - 2 global solves on a problem that grows by one camera, its new landmarks and their observations each time.
 * A session keeps that problem in HBM between the solves -- landmark coordinates, landmark-major observation arrays,
 * the observation-pair lists of the Schur build, the solver workspace -- with the graph (tracks in
 * triangulatedFeatures order; additions append like push_back) and the 12 numbers per camera mirrored on the host.
 * A session solve is rcn_ba_solve's arithmetic on the same problem: identical results, bit for bit.
 *   cameras       index = position in imgIdxOrder; pose6 / intr6 as rcn_ba_problem
 *   observations  (landmark, camera, integer pixel x, y), appended to the END of the landmark's track            */
typedef struct rcn_ba_session rcn_ba_session;
int  rcn_ba_session_create(rcn_ctx *ctx, rcn_ba_session **out);
void rcn_ba_session_destroy(rcn_ba_session *s);
int  rcn_ba_session_add_camera(rcn_ba_session *s, const double *pose6, const double *intr6, int32_t *index_out);
/* read (…_out) and / or overwrite (…_in) every camera's pose6 / intr6; any pointer may be NULL.  Adapters that keep
 * the reference's 4x4 pose matrices between solves (unpack :157-185, re-pack :46-59) round-trip the poses here. */
int  rcn_ba_session_cameras(rcn_ba_session *s, double *poses_out, double *intr_out, const double *poses_in, const double *intr_in);
int  rcn_ba_session_add_points(rcn_ba_session *s, int32_t n, const double *xyz_host, int32_t *first_index_out);
int  rcn_ba_session_add_observations(rcn_ba_session *s, int32_t n, const int32_t *pt, const int32_t *cam, const int32_t *xy);
int  rcn_ba_session_counts(const rcn_ba_session *s, int32_t *n_cams, int32_t *n_points, int64_t *n_obs);
/* the graph, flattened landmark-major in track order: n_obs entries each (xy_out: 2 per entry); any pointer may be NULL */
int  rcn_ba_session_graph(rcn_ba_session *s, int32_t *pt_out, int32_t *cam_out, int32_t *xy_out);
/* options == NULL: rcn_ba_default_options for the current camera count (the reference's choice, BundleAdjuster.cpp:99-142) */
int  rcn_ba_session_solve(rcn_ba_session *s, const rcn_ba_options *options, rcn_ba_summary *summary);
int  rcn_ba_session_read_points(rcn_ba_session *s, double *xyz_host);
const double *rcn_ba_session_points_device(rcn_ba_session *s);      /* n_points x 3 in HBM; valid until the next add / remove */
/* checkLandmarkValidity (SequentialReconstructor.cpp:869-954) on the session's arrays, on the device.  poses34_host:
 * n_cams x 12, rows of [R | t] of imgIdx2camPose as the pipeline holds them.  Observations the sweep erases are erased
 * from the tracks; inlier_out (n_points bytes), n_inliers_out, n_erased_out may be NULL. */
int  rcn_ba_session_validity(rcn_ba_session *s, const double *poses34_host, double max_projection_error, double min_triangulation_angle,
                             uint8_t *inlier_out, int32_t *n_inliers_out, int32_t *n_erased_out);
/* removeOutlierLandmarks (:956-976) for the flags of the last sweep: landmarks compacted on the device.
 * new_index_out (may be NULL): n_points entries before the call, -1 = removed. */
int  rcn_ba_session_remove_outliers(rcn_ba_session *s, int32_t *new_index_out, int32_t *n_removed_out);

/* ---- landmark validity sweep -------------------------------------------------------------
 * SequentialReconstructor::checkLandmarkValidity (SequentialReconstructor.cpp:869-954), the check
 * the reference runs on the observation graph before and after every bundle adjustment:
 *   - walk each landmark's triangulatedFeatures in order; an observation whose L1 reprojection
 *     error |u - x| + |v - y| (calcProjectionError :852-867, PinholeCamera::project Camera.h:59-76)
 *     exceeds max_projection_error, or whose camera-frame depth is negative, is erased -- with the
 *     reference's loop, which skips the element that slides into the erased slot (:877-898);
 *     fewer than two observations left after an erase marks the landmark an outlier;
 *   - a landmark none of whose surviving observation pairs subtends more than
 *     min_triangulation_angle "degrees" (calcTriangulationAngle :815-836, pi = 3.1415) is an outlier.
 * Defaults of the reference: 4.0 px and 1.0 (SequentialReconstructor.h:256-257).
 *   poses34    n_cams x 12   rows of [R | t] of imgIdx2camPose (world -> camera)
 *   intrinsics n_cams x 6    fx fy cx cy k1 k2
 *   points     n_points x 3
 *   pt_off     n_points + 1  CSR over the tracks, observations in triangulatedFeatures order
 *   obs_cam    n_obs ; obs_xy n_obs x 2 integer pixel coordinates (Feature<int>::featCoord)
 * out_inlier[n_points]: 1 = keep the landmark (the vector<bool> the reference returns);
 * out_keep[n_obs]: 1 = the observation is still in its track afterwards (the reference erases in
 * place); out_n_inliers may be NULL. */
typedef struct {
    int32_t        n_cams, n_points, n_obs, reserved;
    const double  *poses34;
    const double  *intrinsics;
    const double  *points;
    const int32_t *pt_off;
    const int32_t *obs_cam;
    const int32_t *obs_xy;
} rcn_landmark_problem;
int rcn_landmark_validity(rcn_ctx *ctx, const rcn_landmark_problem *problem, double max_projection_error,
                          double min_triangulation_angle, uint8_t *out_inlier, uint8_t *out_keep,
                          int32_t *out_n_inliers);
/* Same with every pointer (inputs and outputs, out_n_inliers_dev required) in DEVICE memory, e.g.
 * the arrays a bundle adjustment just left in HBM; asynchronous on the ctx stream, no graph check. */
int rcn_landmark_validity_device(rcn_ctx *ctx, const rcn_landmark_problem *problem_dev, double max_projection_error,
                                 double min_triangulation_angle, uint8_t *out_inlier_dev, uint8_t *out_keep_dev,
                                 int32_t *out_n_inliers_dev);

/* ---- epipolar filter of a pair's matches --------------------------------------------------
 * GeometricFilter::estimateFundamental (GeometricFilter.cpp:39-61) as the pair loop uses it
 * (SequentialReconstructor.cpp:237-269): cv::findFundamentalMat(pts1, pts2, mask) with OpenCV's
 * defaults -- RANSAC over 7-point samples for >= 15 points, LMedS for 8..14, threshold 3 px,
 * confidence 0.99, at most 1000 iterations, cv::RNG seeded the same way at every call -- of which
 * the reference keeps only the inlier mask.
 *   xy1, xy2   n x 2 integer pixel coordinates of the matched features (featuresToCvPoints,
 *              utils.cpp:165-177), in ascending query-feature order (the std::map's order)
 *   out_mask   n bytes, 1 = inlier
 *   out_count  inliers; -1: no model was found (the reference then stores no match for the pair,
 *              :252-255; mask all 0); -2: fewer than 7 points, not filtered (mask all 1, :237).
 *   out_F      may be NULL; 9 doubles, row-major: the winning 7-point hypothesis (F(3,3) = 1), zeros
 *              when there is none.  (OpenCV returns a refit on the inliers instead; no caller in the
 *              reference reads the matrix.) */
int rcn_fmat_filter(rcn_ctx *ctx, const int32_t *xy1, const int32_t *xy2, int32_t n, uint8_t *out_mask,
                    int32_t *out_count, double *out_F);
/* All pairs of a grid in one launch: pair p owns points pair_off[p] .. pair_off[p+1] of xy1 / xy2 /
 * out_mask; out_counts[p] as above; out_iterations (may be NULL) = sampling iterations executed;
 * out_F (may be NULL) = 9 doubles per pair. */
int rcn_fmat_filter_grid(rcn_ctx *ctx, int32_t n_pairs, const int32_t *pair_off, const int32_t *xy1,
                         const int32_t *xy2, uint8_t *out_mask, int32_t *out_counts, int32_t *out_iterations,
                         double *out_F);
/* Same with every pointer in DEVICE memory (all required but out_F_dev); asynchronous on the ctx stream. */
int rcn_fmat_filter_grid_device(rcn_ctx *ctx, int32_t n_pairs, const int32_t *pair_off_dev, const int32_t *xy1_dev,
                                const int32_t *xy2_dev, uint8_t *out_mask_dev, int32_t *out_counts_dev,
                                int32_t *out_iterations_dev, double *out_F_dev);

/* Fused form for a match table that is already in HBM (rcn_match_grid_device): lines :237-269 of
 * the pair loop for every pair, no host round trip.  Needs the integer pixel coordinates of the
 * keypoints of every image involved (Feature<int>::featCoord, K x 2), uploaded once per image. */
int rcn_coords_upload(rcn_ctx *ctx, int32_t img_id, const int32_t *xy_host, int32_t K);
/* n images in one call: one asynchronous copy per image, one host synchronisation (the coordinates' rcn_desc_upload_batch) */
int rcn_coords_upload_batch(rcn_ctx *ctx, int32_t first_img_id, int32_t n_images, const int32_t *const *xy_host, const int32_t *K);
int rcn_coords_clear(rcn_ctx *ctx);
/* pairs_host / table_dev / stride / counts_dev exactly as given to and left by rcn_match_grid_device.
 * The table is filtered in place: in a pair with >= 7 matches only the inliers stay (none when no
 * model was found); pairs with fewer are left alone.  counts_dev is updated; out_status_dev (may be
 * NULL) receives rcn_fmat_filter's out_count per pair.  Asynchronous on the ctx stream after one
 * small host-to-device copy. */
int rcn_match_table_filter_device(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, int32_t *table_dev,
                                  int64_t stride, int32_t *counts_dev, int32_t *out_status_dev);

/* The body of the pair loop (:232-275) for a list of pairs over resident images, results on the HOST: match, then --
 * filter != 0 -- the filter above on the table where it lies in HBM, then the dense table (as rcn_match_grid), the
 * counts and (status_host, may be NULL) the filter's verdict per pair (rcn_fmat_filter's out_count; -2 everywhere when
 * filter == 0).  HipPairGridDriver.h uses it for the pairs the loop matches a second time the other way round. */
int rcn_match_grid_filtered(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, float ratio, int32_t filter,
                            int32_t *out_host, int64_t out_stride, int32_t *counts_host, int32_t *status_host);
/* The same filter on the tables the last rcn_shard_match(sh, ratio, NULL, 0, NULL) left in the shard's ctx, in place,
 * before rcn_shard_lists.  The coordinates of EVERY image of the grid must have been uploaded to this rank's ctx
 * (rcn_coords_upload: 8 bytes per keypoint, handed round by the host like the image list itself).
 * status_host (may be NULL): verdict per pair of this rank, rcn_shard_pairs order. */
int rcn_shard_filter(rcn_shard *sh, int32_t *status_host);

/* ---- store: features + matches on disk -------------------------------------------------------
 * The cache of features / matches the reference lists as a TODO (README.md:39): one versioned binary
 * file (format in reconstructor_amd/csrc/store.hip; FNV-1a checksum; written to <path>.tmp, then
 * renamed) holding per image the K x D fp32 descriptor rows -- the Feature::featDesc.desc vectors,
 * datatypes.h:48-72 -- and optionally the K x 2 integer keypoint coordinates (Feature<int>::featCoord),
 * and per matched pair the (query feature, train feature) lists of featureMatches in the layout
 * rcn_match_compact_begin produces.  Pure host code; a matching stage can be resumed from the file. */
typedef struct {
    int32_t n_images, D, has_coords, n_pairs;
    const int32_t        *img_ids;    /* n_images */
    const int32_t        *img_K;      /* n_images */
    const float  *const  *desc;       /* n_images pointers to K x D fp32 */
    const int32_t *const *coords;     /* n_images pointers to K x 2 int32, or NULL when !has_coords */
    const int32_t        *pairs;      /* n_pairs x (query image id, train image id) */
    const int64_t        *offsets;    /* n_pairs + 1 */
    const int32_t        *qt;         /* offsets[n_pairs] x (query row, train row) */
} rcn_store_contents;
typedef struct rcn_store rcn_store;
int  rcn_store_save(const char *path, const rcn_store_contents *contents);
int  rcn_store_open(const char *path, rcn_store **out);      /* reads and verifies the whole file */
int  rcn_store_contents_of(const rcn_store *store, rcn_store_contents *out);   /* pointers live until rcn_store_close */
void rcn_store_close(rcn_store *store);
/* descriptors (and coordinates) of every stored image into the ctx: rcn_desc_upload / rcn_coords_upload per image */
int  rcn_store_upload(rcn_ctx *ctx, const rcn_store *store);

#ifdef __cplusplus
}
#endif
#endif /* RCN_H */
