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
    int32_t K, Kp;
};

struct ImgHost {
    float *f32 = nullptr;
    _Float16 *f16 = nullptr;
    float *hn = nullptr;
    double *nrm2 = nullptr;
    int32_t K = 0, Kp = 0;
    int32_t slot = -1;  // row in the device image table
    int32_t slab = -1;  // >= 0: buffers are views into ctx->slabs[slab] (batch upload)
    int32_t cap_rows = 0, cap_D = 0;   // capacity of the owned allocations
    bool dirty = true;  // fp16 copy / half-norms not yet built for the current scale
};

// One batch of equally shaped images (rcn_desc_upload_batch_device): one allocation per array.
struct Slab {
    int32_t first_id = 0, n = 0, K = 0, Kp = 0, D = 0;
    const float *f32 = nullptr;   // borrowed from the caller
    _Float16 *f16 = nullptr;
    float *hn = nullptr;
    double *nrm2 = nullptr;
    bool live = false;
};

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
    std::vector<int32_t> all_pairs_host;              // the canonical i < j grid when the caller passes pairs == NULL
    bool prepared = false;
    double scale = 0.0;      // s, power of two (0 = nothing prepared yet)
    double bias = 0.0;       // BIAS in accumulator units
    double max_norm = 0.0;   // max |x| over resident rows
    DevBuf img_table, pairs_dev, groups_dev, cand, owner, fb_list, sv_list, counters, out_tmp, cnt_tmp;
    rcn_match_stats last_stats;
    bool profile = false;
    hipEvent_t ev[64][4];
    bool ev_made = false;
    int ev_n = 0;          // recorded calls since the last stats read (<= 64)
    int ablate = 0;            // RCN_COARSE_ABL (diagnostics)
    int chunks = 1;            // RCN_MATCH_CHUNKS: >1 overlaps re-rank(c) with coarse(c+1) on two streams
    bool force_exact = false;  // RCN_FORCE_EXACT=1: skip the MFMA coarse pass (diagnostics)
    bool no_item_order = false;   // RCN_MATCH_NO_ORDER=1: work items in pair order instead of heaviest-first per XCD (diagnostics)

    // ---- BA state (ba.hip)
    DevBuf ba_ws[40];
    DevBuf lm_ws;              // landmark validity sweep (validity.hip)
    DevBuf fm_ws, fm_state;    // epipolar filter (fmat.hip): host-API staging, per-pair RANSAC state
    DevBuf fm_csr, fm_pairs;   // fused table filter: CSR of the matched points, per-pair coordinate pointers
    std::map<int32_t, std::pair<DevBuf, int32_t>> coords;   // image id -> (K x 2 int32 pixel coordinates in HBM, K)
    bool ba_atomics = false;   // RCN_BA_SCHUR_ATOMICS=1: atomic Schur accumulation instead of the gather form
    bool ba_trsv_fwd = false;  // RCN_BA_TRSV_FWD=1: separate forward substitution instead of the rhs row inside the factorisation
    hipStream_t aux_stream = nullptr;   // lookahead stream of the Cholesky
    hipEvent_t ba_ev[9];
    hipEvent_t ba_tev[4];            // phase timing of rcn_ba_solve
    bool ba_ev_made = false;

    void set_error(const std::string &s) { err = s; }
};

int rcn_match_release(rcn_ctx *ctx);
