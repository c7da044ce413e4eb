// ctx.hip -- context lifetime of the C ABI (include/rcn.h).
#include "rcn_internal.h"
#ifdef RCN_DIAG
extern "C" int rcn_diag_set_poll(int mode, int sleeps);
#endif
#include <vector>

#include <cstdlib>

extern "C" {

const char *rcn_version(void)
{
#ifdef RCN_DIAG
    return "reconstructor_amd 0.3 (gfx950) DIAGNOSTIC BUILD";
#else
    return "reconstructor_amd 0.3 (gfx950)";
#endif
}

int rcn_create(int device_id, rcn_ctx **out)
{
    if (!out) return RCN_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n)
        return RCN_ERR_NO_DEVICE;
    rcn_ctx *ctx = new rcn_ctx();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess ||
        hipGetDeviceProperties(&ctx->prop, device_id) != hipSuccess) {
        delete ctx;
        return RCN_ERR_NO_DEVICE;
    }
    // the code object holds gfx950 ISA only: fail loudly on anything else
    if (std::string(ctx->prop.gcnArchName).rfind("gfx950", 0) != 0) {
        delete ctx;
        return RCN_ERR_NO_DEVICE;
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return RCN_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    // Streams of the dense factorisation (ba.hip, chol_plan.h).  Throughput work -- the bulk trailing updates (aux_stream) and the
    // two-level regime's panel products below the head rows (panel2_stream) -- runs under a CU mask that leaves a few CUs free (mask
    // bits interleave over the XCDs: bit i -> XCD i % 8, so whole rounds of eight keep the XCDs even): the chain's single-workgroup
    // diagonal kernel (132 KB of LDS) never queues behind resident bulk workgroups.  Round 5: the panel stream, which carries the
    // small kernels the chain WAITS for (in-block panels and columns, block rows of a super-block's inverse, the head rows'
    // product and the update of the next super-diagonal block), is NOT masked any more and gets the highest stream priority: with
    // two bulk workgroups per CU holding every vector register of the masked CUs, its kernels could only start where a bulk
    // workgroup retired -- 40-100 us for a 5-us kernel once a bulk tile lives 165 us (K = 512), measured in the device timeline.
    {
        const int ncu = ctx->prop.multiProcessorCount;
        std::vector<uint32_t> mask((ncu + 31) / 32, 0xFFFFFFFFu);
        if (ncu % 32) mask.back() = (1u << (ncu % 32)) - 1u;
        bool carve = ncu >= 64;
        int reserved = 8;
        int panel_mode = 1;      // 0: the panel stream under the bulk streams' mask (rounds 2-4); 1: unmasked, highest priority; 2: masked off the chain's eight CUs only
#ifdef RCN_DIAG
        if (getenv("RCN_NO_CU_MASK")) carve = false;
        if (const char *rc = getenv("RCN_RESERVED_CUS")) reserved = std::max(8, std::min(64, std::atoi(rc) / 8 * 8));
        if (const char *pm = getenv("RCN_PANEL_MODE")) panel_mode = std::atoi(pm);
#endif
        std::vector<uint32_t> mask8 = mask;
        if (carve) {
            for (int i = 0; i < reserved; ++i) mask[(size_t)i / 32] &= ~(1u << (i % 32));
            mask8[0] &= ~0xFFu;
        }
        if (hipExtStreamCreateWithCUMask(&ctx->aux_stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            ctx->aux_stream = nullptr;
        }
        ctx->bulk_cu_mask = mask;      // (a second stream under the same mask is made when a plan asks for it: ensure_chol_plan)
        if (panel_mode != 1) {
            const std::vector<uint32_t> &pm = panel_mode == 2 ? mask8 : mask;
            if (hipExtStreamCreateWithCUMask(&ctx->panel_stream, (uint32_t)pm.size(), pm.data()) != hipSuccess) {
                (void)hipGetLastError();
                ctx->panel_stream = nullptr;
            }
        } else {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
            if (hipStreamCreateWithPriority(&ctx->panel_stream, hipStreamNonBlocking, hi) != hipSuccess) {
                (void)hipGetLastError();
                ctx->panel_stream = nullptr;
            }
        }
    }
    if (!ctx->panel_stream && hipStreamCreateWithFlags(&ctx->panel_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return RCN_ERR_HIP;
    }
#ifdef RCN_DIAG
    {
        const char *pm = std::getenv("RCN_POLL_MODE"), *ps = std::getenv("RCN_POLL_SLEEPS");
        if (pm || ps) (void)rcn_diag_set_poll(pm ? std::atoi(pm) : 0, ps ? std::atoi(ps) : 1);
    }
#endif
    if (!ctx->aux_stream && hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return RCN_ERR_HIP;
    }
    for (auto &e : ctx->ba_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete ctx; return RCN_ERR_HIP; }
    for (auto &e : ctx->ba_tev)
        if (hipEventCreate(&e) != hipSuccess) { delete ctx; return RCN_ERR_HIP; }
    ctx->ba_ev_made = true;
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->ba_host_scal), 16 * sizeof(double), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->ba_host_scal = nullptr; }
    else memset(ctx->ba_host_scal, 0, 16 * sizeof(double));
#ifdef RCN_DIAG
    // Diagnostic build only (tools/librcn_diag.so, -DRCN_DIAG): ablations and alternative device paths.
    // The shipping library reads no environment variable.
    const char *w4 = std::getenv("RCN_COARSE_W4");
    ctx->coarse_w4 = w4 && w4[0] == '1';
    const char *s16 = std::getenv("RCN_COARSE_S16");
    ctx->coarse_shape = s16 ? (s16[0] == '1' ? 1 : 0) : -1;
    const char *fe = std::getenv("RCN_FORCE_EXACT");
    ctx->force_exact = fe && fe[0] == '1';
    const char *nio = getenv("RCN_MATCH_NO_ORDER");
    ctx->no_item_order = nio && nio[0] == '1';
    const char *ab = std::getenv("RCN_COARSE_ABL");
    ctx->ablate = ab ? std::atoi(ab) : 0;
    const char *bat = std::getenv("RCN_BA_SCHUR_ATOMICS");
    ctx->ba_atomics = bat && bat[0] == '1';
    const char *bps = getenv("RCN_PAIR_SMALL");
    if (bps) ctx->ba_pair_small = bps[0] != '0';
    const char *btf = getenv("RCN_BA_TRSV_FWD");
    ctx->ba_trsv_fwd = btf && btf[0] == '1';
    const char *cs = std::getenv("RCN_CHOL_SAFE");
    ctx->chol_safe = cs && cs[0] == '1';
    const char *tch = std::getenv("RCN_TRSV_CHAIN");
    if (tch) ctx->trsv_chain = tch[0] != '0';
    const char *cbk = std::getenv("RCN_CHOL_BREAK");
    ctx->chol_break = cbk ? std::atoi(cbk) : 0;
    const char *cpm = std::getenv("RCN_CHOL_PAIR_MIN");
    if (cpm) ctx->chol_pair_min = std::atoi(cpm);
    const char *cgr = std::getenv("RCN_CHOL_GROUP");
    if (cgr) ctx->chol_group = std::atoi(cgr);
    const char *ctl = std::getenv("RCN_CHOL_TL");
    if (ctl) ctx->chol_tl_g = std::atoi(ctl);
    const char *ctm = std::getenv("RCN_CHOL_TL_MIN");
    if (ctm) ctx->chol_tl_min = std::atoi(ctm);
    const char *cps = std::getenv("RCN_CHOL_PGSTREAM");
    if (cps) ctx->chol_pg_stream = std::atoi(cps);
    const char *ccs = std::getenv("RCN_CHOL_CHAIN_STREAM");
    if (ccs) ctx->chol_chain_stream = std::atoi(ccs);
    if (ctx->chol_chain_stream && ctx->chol_chain_stream != 4 && !ctx->chain_stream) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        if (ctx->chol_chain_stream == 3) {      // a stream with a CU mask of all CUs: a hardware queue of its own?
            const int ncu = ctx->prop.multiProcessorCount;
            std::vector<uint32_t> full((ncu + 31) / 32, 0xFFFFFFFFu);
            if (ncu % 32) full.back() = (1u << (ncu % 32)) - 1u;
            if (hipExtStreamCreateWithCUMask(&ctx->chain_stream, (uint32_t)full.size(), full.data()) != hipSuccess) { (void)hipGetLastError(); ctx->chain_stream = nullptr; }
        } else
        if (hipStreamCreateWithPriority(&ctx->chain_stream, hipStreamNonBlocking, ctx->chol_chain_stream == 2 ? 0 : hi) != hipSuccess) { (void)hipGetLastError(); ctx->chain_stream = nullptr; }
    }
    if (const char *bm = std::getenv("RCN_BA_MIRROR")) if (bm[0] == '0' && ctx->ba_host_scal) { (void)hipHostFree(ctx->ba_host_scal); ctx->ba_host_scal = nullptr; }      // the scalars by copy + synchronisation, as before round 5
    const char *cbb = std::getenv("RCN_CHOL_BULK_BEHIND");
    if (cbb) ctx->chol_bulk_behind = std::atoi(cbb);
    const char *ccv = std::getenv("RCN_CHOL_CARVE");
    if (ccv) ctx->chol_carve_rows = std::atoi(ccv);
    const char *cds = std::getenv("RCN_CHOL_DIAG_SERVER");
    if (cds) ctx->chol_diag_server = std::atoi(cds);
    const char *cwn = std::getenv("RCN_CHOL_WINDOW");
    if (cwn) ctx->chol_window = std::atoi(cwn);
    const char *cts = std::getenv("RCN_CHOL_TL_SERIAL");
    if (cts) ctx->chol_tl_serial = std::atoi(cts);
    const char *chs = std::getenv("RCN_CHOL_HEAD_SMALL");
    if (chs) ctx->chol_head_small = std::atoi(chs);
    const char *cft = std::getenv("RCN_CHOL_FUSE_TAIL");
    if (cft) ctx->chol_fuse_tail = std::atoi(cft);
    const char *cpp = std::getenv("RCN_CHOL_PG_PRIO");
    if (cpp) ctx->chol_pg_prio = cpp[0] != '0';
    const char *cgk = std::getenv("RCN_CHOL_GATE_IN_KERNEL");
    ctx->chol_gate_in_kernel = cgk ? std::atoi(cgk) : 0;
    const char *cht = std::getenv("RCN_CHOL_HOSTTIME");
    ctx->chol_host_time = cht && cht[0] == '1';
    const char *cpi = std::getenv("RCN_CHOL_PIPE_MIN");
    if (cpi) ctx->chol_pipe_min = std::atoi(cpi);
    const char *ch = std::getenv("RCN_CHUNK_ROWS");
    if (ch && std::atoll(ch) > 0) ctx->chunk_rows = std::atoll(ch);
    const char *mr = std::getenv("RCN_MID_ROWS");
    if (mr && std::atoll(mr) >= 0) ctx->mid_rows = std::max<long long>(1, std::atoll(mr));
#endif
    memset(&ctx->last_stats, 0, sizeof(ctx->last_stats));
    *out = ctx;
    return RCN_OK;
}

void rcn_destroy(rcn_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    rcn_match_release(ctx);
    DevBuf *bufs[] = {&ctx->img_table, &ctx->pairs_dev, &ctx->groups_dev, &ctx->cand, &ctx->owner,
                      &ctx->fb_list, &ctx->sv_list, &ctx->counters, &ctx->out_tmp, &ctx->cnt_tmp, &ctx->scale_dev, &ctx->desc_bad, &ctx->bulk_map, &ctx->diag_items};
    for (DevBuf *b : bufs) b->release();
    for (DevBuf &b : ctx->ba_ws) b.release();
    ctx->lm_ws.release();
    ctx->fm_ws.release();
    ctx->fm_state.release();
    ctx->fm_csr.release(); ctx->fm_pairs.release();
    for (auto &kv : ctx->coords) kv.second.first.release();
    if (ctx->ev_made) {
        for (auto &call : ctx->ev_c)
            for (auto &row : call)
                for (auto &e : row) (void)hipEventDestroy(e);
        for (auto &row : ctx->ev_tail)
            for (auto &e : row) (void)hipEventDestroy(e);
    }
    ctx->mid_ws.release();
    if (ctx->ba_ev_made)
        { for (auto &e : ctx->ba_ev) (void)hipEventDestroy(e); for (auto &e : ctx->ba_tev) (void)hipEventDestroy(e); }
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
        for (auto &e : ctx->cmp_ev) (void)hipEventDestroy(e);
        (void)hipEventDestroy(ctx->cmp_filled);
    }
    ctx->cmp_off.release(); ctx->cmp_qt[0].release(); ctx->cmp_qt[1].release();
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->panel_stream) (void)hipStreamDestroy(ctx->panel_stream);
    if (ctx->panel2_stream) (void)hipStreamDestroy(ctx->panel2_stream);
    if (ctx->ba_host_scal) (void)hipHostFree(ctx->ba_host_scal);
    if (ctx->diag_stream) (void)hipStreamDestroy(ctx->diag_stream);
    if (ctx->chain_stream) (void)hipStreamDestroy(ctx->chain_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *rcn_last_error(const rcn_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int rcn_set_stream(rcn_ctx *ctx, void *hip_stream)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return RCN_OK;
}

int rcn_synchronize(rcn_ctx *ctx)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_HIP(hipSetDevice(ctx->device));
    RCN_HIP(hipStreamSynchronize(ctx->stream));
    return RCN_OK;
}

}  // extern "C"
