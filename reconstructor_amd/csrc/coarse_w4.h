// coarse_w4.h -- K1, second form: ONE wave per SIMD, four column blocks per wave, query fragments in AGPRs.
// Included by match.hip (after CoarseArgs / swz); same work items, same LDS ring image, same packed (best,
// second) output as k_coarse_top2 -- only the distribution of the 512 query rows of a work item over the
// workgroup differs:
//     k_coarse_top2   8 waves (2 per SIMD) x 64 query columns  : 1 ds_read_b128 of a train fragment per 2 MFMAs per wave
//     k_coarse_w4     4 waves (1 per SIMD) x 128 query columns : 1 ds_read_b128 per 4 MFMAs per wave -- half the LDS
//                     operand traffic per MFMA, which on this power-limited loop is also clock (MI355X_MICROARCH.md,
//                     DVFS give-back items 3/4; cdna_hip_programming.md rule 28)
// 128 query columns x KS k-steps of B fragments are 4 x KS x 4 = 256 registers at D = 256.  They live in the
// accumulator half of the unified register file under LITERAL names (a[0:255], claimed by a clobber list; hipcc's
// own allocation of 256 "a"-constrained values keeps them in scratch and reloads them inside the loop), written
// once per work item by v_accvgpr_write; the 128 accumulators (two sets: the row block being computed and the one
// whose top-2 epilogue is being folded), fragments and state are ordinary VGPRs -- about 215, so that hipcc never
// needs an accumulator register of its own (audit after every edit: -save-temps, `.vgpr_spill_count 0`,
// `.private_segment_fixed_size 0`, no v_accvgpr_* / scratch_* outside ;;#ASMSTART .. ;;#ASMEND in the loop).
//
// The whole steady state is hand-scheduled: every instruction of the tile loop is an `asm volatile` statement, so
// hipcc only allocates the VGPRs.  What the compiler would otherwise do for us is done by construction
// (cdna_hip_programming.md section 5.7):
//   * LDS reads are counted: a fragment is read two k-steps before its MFMAs, the next row block's half-norm quads at
//     k-step 2; each consumer is preceded by `s_waitcnt lgkmcnt(N)` with N = reads issued after the one it needs (the
//     LDS returns in order), in a statement that names the destination registers "+v".
//   * hazards: an accumulator set is read by VALU no earlier than four MFMA issues (>= 128 cycles) after the last MFMA
//     that wrote it (12 wait states required); `s_nop 15` in front of the drain at the end of a pair; A fragments
//     come from LDS reads, never from a VALU write; the half-norm tuple may have been assembled by v_mov: `s_nop 1`
//     in front of the first MFMA that takes it as C; the half-norm quads are overwritten by a read issued >= 8 MFMA
//     issues after the last MFMA that took them as C; B operands were written hundreds of cycles earlier.
//   * the accumulator initialiser is the half-norm tuple as C of the first k-step with an early-clobber D (D and C
//     of an MFMA must be identical or disjoint).
// Tile hand-over: the LDS-DMA pieces of tile t+2 are issued one at a time, spread over the k-steps of tile t (an
// LDS-DMA instruction holds the wave's issue for ~60 cycles and nothing else feeds a SIMD's matrix pipe here).  Between
// the two row blocks of tile t the wave waits -- counted vmcnt: only the pieces issued in the first row block are
// younger -- until ITS pieces of tile t+1 (issued a whole tile earlier) have landed and joins the workgroup barrier;
// behind it the second row block may read tile t+1's half-norms and first fragments, and everybody is done with tile
// t-1, whose ring buffer is refilled from the top of tile t+1 on.  No LDS or DMA latency is exposed at a tile boundary.
#pragma once
#include <utility>

#define RCN_W4_QCOLS 128     // query columns per wave: 4 column blocks of 32

// every accumulator register, as a clobber list: makes the kernel descriptor allocate a[0:255] and tells hipcc that
// nothing of its own survives there
#define RCN_A10(t) "a" #t "0", "a" #t "1", "a" #t "2", "a" #t "3", "a" #t "4", "a" #t "5", "a" #t "6", "a" #t "7", "a" #t "8", "a" #t "9"
#define RCN_ALL_AGPRS                                                                                                          \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", RCN_A10(1), RCN_A10(2), RCN_A10(3), RCN_A10(4), RCN_A10(5),   \
        RCN_A10(6), RCN_A10(7), RCN_A10(8), RCN_A10(9), RCN_A10(10), RCN_A10(11), RCN_A10(12), RCN_A10(13), RCN_A10(14),       \
        RCN_A10(15), RCN_A10(16), RCN_A10(17), RCN_A10(18), RCN_A10(19), RCN_A10(20), RCN_A10(21), RCN_A10(22), RCN_A10(23),   \
        RCN_A10(24), "a250", "a251", "a252", "a253", "a254", "a255"

// B fragments of column block CB <- the query row's KS 16-byte chunks, sign bits flipped (the MFMA computes hn - q.t).
// All loads of a column block are issued before the first accumulator write (one latency per column block, not per
// fragment: hipcc does not move a load above an earlier asm statement).
template <int DP, int CB, int KSI> __device__ __forceinline__ void w4_write_b(const uint4 &v)
{
    constexpr int KS = DP / 16, A0 = (CB * KS + KSI) * 4;
    asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                 :: "v"(v.x ^ 0x80008000u), "v"(v.y ^ 0x80008000u), "v"(v.z ^ 0x80008000u), "v"(v.w ^ 0x80008000u), "i"(A0), "i"(A0 + 1), "i"(A0 + 2), "i"(A0 + 3));
}
template <int DP, int CB, int... KSIs> __device__ __forceinline__ void w4_load_b_row(const char *base, int h, int sw, std::integer_sequence<int, KSIs...>)
{
    constexpr int KS = DP / 16;
    uint4 v[KS];
    ((v[KSIs] = *reinterpret_cast<const uint4 *>(base + (((KSIs * 2 + h) ^ sw) << 4))), ...);
    (w4_write_b<DP, CB, KSIs>(v[KSIs]), ...);
}

// fold one element of the previous row block's accumulators into the running top-2 of its column block
template <int REG> __device__ __forceinline__ void w4_fold(unsigned &m1, unsigned &m2, const f32x16 &prev, unsigned hmask, unsigned prev_rowbase)
{
    unsigned u;
    asm volatile("v_and_or_b32 %2, %3, %4, %5\n\tv_med3_u32 %1, %0, %1, %2\n\tv_min_u32 %0, %0, %2"
                 : "+v"(m1), "+v"(m2), "=&v"(u)
                 : "v"(prev[REG]), "v"(hmask), "s"(prev_rowbase + (unsigned)((REG & 3) + 8 * (REG >> 2))));
}
template <int REG0, int... Es> __device__ __forceinline__ void w4_fold_n(unsigned &m1, unsigned &m2, const f32x16 &prev, unsigned hmask, unsigned prev_rowbase,
                                                                         std::integer_sequence<int, Es...>)
{
    (w4_fold<REG0 + Es>(m1, m2, prev, hmask, prev_rowbase), ...);
}

// one MFMA of column block CB at k-step KSI + FOLD epilogue elements of the previous row block.  The MFMA and the first
// element's three VALU instructions are ONE statement: hipcc pads `s_nop 0` between two statements when the first
// one's output is a register tuple -- an issue slot per MFMA for nothing (the fold does not touch the accumulator).
#define RCN_W4_FOLD_ASM(u, m1, m2) "v_and_or_b32 %[" #u "], %[p], %[mask], %[idx]\n\tv_med3_u32 %[" #m2 "], %[" #m1 "], %[" #m2 "], %[" #u "]\n\tv_min_u32 %[" #m1 "], %[" #m1 "], %[" #u "]"
template <int DP, int KSI, int CB> __device__ __forceinline__ void w4_mfma(f32x16 &acc, const u32x4 &fa, const f32x16 &hnv, unsigned &m1, unsigned &m2,
                                                                           const f32x16 &prev, unsigned hmask, unsigned prev_rowbase)
{
    constexpr int KS = DP / 16, A0 = (CB * KS + KSI) * 4, FOLD = 16 / KS, R0 = KSI * FOLD;
    const unsigned idx = prev_rowbase + (unsigned)((R0 & 3) + 8 * (R0 >> 2));
    unsigned u;
    if constexpr (KSI == 0 && CB == 0)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %[d], %[a], a[%c[b0]:%c[b1]], %[c]\n\t" RCN_W4_FOLD_ASM(u, m1, m2)
                     : [d] "=&v"(acc), [m1] "+v"(m1), [m2] "+v"(m2), [u] "=&v"(u)
                     : [a] "v"(fa), [c] "v"(hnv), [b0] "i"(A0), [b1] "i"(A0 + 3), [p] "v"(prev[R0]), [mask] "v"(hmask), [idx] "s"(idx));
    else if constexpr (KSI == 0)
        asm volatile("v_mfma_f32_32x32x16_f16 %[d], %[a], a[%c[b0]:%c[b1]], %[c]\n\t" RCN_W4_FOLD_ASM(u, m1, m2)
                     : [d] "=&v"(acc), [m1] "+v"(m1), [m2] "+v"(m2), [u] "=&v"(u)
                     : [a] "v"(fa), [c] "v"(hnv), [b0] "i"(A0), [b1] "i"(A0 + 3), [p] "v"(prev[R0]), [mask] "v"(hmask), [idx] "s"(idx));
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %[d], %[a], a[%c[b0]:%c[b1]], %[d]\n\t" RCN_W4_FOLD_ASM(u, m1, m2)
                     : [d] "+v"(acc), [m1] "+v"(m1), [m2] "+v"(m2), [u] "=&v"(u)
                     : [a] "v"(fa), [b0] "i"(A0), [b1] "i"(A0 + 3), [p] "v"(prev[R0]), [mask] "v"(hmask), [idx] "s"(idx));
    if constexpr (FOLD > 1) w4_fold_n<R0 + 1>(m1, m2, prev, hmask, prev_rowbase, std::make_integer_sequence<int, FOLD - 1>{});
}

__device__ __forceinline__ void w4_read_a(u32x4 &dst, unsigned rbase, unsigned ks_xor)
{
    unsigned ad;
    asm volatile("v_xor_b32 %1, %2, %3\n\tds_read_b128 %0, %1" : "=&v"(dst), "=&v"(ad) : "v"(rbase), "s"(ks_xor));
}
__device__ __forceinline__ void w4_read_h(u32x4 &q0, u32x4 &q1, u32x4 &q2, u32x4 &q3, unsigned hbase)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:96"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(hbase));
    // early-clobber on purpose: an LDS read writes its destination when the data RETURNS, not when the statement ends.
    // With a plain "=v" hipcc may give q0 the register of hbase (dead afterwards in program order): whenever the first
    // read returns before the other three have been issued -- a stall of a few dozen cycles in between is enough -- they
    // fetch from a garbage address.  (Observed: wrong half-norms of the first tiles of a work item on some launches.)
}
// The tile two ahead of the one being computed is staged piece by piece, spread over the k-steps of the current tile:
// an LDS-DMA instruction holds the wave's issue for ~60 cycles, and with one wave per SIMD nothing else feeds that
// SIMD's matrix pipe meanwhile -- nine of them back to back at the top of a tile idle it for ~13 % of the tile.
struct W4Stage {
    const char *src;      // this lane's first source byte of the tile being staged (wave's share, lane offset included)
    const float *hn;      // this lane's half-norm of that tile
    char *dst;            // ring buffer of that tile (generic pointer into LDS) + this wave's share offset
    char *dst_hn;         // this wave's private half-norm copy in that buffer
    bool active;          // wave-uniform: there is a tile to stage
};
template <int DP, int P> __device__ __forceinline__ void w4_stage_piece(const W4Stage &st)
{
    constexpr int NINST = RCN_BT * DP * 2 / 4 / 1024;
    if (!st.active) return;
    asm volatile("" ::: "memory");
    if constexpr (P < NINST)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(st.src + P * 1024),
                                         (__attribute__((address_space(3))) void *)(st.dst + P * 1024), 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)st.hn,
                                         (__attribute__((address_space(3))) void *)st.dst_hn, 4, 0, 0);
    asm volatile("" ::: "memory");
}

// all per-wave state of the steady state, as references to plain locals of the kernel (a struct of arrays handed through
// these inlined helpers is left in scratch memory by hipcc):
//   X, Y   accumulator sets: one computes while the other's epilogue is folded
//   af     A fragments, read two k-steps ahead; slot = k-step & 3 (KS is a multiple of 4)
//   hq     half-norm quads of the NEXT row block (C operand of its first k-step)
//   m1, m2 running (best, second) per column block, train row packed in the low bits
struct W4Refs {
    f32x16 (&X)[4];
    f32x16 (&Y)[4];
    u32x4 (&af)[4];
    u32x4 &hq0, &hq1, &hq2, &hq3;
    unsigned &m1_0, &m1_1, &m1_2, &m1_3, &m2_0, &m2_1, &m2_2, &m2_3;
    unsigned hmask;
};

// k-step KSI of a row block: wait for its fragment, request the fragment two steps ahead (the next row block's past the
// end) and, at k-step 2, the next row block's half-norm quads; then four MFMAs, each with its share of the epilogue.
// LDS queue (in issue order) per row block:  step 0: A(2) | step 1: A(3) | step 2: A(4) H H H H | step 3: A(5) | ...
// so the reads younger than A(ks) number 1, except 5 at k-steps 3 and 4; at k-step 0 the quads requested at the
// previous block's k-step 2 are needed as well and are older than (KS >= 8) or directly behind (KS = 4) A(0): still 1.
template <int DP, int RB, int KSI> __device__ __forceinline__ void w4_kstep(const W4Refs &s, const W4Stage &st, f32x16 (&ACC)[4], f32x16 (&PREV)[4], f32x16 &hnv,
                                                                            unsigned cur_rb, unsigned nxt_rb, unsigned nxt_hbase, unsigned prev_rowbase)
{
    constexpr int KS = DP / 16;
    constexpr int NG = RCN_BT * DP * 2 / 4 / 1024 + 1;        // LDS-DMA pieces per wave and tile (the last one: half-norms)
    constexpr int G = RB * KS + KSI;                          // k-step within the tile, 0 .. 2 KS - 1
    constexpr int PIECE = (G * NG + 2 * KS - 1) / (2 * KS);   // piece p is issued at k-step (p * 2 KS) / NG
    u32x4 &fa = s.af[KSI & 3];
    if constexpr (KSI == 0) {
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa), "+v"(s.hq0), "+v"(s.hq1), "+v"(s.hq2), "+v"(s.hq3));
        hnv[0] = __uint_as_float(s.hq0.x); hnv[1] = __uint_as_float(s.hq0.y); hnv[2] = __uint_as_float(s.hq0.z); hnv[3] = __uint_as_float(s.hq0.w);
        hnv[4] = __uint_as_float(s.hq1.x); hnv[5] = __uint_as_float(s.hq1.y); hnv[6] = __uint_as_float(s.hq1.z); hnv[7] = __uint_as_float(s.hq1.w);
        hnv[8] = __uint_as_float(s.hq2.x); hnv[9] = __uint_as_float(s.hq2.y); hnv[10] = __uint_as_float(s.hq2.z); hnv[11] = __uint_as_float(s.hq2.w);
        hnv[12] = __uint_as_float(s.hq3.x); hnv[13] = __uint_as_float(s.hq3.y); hnv[14] = __uint_as_float(s.hq3.z); hnv[15] = __uint_as_float(s.hq3.w);
    } else if constexpr (KSI == 3 || KSI == 4) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa));
    else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa));
    if constexpr (KSI + 2 < KS) w4_read_a(s.af[(KSI + 2) & 3], cur_rb, (unsigned)((KSI + 2) << 5));
    else w4_read_a(s.af[(KSI + 2) & 3], nxt_rb, (unsigned)((KSI + 2 - KS) << 5));
    if constexpr (KSI == 2) w4_read_h(s.hq0, s.hq1, s.hq2, s.hq3, nxt_hbase);
    w4_mfma<DP, KSI, 0>(ACC[0], fa, hnv, s.m1_0, s.m2_0, PREV[0], s.hmask, prev_rowbase);
    w4_mfma<DP, KSI, 1>(ACC[1], fa, hnv, s.m1_1, s.m2_1, PREV[1], s.hmask, prev_rowbase);
    w4_mfma<DP, KSI, 2>(ACC[2], fa, hnv, s.m1_2, s.m2_2, PREV[2], s.hmask, prev_rowbase);
    w4_mfma<DP, KSI, 3>(ACC[3], fa, hnv, s.m1_3, s.m2_3, PREV[3], s.hmask, prev_rowbase);
    if constexpr (PIECE < NG && (PIECE * 2 * KS) / NG == G) w4_stage_piece<DP, PIECE>(st);
}
// row block: ACC <- hn + A.B over KS k-steps, PREV folded meanwhile.  cur_rb / nxt_rb: LDS addresses (lane part
// included) of this / the next row block's rows; nxt_hbase: of the next row block's half-norm quads.
template <int DP, int RB, int... KSIs> __device__ __forceinline__ void w4_row_block(const W4Refs &s, const W4Stage &st, f32x16 (&ACC)[4], f32x16 (&PREV)[4], unsigned cur_rb,
                                                                                    unsigned nxt_rb, unsigned nxt_hbase, unsigned prev_rowbase, std::integer_sequence<int, KSIs...>)
{
    f32x16 hnv;
    (w4_kstep<DP, RB, KSIs>(s, st, ACC, PREV, hnv, cur_rb, nxt_rb, nxt_hbase, prev_rowbase), ...);
}

template <int DP>
__global__ __launch_bounds__(256, 1) void k_coarse_w4(CoarseArgs a)
{
    constexpr int KS = DP / 16;                       // k-steps of v_mfma_f32_32x32x16_f16 per output tile
    constexpr int ROWB = DP * 2;
    constexpr int TILEB = RCN_BT * ROWB;
    constexpr int PIECE = 64 * 16;                    // bytes per LDS-DMA instruction (16 per lane)
    constexpr int NINST = TILEB / 4 / PIECE;          // tile pieces per wave
    constexpr int BUFB = TILEB + 4 * 256;             // tile + one private half-norm copy per wave
    static_assert(KS >= 4 && KS % 4 == 0 && 16 % KS == 0, "DP must be 64, 128 or 256");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    asm volatile("" ::: RCN_ALL_AGPRS);

    const int b = blockIdx.x;
    const int item = (b & 7) * a.items_per_xcd + (b >> 3);   // XCD x walks a contiguous item range
    if (item >= a.n_groups * a.tiles_per_pair) return;
    const int grp = item / a.tiles_per_pair, qt = item - grp * a.tiles_per_pair;
    const int2 g = a.groups[grp];
    const int p0 = g.x, R = g.y;
    if (R == 0) return;
    const ImgDev qi = a.imgs[a.pairs[2 * p0]];
    if (qt * RCN_QT >= qi.K) return;

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    f32x16 X[4], Y[4];
    u32x4 af[4], hq0, hq1, hq2, hq3;
    unsigned m1[4], m2[4], hmask;
    asm volatile("v_mov_b32 %0, %1" : "=v"(hmask) : "s"(~a.idx_mask));
    const W4Refs s{X, Y, af, hq0, hq1, hq2, hq3, m1[0], m1[1], m1[2], m1[3], m2[0], m2[1], m2[2], m2[3], hmask};

    // ---- per-pair train image records, parked in LDS (nothing but LDS-DMA on the vector-memory queue later)
    struct TrainRec { const char *f16; const float *hn; int nT; int pad; };
    static_assert(sizeof(TrainRec) * RCN_GROUP <= RCN_TBL_BYTES, "train-record table does not fit its LDS slot");
    TrainRec *tbl = reinterpret_cast<TrainRec *>(smem + RCN_NBUF * BUFB);
    if (tid < R) {
        const ImgDev ti = a.imgs[a.pairs[2 * (p0 + tid) + 1]];
        TrainRec rec;
        rec.f16 = reinterpret_cast<const char *>(ti.f16);
        rec.hn = ti.hn;
        rec.nT = ti.K >= 2 ? (ti.K + RCN_BT - 1) / RCN_BT : 0;
        rec.pad = 0;
        tbl[tid] = rec;
    }
    __syncthreads();
    auto tiles_of = [&](int rr) -> int { return __builtin_amdgcn_readfirstlane(tbl[rr].nT); };

    // ---- staging cursor: runs ahead of the compute cursor, across pairs
    int s_pair = 0, s_tile = 0, s_nT = 0, staged = 0;
    const char *s_timg = nullptr;
    const float *s_hn = nullptr;
    auto s_seek = [&]() {
        while (s_pair < R) {
            s_nT = tiles_of(s_pair);
            if (s_nT > 0) {
                const unsigned long long pf = reinterpret_cast<unsigned long long>(tbl[s_pair].f16);
                const unsigned long long ph = reinterpret_cast<unsigned long long>(tbl[s_pair].hn);
                s_timg = reinterpret_cast<const char *>(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(pf >> 32)) << 32) |
                                                        (unsigned)__builtin_amdgcn_readfirstlane((unsigned)pf));
                s_hn = reinterpret_cast<const float *>(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(ph >> 32)) << 32) |
                                                       (unsigned)__builtin_amdgcn_readfirstlane((unsigned)ph));
                s_tile = 0;
                return;
            }
            ++s_pair;
        }
    };
    auto stage_next = [&]() {
        if (s_pair >= R) return;
        char *bbase = smem + (staged % RCN_NBUF) * BUFB;
#pragma unroll
        for (int i = 0; i < NINST; ++i) {
            const int off = (w * NINST + i) * PIECE;
            const char *src = s_timg + (size_t)s_tile * TILEB + off + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(bbase + off), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s_hn + s_tile * RCN_BT + lane),
                                         (__attribute__((address_space(3))) void *)(bbase + TILEB + w * 256), 4, 0, 0);
        ++staged;
        if (++s_tile == s_nT) { ++s_pair; s_seek(); }
    };
    int total_tiles = 0;
    for (int rr = 0; rr < R; ++rr) total_tiles += tiles_of(rr);
    if (total_tiles == 0) return;
    s_seek();
    stage_next();
    stage_next();
    // ---- query fragments, negated, resident in a[0:16*KS) for the whole item (loaded behind the first two tiles'
    //      LDS-DMA: their latency hides the fragments')
    {
        const char *qbase = reinterpret_cast<const char *>(qi.f16) + (size_t)(qt * RCN_QT + w * RCN_W4_QCOLS + r) * ROWB;   // rows < Kp (a multiple of 512)
        const int sw = swz<DP>(r);                 // swz depends on the row's low bits only: the same for every column block
        w4_load_b_row<DP, 0>(qbase, h, sw, std::make_integer_sequence<int, KS>{});
        w4_load_b_row<DP, 1>(qbase + 32 * ROWB, h, sw, std::make_integer_sequence<int, KS>{});
        w4_load_b_row<DP, 2>(qbase + 64 * ROWB, h, sw, std::make_integer_sequence<int, KS>{});
        w4_load_b_row<DP, 3>(qbase + 96 * ROWB, h, sw, std::make_integer_sequence<int, KS>{});
    }

    const unsigned smem_base = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)smem;
    // lane constants of the fragment address:  addr(tile, rb, ks) = (tile + rb*32*ROWB + lane_a) ^ (ks << 5)
    //   lane_a = r*ROWB + ((h ^ (sw & 1)) << 4) + ((sw & ~1) << 4), sw = swz(r)   (the XOR swizzle of the image;
    //   tile and rb*32*ROWB are multiples of ROWB, so the XOR only ever touches lane_a's chunk bits)
    const int swl = swz<DP>(r);
    const unsigned lane_a = (unsigned)(r * ROWB + (((h ^ (swl & 1)) + (swl & ~1)) << 4));
    const unsigned lane_h = (unsigned)(TILEB + w * 256 + 16 * h);      // half-norm quads of this lane's accumulator rows

    // tiles 0 and 1 have landed for everybody before the first fragment is read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int NGP = NINST + 1;                                       // pieces per wave and tile
    constexpr int MID_PIECES = (KS * NGP + 2 * KS - 1) / (2 * KS);       // pieces p with (p * 2 KS) / NGP < KS: issued in row block 0
    static_assert(((MID_PIECES - 1) * 2 * KS) / NGP < KS && (MID_PIECES * 2 * KS) / NGP >= KS, "piece schedule");
    int done = 0;        // tiles computed so far, flat over the item's pairs
    bool primed = false;
    for (int rr = 0; rr < R; ++rr) {
        const int nT = tiles_of(rr);
        if (nT == 0) continue;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            m1[cb] = m2[cb] = 0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i < 16; ++i) Y[cb][i] = 3.0e38f;    // folded beside the first row block: never a candidate
        }
        for (int t = 0; t < nT; ++t, ++done) {
            // tile done+2 goes out piece by piece beside this tile's MFMAs (its ring buffer was last read as tile done-2)
            W4Stage st;
            st.active = s_pair < R;
            st.src = s_timg + (size_t)s_tile * TILEB + w * NINST * PIECE + lane * 16;
            st.hn = s_hn + s_tile * RCN_BT + lane;
            st.dst = smem + (staged % RCN_NBUF) * BUFB + w * NINST * PIECE;
            st.dst_hn = smem + (staged % RCN_NBUF) * BUFB + TILEB + w * 256;
            const unsigned tile = smem_base + (unsigned)((done % RCN_NBUF) * BUFB);
            const unsigned ntile = smem_base + (unsigned)(((done + 1) % RCN_NBUF) * BUFB);
            const unsigned rb0 = tile + lane_a, rb1 = rb0 + 32 * ROWB, nrb0 = ntile + lane_a;
            const unsigned hb0 = tile + lane_h, hb1 = hb0 + 128, nhb0 = ntile + lane_h;
            if (!primed) {      // first tile of the item: nothing was prefetched yet.  Queue: A0, H x4, A1
                w4_read_a(af[0], rb0, 0u);
                w4_read_h(hq0, hq1, hq2, hq3, hb0);
                w4_read_a(af[1], rb0, 32u);
                primed = true;
            }
            const unsigned base = (unsigned)(t * RCN_BT);
            w4_row_block<DP, 0>(s, st, X, Y, rb0, rb1, hb1, base - 32u, std::make_integer_sequence<int, KS>{});      // epilogue of (t-1, rb 1)
            // middle of tile `done`: tile done+1 (issued during tile done-1, a tile's worth of cycles ago) must have landed for
            // everybody before the second row block reads its half-norms and first fragments.  Only the pieces of tile done+2
            // issued in the first row block are younger: a counted wait, so nothing recent is waited for.  (With the wait at
            // the top of the tile instead, the last pieces -- issued a few hundred cycles earlier -- were still in flight:
            // SQ_WAIT_ANY was 19 % of the wave cycles.)  The barrier also tells every wave that tile done-1's ring buffer
            // is free: it is refilled from the top of tile done+1 on.
            if (st.active) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(MID_PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            w4_row_block<DP, 1>(s, st, Y, X, rb1, nrb0, nhb0, base, std::make_integer_sequence<int, KS>{});          // epilogue of (t, rb 0)
            if (st.active) {            // the staging cursor moves on
                ++staged;
                if (++s_tile == s_nT) { ++s_pair; s_seek(); }
            }
        }
        {   // drain: epilogue of the pair's last row block (just written by MFMAs: 12 wait states first)
            asm volatile("s_nop 15" : "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]));
            const unsigned last = (unsigned)((nT - 1) * RCN_BT + 32);
            w4_fold_n<0>(m1[0], m2[0], Y[0], hmask, last, std::make_integer_sequence<int, 16>{});
            w4_fold_n<0>(m1[1], m2[1], Y[1], hmask, last, std::make_integer_sequence<int, 16>{});
            w4_fold_n<0>(m1[2], m2[2], Y[2], hmask, last, std::make_integer_sequence<int, 16>{});
            w4_fold_n<0>(m1[3], m2[3], Y[3], hmask, last, std::make_integer_sequence<int, 16>{});
        }
        // lane l and l^32 hold the same query, disjoint train rows: merge, then lanes 0..31 store
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            unsigned a1 = m1[cb] | (unsigned)(4 * h), a2 = m2[cb] | (unsigned)(4 * h);
            if (m1[cb] == 0xFFFFFFFFu) a1 = 0xFFFFFFFFu;
            if (m2[cb] == 0xFFFFFFFFu) a2 = 0xFFFFFFFFu;
            unsigned b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2, 32);
            unsigned r1 = min(a1, b1);
            unsigned r2 = min(max(a1, b1), min(a2, b2));
            const int qrow = qt * RCN_QT + w * RCN_W4_QCOLS + cb * 32 + r;
            if (h == 0 && qrow < qi.K) a.cand[(size_t)(p0 + rr) * a.kq_stride + qrow] = make_uint2(r1, r2);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the prefetches past the last row block
}
