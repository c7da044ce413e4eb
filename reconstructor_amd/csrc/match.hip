// match.hip -- all-pairs descriptor matching on MI355X (gfx950): kernels + C ABI.
//
// Replaces, behind include/rcn.h:
//   FlannMatcher::matchFeatures          FeatureMatcher.cpp:32-65   (per pair)
//   cv::DescriptorMatcher::knnMatch(k=2) call site FeatureMatcher.cpp:49
//   SequentialReconstructor::matchFeatures pair loop, SequentialReconstructor.cpp:199-279
//
// Pipeline per grid call:
//   K1 k_coarse_top2   fp16 MFMA (v_mfma_f32_32x32x16_f16) distance tiles, running top-2 per
//                      query with the train index packed into the low mantissa bits
//   K2 k_rerank        exact fp64 re-computation of the two candidates in the canonical
//                      order, certified against a rigorous bound on the coarse error;
//                      rows that cannot be certified go to the fallback list
//   K2b k_exact_rows   exact brute force of the listed rows (also the generic path)
//   K3 k_unique_claim / k_unique_emit   lowest query index keeps a contested train row
// The result is bit-identical to the canonical exact matcher whatever the coarse pass does:
// the coarse pass only ever *proposes*.
#include "rcn_internal.h"

#include <algorithm>
#include <cmath>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define RCN_QT 512          // query rows per workgroup (8 waves x 64)
#define RCN_GROUP 4         // consecutive pairs (same query image) swept by one workgroup
#define RCN_TBL_BYTES ((RCN_GROUP * 24 + 127) / 128 * 128)   // LDS table of the group's train-image records
#define RCN_NBUF 4          // LDS ring depth (train tiles)
#define RCN_PD 2            // prefetch distance, tiles
#define RCN_CHUNK_ROWS (1ll << 27)   // default query-row slots of the candidate table per pipeline chunk (ctx->chunk_rows: 1 GiB of candidates; the row lists are at most as long).
                                     // cfg 3 on one box (tools/chunk_sweep.py): 1 chunk 3104 ms per call, library holds 57.8 GB; 8 chunks 3104 ms, 9.8 GB; 16 chunks (this
                                     // setting) 3116 ms, 6.2 GB; 31 chunks 3135 ms, 4.4 GB
#define RCN_BT 64           // train rows per LDS tile (k_coarse_w4; k_coarse_top2: coarse_bt)
#define RCN_PAD_HN 1.0e30f  // half-norm of padded train rows: never a candidate

// ---------------------------------------------------------------------------------------
// 16-byte-chunk swizzle of the fp16 image.  A row of DP halfs has DP/8 chunks; rows that a
// ds_read_b128 lane group reads together (16 different rows, same logical chunk) must land
// on 16 different 16-B slots of the 256-B LDS bank row.
template <int DP> __host__ __device__ __forceinline__ int swz(int row)
{
    if (DP >= 128) return row & 15;
    if (DP == 64) return (row >> 1) & 7;
    return (row >> 2) & 3;  // DP == 32
}

// ---------------------------------------------------------------------------------------
// Row statistics at upload: |x|^2 in fp64 and the running maxima that fix the global
// power-of-two scale.  One wave per row, coalesced; |x|^2 only feeds error bounds (which carry
// their own slack), so the summation order is free.
// g_hist (RCN_HIST_BINS words): rows per octave of |x|^2 (bin = floor(log2 |x|^2) + RCN_HIST_BINS / 2, clamped; zero rows in
// bin 0), collected in LDS and flushed once per workgroup -- what fix_scale needs to tell a few out-of-scale rows from the rest.
__device__ __forceinline__ int norm_bin(double n2)
{
    if (!(n2 > 0.0)) return 0;
    if (!(n2 <= 1.7976931348623157e308)) return RCN_HIST_BINS - 1;       // infinity / NaN
    const int e = ilogb(n2) + RCN_HIST_BINS / 2;
    return e < 1 ? 1 : (e > RCN_HIST_BINS - 1 ? RCN_HIST_BINS - 1 : e);
}
template <bool VEC4>
__global__ __launch_bounds__(256) void k_rowstats(const float *__restrict__ x, int K, int D, double *__restrict__ nrm2,
                                                  unsigned *__restrict__ g_maxabs_bits,
                                                  unsigned long long *__restrict__ g_maxnrm2_bits, unsigned *__restrict__ g_hist)
{
    __shared__ unsigned s_hist[RCN_HIST_BINS];
    for (int i = threadIdx.x; i < RCN_HIST_BINS; i += 256) s_hist[i] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    float ma = 0.f;
    double mx = 0.0;
    if (VEC4) {
        // D % 4 == 0, 16-B aligned rows: 16 lanes per row, 4 rows per wave step, float4 loads
        const int sub = lane & 15, rsel = lane >> 4, nv = D >> 2;
        for (int j0 = wave * 4; j0 < K; j0 += nwaves * 4) {
            const int j = j0 + rsel;
            double acc = 0.0;
            if (j < K) {
                const float4 *row = reinterpret_cast<const float4 *>(x + (size_t)j * D);
                for (int k = sub; k < nv; k += 16) {
                    const float4 v = row[k];
                    ma = fmaxf(fmaxf(ma, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                    acc = fma((double)v.x, (double)v.x, acc);
                    acc = fma((double)v.y, (double)v.y, acc);
                    acc = fma((double)v.z, (double)v.z, acc);
                    acc = fma((double)v.w, (double)v.w, acc);
                }
            }
            for (int o = 8; o; o >>= 1) acc += __shfl_xor(acc, o);
            if (sub == 0 && j < K) { nrm2[j] = acc; atomicAdd(&s_hist[norm_bin(acc)], 1u); }
            mx = fmax(mx, acc);
        }
        for (int o = 32; o >= 16; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    } else {
        for (int j = wave; j < K; j += nwaves) {
            const float *row = x + (size_t)j * D;
            double acc = 0.0;
            for (int k = lane; k < D; k += 64) {
                const float v = row[k];
                ma = fmaxf(ma, fabsf(v));
                acc = fma((double)v, (double)v, acc);
            }
            for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) { nrm2[j] = acc; atomicAdd(&s_hist[norm_bin(acc)], 1u); }
            mx = fmax(mx, acc);
        }
    }
    // non-negative floats / doubles order like their bit patterns.  One atomic pair per
    // workgroup, and only when it would raise the maximum (a single word takes ~100 atomics/us).
    __shared__ float s_ma[4];
    __shared__ double s_mx[4];
    for (int o = 32; o; o >>= 1) ma = fmaxf(ma, __shfl_xor(ma, o));
    if (lane == 0) { s_ma[threadIdx.x >> 6] = ma; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { ma = fmaxf(ma, s_ma[w]); mx = fmax(mx, s_mx[w]); }
        const unsigned a = __float_as_uint(ma);
        const unsigned long long m = (unsigned long long)__double_as_longlong(mx);
        if (a > __hip_atomic_load(g_maxabs_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(g_maxabs_bits, a);
        if (m > __hip_atomic_load(g_maxnrm2_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(g_maxnrm2_bits, m);
    }
    // (the __syncthreads above ordered every wave's LDS histogram updates before this flush)
    for (int i = threadIdx.x; i < RCN_HIST_BINS; i += 256)
        if (s_hist[i]) atomicAdd(g_hist + i, s_hist[i]);
}

// the histogram again from norms already in HBM (n rows, zero = unused slot row): rebuilt when replaced images have left
// more history in it than there are rows resident (rcn_int_prepare_all)
__global__ __launch_bounds__(256) void k_norm_hist(const double *__restrict__ nrm2, long n, unsigned *__restrict__ g_hist)
{
    __shared__ unsigned s_hist[RCN_HIST_BINS];
    for (int i = threadIdx.x; i < RCN_HIST_BINS; i += 256) s_hist[i] = 0u;
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) atomicAdd(&s_hist[norm_bin(nrm2[i])], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < RCN_HIST_BINS; i += 256)
        if (s_hist[i]) atomicAdd(g_hist + i, s_hist[i]);
}

// fp32 rows -> scaled fp16 rows (chunk-swizzled) + biased half-norms.
template <int DP>
__global__ void k_prepare(const float *__restrict__ x, const double *__restrict__ nrm2, int K,
                          int Kp, int D, const ScaleDev *__restrict__ sc,
                          _Float16 *__restrict__ f16, float *__restrict__ hn, unsigned long long *__restrict__ bigmin)
{
    constexpr int CPR = DP / 8;
    const float scale = sc->sf;
    const double half_s2 = sc->hs2, bias = sc->bias, thr2 = sc->thr2;
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int row = gid / CPR, c = gid % CPR;
    if (row >= Kp) return;
    // a BIG row (|x|^2 >= thr2, fix_scale): no fp16 copy (zeros), never a coarse candidate (half-norm = padding value);
    // the image remembers the smallest such norm
    const bool big = row < K && thr2 <= 1.7976931348623157e308 && !(nrm2[row] < thr2);      // only a finite threshold splits (a NaN norm is not a BIG row)
    half8 v;
    if (big) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
        if (c == 0) atomicMin(bigmin, (unsigned long long)__double_as_longlong(nrm2[row] == nrm2[row] ? nrm2[row] : 0.0));
    } else if (row < K && c * 8 + 8 <= D && (D & 3) == 0) {      // whole chunk inside the row, 16-B aligned: two float4 loads
        const float4 lo = *reinterpret_cast<const float4 *>(x + (size_t)row * D + c * 8);
        const float4 hi = *reinterpret_cast<const float4 *>(x + (size_t)row * D + c * 8 + 4);
        v[0] = (_Float16)(lo.x * scale); v[1] = (_Float16)(lo.y * scale); v[2] = (_Float16)(lo.z * scale); v[3] = (_Float16)(lo.w * scale);
        v[4] = (_Float16)(hi.x * scale); v[5] = (_Float16)(hi.y * scale); v[6] = (_Float16)(hi.z * scale); v[7] = (_Float16)(hi.w * scale);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int k = c * 8 + e;
            float f = (row < K && k < D) ? x[(size_t)row * D + k] * scale : 0.f;
            v[e] = (_Float16)f;  // round to nearest even
        }
    }
    *reinterpret_cast<half8 *>(f16 + (size_t)row * DP + ((c ^ swz<DP>(row)) * 8)) = v;
    if (c == 0) hn[row] = (row < K && !big) ? (float)(half_s2 * nrm2[row] + bias) : RCN_PAD_HN;
}

// Same for a batch of n equally shaped images ([n][K][D] fp32 -> [n][Kp][DP] fp16).
template <int DP>
__global__ void k_prepare_batch(const float *__restrict__ x, const double *__restrict__ nrm2,
                                int n, int Kslot, int Kp, int D, const ScaleDev *__restrict__ sc,
                                _Float16 *__restrict__ f16, float *__restrict__ hn,
                                const int32_t *__restrict__ Ks, unsigned long long *__restrict__ bigmin)
{
    // every image owns a slot of Kslot fp32 rows; Ks (may be NULL = Kslot everywhere) holds the rows in use
    constexpr int CPR = DP / 8;
    const float scale = sc->sf;
    const double half_s2 = sc->hs2, bias = sc->bias, thr2 = sc->thr2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long grow = gid / CPR;
    const int c = (int)(gid % CPR);
    if (grow >= (long)n * Kp) return;
    const int img = (int)(grow / Kp), row = (int)(grow % Kp);
    const float *xi = x + (size_t)img * Kslot * D;
    const int K = Ks ? Ks[img] : Kslot;
    const double n2 = row < K ? nrm2[(size_t)img * Kslot + row] : 0.0;
    const bool big = row < K && thr2 <= 1.7976931348623157e308 && !(n2 < thr2);            // BIG row: see k_prepare
    half8 v;
    if (big) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
        if (c == 0) atomicMin(bigmin + img, (unsigned long long)__double_as_longlong(n2 == n2 ? n2 : 0.0));
    } else if (row < K && c * 8 + 8 <= D && (D & 3) == 0) {      // whole chunk inside the row, 16-B aligned: two float4 loads
        const float4 lo = *reinterpret_cast<const float4 *>(xi + (size_t)row * D + c * 8);
        const float4 hi = *reinterpret_cast<const float4 *>(xi + (size_t)row * D + c * 8 + 4);
        v[0] = (_Float16)(lo.x * scale); v[1] = (_Float16)(lo.y * scale); v[2] = (_Float16)(lo.z * scale); v[3] = (_Float16)(lo.w * scale);
        v[4] = (_Float16)(hi.x * scale); v[5] = (_Float16)(hi.y * scale); v[6] = (_Float16)(hi.z * scale); v[7] = (_Float16)(hi.w * scale);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int k = c * 8 + e;
            float f = (row < K && k < D) ? xi[(size_t)row * D + k] * scale : 0.f;
            v[e] = (_Float16)f;
        }
    }
    *reinterpret_cast<half8 *>(f16 + (size_t)grow * DP + ((c ^ swz<DP>(row)) * 8)) = v;
    if (c == 0) hn[grow] = (row < K && !big) ? (float)(half_s2 * n2 + bias) : RCN_PAD_HN;
}

// +infinity (as bits) into n words: "no BIG row" before a conversion records any
__global__ void k_fill_inf(unsigned long long *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0x7FF0000000000000ull;
}

// The global scale from the row statistics (counters[0]: max |x| as fp32 bits, counters[2..3]: max |x|^2 as fp64
// bits), on the device: the arithmetic of rcn_int_prepare_all's host path, operation for operation (frexp / ldexp /
// ceil are exact; sqrt is correctly rounded on both sides), so that every GPU of a sharded grid -- and a one-GPU run
// through the host path -- arrives at the same constants.
// Out-of-scale rows.  The scale is one power of two for every resident image, so a handful of rows a few thousand times larger
// than the rest would push everybody else's fp16 copy towards the subnormals (correct, but nothing certifies any more and
// every row goes through the exact kernels).  The histogram tells that situation apart: if at most 1 / 8 of the rows lie
// above some octave and the next occupied octave above it is at least 6 further up (norms 8 times larger), the rows above
// are BIG rows -- they get no fp16 copy, never become coarse candidates, go to the exact kernel as queries, and bound every
// other query's certificates through their smallest norm (k_filter) -- and the scale is fixed for the rows below:
// o->thr2 = 2^(top normal octave + 1), s from sqrt(thr2), |x|^2 bound = thr2.  Without such a split thr2 = +infinity and
// everything is as it was (same s, same BIAS, bit for bit).
__host__ __device__ inline void fix_scale(float maxabs, double maxn2, const unsigned *hist, int DPa, ScaleDev *o)
{
    if (!(maxabs > 0.f && maxabs <= 3.4028234e38f)) maxabs = 1.f;          // zero, NaN, infinity
    if (!(maxn2 > 0.0 && maxn2 <= 1.7976931348623157e308)) maxn2 = 1.0;
    double thr2 = 1.0e308 * 10.0;                                            // +infinity
    if (hist) {
        unsigned long long total = 0;
        int top = 0;
        for (int b = 1; b < RCN_HIST_BINS; ++b) { total += hist[b]; if (hist[b]) top = b; }
        // walk down from the top: `above` = rows in octaves > b
        unsigned long long above = 0;
        int lowest_big = top + 1;                                            // lowest occupied octave among the rows above b
        for (int b = top; b >= 1 && total >= 64; --b) {
            if (above * 8ull > total) break;
            if (hist[b] && above > 0 && lowest_big - b >= 6)                 // b is occupied, little lies above it, and far above:
                thr2 = ldexp(1.0, b - RCN_HIST_BINS / 2 + 1);                // a split (the lowest one within the budget wins)
            if (hist[b]) { above += hist[b]; lowest_big = b; }
        }
    }
    if (thr2 <= 1.7976931348623157e308) {
        maxn2 = thr2;                                                        // every normal row has |x|^2 < thr2
        maxabs = (float)sqrt(thr2);                                          // ... and every element below |x|
    }
    // s = 2^e with s*maxabs in (2^13, 2^14]  (fp16 max is 65504; the query side is negated only)
    int ex;
    (void)frexp((double)maxabs, &ex);  // maxabs = m * 2^ex, m in [0.5,1)
    const double s = ldexp(1.0, 14 - ex);
    // the norm bound is rounded UP to a 1/16-octave grid so that it (and BIAS) stays put while
    // images of similar norm come and go: then only new images need converting
    int en;
    const double mn = frexp(maxn2 * (1.0 + 1e-12), &en);            // in [0.5, 1)
    const double maxn2q = ldexp(ceil(mn * 32.0) / 32.0, en);
    const double bias = 0.5625 * s * s * maxn2q + 1.0;  // accumulator >= s^2 Nmax^2/16 > 0 for every (q,t)
    const double n_max = sqrt(maxn2q), u = ldexp(1.0, -11);
    o->s = s; o->s2 = s * s; o->hs2 = 0.5 * s * s; o->bias = bias;
    o->c_in = (2 * u + u * u) * s * s;
    o->c_sub = ldexp(1.0, -14) * sqrt((double)DPa) * s;
    o->c_acc = (DPa + 8) * ldexp(1.0, -23);
    o->n_max = n_max;
    o->hn_max = 0.5 * s * s * n_max * n_max + bias;
    o->rel_slack = 1e-9;
    o->thr2 = thr2;
    o->sf = (float)s; o->pad = 0.f;
}
__global__ void k_fix_scale(const unsigned *__restrict__ counters, int DPa, ScaleDev *__restrict__ out)
{
    if (threadIdx.x || blockIdx.x) return;
    const float maxabs = __uint_as_float(counters[0]);
    const double maxn2 = __longlong_as_double(*reinterpret_cast<const long long *>(counters + 2));
    fix_scale(maxabs, maxn2, counters + RCN_HIST_WORD, DPa, out);
}

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// train rows per LDS tile of k_coarse_top2
__host__ __device__ constexpr int coarse_bt(int DP) { return DP == 256 ? 64 : 128; }

struct CoarseArgs {
    const ImgDev *imgs;
    const int32_t *pairs;   // n_pairs x (query slot, train slot)
    const int2 *groups;     // n_groups x (first pair, count <= RCN_GROUP): consecutive pairs sharing the query image
    uint2 *cand;            // [n_pairs][kq_stride] packed (best, second)
    int32_t n_groups, tiles_per_pair, items_per_xcd, kq_stride;
    uint32_t idx_mask;      // low bits that carry the train row
};

#include "coarse_w4.h"

// K1: one workgroup (8 waves, 2 per SIMD) = 512 query rows of one image against every train row
//   of up to RCN_GROUP consecutive pairs that share that query image; query fragments are
//   loaded once per work item.
//   MFMA orientation: A = train tile (rows -> accumulator registers), B = -query (columns ->
//   lanes), C initialised with the train rows' biased half-norms, so each accumulator
//   element is  s^2 * (|t|^2/2 - q.t) + BIAS  > 0  and orders like the squared distance for a
//   fixed query.  Positive floats order like unsigned integers, so the running top-2 per
//   (lane, column block) is v_and_or + v_med3_u32 + v_min_u32 per element, software-pipelined
//   into the MFMA issue gaps of the next row block.
//   Train tiles (64 rows + a private copy of their half-norms per wave) stream through an LDS
//   ring of RCN_NBUF buffers by LDS-DMA, RCN_PD tiles ahead, across pair boundaries; one raw
//   s_barrier per tile behind a counted s_waitcnt vmcnt (never 0 in steady state).
// ABL (ablation bits, diagnostics only -- results are wrong unless ABL == 0):
//   1 skip the top-2 epilogue, 2 fold the VALUES only (two vector operations per element instead of three: what an exact top-2
//   without the index in the key would cost), 4 (with 2) keep the running MINIMUM only -- one operation per element, 8 stage only the first tiles.
// SH: MFMA shape.  0: v_mfma_f32_32x32x16_f16 (two 32-column blocks per wave, 16 accumulators per lane and tile);
//   1: v_mfma_f32_16x16x32_f16 -- the same 64 query columns per wave as four 16-column blocks, a 32-row train block as
//   two 16-row halves, 4 accumulators per lane and tile; same operand bytes per MAC from LDS, same registers, same
//   VALU per element, twice the MFMA instructions at half the passes.  The chip holds a different clock on the two
//   shapes under load (MI355X_MICROARCH.md, DVFS give-back item 7), which is why both exist.
template <int DP, int ABL, int SH = 0>
__global__ __launch_bounds__(512, 2) void k_coarse_top2(CoarseArgs a)
{
    constexpr int KS = DP / 16;
    constexpr int NCB = SH ? 4 : 2;                 // column (query) blocks per wave
    constexpr int CW = SH ? 16 : 32;                // their width
    constexpr int NKS = SH ? DP / 32 : DP / 16;     // k-steps per tile row block
    constexpr int CPK = SH ? 4 : 2;                 // 16-byte chunks of a row one k-step consumes
    constexpr int ROWB = DP * 2;
    // train rows per LDS tile = per stage barrier: 64 at D = 256, 128 below -- the same 32 KB of operands and the same 64 MFMA
    // per wave between two barriers at D = 128 as at D = 256 (with 64-row tiles the D = 128 kernel met a barrier and a DMA wait
    // every 32 MFMAs: round 2 measured 45.8 % there against 56 % at D = 256 with the same VALU work per element)
    constexpr int BT = coarse_bt(DP);
    constexpr int TILEB = BT * ROWB;
    constexpr int GLSZ = DP == 32 ? 4 : 16;          // bytes per lane per LDS-DMA instruction
    constexpr int PIECE = 64 * GLSZ;
    constexpr int NINST = TILEB / 8 / PIECE;         // tile pieces per wave
    constexpr int HNP = BT / 64;                     // half-norm pieces (64 rows each) per wave
    constexpr int NG = NINST + HNP;
    constexpr int BUFB = TILEB + 8 * BT * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int b = blockIdx.x;
    const int item = (b & 7) * a.items_per_xcd + (b >> 3);  // XCD x walks a contiguous item range
    if (item >= a.n_groups * a.tiles_per_pair) return;
    const int grp = item / a.tiles_per_pair, qt = item - grp * a.tiles_per_pair;
    const int2 g = a.groups[grp];
    const int p0 = g.x, R = g.y;
    if (R == 0) return;                                      // padding slot of the item order
    const ImgDev qi = a.imgs[a.pairs[2 * p0]];
    if (qt * RCN_QT >= qi.K) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int r = SH ? (lane & 15) : (lane & 31), h = SH ? (lane >> 4) : (lane >> 5);   // row / column in the block, k group
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // query fragments, negated, resident for the whole item
    half8 bq[NCB][NKS];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int qrow = qt * RCN_QT + w * 64 + cb * CW + r;  // < Kp (Kp is a multiple of 512)
        const char *base = reinterpret_cast<const char *>(qi.f16) + (size_t)qrow * ROWB;
        const int sw = swz<DP>(qrow);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            uint4 v = *reinterpret_cast<const uint4 *>(base + (((ks * CPK + h) ^ sw) << 4));
            v.x ^= 0x80008000u; v.y ^= 0x80008000u; v.z ^= 0x80008000u; v.w ^= 0x80008000u;
            bq[cb][ks] = __builtin_bit_cast(half8, v);
        }
    }
    // the mask lives in a VGPR so that (acc & mask) | idx is ONE v_and_or_b32 with idx in an SGPR
    unsigned hmask;
    asm volatile("v_mov_b32 %0, %1" : "=v"(hmask) : "s"(~a.idx_mask));

    // per-pair train image records, read once with ordinary loads and parked in LDS: inside the
    // tile loop nothing but LDS-DMA may sit on the vector-memory queue (counted vmcnt)
    struct TrainRec { const char *f16; const float *hn; int nT; int pad; };
    static_assert(sizeof(TrainRec) * RCN_GROUP <= RCN_TBL_BYTES, "train-record table does not fit its LDS slot");
    TrainRec *tbl = reinterpret_cast<TrainRec *>(smem + RCN_NBUF * BUFB);
    if (tid < R) {
        const ImgDev ti = a.imgs[a.pairs[2 * (p0 + tid) + 1]];
        TrainRec rec;
        rec.f16 = reinterpret_cast<const char *>(ti.f16);
        rec.hn = ti.hn;
        rec.nT = ti.K >= 2 ? (ti.K + BT - 1) / BT : 0;
        rec.pad = 0;
        tbl[tid] = rec;
    }
    __syncthreads();
    auto tiles_of = [&](int rr) -> int { return __builtin_amdgcn_readfirstlane(tbl[rr].nT); };

    // ---- staging cursor (runs RCN_PD tiles ahead of the compute cursor, across pairs)
    int s_pair = 0, s_tile = 0, s_nT = 0, staged = 0;
    const char *s_timg = nullptr;
    const float *s_hn = nullptr;
    auto s_seek = [&]() {   // move to the next pair that has tiles
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
        if ((ABL & 8) && staged >= RCN_NBUF) { ++staged; }
        else {
            char *bbase = smem + (staged % RCN_NBUF) * BUFB;
#pragma unroll
            for (int i = 0; i < NINST; ++i) {
                const int off = (w * NINST + i) * PIECE;
                const char *src = s_timg + (size_t)s_tile * TILEB + off + lane * GLSZ;
                if constexpr (DP == 32)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)src,
                        (__attribute__((address_space(3))) void *)(bbase + off), 4, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)src,
                        (__attribute__((address_space(3))) void *)(bbase + off), 16, 0, 0);
            }
#pragma unroll
            for (int hp = 0; hp < HNP; ++hp)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(s_hn + s_tile * BT + 64 * hp + lane),
                    (__attribute__((address_space(3))) void *)(bbase + TILEB + w * (BT * 4) + 256 * hp), 4, 0, 0);
            ++staged;
        }
        if (++s_tile == s_nT) { ++s_pair; s_seek(); }
    };
    s_seek();
#pragma unroll
    for (int i = 0; i < RCN_PD; ++i) stage_next();

    unsigned m1[NCB], m2[NCB];
    f32x16 pX0, pX1, pY0, pY1;          // SH 0: current / previous row block, two column blocks
    f32x4 qX[2][4], qY[2][4];           // SH 1: [16-row half][column block]
    auto top2 = [&](int cb, unsigned u) {
        if constexpr ((ABL & 4) != 0) { m1[cb] = min(m1[cb], u); return; }      // ONE operation per element: the running minimum alone (round 5: what a fold that leaves the second neighbour to a later sweep would cost)
        // med3(m1,m2,u) spelled so that isel forms v_med3_u32 (scheduler sees a plain VALU op)
        const unsigned lo = min(m1[cb], m2[cb]), hi = max(m1[cb], m2[cb]);
        m2[cb] = max(lo, min(hi, u));
        m1[cb] = min(m1[cb], u);
    };
    // LDS reads of the ring go through inline asm: hipcc would otherwise put s_waitcnt vmcnt(0)
    // in front of every ds_read that may alias an in-flight LDS-DMA.  Each consumer is preceded
    // by a wait statement that names the registers it is about to use ("+v"), so neither the
    // MFMAs nor copies of those registers can be scheduled above the wait.
    auto lds_read = [&](unsigned addr) -> u32x4 {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
        return v;
    };
    // cur <- hn + A.B for row block (tile, rb); prev (row block before it) is folded into top-2
    auto step = [&](f32x16 &c0, f32x16 &c1, const f32x16 &p0v, const f32x16 &p1v, unsigned tile,
                    unsigned hnl, int rb, unsigned prev_rowbase) {
        if constexpr (SH == 0) {
        const int lrow = rb * 32 + r;
        const unsigned arow = tile + lrow * ROWB;
        const int sw = swz<DP>(lrow);
        u32x4 h0, h1, h2, h3, f0, f1;
        h0 = lds_read(hnl + (rb * 32 + 0 + 4 * h) * 4);
        h1 = lds_read(hnl + (rb * 32 + 8 + 4 * h) * 4);
        h2 = lds_read(hnl + (rb * 32 + 16 + 4 * h) * 4);
        h3 = lds_read(hnl + (rb * 32 + 24 + 4 * h) * 4);
        f0 = lds_read(arow + ((h ^ sw) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(f0));
        // the half-norm tuple is the C input of the first k-step of BOTH column blocks (D may differ
        // from C), so no accumulator is initialised by copies
        f32x16 hnv;
        hnv[0] = __uint_as_float(h0.x); hnv[1] = __uint_as_float(h0.y); hnv[2] = __uint_as_float(h0.z); hnv[3] = __uint_as_float(h0.w);
        hnv[4] = __uint_as_float(h1.x); hnv[5] = __uint_as_float(h1.y); hnv[6] = __uint_as_float(h1.z); hnv[7] = __uint_as_float(h1.w);
        hnv[8] = __uint_as_float(h2.x); hnv[9] = __uint_as_float(h2.y); hnv[10] = __uint_as_float(h2.z); hnv[11] = __uint_as_float(h2.w);
        hnv[12] = __uint_as_float(h3.x); hnv[13] = __uint_as_float(h3.y); hnv[14] = __uint_as_float(h3.z); hnv[15] = __uint_as_float(h3.w);
        auto kstep = [&](int ks, u32x4 &cur, u32x4 &nxt) {
            // naming c1 ties the wait BEHIND the previous k-step's second MFMA
            if (ks > 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur), "+v"(c1));
            // fragment of k-step ks+1 is in flight while ks computes
            if (ks + 1 < KS) nxt = lds_read(arow + ((((ks + 1) * 2 + h) ^ sw) << 4));
            const half8 av = __builtin_bit_cast(half8, cur);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bq[0][ks], ks == 0 ? hnv : c0, 0, 0, 0);
            if (!(ABL & 1)) {
#pragma unroll
                for (int reg = ks * 16 / KS; reg < (ks + 1) * 16 / KS; ++reg)
                    top2(0, (ABL & 2) ? __float_as_uint(p0v[reg]) : ((__float_as_uint(p0v[reg]) & hmask) | (prev_rowbase + (reg & 3) + 8 * (reg >> 2))));
            }
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bq[1][ks], ks == 0 ? hnv : c1, 0, 0, 0);
            if (!(ABL & 1)) {
#pragma unroll
                for (int reg = ks * 16 / KS; reg < (ks + 1) * 16 / KS; ++reg)
                    top2(1, (ABL & 2) ? __float_as_uint(p1v[reg]) : ((__float_as_uint(p1v[reg]) & hmask) | (prev_rowbase + (reg & 3) + 8 * (reg >> 2))));
            }
        };
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            kstep(ks, f0, f1);
            kstep(ks + 1, f1, f0);
        }
        }
    };
    // SH 1.  cur <- hn + A.B for row block (tile, rb) as 2 halves x 4 column blocks of 16x16x32 MFMAs; the 32 values a
    // lane holds of the row block before it are folded into the top-2 between them.  Element (half s, block cb, reg)
    // is train row 16 s + 4 h + reg of the block: 16 s + reg goes into the packed index here, 4 h at the very end.
    auto step16 = [&](f32x4 (&c)[2][4], const f32x4 (&pv)[2][4], unsigned tile, unsigned hnl, int rb, unsigned prev_rowbase) {
        if constexpr (SH == 1) {
            const int lrow = rb * 32 + r;                    // + 16 for the second half: same swizzle (period <= 16 rows)
            const unsigned arow0 = tile + lrow * ROWB, arow1 = arow0 + 16 * ROWB;
            const int sw = swz<DP>(lrow);
            u32x4 hA, hB, f0a, f0b, f1a, f1b;
            hA = lds_read(hnl + (rb * 32 + 0 + 4 * h) * 4);
            hB = lds_read(hnl + (rb * 32 + 16 + 4 * h) * 4);
            f0a = lds_read(arow0 + ((h ^ sw) << 4));
            f0b = lds_read(arow1 + ((h ^ sw) << 4));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hA), "+v"(hB), "+v"(f0a), "+v"(f0b));
            f32x4 hn[2];
            hn[0] = __builtin_bit_cast(f32x4, hA);
            hn[1] = __builtin_bit_cast(f32x4, hB);
            constexpr int EPK = 32 / NKS;                    // previous-block elements folded per k-step
            auto kstep = [&](int ks, u32x4 &ca_, u32x4 &cb_, u32x4 &na_, u32x4 &nb_) {
                // naming the last accumulator ties the wait BEHIND the previous k-step's MFMAs
                if (ks > 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ca_), "+v"(cb_), "+v"(c[1][3]));
                if (ks + 1 < NKS) {
                    na_ = lds_read(arow0 + ((((ks + 1) * 4 + h) ^ sw) << 4));
                    nb_ = lds_read(arow1 + ((((ks + 1) * 4 + h) ^ sw) << 4));
                }
                const half8 av[2] = {__builtin_bit_cast(half8, ca_), __builtin_bit_cast(half8, cb_)};
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int sh = m >> 2, cb = m & 3;
                    c[sh][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[sh], bq[cb][ks], ks == 0 ? hn[sh] : c[sh][cb], 0, 0, 0);
                    if (!(ABL & 1)) {
#pragma unroll
                        for (int e = ks * EPK + m * EPK / 8; e < ks * EPK + (m + 1) * EPK / 8; ++e) {
                            const int es = e >> 4, ecb = (e >> 2) & 3, er = e & 3;
                            top2(ecb, (__float_as_uint(pv[es][ecb][er]) & hmask) | (prev_rowbase + 16 * es + er));
                        }
                    }
                }
            };
#pragma unroll
            for (int ks = 0; ks < NKS; ks += 2) {
                kstep(ks, f0a, f0b, f1a, f1b);
                if (ks + 1 < NKS) kstep(ks + 1, f1a, f1b, f0a, f0b);
            }
        }
    };
    const unsigned smem_base = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)smem;

    int done = 0;   // tiles computed so far (flat over the item's pairs)
    for (int rr = 0; rr < R; ++rr) {
        const int nT = tiles_of(rr);
        if (nT == 0) continue;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) m1[cb] = m2[cb] = 0xFFFFFFFFu;
        if constexpr (SH == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { pY0[i] = 3.0e38f; pY1[i] = 3.0e38f; }
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) qY[i >> 4][(i >> 2) & 3][i & 3] = 3.0e38f;
        }
        for (int t = 0; t < nT; ++t, ++done) {
            stage_next();
            const int ahead = staged - done - 1;   // tiles issued after the one about to be read
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NG) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NG) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned tile = smem_base + (done % RCN_NBUF) * BUFB;
            const unsigned hnl = tile + TILEB + w * (BT * 4);
            const unsigned base = (unsigned)(t * BT);
#pragma unroll
            for (int rb = 0; rb < BT / 32; rb += 2) {
                if constexpr (SH == 0) {
                    step(pX0, pX1, pY0, pY1, tile, hnl, rb, base + 32u * rb - 32u);   // epilogue of the row block before (t, rb)
                    step(pY0, pY1, pX0, pX1, tile, hnl, rb + 1, base + 32u * rb);     // epilogue of (t, rb)
                } else {
                    step16(qX, qY, tile, hnl, rb, base + 32u * rb - 32u);
                    step16(qY, qX, tile, hnl, rb + 1, base + 32u * rb);
                }
            }
        }
        if (ABL & 1) {
            if constexpr (SH == 0) asm volatile("" ::"v"(pY0), "v"(pY1), "v"(pX0), "v"(pX1));
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(qY[i >> 2][i & 3]), "v"(qX[i >> 2][i & 3]));
            }
        }
        {   // drain: epilogue of the pair's last row block
            const unsigned rowbase = (unsigned)((nT - 1) * BT + BT - 32);
            if constexpr (SH == 0) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const unsigned idx = rowbase + (reg & 3) + 8 * (reg >> 2);
                    top2(0, (__float_as_uint(pY0[reg]) & hmask) | idx);
                    top2(1, (__float_as_uint(pY1[reg]) & hmask) | idx);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 32; ++e) {
                    const int es = e >> 4, ecb = (e >> 2) & 3, er = e & 3;
                    top2(ecb, (__float_as_uint(qY[es][ecb][er]) & hmask) | (rowbase + 16 * es + er));
                }
            }
        }
        if constexpr (SH == 0) {
            // lane l and l^32 hold the same query, disjoint train rows: merge, then lanes 0..31 store
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                unsigned a1 = m1[cb] | (unsigned)(4 * h), a2 = m2[cb] | (unsigned)(4 * h);
                if (m1[cb] == 0xFFFFFFFFu) a1 = 0xFFFFFFFFu;
                if (m2[cb] == 0xFFFFFFFFu) a2 = 0xFFFFFFFFu;
                unsigned b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2, 32);
                unsigned r1 = min(a1, b1);
                unsigned r2 = min(max(a1, b1), min(a2, b2));
                const int qrow = qt * RCN_QT + w * 64 + cb * 32 + r;
                if (h == 0 && qrow < qi.K) a.cand[(size_t)(p0 + rr) * a.kq_stride + qrow] = make_uint2(r1, r2);
            }
        } else {
            // lanes l, l^16, l^32, l^48 hold the same query, disjoint train rows (4 h + reg of every 16): two merges
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                unsigned a1 = m1[cb] | (unsigned)(4 * h), a2 = m2[cb] | (unsigned)(4 * h);
                if (m1[cb] == 0xFFFFFFFFu) a1 = 0xFFFFFFFFu;
                if (m2[cb] == 0xFFFFFFFFu) a2 = 0xFFFFFFFFu;
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const unsigned b1 = __shfl_xor(a1, o), b2 = __shfl_xor(a2, o);
                    const unsigned r1 = min(a1, b1);
                    a2 = min(max(a1, b1), min(a2, b2));
                    a1 = r1;
                }
                const int qrow = qt * RCN_QT + w * 64 + cb * 16 + r;
                if (h == 0 && qrow < qi.K) a.cand[(size_t)(p0 + rr) * a.kq_stride + qrow] = make_uint2(a1, a2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// canonical squared distance: fp64 fma chain in ascending k
template <bool VEC4>
__device__ __forceinline__ double exact_d2(const float *__restrict__ q, const float *__restrict__ t, int D)
{
    double acc = 0.0;
    if (VEC4) {
        const float4 *q4 = reinterpret_cast<const float4 *>(q);
        const float4 *t4 = reinterpret_cast<const float4 *>(t);
#pragma unroll 8
        for (int k = 0; k < D / 4; ++k) {
            float4 a = q4[k], b = t4[k];
            double d;
            d = (double)a.x - (double)b.x; acc = fma(d, d, acc);
            d = (double)a.y - (double)b.y; acc = fma(d, d, acc);
            d = (double)a.z - (double)b.z; acc = fma(d, d, acc);
            d = (double)a.w - (double)b.w; acc = fma(d, d, acc);
        }
    } else {
        for (int k = 0; k < D; ++k) {
            double d = (double)q[k] - (double)t[k];
            acc = fma(d, d, acc);
        }
    }
    return acc;
}

// Lowe ratio test exactly as FeatureMatcher.cpp:55 on float distances (sqrt of squared L2).
__device__ __forceinline__ bool ratio_pass(double d2_best, double d2_second, float ratio)
{
    float dist0 = sqrtf((float)d2_best);
    float dist1 = sqrtf((float)d2_second);
    return dist0 < ratio * dist1;
}

struct RerankArgs {
    const ImgDev *imgs;
    const int32_t *pairs;
    const uint2 *cand;
    int32_t *out;                 // [n_pairs][out_stride]: >=0 train row, -1 none (before uniqueness)
    int64_t out_stride;
    unsigned long long *fb_list;  // (pair << 32 | query) rows needing the exact kernel
    unsigned *fb_count;
    unsigned long long *sv_list;  // rows that survive the coarse filter: exact re-rank
    unsigned *sv_count;
    int32_t n_pairs, kq_stride, D, qblocks, pair_base;
    uint32_t idx_mask;
    float ratio;
    const ScaleDev *sc;  // error model of the coarse pass, in HBM: see rcn_internal.h
    int32_t all_to_fallback;
};

// ---- certified decision from exact candidate distances --------------
// ea<=eb exact (fp64 chain) distances of the two candidates (ia, ib), ordered by (value, idx);
// lbnc = lower bound on the exact distance of every non-candidate row.
// returns  >=0 : certified match index,  -1 : certified "no match",  -2 : cannot certify.
__device__ __forceinline__ int certify(double ea, int ia, double eb, double lbnc, float ratio)
{
    const bool nn_certain = ea < lbnc;
    double lb1 = fmin(eb, lbnc), lb0 = fmin(ea, lbnc);
    if (lb1 < 0.0) lb1 = 0.0;
    if (lb0 < 0.0) lb0 = 0.0;
    if (nn_certain && ratio_pass(ea, lb1, ratio)) return ia;
    if (!ratio_pass(lb0, eb, ratio)) return -1;
    return -2;
}

// coarse-error bound for one query, accumulator units
__device__ __forceinline__ double coarse_eps(const ScaleDev &a, double nq2)
{
    const double nq = sqrt(nq2) * (1.0 + 1e-12);
    const double mag = a.s2 * nq * a.n_max;
    return a.c_in * nq * a.n_max + a.c_sub * (nq + a.n_max) + 1e-9 + a.c_acc * (a.hn_max + mag) +
           6.0e-8 * a.hn_max;
}
// accumulator value -> squared distance (both real-valued), minus/plus the evaluation slack
__device__ __forceinline__ double acc_to_d2(const ScaleDev &a, double nq2, double acc)
{
    return nq2 + (2.0 / a.s2) * (acc - a.bias);
}

// lower bound on the exact squared distance from a query of squared norm nq2 to ANY BIG row of an image: (|t| - |q|)^2 with
// |t| >= the image's smallest BIG norm, evaluated with a relative slack; +infinity when the image has no BIG row
__device__ __forceinline__ double big_lower_bound(const unsigned long long *bigmin, double nq2)
{
    const double b2 = __longlong_as_double((long long)*bigmin);
    if (!(b2 <= 1.7976931348623157e308)) return b2;                 // +infinity
    const double d = sqrt(b2) * (1.0 - 1e-12) - sqrt(nq2) * (1.0 + 1e-12);
    return d > 0.0 ? d * d * (1.0 - 1e-9) : 0.0;
}

__device__ __forceinline__ void list_append(unsigned long long *list, unsigned *count, bool want,
                                            unsigned long long entry)
{
    const unsigned long long m = __ballot(want);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    if (want) list[base + __popcll(m & ((1ull << lane) - 1ull))] = entry;
}

// K2a: one thread per (pair, query row): decide from the coarse values alone whenever the
// ratio test fails for every distance pair compatible with the error bound (the common case:
// queries without a true counterpart have dist0 ~ dist1) or passes for every such pair (clear
// matches: dist0 << dist1); only the undecided rows go to the survivor list.
#define RCN_FT 1024   // threads of a k_filter workgroup
#define RCN_FR 4      // rows per thread, RCN_FT apart (round 4, second session: a workgroup of one row per thread lived ~4 us -- one load, the
                      // arithmetic, two barriers, one atomic -- and a CU holds two of them: the kernel was bound by that latency, 982 us per cfg-3
                      // chunk at 1.6 TB/s of HBM traffic; four independent rows per thread overlap their loads)
#define RCN_FB (RCN_FT * RCN_FR)   // rows per k_filter workgroup: one atomic per list per workgroup
__global__ __launch_bounds__(RCN_FT) void k_filter(RerankArgs a)
{
    __shared__ unsigned wcnt[2][RCN_FT / 64];
    __shared__ unsigned wbase[2][RCN_FT / 64];
    const int pl = blockIdx.x / a.qblocks, pair = a.pair_base + pl;
    const int q0 = (blockIdx.x - pl * a.qblocks) * RCN_FB + threadIdx.x;
    const ScaleDev S = *a.sc;
    const ImgDev qi = a.imgs[a.pairs[2 * pair]];
    const ImgDev ti = a.imgs[a.pairs[2 * pair + 1]];
    unsigned survm = 0u, fbm = 0u;      // bit r: row q0 + r * RCN_FT
    double nq2v[RCN_FR];
    uint2 cv[RCN_FR];
#pragma unroll
    for (int r = 0; r < RCN_FR; ++r) {
        const int q = q0 + r * RCN_FT;
        nq2v[r] = 0.0; cv[r] = make_uint2(0u, 0u);
        if (q < qi.K && ti.K >= 2) {
            nq2v[r] = qi.nrm2[q];
            if (!a.all_to_fallback) cv[r] = a.cand[(size_t)pair * a.kq_stride + q];
        }
    }
#pragma unroll
    for (int r = 0; r < RCN_FR; ++r) {
        const int q = q0 + r * RCN_FT;
        bool surv = false, fb = false;
        if (q < qi.K) {
            int32_t res = -1;
            if (ti.K >= 2) {
                const double nq2 = nq2v[r];
                const uint2 c = cv[r];
                // exact kernel at once: no coarse pass at all; a BIG query row (no fp16 copy: fix_scale); fewer than two ordinary train
                // rows (the second candidate is a padding / BIG row, whose accumulator says nothing about its distance)
                if (a.all_to_fallback || !(nq2 < S.thr2) /* also a NaN norm: the exact kernel restates the oracle's arithmetic for it */ || !(__uint_as_float(c.y & ~a.idx_mask) < 1.0e29f)) fb = true;
                else {
                    const double eps = coarse_eps(S, nq2);
                    // the train image's BIG rows never were candidates; every one of them is at least this far from the query
                    const double lb_big = big_lower_bound(ti.bigmin, nq2);
                    const double slack = S.rel_slack * (nq2 + S.n_max * S.n_max);
                    // every row has acc >= trunc(best); both candidates have acc < trunc(second)+quantum
                    const double lo = (double)__uint_as_float(c.x & ~a.idx_mask);
                    const double hi = (double)__uint_as_float((c.y & ~a.idx_mask) + a.idx_mask + 1u);
                    double lb0 = fmin(acc_to_d2(S, nq2, lo - eps) - slack, lb_big);
                    const double ub1 = acc_to_d2(S, nq2, hi + eps) + slack;
                    if (lb0 < 0.0) lb0 = 0.0;
                    surv = ratio_pass(lb0, ub1, a.ratio);
                    // certified PASS from the coarse values alone: the best candidate's exact
                    // distance is <= ub0, every other row's is >= lbnc (acc >= trunc(second)).
                    // sqrtf, the fp32 conversion and the product with a positive ratio are monotone,
                    // so the test holding at (ub0, lbnc) holds for the exact pair; ub0 < lbnc makes
                    // the candidate the unique nearest neighbour.  No distance needs recomputing.
                    if (surv && ti.K > 2) {
                        const double hi0 = (double)__uint_as_float((c.x & ~a.idx_mask) + a.idx_mask + 1u);
                        const double lo1 = (double)__uint_as_float(c.y & ~a.idx_mask);
                        const double ub0 = acc_to_d2(S, nq2, hi0 + eps) + slack;
                        double lbnc = fmin(acc_to_d2(S, nq2, lo1 - eps) - slack, lb_big);
                        if (lbnc < 0.0) lbnc = 0.0;
                        if (ub0 >= 0.0 && ub0 < lbnc && ratio_pass(ub0, lbnc, a.ratio)) {
                            res = (int32_t)(c.x & a.idx_mask);
                            surv = false;
                        }
                    }
                }
            }
            a.out[(size_t)pair * a.out_stride + q] = res;
        }
        survm |= surv ? 1u << r : 0u;
        fbm |= fb ? 1u << r : 0u;
    }
    // workgroup-aggregated append (a single counter word takes only ~88 atomics per microsecond)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned ns = 0, nf = 0;
#pragma unroll
    for (int r = 0; r < RCN_FR; ++r) { ns += (unsigned)__popcll(__ballot(survm >> r & 1u)); nf += (unsigned)__popcll(__ballot(fbm >> r & 1u)); }
    if (lane == 0) { wcnt[0][w] = ns; wcnt[1][w] = nf; }
    __syncthreads();
    if (threadIdx.x < 2) {
        unsigned tot = 0;
        for (int i = 0; i < nw; ++i) { wbase[threadIdx.x][i] = tot; tot += wcnt[threadIdx.x][i]; }
        const unsigned base = tot ? atomicAdd(threadIdx.x == 0 ? a.sv_count : a.fb_count, tot) : 0u;
        for (int i = 0; i < nw; ++i) wbase[threadIdx.x][i] += base;
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned bs = wbase[0][w], bf = wbase[1][w];
#pragma unroll
    for (int r = 0; r < RCN_FR; ++r) {
        const unsigned long long entry = ((unsigned long long)pair << 32) | (unsigned)(q0 + r * RCN_FT);
        const unsigned long long ms = __ballot(survm >> r & 1u), mf = __ballot(fbm >> r & 1u);
        if (survm >> r & 1u) a.sv_list[bs + __popcll(ms & lt)] = entry;
        if (fbm >> r & 1u) a.fb_list[bf + __popcll(mf & lt)] = entry;
        bs += (unsigned)__popcll(ms); bf += (unsigned)__popcll(mf);
    }
}

// K2: exact fp64 re-rank of the survivors.  Only the BEST coarse candidate is re-computed:
// every other row -- the second candidate included -- has accumulator >= trunc(second), hence
// exact distance >= lbnc, so  ea < lbnc  proves the candidate is the nearest neighbour and
// ratio_pass(ea, lbnc) proves the ratio test for whatever the true second distance is.  Rows
// that this does not certify (a few hundred in ten million) go to the exact kernel.
// One wave = 64 survivors.  Rows are staged through LDS in 32-float chunks so that global reads
// are whole 128-B segments (8 lanes per row) instead of 64 lanes striding 1-KiB rows; each
// lane then walks its (query, candidate) chain in ascending k out of LDS.
// (round 4, second session) BOTH coarse candidates go through the chain: with the second exact distance in hand `certify` can also
// say "no match" -- every row but the two candidates is at least lbnc away, so the exact second neighbour lies in [min(eb, lbnc), eb],
// and a ratio test that fails against eb fails for whatever it is -- where round 3 sent every row it could not PASS to the next tier.
#define RR_ROWS 192
#define RR_LD 36  // floats per LDS row: 32 + 4 pad (conflict-free ds_read_b128 by row)
__global__ __launch_bounds__(64) void k_rerank_lds(RerankArgs a)
{
    __shared__ __attribute__((aligned(16))) float tile[RR_ROWS * RR_LD];
    __shared__ const float *rowptr[RR_ROWS];
    const int lane = threadIdx.x;
    const unsigned n = *a.sv_count;
    const int D = a.D;
    const int nchunk = (D + 31) / 32;
    const ScaleDev S = *a.sc;
    for (unsigned g = blockIdx.x; g * 64u < n; g += gridDim.x) {
        const unsigned sidx = g * 64u + lane;
        const bool live = sidx < n;
        const unsigned long long e = a.sv_list[live ? sidx : g * 64u];
        const int pair = (int)(e >> 32), q = (int)(e & 0xFFFFFFFFu);
        const ImgDev qi = a.imgs[a.pairs[2 * pair]];
        const ImgDev ti = a.imgs[a.pairs[2 * pair + 1]];
        const uint2 c = a.cand[(size_t)pair * a.kq_stride + q];
        int ia = (int)(c.x & a.idx_mask), ib = (int)(c.y & a.idx_mask);
        if (ib >= ti.K) ib = ia;      // (a padding row as second candidate: k_filter keeps such rows off this list; never an address)
        __syncthreads();  // previous group's reads of rowptr/tile are done
        rowptr[lane] = qi.f32 + (size_t)q * D;
        rowptr[64 + lane] = ti.f32 + (size_t)ia * D;
        rowptr[128 + lane] = ti.f32 + (size_t)ib * D;
        __syncthreads();
        double acc = 0.0, acc2 = 0.0;
        const float *qrow = tile + lane * RR_LD;
        const float *trow = tile + (64 + lane) * RR_LD;
        const float *trow2 = tile + (128 + lane) * RR_LD;
        for (int ch = 0; ch < nchunk; ++ch) {
            const int col = ch * 32 + (lane & 7) * 4;
#pragma unroll
            for (int i = 0; i < RR_ROWS / 8; ++i) {
                const int r = i * 8 + (lane >> 3);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col < D) v = *reinterpret_cast<const float4 *>(rowptr[r] + col);
                *reinterpret_cast<float4 *>(tile + r * RR_LD + (lane & 7) * 4) = v;
            }
            __syncthreads();
            const int kmax = min(32, D - ch * 32);
            for (int k4 = 0; k4 < kmax; k4 += 4) {
                const float4 x = *reinterpret_cast<const float4 *>(qrow + k4);
                const float4 y = *reinterpret_cast<const float4 *>(trow + k4);
                const float4 z = *reinterpret_cast<const float4 *>(trow2 + k4);
                double d;
                d = (double)x.x - (double)y.x; acc = fma(d, d, acc);
                d = (double)x.y - (double)y.y; acc = fma(d, d, acc);
                d = (double)x.z - (double)y.z; acc = fma(d, d, acc);
                d = (double)x.w - (double)y.w; acc = fma(d, d, acc);
                d = (double)x.x - (double)z.x; acc2 = fma(d, d, acc2);
                d = (double)x.y - (double)z.y; acc2 = fma(d, d, acc2);
                d = (double)x.z - (double)z.z; acc2 = fma(d, d, acc2);
                d = (double)x.w - (double)z.w; acc2 = fma(d, d, acc2);
            }
            __syncthreads();
        }
        bool fb = false;
        if (live) {
            // the two candidates ordered by (exact value, index), as the oracle's scan would meet them
            double ea = acc, eb = acc2;
            if (ib != ia && (eb < ea || (eb == ea && ib < ia))) { ea = acc2; eb = acc; const int t = ia; ia = ib; ib = t; }
            int res = -2;
            if (ti.K > 2 && ib != ia) {
                const double nq2 = qi.nrm2[q];
                const double lbacc = (double)__uint_as_float(c.y & ~a.idx_mask);
                const double lbnc = fmin(acc_to_d2(S, nq2, lbacc - coarse_eps(S, nq2)) - S.rel_slack * (nq2 + S.n_max * S.n_max), big_lower_bound(ti.bigmin, nq2));
                res = certify(ea, ia, eb, lbnc, a.ratio);
            }
            if (res >= 0) a.out[(size_t)pair * a.out_stride + q] = res;
            else if (res == -2) fb = true;   // out stays -1 (which is also the certified "no match") until the next tier decides
        }
        list_append(a.fb_list, a.fb_count, fb, e);
    }
}

// generic survivor re-rank (D not a multiple of 4): one thread per survivor
__global__ void k_rerank_generic(RerankArgs a)
{
    const unsigned n = *a.sv_count;
    const ScaleDev S = *a.sc;
    for (unsigned s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        const unsigned long long e = a.sv_list[s];
        const int pair = (int)(e >> 32), q = (int)(e & 0xFFFFFFFFu);
        const ImgDev qi = a.imgs[a.pairs[2 * pair]];
        const ImgDev ti = a.imgs[a.pairs[2 * pair + 1]];
        const uint2 c = a.cand[(size_t)pair * a.kq_stride + q];
        int ia = (int)(c.x & a.idx_mask), ib = (int)(c.y & a.idx_mask);
        const float *qrow = qi.f32 + (size_t)q * a.D;
        double ea = exact_d2<false>(qrow, ti.f32 + (size_t)ia * a.D, a.D);
        double eb = exact_d2<false>(qrow, ti.f32 + (size_t)ib * a.D, a.D);
        if (eb < ea || (eb == ea && ib < ia)) { double te = ea; ea = eb; eb = te; ia = ib; }
        double lbnc = INFINITY;
        if (ti.K > 2) {
            const double nq2 = qi.nrm2[q];
            const double lbacc = (double)__uint_as_float(c.y & ~a.idx_mask);
            lbnc = fmin(acc_to_d2(S, nq2, lbacc - coarse_eps(S, nq2)) - S.rel_slack * (nq2 + S.n_max * S.n_max), big_lower_bound(ti.bigmin, nq2));
        }
        int res = certify(ea, ia, eb, lbnc, a.ratio);
        if (res == -2) {
            unsigned slot = atomicAdd(a.fb_count, 1u);
            a.fb_list[slot] = e;
            res = -1;
        }
        a.out[(size_t)pair * a.out_stride + q] = res;
    }
}

// K2b: exact brute force of listed rows, one workgroup (4 waves) per row, grid-stride over the
// list; each thread walks the canonical chain of its train rows, then (value, index)-ordered
// top-2 pairs are merged across the wave and across the 4 waves.
__device__ __forceinline__ void merge_top2(double &b0, int &i0, double &b1, int &i1, double c0, int j0, double c1, int j1)
{
    const bool mine = (b0 < c0) || (b0 == c0 && i0 < j0);
    const double f0 = mine ? b0 : c0; const int fi0 = mine ? i0 : j0;
    const double l0 = mine ? c0 : b0; const int li0 = mine ? j0 : i0;   // the loser of the firsts
    const double s0 = mine ? b1 : c1; const int si0 = mine ? i1 : j1;   // the winner's own second
    const bool pick = (l0 < s0) || (l0 == s0 && li0 < si0);
    b0 = f0; i0 = fi0;
    b1 = pick ? l0 : s0; i1 = pick ? li0 : si0;
}

template <bool VEC4>
__global__ __launch_bounds__(256) void k_exact_rows(const ImgDev *__restrict__ imgs,
                                                     const int32_t *__restrict__ pairs,
                                                     const unsigned long long *__restrict__ list,
                                                     const unsigned *__restrict__ count, int D,
                                                     float ratio, int32_t *__restrict__ out,
                                                     int64_t out_stride, unsigned skip = 0)
{
    __shared__ double sb[4][2];
    __shared__ int si[4][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned n = *count;
    for (unsigned it = skip + blockIdx.x; it < n; it += gridDim.x) {
        const unsigned long long e = list[it];
        const int pair = (int)(e >> 32), q = (int)(e & 0xFFFFFFFFu);
        const ImgDev qi = imgs[pairs[2 * pair]];
        const ImgDev ti = imgs[pairs[2 * pair + 1]];
        const float *qrow = qi.f32 + (size_t)q * D;
        double b0 = INFINITY, b1 = INFINITY;
        int i0 = 0x7FFFFFFF, i1 = 0x7FFFFFFF;
        for (int j = threadIdx.x; j < ti.K; j += 256) {
            double d = exact_d2<VEC4>(qrow, ti.f32 + (size_t)j * D, D);
            if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }          // ascending j: strict <
            else if (d < b1) { b1 = d; i1 = j; }
        }
        for (int o = 32; o; o >>= 1) {
            const double c0 = __shfl_xor(b0, o), c1 = __shfl_xor(b1, o);
            const int j0 = __shfl_xor(i0, o), j1 = __shfl_xor(i1, o);
            merge_top2(b0, i0, b1, i1, c0, j0, c1, j1);
        }
        __syncthreads();
        if (lane == 0) { sb[w][0] = b0; sb[w][1] = b1; si[w][0] = i0; si[w][1] = i1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; ++k) merge_top2(b0, i0, b1, i1, sb[k][0], si[k][0], sb[k][1], si[k][1]);
            out[(size_t)pair * out_stride + q] = ratio_pass(b0, b1, ratio) ? i0 : -1;
        }
    }
}

// K2b for D % 4 == 0: same result as k_exact_rows, but train rows reach the lanes through LDS.
// Each wave stages 64 consecutive train rows (and the query row) in 32-float chunks with whole
// 128-B global segments (8 lanes per row) and every lane walks its own row's chain out of LDS,
// carrying the fp64 accumulator across chunks -- the row-per-lane global access pattern of the
// simple kernel is latency-bound (0.22 ms for 351 rows at cfg 2).
#define EX_LD 36   // floats per staged row: 32 + 4 pad
#define EX_WAVES 8
__global__ __launch_bounds__(64 * EX_WAVES) void k_exact_rows_lds(const ImgDev *__restrict__ imgs,
                                                         const int32_t *__restrict__ pairs,
                                                         const unsigned long long *__restrict__ list,
                                                         const unsigned *__restrict__ count, int D,
                                                         float ratio, int32_t *__restrict__ out,
                                                         int64_t out_stride, unsigned skip = 0)
{
    __shared__ __attribute__((aligned(16))) float tile[EX_WAVES][65 * EX_LD];
    __shared__ double sb[EX_WAVES][2];
    __shared__ int si[EX_WAVES][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned n = *count;
    const int nchunk = (D + 31) / 32;
    float *tw = tile[w];
    for (unsigned it = skip + blockIdx.x; it < n; it += gridDim.x) {      // skip: the leading entries went through the middle tier
        const unsigned long long e = list[it];
        const int pair = (int)(e >> 32), q = (int)(e & 0xFFFFFFFFu);
        const ImgDev qi = imgs[pairs[2 * pair]];
        const ImgDev ti = imgs[pairs[2 * pair + 1]];
        const float *qrow = qi.f32 + (size_t)q * D;
        double b0 = INFINITY, b1 = INFINITY;
        int i0 = 0x7FFFFFFF, i1 = 0x7FFFFFFF;
        for (int base0 = 0; base0 < ti.K; base0 += 64 * EX_WAVES) {      // uniform trip count over the waves
            const int base = base0 + w * 64;
            const int j = base + lane;
            double acc = 0.0;
            for (int ch = 0; ch < nchunk; ++ch) {
                const int col = ch * 32 + (lane & 7) * 4;
                float4 v[9];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = base + i * 8 + (lane >> 3);
                    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (col < D && r < ti.K) v[i] = *reinterpret_cast<const float4 *>(ti.f32 + (size_t)r * D + col);
                }
                v[8] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane < 8 && col < D) v[8] = *reinterpret_cast<const float4 *>(qrow + col);
                __syncthreads();                                // previous chunk's LDS reads are done
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<float4 *>(tw + (i * 8 + (lane >> 3)) * EX_LD + (lane & 7) * 4) = v[i];
                if (lane < 8) *reinterpret_cast<float4 *>(tw + 64 * EX_LD + lane * 4) = v[8];
                __syncthreads();
                const int kmax = min(32, D - ch * 32);
                const float *trow = tw + lane * EX_LD, *qs = tw + 64 * EX_LD;
                for (int k4 = 0; k4 < kmax; k4 += 4) {
                    const float4 x = *reinterpret_cast<const float4 *>(qs + k4);
                    const float4 y = *reinterpret_cast<const float4 *>(trow + k4);
                    double d;
                    d = (double)x.x - (double)y.x; acc = fma(d, d, acc);
                    d = (double)x.y - (double)y.y; acc = fma(d, d, acc);
                    d = (double)x.z - (double)y.z; acc = fma(d, d, acc);
                    d = (double)x.w - (double)y.w; acc = fma(d, d, acc);
                }
            }
            if (j < ti.K) {
                if (acc < b0) { b1 = b0; i1 = i0; b0 = acc; i0 = j; }   // ascending j per lane: strict <
                else if (acc < b1) { b1 = acc; i1 = j; }
            }
        }
        for (int o = 32; o; o >>= 1) {
            const double c0 = __shfl_xor(b0, o), c1 = __shfl_xor(b1, o);
            const int j0 = __shfl_xor(i0, o), j1 = __shfl_xor(i1, o);
            merge_top2(b0, i0, b1, i1, c0, j0, c1, j1);
        }
        __syncthreads();
        if (lane == 0) { sb[w][0] = b0; sb[w][1] = b1; si[w][0] = i0; si[w][1] = i1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < EX_WAVES; ++k) merge_top2(b0, i0, b1, i1, sb[k][0], si[k][0], sb[k][1], si[k][1]);
            out[(size_t)pair * out_stride + q] = ratio_pass(b0, b1, ratio) ? i0 : -1;
        }
    }
}

// ---- middle tier (round 4): the rows K2 could not certify, between "one candidate" and "every train row in fp64" -------
// An uncertified row needs its EXACT two nearest neighbours.  Round 3 sent it straight to K2b: one workgroup per row, all
// K2 train rows through the fp64 chain -- and, what costs more, all K2 x D x 4 bytes of the train image read for ONE query
// (cfg 3: 114 476 rows x 4 MB = 480 GB per step through L2 / Infinity Cache: 68.8 ms).  Here the rows are binned by TRAIN
// image, so that sixteen of them share one sweep over that image, and the sweep runs in fp32 (sub + fma per element, eight
// times the fp64 chain's rate).  fp32 values cannot decide anything, but they can EXCLUDE: with
//     g = (D + 8) 2^-23   (relative error of a D-term fp32 sum of squares of fp32 differences, with room)
// every row whose exact distance is <= U has fp32 value <= U (1 + g) + tiny, so the rows with value <= thr = U (1 + g) + tiny
// are a superset of the rows within U.  U = an upper bound of the exact SECOND-nearest distance:
//   * from the coarse pass when the row has one (the second candidate's accumulator + quantum + eps, as k_filter's ub1):
//     known before the sweep -- one sweep, candidates emitted on the way;
//   * otherwise (BIG query rows, train images with fewer than two ordinary rows, grids without a coarse pass): the second
//     smallest fp32 value of a first sweep, inflated by the same error model -- two sweeps.
// The candidates (a handful per row; the list holds RCN_MIDCAP) then go through the canonical fp64 chain and the
// (value, index)-ordered top-2 of THOSE is the top-2 of all rows: same bits as K2b.  A list that overflows, and whatever
// exceeds the tier's row budget, still goes to K2b.
#define RCN_MIDCAP 32          // candidates kept per row
#define RCN_MIDQ 16            // query rows per sweep of a train image
#define RCN_MIDROWS (1 << 21)  // rows per pipeline chunk the tier takes (the rest: K2b)
struct MidArgs {
    const ImgDev *imgs;
    const int32_t *pairs;
    const uint2 *cand;                     // coarse candidates (NULL / all_to_fallback: none)
    const unsigned long long *fb_list;     // (pair << 32 | query) in discovery order
    const unsigned *fb_count;
    unsigned long long *sorted;            // the same entries binned by train image
    float *thr;                            // per sorted row: fp32 threshold, +infinity = not known before the sweep
    unsigned *ccount;                      // per sorted row: candidates found
    int32_t *clist;                        // [rows][RCN_MIDCAP] train rows
    unsigned *hist, *offs, *cursor, *ibase; // per image slot (+ 1): rows, first sorted row, fill cursor, first work item
    unsigned *n_items;
    const float *thr_in;                   // deferred rows (k_mid_defer): their thresholds, computed while their chunk's candidate table was there
    unsigned long long *fb2_list;          // overflowed rows -> K2b
    unsigned *fb2_count;
    int32_t *out;
    int64_t out_stride;
    const ScaleDev *sc;
    int32_t n_slots, kq_stride, D, all_to_fallback;
    uint32_t idx_mask, midrows;            // midrows: rows per pipeline chunk the tier takes
    float ratio;
};
__device__ __forceinline__ unsigned mid_rows(const MidArgs &a) { const unsigned n = *a.fb_count; return n < a.midrows ? n : a.midrows; }

// one atomic per distinct key of a wave (the list comes in runs of one pair: a wave usually holds ONE train image, and a single
// word takes only ~88 atomics per microsecond): returns the value of the counter before this wave's rows + the lane's rank
__device__ __forceinline__ unsigned wave_count(unsigned *counters, int key, bool live)
{
    unsigned res = 0;
    unsigned long long todo = __ballot(live);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader);
        const unsigned long long peers = __ballot(live && key == k) & todo;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(counters + k, (unsigned)__popcll(peers));
        base = __shfl(base, leader);
        if (peers >> lane & 1ull) res = base + (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        todo &= ~peers;
    }
    return res;
}
__global__ __launch_bounds__(256) void k_mid_hist(MidArgs a)
{
    const unsigned n = mid_rows(a);
    for (unsigned i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const unsigned i = i0 + threadIdx.x;
        const bool live = i < n;
        const int ts = live ? a.pairs[2 * (int)(a.fb_list[i] >> 32) + 1] : 0;
        (void)wave_count(a.hist, ts, live);
    }
}
// one workgroup: exclusive scans of the bins -- rows (offs) and work items (ibase: a train image x up to RCN_MIDQ rows is one item;
// k_mid_eval finds the bin of item number i by bisection) -- cursors and bins reset
__global__ __launch_bounds__(1024) void k_mid_bins(MidArgs a)
{
    __shared__ unsigned s1[1024], s2[1024];
    const int t = threadIdx.x, per = (a.n_slots + 1023) / 1024;
    const int lo = min(a.n_slots, t * per), hi = min(a.n_slots, lo + per);
    unsigned rows = 0, its = 0;
    for (int i = lo; i < hi; ++i) { const unsigned h = a.hist[i]; rows += h; its += (h + RCN_MIDQ - 1) / RCN_MIDQ; }
    s1[t] = rows; s2[t] = its;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned v1 = t >= o ? s1[t - o] : 0u, v2 = t >= o ? s2[t - o] : 0u;
        __syncthreads();
        s1[t] += v1; s2[t] += v2;
        __syncthreads();
    }
    unsigned r = s1[t] - rows, k = s2[t] - its;
    for (int i = lo; i < hi; ++i) {
        const unsigned h = a.hist[i];
        a.offs[i] = r; a.ibase[i] = k; a.cursor[i] = 0u; a.hist[i] = 0u;
        r += h; k += (h + RCN_MIDQ - 1) / RCN_MIDQ;
    }
    if (t == 1023) { *a.n_items = s2[1023]; a.offs[a.n_slots] = s1[1023]; a.ibase[a.n_slots] = s2[1023]; }
}
// fp32 threshold of the one-sweep path for row (pair, q), from the chunk's candidate table; +infinity: none (two sweeps)
__device__ __forceinline__ float mid_threshold(const MidArgs &a, const ScaleDev &S, int pair, int q)
{
    float thr = INFINITY;
    if (!a.all_to_fallback) {
        const double g = (double)(a.D + 8) * 1.1920928955078125e-7;
        const ImgDev qi = a.imgs[a.pairs[2 * pair]];
        const double nq2 = qi.nrm2[q];
        const uint2 c = a.cand[(size_t)pair * a.kq_stride + q];
        if (nq2 < S.thr2 && __uint_as_float(c.y & ~a.idx_mask) < 1.0e29f) {
            // two ordinary rows (the coarse candidates) are within ub1 of the query: so is the exact second neighbour
            const double hi = (double)__uint_as_float((c.y & ~a.idx_mask) + a.idx_mask + 1u);
            const double ub1 = acc_to_d2(S, nq2, hi + coarse_eps(S, nq2)) + S.rel_slack * (nq2 + S.n_max * S.n_max);
            if (ub1 >= 0.0) {
                const double tv = (ub1 * (1.0 + g) + 1.0e-30) * (1.0 + 1.0e-6);
                thr = tv < 3.0e38 ? (float)tv : INFINITY;      // a threshold that does not fit fp32 decides nothing: two sweeps
            }
        }
    }
    return thr;
}
__global__ __launch_bounds__(256) void k_mid_scatter(MidArgs a)
{
    const unsigned n = mid_rows(a);
    const ScaleDev S = *a.sc;
    for (unsigned i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const unsigned i = i0 + threadIdx.x;
        const bool live = i < n;
        const unsigned long long e = live ? a.fb_list[i] : 0ull;
        const int pair = (int)(e >> 32), q = (int)(e & 0xFFFFFFFFu);
        const int ts = live ? a.pairs[2 * pair + 1] : 0;
        const unsigned rank = wave_count(a.cursor, ts, live);
        if (!live) continue;
        const unsigned pos = a.offs[ts] + rank;
        a.sorted[pos] = e;
        a.thr[pos] = a.thr_in ? a.thr_in[i] : mid_threshold(a, S, pair, q);
        a.ccount[pos] = 0u;
    }
}
// Deferral (round 4): on a grid of several pipeline chunks the rows a chunk leaves for this tier are few per TRAIN image (cfg 3,
// sixteen chunks: two per image and chunk), and a sweep over a train image costs the same for one row as for sixteen.  So the
// chunk's rows -- with the one thing the tier needs of the chunk's candidate table, the threshold -- join ONE list of the call, as
// long as it has room (cap rows; a chunk that does not fit goes through the tier at once, as before), and the tier runs once, behind
// the last chunk, where an image's rows of all chunks share its sweeps.  c = the chunk counters (c[0] rows, c[7] rows deferred so far).
__global__ __launch_bounds__(256) void k_mid_defer(MidArgs a, unsigned long long *def_e, float *def_thr, const unsigned *c, unsigned cap)
{
    const unsigned n = c[0], base = c[7];
    if (n == 0 || base + n > cap || base + n < base) return;
    const ScaleDev S = *a.sc;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long e = a.fb_list[i];
        def_e[base + i] = e;
        def_thr[base + i] = mid_threshold(a, S, (int)(e >> 32), (int)(e & 0xFFFFFFFFu));
    }
}
// (one thread, behind k_mid_defer: the same test; the rows count as fallback rows of the call and leave the chunk's list)
__global__ void k_mid_defer_commit(unsigned *c, unsigned cap)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned n = c[0], base = c[7];
    if (n == 0 || base + n > cap || base + n < base) return;
    c[7] = base + n;
    c[4] += n;
    c[0] = 0u;
}
// (one thread, behind the deferred pass: what it passed on to K2b)
__global__ void k_stats_acc_deferred(unsigned *c)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) c[6] += c[2];
}
// One workgroup (256 threads) per work item.  A sweep: the train image in tiles of 256 rows (one per thread), every tile in
// 32-float chunks through LDS with whole 128-byte global segments, the item's query rows beside them; thread t accumulates its train row against each query in fp32, two
// elements per instruction (v_pk_fma_f32).  Two forms of the value:
//   DIRECT (MODE 0)   v = sum (x - y)^2            relative error g: sharp for near neighbours; the form of the one-sweep path,
//                                                  whose threshold comes from the coarse pass
//   PRODUCT (MODE 1, 2)  v = |t|^2 - 2 sum x y     = d2 - |q|^2, absolute error E(q, t) ~ g |q| |t|: the form of the rows that come
//                     without a threshold.  A query a thousand million times larger than the train rows (a BIG query row,
//                     fix_scale) has d2 = |q|^2 (1 + O(1e-9)) for EVERY train row: the direct form cannot tell them apart
//                     in fp32 (all of them would be candidates), the product form orders them by q.t as well as it orders unit rows.
// MODE 0 / 2 emit the rows under the query's threshold into its candidate list (MODE 2: v - E(q, t) <= U, the row's own error
// bound); MODE 1 keeps the two smallest UPPER bounds v + E(q, t) the thread saw per query, over the first `rows` train rows only.
#define MID_LD 36
// |v - (d2 - |q|^2)| <= E: the fp32 dot product ((D + 2) 2^-24 |q| |t|, doubled by the factor 2, with room), the rounding of |t|^2 to
// fp32 and of the final fma, and the rounding of the oracle's own fp64 chain (the ORDER to reproduce is that of its rounded values)
__device__ __forceinline__ float mid_err(int D, float qn, float tn)
{
    const float s = qn + tn;
    return ((float)(D + 8) * 2.3841858e-7f * qn * tn + 4.7683716e-7f * tn * tn + (float)(D + 8) * 2.220446e-16f * s * s) * 1.0001f + 1.0e-30f;
}
template <int MODE>
__device__ __forceinline__ void mid_sweep(const MidArgs &a, const ImgDev &ti, const float *const *qptr, int nq, int first, const float *thr, const float *qnorm,
                                          int rows, float *tt, float *tq, float (&m1)[RCN_MIDQ], float (&m2)[RCN_MIDQ])
{
    const int t = threadIdx.x, D = a.D, nchunk = (D + 31) / 32;
    const int ntile = (rows + 255) / 256, nit = ntile * nchunk;
    float4 v[8], vq;
    auto fetch = [&](int it) {
        const int base = (it / nchunk) * 256, col = (it % nchunk) * 32 + (t & 7) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = base + i * 32 + (t >> 3);
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < D && r < rows) v[i] = *reinterpret_cast<const float4 *>(ti.f32 + (size_t)r * D + col);
        }
        vq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < 8 * RCN_MIDQ && (t >> 3) < nq && col < D) vq = *reinterpret_cast<const float4 *>(qptr[t >> 3] + col);
    };
    float acc[RCN_MIDQ];
    for (int it = 0; it < nit; ++it) {
        const int ch = it % nchunk, base = (it / nchunk) * 256;
        if (ch == 0) {
#pragma unroll
            for (int q = 0; q < RCN_MIDQ; ++q) acc[q] = 0.f;
        }
        fetch(it);                                          // (a prefetch of chunk it + 1 behind the arithmetic costs 32 registers and, with them, three of
                                                            //  the four workgroups a CU holds: 5.9 ms instead of 3.8 on the mixed-magnitudes grid; packed fp32
                                                            //  -- v_pk_fma_f32 over pairs of elements -- doubles the accumulators and spills: no faster either)
        __syncthreads();                                    // the previous chunk's LDS reads are done
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4 *>(tt + (i * 32 + (t >> 3)) * MID_LD + (t & 7) * 4) = v[i];
        if (t < 8 * RCN_MIDQ) *reinterpret_cast<float4 *>(tq + (t >> 3) * MID_LD + (t & 7) * 4) = vq;
        __syncthreads();
        float y[32];
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const float4 w4 = *reinterpret_cast<const float4 *>(tt + t * MID_LD + 4 * k4);
            y[4 * k4] = w4.x; y[4 * k4 + 1] = w4.y; y[4 * k4 + 2] = w4.z; y[4 * k4 + 3] = w4.w;
        }
#pragma unroll
        for (int q = 0; q < RCN_MIDQ; ++q) {
            if (q >= nq) break;                               // nq is uniform over the workgroup
            float s = acc[q];
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const float4 x = *reinterpret_cast<const float4 *>(tq + q * MID_LD + 4 * k4);      // one address for the whole wave: a broadcast
                if (MODE == 0) {
                    float d;
                    d = x.x - y[4 * k4]; s = fmaf(d, d, s);
                    d = x.y - y[4 * k4 + 1]; s = fmaf(d, d, s);
                    d = x.z - y[4 * k4 + 2]; s = fmaf(d, d, s);
                    d = x.w - y[4 * k4 + 3]; s = fmaf(d, d, s);
                } else {
                    s = fmaf(x.x, y[4 * k4], s); s = fmaf(x.y, y[4 * k4 + 1], s);
                    s = fmaf(x.z, y[4 * k4 + 2], s); s = fmaf(x.w, y[4 * k4 + 3], s);
                }
            }
            acc[q] = s;
        }
        if (ch != nchunk - 1) continue;
        const int j = base + t;
        if (j < rows) {
            const float nt = MODE == 0 ? 0.f : (float)ti.nrm2[j];
            const float tn = MODE == 0 ? 0.f : sqrtf(nt) * 1.000001f;
#pragma unroll
            for (int q = 0; q < RCN_MIDQ; ++q) {
                if (q >= nq) break;
                const float dot = acc[q];
                const float val = MODE == 0 ? dot : fmaf(-2.f, dot, nt);
                if (MODE == 0) {
                    if (val <= thr[q]) {
                        const unsigned pos = atomicAdd(a.ccount + first + q, 1u);
                        if (pos < (unsigned)RCN_MIDCAP) a.clist[(size_t)(first + q) * RCN_MIDCAP + pos] = j;
                    }
                } else {
                    const float E = mid_err(D, qnorm[q], tn);
                    if (MODE == 2) {
                        // (a NaN value fails the test: such a row is never a neighbour, K2b would not select it either; an infinite
                        //  bound lets everything through, the list overflows and K2b decides)
                        if (val - E <= thr[q]) {
                            const unsigned pos = atomicAdd(a.ccount + first + q, 1u);
                            if (pos < (unsigned)RCN_MIDCAP) a.clist[(size_t)(first + q) * RCN_MIDCAP + pos] = j;
                        }
                    } else {
                        const float ub = val + E;            // the row's exact value is at most this
                        if (ub < m1[q]) { m2[q] = m1[q]; m1[q] = ub; }
                        else if (ub < m2[q]) m2[q] = ub;
                    }
                }
            }
        }
    }
}
__global__ __launch_bounds__(256, 4) void k_mid_eval(MidArgs a)
{
    __shared__ __attribute__((aligned(16))) float tt[256 * MID_LD];
    __shared__ __attribute__((aligned(16))) float tq[RCN_MIDQ * MID_LD];
    __shared__ const float *qptr[RCN_MIDQ];
    __shared__ float thr[RCN_MIDQ], qnorm[RCN_MIDQ];
    __shared__ float red[4][RCN_MIDQ][2];
    __shared__ int s_need2, s_bin;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned n_items = *a.n_items;
    for (unsigned it = blockIdx.x; it < n_items; it += gridDim.x) {
        __syncthreads();                                        // the previous item's readers of qptr / thr are done
        if (t == 0) {
            int lo = 0, hi = a.n_slots;                         // the last bin b with ibase[b] <= it (bins without rows share their successor's base)
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.ibase[mid] <= it) lo = mid; else hi = mid; }
            s_bin = lo;
            s_need2 = 0;
        }
        __syncthreads();
        const int bin = s_bin;
        const unsigned k = it - a.ibase[bin], h = a.offs[bin + 1] - a.offs[bin];
        const int first = (int)(a.offs[bin] + RCN_MIDQ * k), nq = (int)min((unsigned)RCN_MIDQ, h - RCN_MIDQ * k);
        const ImgDev ti = a.imgs[bin];
        if (t < nq) {
            const unsigned long long e = a.sorted[first + t];
            const int pair = (int)(e >> 32), q = (int)(e & 0xFFFFFFFFu);
            const ImgDev qi = a.imgs[a.pairs[2 * pair]];
            qptr[t] = qi.f32 + (size_t)q * a.D;
            qnorm[t] = (float)(sqrt(qi.nrm2[q]) * (1.0 + 1e-6));
            const float th = a.thr[first + t];
            thr[t] = th;
            if (!(th < INFINITY)) s_need2 = 1;
        }
        __syncthreads();
        float m1[RCN_MIDQ], m2[RCN_MIDQ];
        if (!s_need2) {
            mid_sweep<0>(a, ti, qptr, nq, first, thr, qnorm, ti.K, tt, tq, m1, m2);
            continue;
        }
        // No bound on the second neighbour before the sweep.  ANY two rows give one: the second smallest upper bound over a
        // prefix of the train image (a quarter of it, at least 512 rows: the expected number of rows of the whole image under
        // that bound is about eight), then one sweep over the whole image against it.
        const int prefix = min(ti.K, max(512, ti.K / 4));
#pragma unroll
        for (int q = 0; q < RCN_MIDQ; ++q) { m1[q] = INFINITY; m2[q] = INFINITY; }
        mid_sweep<1>(a, ti, qptr, nq, first, thr, qnorm, prefix, tt, tq, m1, m2);
#pragma unroll
        for (int q = 0; q < RCN_MIDQ; ++q) {
            float b1 = m1[q], b2 = m2[q];
            for (int o = 32; o; o >>= 1) {
                const float c1 = __shfl_xor(b1, o), c2 = __shfl_xor(b2, o);
                const float lo = fminf(b1, c1), hi = fmaxf(b1, c1);
                b2 = fminf(hi, fminf(b2, c2)); b1 = lo;
            }
            if (lane == 0) { red[w][q][0] = b1; red[w][q][1] = b2; }
        }
        __syncthreads();
        if (t < nq) {
            float b1 = red[0][t][0], b2 = red[0][t][1];
            for (int kk = 1; kk < 4; ++kk) {
                const float c1 = red[kk][t][0], c2 = red[kk][t][1];
                const float lo = fminf(b1, c1), hi = fmaxf(b1, c1);
                b2 = fminf(hi, fminf(b2, c2)); b1 = lo;
            }
            // two rows have exact values <= b2: so has the second neighbour.  A row within that has v - E(q, t) <= b2.
            thr[t] = b2 < INFINITY ? b2 + fabsf(b2) * 1.0e-6f : INFINITY;      // (a NaN bound compares false: INFINITY, K2b decides)
        }
        __syncthreads();
        mid_sweep<2>(a, ti, qptr, nq, first, thr, qnorm, ti.K, tt, tq, m1, m2);
    }
}
// one wave per row of the tier: the canonical fp64 chain of its candidates, (value, index)-ordered top-2, ratio test
__global__ __launch_bounds__(256) void k_mid_exact(MidArgs a)
{
    const unsigned n = mid_rows(a);
    const int lane = threadIdx.x & 63;
    for (unsigned r = blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += gridDim.x * 4) {
        const unsigned long long e = a.sorted[r];
        const int pair = (int)(e >> 32), q = (int)(e & 0xFFFFFFFFu);
        const unsigned cnt = a.ccount[r];
        if (cnt > (unsigned)RCN_MIDCAP) {                       // more rows under the threshold than the list holds: every train row, K2b
            if (lane == 0) a.fb2_list[atomicAdd(a.fb2_count, 1u)] = e;
            continue;
        }
        const ImgDev qi = a.imgs[a.pairs[2 * pair]];
        const ImgDev ti = a.imgs[a.pairs[2 * pair + 1]];
        double b0 = INFINITY, b1 = INFINITY;
        int i0 = 0x7FFFFFFF, i1 = 0x7FFFFFFF;
        if (lane < (int)cnt) {
            i0 = a.clist[(size_t)r * RCN_MIDCAP + lane];
            b0 = exact_d2<true>(qi.f32 + (size_t)q * a.D, ti.f32 + (size_t)i0 * a.D, a.D);
            if (!(b0 < INFINITY)) { b0 = INFINITY; i0 = 0x7FFFFFFF; }      // K2b never selects a row whose distance is not below +infinity
        }
        for (int o = 32; o; o >>= 1) {
            const double c0 = __shfl_xor(b0, o), c1 = __shfl_xor(b1, o);
            const int j0 = __shfl_xor(i0, o), j1 = __shfl_xor(i1, o);
            merge_top2(b0, i0, b1, i1, c0, j0, c1, j1);
        }
        if (lane == 0) a.out[(size_t)pair * a.out_stride + q] = ratio_pass(b0, b1, a.ratio) ? i0 : -1;
    }
}

// K3: uniqueness (FeatureMatcher.cpp:58-62): ascending query order, first claim wins
//     == the smallest claiming query index per train row.
__global__ void k_unique_claim(const ImgDev *__restrict__ imgs, const int32_t *__restrict__ pairs,
                               const int32_t *__restrict__ out, int64_t out_stride,
                               int32_t *__restrict__ owner, int owner_stride, int qblocks, int pair_base)
{
    const int pl = blockIdx.x / qblocks, pair = pair_base + pl;
    const int q = (blockIdx.x - pl * qblocks) * blockDim.x + threadIdx.x;
    if (q >= imgs[pairs[2 * pair]].K) return;
    const int t = out[(size_t)pair * out_stride + q];
    if (t >= 0) atomicMin(owner + (size_t)pair * owner_stride + t, q);
}

__global__ void k_unique_emit(const ImgDev *__restrict__ imgs, const int32_t *__restrict__ pairs,
                              int32_t *__restrict__ out, int64_t out_stride,
                              const int32_t *__restrict__ owner, int owner_stride,
                              int32_t *__restrict__ counts, int qblocks, int pair_base)
{
    const int pl = blockIdx.x / qblocks, pair = pair_base + pl;
    const int q = (blockIdx.x - pl * qblocks) * blockDim.x + threadIdx.x;
    if (q >= out_stride) return;
    int32_t *o = out + (size_t)pair * out_stride + q;
    bool keep = false;
    if (q < imgs[pairs[2 * pair]].K) {
        const int t = *o;
        keep = t >= 0 && owner[(size_t)pair * owner_stride + t] == q;
    }
    if (!keep) *o = -1;
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(counts + pair, (int)__popcll(m));
}

// K3, one launch: a workgroup per pair with the owner table of the pair's train rows in LDS
// (K2 <= 8192): claim with ds_min, emit, count -- the match table is read once and only the
// entries that change are rewritten.
#define RCN_UNIQ_LDS 8192
__global__ __launch_bounds__(256) void k_unique_pair(const ImgDev *__restrict__ imgs, const int32_t *__restrict__ pairs,
                                                     int32_t *__restrict__ out, int64_t out_stride,
                                                     int32_t *__restrict__ counts, int pair_base)
{
    __shared__ int owner[RCN_UNIQ_LDS];
    __shared__ int wsum[4];
    const int pair = pair_base + blockIdx.x, t = threadIdx.x;
    const int Kq = imgs[pairs[2 * pair]].K, Kt = imgs[pairs[2 * pair + 1]].K;
    for (int i = t; i < Kt; i += 256) owner[i] = 0x7fffffff;
    __syncthreads();
    int32_t *row = out + (size_t)pair * out_stride;
    for (int q = t; q < Kq; q += 256) {
        const int tr = row[q];
        if (tr >= 0) atomicMin(&owner[tr], q);
    }
    __syncthreads();
    int kept = 0;
    for (int q0 = 0; q0 < out_stride; q0 += 256) {
        const int q = q0 + t;
        bool keep = false;
        if (q < out_stride) {
            const int tr = q < Kq ? row[q] : -1;
            keep = tr >= 0 && owner[tr] == q;
            if (!keep && (q >= Kq || tr >= 0)) row[q] = -1;      // the tail beyond Kq is defined as -1
        }
        kept += (int)__popcll(__ballot(keep));
    }
    if ((t & 63) == 0) wsum[t >> 6] = kept;
    __syncthreads();
    if (t == 0) counts[pair] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// per pipeline chunk: the chunk's counts into the totals of the call.  c[0] rows past the re-rank, c[1] survivors, c[2] rows the
// middle tier passed on; c[4..6]: fallback rows, re-ranked rows, rows that went through K2b
__global__ void k_stats_acc(unsigned *c, int mid, unsigned midrows)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    c[4] += c[0];
    c[5] += c[1];
    c[6] += mid ? c[2] + (c[0] > midrows ? c[0] - midrows : 0u) : c[0];
}

// =========================================================================================
// host side
// =========================================================================================
static int pad_dim(int D)
{
    if (D <= 32) return 32;
    if (D <= 64) return 64;
    if (D <= 128) return 128;
    if (D <= 256) return 256;
    return 0;  // no MFMA path
}

static hipError_t launch_rowstats(rcn_ctx *ctx, const float *x, int rows, int D, double *nrm2, unsigned *cnt)
{
    ctx->hist_rows += rows;
    unsigned long long *mx = reinterpret_cast<unsigned long long *>(cnt + 2);
    if (D % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        k_rowstats<true><<<std::max(1, std::min((rows + 15) / 16, 4096)), 256, 0, ctx->stream>>>(x, rows, D, nrm2, cnt, mx, cnt + RCN_HIST_WORD);
    else
        k_rowstats<false><<<std::max(1, std::min((rows + 3) / 4, 4096)), 256, 0, ctx->stream>>>(x, rows, D, nrm2, cnt, mx, cnt + RCN_HIST_WORD);
    return hipGetLastError();
}

static void free_image(ImgHost &im)
{
    if (im.slab >= 0) return;  // views into a slab
    if (im.f32) (void)hipFree(im.f32);
    if (im.f16) (void)hipFree(im.f16);
    if (im.hn) (void)hipFree(im.hn);
    if (im.nrm2) (void)hipFree(im.nrm2);
    if (im.bigmin) (void)hipFree(im.bigmin);
    im.f32 = nullptr; im.f16 = nullptr; im.hn = nullptr; im.nrm2 = nullptr; im.bigmin = nullptr;
}

static void free_slab(Slab &sl)
{
    if (sl.f16) (void)hipFree(sl.f16);
    if (sl.hn) (void)hipFree(sl.hn);
    if (sl.nrm2) (void)hipFree(sl.nrm2);
    if (sl.bigmin) (void)hipFree(sl.bigmin);
    if (sl.own_f32) (void)hipFree(sl.own_f32);
    if (sl.own_Ks) (void)hipFree(sl.own_Ks);
    sl = Slab();
}

// Slabs that no resident image refers to any more (their ids were re-uploaded with another shape).
// The caller has synchronised the stream.
static void retire_unreferenced_slabs(rcn_ctx *ctx)
{
    std::vector<char> used(ctx->slabs.size(), 0);
    for (const auto &kv : ctx->images)
        if (kv.second.slab >= 0) used[kv.second.slab] = 1;
    for (size_t i = 0; i < ctx->slabs.size(); ++i)
        if (ctx->slabs[i].live && !used[i]) free_slab(ctx->slabs[i]);   // the index stays: images refer to slabs by position
}

int rcn_match_release(rcn_ctx *ctx)
{
    for (auto &kv : ctx->images) free_image(kv.second);
    ctx->images.clear();
    for (Slab &sl : ctx->slabs) free_slab(sl);
    ctx->slabs.clear();
    ctx->table_host.clear();
    ctx->prepared = false;
    ctx->D = ctx->DP = 0;
    return RCN_OK;
}

static int ensure_counters(rcn_ctx *ctx)
{
    if (!ctx->counters.p) {
        RCN_HIP(ctx->counters.reserve(RCN_COUNTER_BYTES));
        RCN_HIP(hipMemsetAsync(ctx->counters.p, 0, RCN_COUNTER_BYTES, ctx->stream));
    }
    return RCN_OK;
}
// counters layout (words): [0] u32 maxabs bits, [2..3] u64 max nrm2 bits, [8 + 2c] fallback / [9 + 2c] survivor count of chunk c,
// [RCN_HIST_WORD .. + RCN_HIST_BINS) rows per octave of |x|^2

// rcn_match_pair keeps two scratch images resident (ids INT32_MIN, INT32_MIN+1) so that the next
// call can reuse their allocations; they must not pin D when nothing else is resident.
static int drop_scratch_if_alone(rcn_ctx *ctx, int32_t D)
{
    if (ctx->images.empty() || ctx->D == D) return RCN_OK;
    for (const auto &kv : ctx->images)
        if (kv.first != INT32_MIN && kv.first != INT32_MIN + 1) return RCN_OK;   // user images decide
    RCN_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &kv : ctx->images) free_image(kv.second);
    ctx->images.clear();
    ctx->prepared = false;
    return RCN_OK;
}

static int upload_common(rcn_ctx *ctx, int32_t img_id, const float *src, bool src_is_device,
                         int32_t K, int32_t D)
{
    if (!ctx || K < 0 || D <= 0 || (K > 0 && !src)) {
        if (ctx) ctx->set_error("rcn_desc_upload: bad argument");
        return RCN_ERR_ARG;
    }
    { int rcd = drop_scratch_if_alone(ctx, D); if (rcd) return rcd; }
    if (!ctx->images.empty() && ctx->D != D) {
        ctx->set_error("rcn_desc_upload: all resident images must share D");
        return RCN_ERR_ARG;
    }
    RCN_HIP(hipSetDevice(ctx->device));
    int rc = ensure_counters(ctx);
    if (rc) return rc;
    ctx->D = D;
    ctx->DP = pad_dim(D);
    int Kp = (K + RCN_QT - 1) / RCN_QT * RCN_QT;
    if (Kp == 0) Kp = RCN_QT;
    const int DPa = ctx->DP ? ctx->DP : 32;
    ImgHost im;
    auto it = ctx->images.find(img_id);
    if (it != ctx->images.end()) {
        // same id again (the per-pair plugin path re-uploads on every call): keep the
        // allocations when they are large enough; stream order protects in-flight readers
        ImgHost &o = it->second;
        if (o.slab < 0 && o.cap_rows >= Kp && o.cap_D == D) im = o;
        else {
            RCN_HIP(hipStreamSynchronize(ctx->stream));
            free_image(o);
        }
        ctx->images.erase(it);
    }
    im.K = K;
    im.Kp = Kp;
    im.dirty = true;
    if (!im.f32) {
        im.slab = -1;
        im.cap_rows = Kp; im.cap_D = D;
        RCN_HIP(hipMalloc(&im.f32, (size_t)Kp * D * sizeof(float)));
        RCN_HIP(hipMalloc(&im.f16, (size_t)Kp * DPa * sizeof(_Float16)));
        RCN_HIP(hipMalloc(&im.hn, (size_t)Kp * sizeof(float)));
        RCN_HIP(hipMalloc(&im.nrm2, (size_t)Kp * sizeof(double)));
        RCN_HIP(hipMalloc(&im.bigmin, sizeof(unsigned long long)));
    }
    if (K > 0) {
        RCN_HIP(hipMemcpyAsync(im.f32, src, (size_t)K * D * sizeof(float),
                               src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                               ctx->stream));
        unsigned *cnt = ctx->counters.as<unsigned>();
        RCN_HIP(launch_rowstats(ctx, im.f32, K, D, im.nrm2, cnt));
        if (!src_is_device) RCN_HIP(hipStreamSynchronize(ctx->stream));  // host rows are borrowed
    }
    ctx->images[img_id] = im;
    ctx->prepared = false;
    return RCN_OK;
}

// Attach a block of equally shaped images ([n_slots][K][D] fp32 in HBM, borrowed) as ids
// first_id .. first_id + n_images - 1.  n_slots >= n_images: the block may carry unused slots at
// its end (the landing buffer of an all-gather whose image count is not a multiple of the rank
// count).  [conv_first, conv_first + conv_n) are the slots THIS ctx converts to fp16 itself; the
// other slots' fp16 rows / half-norms / norms are filled by the caller (RCCL all-gather, shard.hip).
// Same shape again reuses every allocation.
// Ks_host / Ks_dev (both or neither): rows in use per slot (<= K) when the images are ragged; the unused
// tail rows of a slot must hold zeros (they take part in the row statistics) and are never matched.
int rcn_int_slab_attach(rcn_ctx *ctx, int32_t first_id, int32_t n_images, int32_t n_slots, const float *src,
                        int32_t K, int32_t D, int32_t conv_first, int32_t conv_n, int *slab_out,
                        const int32_t *Ks_host, const int32_t *Ks_dev)
{
    if (n_images < 0 || n_slots < n_images || K <= 0 || D <= 0 || (n_slots > 0 && !src) || (int64_t)n_slots * K > 0x7fffffffLL ||
        conv_first < 0 || conv_n < 0 || conv_first + conv_n > n_slots) {
        ctx->set_error("rcn_desc_upload_batch_device: bad argument");
        return RCN_ERR_ARG;
    }
    if (slab_out) *slab_out = -1;
    if (n_slots == 0) return RCN_OK;
    if (D % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) != 0) {
        ctx->set_error("rcn_desc_upload_batch_device: the borrowed block must be 16-byte aligned when D % 4 == 0");
        return RCN_ERR_ARG;
    }
    { int rcd = drop_scratch_if_alone(ctx, D); if (rcd) return rcd; }
    if (!ctx->images.empty() && ctx->D != D) {
        ctx->set_error("rcn_desc_upload_batch_device: all resident images must share D");
        return RCN_ERR_ARG;
    }
    RCN_HIP(hipSetDevice(ctx->device));
    int rc = ensure_counters(ctx);
    if (rc) return rc;
    ctx->D = D;
    ctx->DP = pad_dim(D);
    const int DPa = ctx->DP ? ctx->DP : 32;
    const int Kp = (K + RCN_QT - 1) / RCN_QT * RCN_QT;
    int si = -1;
    for (size_t i = 0; i < ctx->slabs.size(); ++i) {
        const Slab &sl = ctx->slabs[i];
        if (sl.live && sl.first_id == first_id && sl.n == n_slots && sl.K == K && sl.D == D) si = (int)i;
    }
    if (si < 0) {
        // new shape: drop whatever these ids held, then allocate once
        RCN_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < n_images; ++i) {
            auto it = ctx->images.find(first_id + i);
            if (it != ctx->images.end()) { free_image(it->second); ctx->images.erase(it); }
        }
        retire_unreferenced_slabs(ctx);      // a slab none of whose images is left gives its HBM back
        Slab sl;
        sl.first_id = first_id; sl.n = n_slots; sl.K = K; sl.Kp = Kp; sl.D = D; sl.live = true;
        RCN_HIP(hipMalloc(&sl.f16, (size_t)n_slots * Kp * DPa * sizeof(_Float16)));
        RCN_HIP(hipMalloc(&sl.hn, (size_t)n_slots * Kp * sizeof(float)));
        RCN_HIP(hipMalloc(&sl.nrm2, (size_t)n_slots * K * sizeof(double)));
        RCN_HIP(hipMalloc(&sl.bigmin, (size_t)n_slots * sizeof(unsigned long long)));
        for (size_t i = 0; i < ctx->slabs.size() && si < 0; ++i)
            if (!ctx->slabs[i].live) { ctx->slabs[i] = sl; si = (int)i; }      // reuse a retired position
        if (si < 0) { ctx->slabs.push_back(sl); si = (int)ctx->slabs.size() - 1; }
    }
    Slab &sl = ctx->slabs[si];
    sl.f32 = src;
    sl.n_images = n_images;
    sl.conv_first = conv_first; sl.conv_n = conv_n;
    sl.Ks_dev = Ks_dev;
    for (int i = 0; i < n_images; ++i) {
        if (Ks_host && (Ks_host[i] < 0 || Ks_host[i] > K)) { ctx->set_error("slab attach: an image's row count exceeds its slot"); return RCN_ERR_ARG; }
        auto old = ctx->images.find(first_id + i);
        if (old != ctx->images.end() && old->second.slab != si) {
            // the id was re-uploaded on its own (or through another slab) in between: release what it owns
            if (old->second.slab < 0) { RCN_HIP(hipStreamSynchronize(ctx->stream)); free_image(old->second); }
            ctx->images.erase(old);
        }
        ImgHost im;
        im.K = Ks_host ? Ks_host[i] : K; im.Kp = Kp; im.slab = si; im.dirty = true;
        im.f32 = const_cast<float *>(src) + (size_t)i * K * D;
        im.f16 = sl.f16 + (size_t)i * Kp * DPa;
        im.hn = sl.hn + (size_t)i * Kp;
        im.nrm2 = sl.nrm2 + (size_t)i * K;
        im.bigmin = sl.bigmin + i;
        auto it = ctx->images.find(first_id + i);
        if (it != ctx->images.end()) im.slot = it->second.slot;
        ctx->images[first_id + i] = im;
    }
    // images of a previous, larger attach of the same slab that are no longer part of it
    for (auto it = ctx->images.begin(); it != ctx->images.end();) {
        if (it->second.slab == si && (it->first < first_id || it->first >= first_id + n_images)) it = ctx->images.erase(it);
        else ++it;
    }
    ctx->prepared = false;
    if (slab_out) *slab_out = si;
    return RCN_OK;
}

// Row statistics (|x|^2 per row into the slab's norm array, running maxima into the ctx counters) of
// slots [first, first + n) of an attached slab.
int rcn_int_slab_rowstats(rcn_ctx *ctx, int si, int32_t first, int32_t n)
{
    if (si < 0 || n <= 0) return RCN_OK;
    const Slab &sl = ctx->slabs[si];
    unsigned *cnt = ctx->counters.as<unsigned>();
    RCN_HIP(launch_rowstats(ctx, sl.f32 + (size_t)first * sl.K * sl.D, n * sl.K, sl.D, sl.nrm2 + (size_t)first * sl.K, cnt));
    return RCN_OK;
}

static int upload_batch(rcn_ctx *ctx, int32_t first_id, int32_t n, const float *src, int32_t K, int32_t D)
{
    int si = -1;
    int rc = rcn_int_slab_attach(ctx, first_id, n, n, src, K, D, 0, n, &si, nullptr, nullptr);
    if (rc || si < 0) return rc;
    return rcn_int_slab_rowstats(ctx, si, 0, n);
}

// rcn_desc_upload_batch: ragged host images into one ctx-owned block.  The slab machinery above does the rest (one stats
// launch here, one conversion launch at the next prepare); what this adds is the block itself and the copies.
static int upload_batch_host(rcn_ctx *ctx, int32_t first_id, int32_t n, const float *const *rows, const int32_t *Ks, int32_t D)
{
    if (n < 0 || D <= 0 || (n > 0 && (!rows || !Ks))) { ctx->set_error("rcn_desc_upload_batch: bad argument"); return RCN_ERR_ARG; }
    if (n == 0) return RCN_OK;
    int32_t Kmax = 1;
    for (int i = 0; i < n; ++i) {
        if (Ks[i] < 0 || (Ks[i] > 0 && !rows[i])) { ctx->set_error("rcn_desc_upload_batch: bad row count or NULL rows"); return RCN_ERR_ARG; }
        Kmax = std::max(Kmax, Ks[i]);
    }
    if ((int64_t)n * Kmax > 0x7fffffffLL) { ctx->set_error("rcn_desc_upload_batch: batch too large"); return RCN_ERR_ARG; }
    RCN_HIP(hipSetDevice(ctx->device));
    // a block of this shape from an earlier call is reused; otherwise the new one is handed to the slab once it is attached
    float *blk = nullptr;
    int32_t *ksd = nullptr;
    for (const Slab &sl : ctx->slabs)
        if (sl.live && sl.own_f32 && sl.first_id == first_id && sl.n == n && sl.K == Kmax && sl.D == D) { blk = sl.own_f32; ksd = sl.own_Ks; }
    const bool fresh = !blk;
    const size_t bytes = (size_t)n * Kmax * D * sizeof(float);
    if (fresh) {
        RCN_HIP(hipMalloc(&blk, bytes));
        if (hipMalloc(&ksd, (size_t)n * sizeof(int32_t)) != hipSuccess) { (void)hipFree(blk); ctx->set_error("rcn_desc_upload_batch: out of device memory"); return RCN_ERR_HIP; }
    }
    auto fail = [&](int rc) { if (fresh) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(blk); (void)hipFree(ksd); } return rc; };
    // The copies alternate between the ctx stream and the ctx's copy stream: one stream's copies run on one DMA engine
    // (~27 GB/s measured from pinned rows), two streams' on two.  The side stream starts behind the zero fill and joins
    // before the statistics.
    hipError_t e = hipMemsetAsync(blk, 0, bytes, ctx->stream);                     // tails must be zero: they take part in the row statistics
    hipStream_t side = nullptr;
    if (e == hipSuccess && n > 1) {
        if (!ctx->copy_stream) {
            e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
            for (auto &ev : ctx->cmp_ev) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->cmp_filled, hipEventDisableTiming);
        }
        side = ctx->copy_stream;
        if (e == hipSuccess) e = hipEventRecord(ctx->cmp_filled, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(side, ctx->cmp_filled, 0);
    }
    for (int i = 0; i < n && e == hipSuccess; ++i)
        if (Ks[i] > 0) e = hipMemcpyAsync(blk + (size_t)i * Kmax * D, rows[i], (size_t)Ks[i] * D * sizeof(float), hipMemcpyHostToDevice, (side && (i & 1)) ? side : ctx->stream);
    if (e == hipSuccess && side) {
        e = hipEventRecord(ctx->cmp_filled, side);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->cmp_filled, 0);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ksd, Ks, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { ctx->set_error(std::string("rcn_desc_upload_batch: ") + hipGetErrorString(e)); return fail(RCN_ERR_HIP); }
    int si = -1;
    int rc = rcn_int_slab_attach(ctx, first_id, n, n, blk, Kmax, D, 0, n, &si, Ks, ksd);
    if (rc || si < 0) return fail(rc ? rc : RCN_ERR_ARG);
    ctx->slabs[si].own_f32 = blk;
    ctx->slabs[si].own_Ks = ksd;
    rc = rcn_int_slab_rowstats(ctx, si, 0, n);
    RCN_HIP(hipStreamSynchronize(ctx->stream));       // the host rows are borrowed for the call only
    return rc;
}

template <int DP> static void launch_prepare_batch(rcn_ctx *ctx, const Slab &sl)
{
    // only the slots this ctx converts itself (all of them unless the slab is an all-gather landing buffer)
    if (sl.conv_n <= 0) return;
    const long nthr = (long)sl.conv_n * sl.Kp * (DP / 8);
    const size_t f = sl.conv_first;
    k_fill_inf<<<(sl.conv_n + 255) / 256, 256, 0, ctx->stream>>>(sl.bigmin + f, sl.conv_n);
    k_prepare_batch<DP><<<(unsigned)((nthr + 255) / 256), 256, 0, ctx->stream>>>(
        sl.f32 + f * sl.K * sl.D, sl.nrm2 + f * sl.K, sl.conv_n, sl.K, sl.Kp, sl.D, ctx->scale_dev.as<ScaleDev>(),
        sl.f16 + f * sl.Kp * DP, sl.hn + f * sl.Kp, sl.Ks_dev ? sl.Ks_dev + f : nullptr, sl.bigmin + f);
}

template <int DP> static void launch_prepare(rcn_ctx *ctx, const ImgHost &im)
{
    const int n = im.Kp * (DP / 8);
    k_fill_inf<<<1, 64, 0, ctx->stream>>>(im.bigmin, 1);
    k_prepare<DP><<<(n + 255) / 256, 256, 0, ctx->stream>>>(im.f32, im.nrm2, im.K, im.Kp, ctx->D, ctx->scale_dev.as<ScaleDev>(),
                                                            im.f16, im.hn, im.bigmin);
}

// Host copies of the scale constants after a device-side fix (k_fix_scale): one small read behind the stream.
int rcn_int_resolve_scale(rcn_ctx *ctx)
{
    if (!ctx->scale_on_device) return RCN_OK;
    RCN_HIP(hipMemcpyAsync(&ctx->scale_host, ctx->scale_dev.p, sizeof(ScaleDev), hipMemcpyDeviceToHost, ctx->stream));
    RCN_HIP(rcn_int_stream_wait(ctx));
    ctx->scale = ctx->scale_host.s;
    ctx->bias = ctx->scale_host.bias;
    ctx->max_norm = ctx->scale_host.n_max;
    ctx->thr2 = ctx->scale_host.thr2;
    ctx->scale_on_device = false;
    return RCN_OK;
}

// Fix the global scale / bias, (re)build every image's fp16 copy and the device image table.
// Two ways to the scale.  Host (default): read the row statistics, compare with the scale in force, convert only what
// is new or what a moved scale invalidates -- two host synchronisations.  Device (ctx->want_dev_scale, the sharded
// exchange): k_fix_scale computes the same constants in HBM behind the all-reduce of the statistics and every dirty
// image is converted with them; the host reads nothing and does not wait (it may only do so when every resident image
// is dirty, i.e. about to be converted anyway -- always the case for a landing buffer).
int rcn_int_prepare_all(rcn_ctx *ctx)
{
    if (ctx->prepared) return RCN_OK;
    RCN_HIP(ctx->scale_dev.reserve(sizeof(ScaleDev)));
    const int DPa = ctx->DP ? ctx->DP : 32;
    bool on_device = ctx->want_dev_scale;
    if (on_device)
        for (const auto &kv : ctx->images) on_device = on_device && kv.second.dirty;
    bool moved = false;
    if (on_device) {
        k_fix_scale<<<1, 64, 0, ctx->stream>>>(ctx->counters.as<unsigned>(), DPa, ctx->scale_dev.as<ScaleDev>());
        RCN_HIP(hipGetLastError());
        ctx->scale_on_device = true;
    } else {
        { int rcs = rcn_int_resolve_scale(ctx); if (rcs) return rcs; }
        {   // the histogram counts every row ever uploaded since the last clear: once replaced images (the per-pair plugin call
            // re-uploads two scratch images every time) outweigh the resident ones, count the resident rows afresh
            long resident = 0;
            for (const auto &kv : ctx->images) resident += kv.second.slab < 0 ? kv.second.K : 0;
            for (const Slab &sl : ctx->slabs) resident += sl.live ? (long)sl.n * sl.K : 0;
            if (ctx->hist_rows > 2 * resident + 1024) {
                unsigned *hist = ctx->counters.as<unsigned>() + RCN_HIST_WORD;
                RCN_HIP(hipMemsetAsync(hist, 0, RCN_HIST_BINS * sizeof(unsigned), ctx->stream));
                for (const auto &kv : ctx->images)
                    if (kv.second.slab < 0 && kv.second.K > 0)
                        k_norm_hist<<<std::min(64, (kv.second.K + 255) / 256), 256, 0, ctx->stream>>>(kv.second.nrm2, kv.second.K, hist);
                for (const Slab &sl : ctx->slabs)
                    if (sl.live) k_norm_hist<<<std::min(1024L, ((long)sl.n * sl.K + 255) / 256), 256, 0, ctx->stream>>>(sl.nrm2, (long)sl.n * sl.K, hist);
                RCN_HIP(hipGetLastError());
                ctx->hist_rows = resident;
            }
        }
        std::vector<unsigned> hc(RCN_HIST_WORD + RCN_HIST_BINS, 0u);
        RCN_HIP(hipMemcpyAsync(hc.data(), ctx->counters.p, RCN_COUNTER_BYTES, hipMemcpyDeviceToHost, ctx->stream));
        RCN_HIP(rcn_int_stream_wait(ctx));
        float maxabs;
        double maxn2;
        memcpy(&maxabs, &hc[0], 4);
        memcpy(&maxn2, &hc[2], 8);
        fix_scale(maxabs, maxn2, hc.data() + RCN_HIST_WORD, DPa, &ctx->scale_host);
        const double s = ctx->scale_host.s, bias = ctx->scale_host.bias;
        moved = s != ctx->scale || bias != ctx->bias || ctx->scale_host.thr2 != ctx->thr2;
        ctx->thr2 = ctx->scale_host.thr2;
        ctx->scale = s;
        ctx->max_norm = ctx->scale_host.n_max;
        ctx->bias = bias;
        RCN_HIP(hipMemcpyAsync(ctx->scale_dev.p, &ctx->scale_host, sizeof(ScaleDev), hipMemcpyHostToDevice, ctx->stream));
    }

    std::vector<ImgDev> table;
    table.reserve(ctx->images.size());
    int slot = 0;
    std::vector<char> slab_dirty(ctx->slabs.size(), 0);
    for (auto &kv : ctx->images) {
        ImgHost &im = kv.second;
        im.slot = slot++;
        const bool need = moved || im.dirty;
        if (need && im.slab >= 0) slab_dirty[im.slab] = 1;
        if (ctx->DP && im.slab < 0 && need) {
            switch (ctx->DP) {
            case 32: launch_prepare<32>(ctx, im); break;
            case 64: launch_prepare<64>(ctx, im); break;
            case 128: launch_prepare<128>(ctx, im); break;
            default: launch_prepare<256>(ctx, im); break;
            }
            RCN_HIP(hipGetLastError());
        }
        im.dirty = false;
        table.push_back(ImgDev{im.f32, im.f16, im.hn, im.nrm2, im.bigmin, im.K, im.Kp});
    }
    if (ctx->DP)
        for (size_t si = 0; si < ctx->slabs.size(); ++si) {
            const Slab &sl = ctx->slabs[si];
            if (!sl.live || !slab_dirty[si]) continue;
            switch (ctx->DP) {
            case 32: launch_prepare_batch<32>(ctx, sl); break;
            case 64: launch_prepare_batch<64>(ctx, sl); break;
            case 128: launch_prepare_batch<128>(ctx, sl); break;
            default: launch_prepare_batch<256>(ctx, sl); break;
            }
            RCN_HIP(hipGetLastError());
        }
    // the image table: uploaded only when it differs from what the device already holds (a per-step exchange into the
    // same landing buffer leaves it unchanged); the staging vector lives in the ctx, so nothing waits for the copy
    const bool same = table.size() == ctx->table_host.size() && ctx->img_table.p &&
                      (table.empty() || memcmp(table.data(), ctx->table_host.data(), table.size() * sizeof(ImgDev)) == 0);
    if (!same) {
        RCN_HIP(rcn_int_stream_wait(ctx));       // a previous upload may still read the staging vector
        ctx->table_host.swap(table);
        RCN_HIP(ctx->img_table.reserve(std::max<size_t>(1, ctx->table_host.size()) * sizeof(ImgDev)));
        if (!ctx->table_host.empty())
            RCN_HIP(hipMemcpyAsync(ctx->img_table.p, ctx->table_host.data(), ctx->table_host.size() * sizeof(ImgDev),
                                   hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->prepared = true;
    return RCN_OK;
}

template <int DP> static hipError_t launch_coarse_w4(rcn_ctx *ctx, const CoarseArgs &ca, int blocks)
{
    const size_t lds = (size_t)RCN_NBUF * (RCN_BT * DP * 2 + 4 * 256) + RCN_TBL_BYTES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_coarse_w4<DP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k_coarse_w4<DP><<<blocks, 256, lds, ctx->stream>>>(ca);
    return hipGetLastError();
}

template <int DP, int ABL = 0, int SH = 0> static hipError_t launch_coarse(rcn_ctx *ctx, const CoarseArgs &ca, int blocks)
{
    const size_t lds = (size_t)RCN_NBUF * (coarse_bt(DP) * DP * 2 + 8 * coarse_bt(DP) * 4) + RCN_TBL_BYTES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_coarse_top2<DP, ABL, SH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k_coarse_top2<DP, ABL, SH><<<blocks, 512, lds, ctx->stream>>>(ca);
    return hipGetLastError();
}

int rcn_int_match_grid(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, float ratio,
                           int32_t *out_dev, int64_t out_stride, int32_t *counts_dev)
{
    // pairs == NULL: the canonical grid of the reference's pair loop (SequentialReconstructor.cpp:202-227
    // with FakeImgMatcher, ImageMatcher.cpp:6-23): every i < j over the resident image ids in ascending order
    if (!pairs_host && n_pairs > 0) {
        std::vector<int32_t> ids;
        for (const auto &kv : ctx->images)
            if (kv.first != INT32_MIN && kv.first != INT32_MIN + 1) ids.push_back(kv.first);
        const int64_t want = (int64_t)ids.size() * ((int64_t)ids.size() - 1) / 2;
        if (want != n_pairs) { ctx->set_error("rcn_match_grid: pairs == NULL needs n_pairs = n (n - 1) / 2 over the resident images"); return RCN_ERR_ARG; }
        std::vector<int32_t> &all = ctx->all_pairs_host;
        all.clear();
        for (size_t i = 0; i < ids.size(); ++i)
            for (size_t j = i + 1; j < ids.size(); ++j) { all.push_back(ids[i]); all.push_back(ids[j]); }
        pairs_host = all.data();
    }
    if (n_pairs < 0 || (n_pairs > 0 && (!pairs_host || !out_dev || !counts_dev))) {
        ctx->set_error("rcn_match_grid: bad argument");
        return RCN_ERR_ARG;
    }
    memset(&ctx->last_stats, 0, sizeof(ctx->last_stats));
    if (n_pairs == 0) return RCN_OK;
    RCN_HIP(hipSetDevice(ctx->device));
    int rc = rcn_int_prepare_all(ctx);
    if (rc) return rc;

    // image ids -> table slots; shape bookkeeping
    std::vector<int32_t> &slots = ctx->slots_host;     // kept in the ctx: uploaded asynchronously
    slots.assign(2 * (size_t)n_pairs, 0);
    int kq_max = 0, kt_max = 0, ktp_max = 0;
    int64_t rows = 0, pd = 0;
    // id -> image record through a flat table when the ids are small non-negative integers (they are image indices):
    // a 500 000-pair list costs three million std::map walks otherwise, with the GPU idle behind the host
    std::vector<const ImgHost *> flat;
    {
        int32_t lo = INT32_MAX, hi = INT32_MIN;
        for (const auto &kv : ctx->images)
            if (kv.first != INT32_MIN && kv.first != INT32_MIN + 1) { lo = std::min(lo, kv.first); hi = std::max(hi, kv.first); }
        if (lo >= 0 && hi >= lo && hi < (1 << 24)) {
            flat.assign((size_t)hi + 1, nullptr);
            for (const auto &kv : ctx->images)
                if (kv.first >= 0) flat[kv.first] = &kv.second;
        }
    }
    auto lookup = [&](int32_t id) -> const ImgHost * {
        if (!flat.empty() && id >= 0 && (size_t)id < flat.size()) return flat[id];
        auto it = ctx->images.find(id);
        return it == ctx->images.end() ? nullptr : &it->second;
    };
    for (int p = 0; p < n_pairs; ++p) {
        const ImgHost *a = lookup(pairs_host[2 * p]), *b = lookup(pairs_host[2 * p + 1]);
        if (!a || !b) {
            ctx->set_error("rcn_match_grid: image id not resident");
            return RCN_ERR_NOT_FOUND;
        }
        slots[2 * p] = a->slot;
        slots[2 * p + 1] = b->slot;
        kq_max = std::max(kq_max, a->K);
        kt_max = std::max(kt_max, b->K);
        ktp_max = std::max(ktp_max, b->Kp);
        rows += a->K;
        pd += (int64_t)a->K * b->K;
    }
    if (out_stride < kq_max) {
        ctx->set_error("rcn_match_grid: out_stride smaller than a query image's K");
        return RCN_ERR_ARG;
    }
    const int tiles = std::max(1, (kq_max + RCN_QT - 1) / RCN_QT);
    const int kq_stride = tiles * RCN_QT;
    int idx_bits = 1;
    while ((1 << idx_bits) < ktp_max) ++idx_bits;
    // The packed candidate keeps the train row in its low idx_bits and the accumulator's leading 32 - idx_bits bits
    // above them: up to 16 index bits (65536 rows per image) leave sign + exponent + 7 mantissa bits, and the
    // certificates price that quantum (k_filter: upper bounds are truncated value + one quantum), so a coarser
    // value only sends more rows to the exact re-rank -- it never changes a result.
    const bool mfma = ctx->DP != 0 && idx_bits <= 16 && !ctx->force_exact && kq_max > 0 && kt_max >= 2;
    const uint32_t idx_mask = (1u << idx_bits) - 1u;
    const int owner_stride = std::max(1, kt_max);

    // groups: runs of consecutive pairs that share the query image, cut at RCN_GROUP
    std::vector<int2> &groups = ctx->groups_host;
    groups.clear();
    for (int p = 0; p < n_pairs;) {
        int cnt = 1;
        while (p + cnt < n_pairs && cnt < RCN_GROUP && slots[2 * (p + cnt)] == slots[2 * p]) ++cnt;
        groups.push_back(make_int2(p, cnt));
        p += cnt;
    }
    const int n_groups = (int)groups.size();
    // Pipeline chunks (round 4): the candidate table and the two row lists are sized for at most RCN_CHUNK_ROWS query-row slots
    // and REUSED by consecutive ranges of the pair list, in stream order -- coarse(c), filter(c), re-rank(c), uniqueness(c),
    // coarse(c+1), ...  Round 3 sized them for the whole grid: 16.4 GB each at cfg 3, three of them, and a 2000-image grid
    // would not have fitted the device at all.  A chunk never exceeds the budget (the lists can therefore not overflow) and
    // ends at a group boundary; a small grid is one chunk and runs exactly as before.  (The two-stream overlap of chunks that
    // the diagnostic build once had measured slower in rounds 1 and 2 and is gone.)
    std::vector<int> chunk_g0;      // first group of every chunk, then n_groups
    int64_t cap_slots = 0, cap_rows = 0;
    {
        int64_t slots_c = 0, rows_c = 0;
        chunk_g0.push_back(0);
        for (int g = 0; g < n_groups; ++g) {
            int64_t gs = (int64_t)groups[g].y * kq_stride, gr = 0;
            for (int r = 0; r < groups[g].y; ++r) gr += lookup(pairs_host[2 * (groups[g].x + r)])->K;
            if (slots_c > 0 && slots_c + gs > ctx->chunk_rows) {
                chunk_g0.push_back(g);
                cap_slots = std::max(cap_slots, slots_c); cap_rows = std::max(cap_rows, rows_c);
                slots_c = 0; rows_c = 0;
            }
            slots_c += gs; rows_c += gr;
        }
        cap_slots = std::max(cap_slots, slots_c); cap_rows = std::max(cap_rows, rows_c);
        chunk_g0.push_back(n_groups);
    }
    const int n_chunks = (int)chunk_g0.size() - 1;
    const bool vec4 = (ctx->D % 4) == 0;
    const bool mid = vec4 && kt_max >= 2;            // the middle tier reads rows as float4
    const int n_slots = (int)ctx->table_host.size();

    RCN_HIP(ctx->pairs_dev.reserve(slots.size() * sizeof(int32_t)));
    RCN_HIP(ctx->cand.reserve((size_t)std::max<int64_t>(1, cap_slots) * sizeof(uint2)));
    const bool uniq_lds = kt_max <= RCN_UNIQ_LDS;     // owner table of a pair fits LDS: one fused launch
    if (!uniq_lds) RCN_HIP(ctx->owner.reserve((size_t)n_pairs * owner_stride * sizeof(int32_t)));
    RCN_HIP(ctx->fb_list.reserve(std::max<int64_t>(1, cap_rows) * sizeof(unsigned long long)));
    RCN_HIP(ctx->sv_list.reserve(std::max<int64_t>(1, cap_rows) * sizeof(unsigned long long)));
    // middle tier: binned rows, thresholds, candidate lists, work items, the overflow list, three words per image slot
    const size_t mid_rows_cap = (size_t)std::min<int64_t>(std::max<int64_t>(1, cap_rows), ctx->mid_rows);
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    // (the bins come FIRST: their place must not move with the row count of the grid, they carry state -- zeros -- from call to call)
    const size_t mo_bins = 0, mo_sorted = mo_bins + al(4 * 4 * ((size_t)n_slots + 1)), mo_thr = mo_sorted + al(8 * mid_rows_cap), mo_cc = mo_thr + al(4 * mid_rows_cap),
                 mo_cl = mo_cc + al(4 * mid_rows_cap), mo_fb2 = mo_cl + al(4 * mid_rows_cap * RCN_MIDCAP), mo_def = mo_fb2 + al(8 * mid_rows_cap),
                 mo_defthr = mo_def + (n_chunks > 1 ? al(8 * mid_rows_cap) : 0), mo_end = mo_defthr + (n_chunks > 1 ? al(4 * mid_rows_cap) : 0);
    // the call's list of deferred rows (k_mid_defer): as many rows as one pass of the tier takes
    const bool defer = mid && n_chunks > 1 && uniq_lds && kq_max > 0;
    if (mid) {
        const bool fresh = mo_end > ctx->mid_ws.cap;
        RCN_HIP(ctx->mid_ws.reserve(mo_end));
        if (fresh || ctx->mid_bins_slots != n_slots) {       // the bins are left zeroed by every pass (k_mid_bins); a new buffer, or another slot count, is not
            RCN_HIP(hipMemsetAsync(ctx->mid_ws.as<char>() + mo_bins, 0, mo_sorted - mo_bins, ctx->stream));
            ctx->mid_bins_slots = n_slots;
        }
    }
    RCN_HIP(hipMemcpyAsync(ctx->pairs_dev.p, slots.data(), slots.size() * sizeof(int32_t),
                           hipMemcpyHostToDevice, ctx->stream));
    if (!uniq_lds) RCN_HIP(hipMemsetAsync(ctx->owner.p, 0x7f, (size_t)n_pairs * owner_stride * sizeof(int32_t), ctx->stream));
    RCN_HIP(hipMemsetAsync(counts_dev, 0, (size_t)n_pairs * sizeof(int32_t), ctx->stream));

    const ImgDev *imgs = ctx->img_table.as<ImgDev>();
    const int32_t *pairs = ctx->pairs_dev.as<int32_t>();

    const int evi = ctx->ev_n % 64;
    const bool prof = ctx->profile && ctx->ev_made;

    RerankArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.imgs = imgs; ra.pairs = pairs;
    ra.out = out_dev; ra.out_stride = out_stride;
    ra.n_pairs = n_pairs; ra.kq_stride = kq_stride; ra.D = ctx->D; ra.idx_mask = idx_mask;
    ra.ratio = ratio;
    ra.sc = ctx->scale_dev.as<ScaleDev>();
    ra.all_to_fallback = mfma ? 0 : 1;

    // Order of the work items of a coarse launch: every XCD walks its own contiguous range of the chunk's
    // group list, heaviest groups first, so that the last items to start are the short ones (the tail
    // of a launch is one item long: 5 % of a cfg-2 grid for a four-pair group, 1 % for a one-pair group).
    // Groups are dealt to the XCD ranges round-robin in descending weight; unused slots hold empty groups.
    // (Measured and dropped in round 2: a train-block-major order -- every XCD runs the query tiles of ~8 query
    // images against the SAME four train images at a time, so that all but one of them hit in its L2 -- is 0.5-1.5 %
    // slower: the kernel is power-limited, not fabric-limited, and the order above has the shorter tail.)
    std::vector<int> chunk_off((size_t)n_chunks, 0), chunk_ng((size_t)n_chunks, 0);
    {
        std::vector<int2> &arr = ctx->groups_arranged;
        arr.clear();
        std::vector<std::pair<int64_t, int>> order;
        for (int c = 0; c < n_chunks; ++c) {
            const int g0 = chunk_g0[c], g1 = chunk_g0[c + 1], ng = g1 - g0;
            chunk_off[c] = (int)arr.size();
            if (mfma && ng >= 64 && !ctx->no_item_order) {
                order.resize((size_t)ng);
                for (int g = 0; g < ng; ++g) {
                    int64_t wgt = 0;
                    for (int r = 0; r < groups[g0 + g].y; ++r) wgt += lookup(pairs_host[2 * (groups[g0 + g].x + r) + 1])->K;
                    order[g] = std::make_pair(-wgt, g0 + g);
                }
                std::stable_sort(order.begin(), order.end());
                const int gpx = (ng + 7) / 8;
                const size_t base = arr.size();
                arr.resize(base + (size_t)8 * gpx, make_int2(0, 0));
                for (int k = 0; k < ng; ++k) arr[base + (size_t)(k % 8) * gpx + k / 8] = groups[order[k].second];
                chunk_ng[c] = 8 * gpx;
            } else {
                arr.insert(arr.end(), groups.begin() + g0, groups.begin() + g1);
                chunk_ng[c] = ng;
            }
        }
        RCN_HIP(ctx->groups_dev.reserve(std::max<size_t>(1, arr.size()) * sizeof(int2)));
        RCN_HIP(hipMemcpyAsync(ctx->groups_dev.p, arr.data(), arr.size() * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    }
    hipStream_t st = ctx->stream;
    // counters (words from +8): [0] fallback rows of the chunk, [1] survivors, [2] rows past the middle tier, [3] its work items;
    // [4..6] the same three summed over the chunks of the call (rcn_match_last_stats)
    unsigned *ccnt = ctx->counters.as<unsigned>() + 8;
    RCN_HIP(hipMemsetAsync(ccnt, 0, 8 * sizeof(unsigned), st));
    ctx->ev_chunks[evi] = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const int g0 = chunk_g0[c], g1 = chunk_g0[c + 1];
        if (g1 <= g0) continue;
        const int p0 = groups[g0].x, p1 = g1 < n_groups ? groups[g1].x : n_pairs, np_c = p1 - p0;
        // the chunk's candidates sit at the START of the buffer; the kernels index by grid-wide pair number
        // (indexed by grid-wide pair number: the chunk's table starts at pair p0.  The bias is applied to the ADDRESS -- a pointer in front
        //  of its allocation is not something C++ pointer arithmetic may form; the kernels only ever index pairs p0 .. p1 - 1 of it)
        uint2 *cand_c = reinterpret_cast<uint2 *>(reinterpret_cast<uintptr_t>(ctx->cand.as<uint2>()) - (uintptr_t)((size_t)p0 * kq_stride * sizeof(uint2)));
        const bool tm = prof && c < RCN_EV_CHUNKS;
        if (tm) { RCN_HIP(hipEventRecord(ctx->ev_c[evi][c][0], st)); ctx->ev_chunks[evi] = c + 1; }
        if (mfma) {
            CoarseArgs ca;
            ca.imgs = imgs; ca.pairs = pairs; ca.cand = cand_c;
            const int ng_c = chunk_ng[c];
            ca.groups = ctx->groups_dev.as<int2>() + chunk_off[c];
            ca.n_groups = ng_c; ca.tiles_per_pair = tiles; ca.kq_stride = kq_stride;
            const int64_t items = (int64_t)ng_c * tiles;
            ca.items_per_xcd = (int)((items + 7) / 8);
            ca.idx_mask = idx_mask;
            const int blocks = ca.items_per_xcd * 8;
            hipError_t e;
            if (ctx->coarse_w4 && ctx->DP >= 64) {
                switch (ctx->DP) {
                case 64: e = launch_coarse_w4<64>(ctx, ca, blocks); break;
                case 128: e = launch_coarse_w4<128>(ctx, ca, blocks); break;
                default: e = launch_coarse_w4<256>(ctx, ca, blocks); break;
                }
            } else {
                // MFMA shape: v_mfma_f32_16x16x32_f16 at D = 256 (+4 % by wall, A/B on one box: the chip holds a higher
                // clock on it under this kernel's load), v_mfma_f32_32x32x16_f16 below (D = 128: -2.5 % on the other
                // shape -- half the MFMA work per tile makes that kernel issue-bound, and 16x16x32 issues twice as many)
                const int shape = ctx->coarse_shape >= 0 ? ctx->coarse_shape : (ctx->DP == 256 ? 1 : 0);
                if (shape == 1) {
                    switch (ctx->DP) {
                    case 32: e = launch_coarse<32, 0, 1>(ctx, ca, blocks); break;
                    case 64: e = launch_coarse<64, 0, 1>(ctx, ca, blocks); break;
                    case 128: e = launch_coarse<128, 0, 1>(ctx, ca, blocks); break;
                    default:
#ifdef RCN_DIAG
                        switch (ctx->ablate) {   // RCN_COARSE_ABL: timing experiments only (diagnostic build)
                        case 1: e = launch_coarse<256, 1, 1>(ctx, ca, blocks); break;
                        case 9: e = launch_coarse<256, 9, 1>(ctx, ca, blocks); break;
                        default: e = launch_coarse<256, 0, 1>(ctx, ca, blocks); break;
                        }
#else
                        e = launch_coarse<256, 0, 1>(ctx, ca, blocks);
#endif
                        break;
                    }
                } else {
                    switch (ctx->DP) {
                    case 32:
#ifdef RCN_DIAG
                        if (ctx->ablate == 1) { e = launch_coarse<32, 1, 0>(ctx, ca, blocks); break; }
                        if (ctx->ablate == 2) { e = launch_coarse<32, 2, 0>(ctx, ca, blocks); break; }
                        if (ctx->ablate == 6) { e = launch_coarse<32, 6, 0>(ctx, ca, blocks); break; }
#endif
                        e = launch_coarse<32>(ctx, ca, blocks); break;
                    case 64: e = launch_coarse<64>(ctx, ca, blocks); break;
                    case 128:
#ifdef RCN_DIAG
                        if (ctx->ablate == 1) { e = launch_coarse<128, 1, 0>(ctx, ca, blocks); break; }   // timing only: no top-2 fold
                        if (ctx->ablate == 2) { e = launch_coarse<128, 2, 0>(ctx, ca, blocks); break; }   // timing only: values-only fold
                        if (ctx->ablate == 6) { e = launch_coarse<128, 6, 0>(ctx, ca, blocks); break; }   // timing only: minimum-only fold (one operation per element)
#endif
                        e = launch_coarse<128>(ctx, ca, blocks); break;
                    default: e = launch_coarse<256>(ctx, ca, blocks); break;
                    }
                }
            }
            RCN_HIP(e);
        }
        if (tm) RCN_HIP(hipEventRecord(ctx->ev_c[evi][c][1], st));
        ra.cand = cand_c;
        ra.pair_base = p0;
        ra.fb_list = ctx->fb_list.as<unsigned long long>(); ra.fb_count = ccnt;
        ra.sv_list = ctx->sv_list.as<unsigned long long>(); ra.sv_count = ccnt + 1;
        if (c > 0) RCN_HIP(hipMemsetAsync(ccnt, 0, 4 * sizeof(unsigned), st));
        if (kq_max > 0) {
            const int qblocks = (kq_max + 255) / 256;
            dim3 g((unsigned)qblocks * (unsigned)np_c);
            ra.qblocks = (kq_max + RCN_FB - 1) / RCN_FB;
            k_filter<<<(unsigned)ra.qblocks * (unsigned)np_c, RCN_FT, 0, st>>>(ra);
            RCN_HIP(hipGetLastError());
            // sharded grid: the fp32 rows of the other ranks' images travel on a side stream while the
            // coarse pass runs on the fp16 payload; the exact stages are the first to read them
            if (ctx->f32_ready) RCN_HIP(hipStreamWaitEvent(st, ctx->f32_ready, 0));
            if (mfma) {
                if (vec4) k_rerank_lds<<<ctx->prop.multiProcessorCount * 16, 64, 0, st>>>(ra);
                else k_rerank_generic<<<ctx->prop.multiProcessorCount * 8, 256, 0, st>>>(ra);
                RCN_HIP(hipGetLastError());
            }
            const int fb_blocks = ctx->prop.multiProcessorCount * 8;
            const unsigned long long *brute = ra.fb_list;
            const unsigned *brute_n = ccnt;
            if (mid) {
                // the middle tier takes the first RCN_MIDROWS rows of the list; what overflows its candidate lists lands in fb2
                char *mw = ctx->mid_ws.as<char>();
                MidArgs ma;
                memset(&ma, 0, sizeof(ma));
                ma.imgs = imgs; ma.pairs = pairs; ma.cand = cand_c; ma.fb_list = ra.fb_list; ma.fb_count = ccnt;
                if (defer) {
                    // the chunk's rows join the call's list if they fit (then ccnt[0] is 0 and the kernels below find nothing to do)
                    MidArgs md = ma;
                    md.sc = ra.sc; md.kq_stride = kq_stride; md.D = ctx->D; md.all_to_fallback = ra.all_to_fallback; md.idx_mask = idx_mask;
                    k_mid_defer<<<ctx->prop.multiProcessorCount * 2, 256, 0, st>>>(md, (unsigned long long *)(mw + mo_def), (float *)(mw + mo_defthr), ccnt, (unsigned)mid_rows_cap);
                    k_mid_defer_commit<<<1, 64, 0, st>>>(ccnt, (unsigned)mid_rows_cap);
                }
                ma.sorted = (unsigned long long *)(mw + mo_sorted); ma.thr = (float *)(mw + mo_thr); ma.ccount = (unsigned *)(mw + mo_cc);
                ma.clist = (int32_t *)(mw + mo_cl); ma.fb2_list = (unsigned long long *)(mw + mo_fb2);
                ma.hist = (unsigned *)(mw + mo_bins); ma.offs = ma.hist + (n_slots + 1); ma.cursor = ma.offs + (n_slots + 1); ma.ibase = ma.cursor + (n_slots + 1);
                ma.n_items = ccnt + 3; ma.fb2_count = ccnt + 2;
                ma.out = out_dev; ma.out_stride = out_stride; ma.sc = ra.sc; ma.n_slots = n_slots; ma.kq_stride = kq_stride; ma.D = ctx->D;
                ma.all_to_fallback = ra.all_to_fallback; ma.idx_mask = idx_mask; ma.ratio = ratio; ma.midrows = (uint32_t)ctx->mid_rows;
                const int mb = ctx->prop.multiProcessorCount * 4;
                k_mid_hist<<<mb, 256, 0, st>>>(ma);
                k_mid_bins<<<1, 1024, 0, st>>>(ma);
                k_mid_scatter<<<mb, 256, 0, st>>>(ma);
                k_mid_eval<<<ctx->prop.multiProcessorCount * 4, 256, 0, st>>>(ma);
                k_mid_exact<<<mb, 256, 0, st>>>(ma);
                RCN_HIP(hipGetLastError());
                // K2b: the overflow list, then the rows beyond the tier's budget (none unless a chunk leaves more than RCN_MIDROWS rows)
                if (vec4) k_exact_rows_lds<<<fb_blocks, 64 * EX_WAVES, 0, st>>>(imgs, pairs, ma.fb2_list, ma.fb2_count, ctx->D, ratio, out_dev, out_stride);
                RCN_HIP(hipGetLastError());
                if (cap_rows > ctx->mid_rows) k_exact_rows_lds<<<fb_blocks, 64 * EX_WAVES, 0, st>>>(imgs, pairs, brute, brute_n, ctx->D, ratio, out_dev, out_stride, (unsigned)ctx->mid_rows);
            } else if (vec4) k_exact_rows_lds<<<fb_blocks, 64 * EX_WAVES, 0, st>>>(imgs, pairs, brute, brute_n, ctx->D, ratio, out_dev, out_stride);
            else k_exact_rows<false><<<fb_blocks, 256, 0, st>>>(imgs, pairs, brute, brute_n, ctx->D, ratio, out_dev, out_stride);
            RCN_HIP(hipGetLastError());
            k_stats_acc<<<1, 64, 0, st>>>(ccnt, mid ? 1 : 0, (unsigned)ctx->mid_rows);
            if (tm) RCN_HIP(hipEventRecord(ctx->ev_c[evi][c][2], st));
            if (!uniq_lds) k_unique_claim<<<g, 256, 0, st>>>(imgs, pairs, out_dev, out_stride, ctx->owner.as<int32_t>(), owner_stride, qblocks, p0);
            RCN_HIP(hipGetLastError());
        }
        if (defer) { /* uniqueness follows the deferred pass, for every pair at once */ }
        else if (uniq_lds) k_unique_pair<<<np_c, 256, 0, st>>>(imgs, pairs, out_dev, out_stride, counts_dev, p0);
        else {
            const int eblocks = (int)((out_stride + 255) / 256);
            dim3 g((unsigned)eblocks * (unsigned)np_c);
            k_unique_emit<<<g, 256, 0, st>>>(imgs, pairs, out_dev, out_stride, ctx->owner.as<int32_t>(), owner_stride, counts_dev, eblocks, p0);
        }
        RCN_HIP(hipGetLastError());
        if (tm) RCN_HIP(hipEventRecord(ctx->ev_c[evi][c][kq_max > 0 ? 3 : 2], st));
        if (tm && kq_max <= 0) RCN_HIP(hipEventRecord(ctx->ev_c[evi][c][3], st));
    }
    ctx->ev_tail_on[evi] = false;
    if (defer) {
        // the deferred rows of every chunk in one pass of the tier (their thresholds came with them), K2b for what overflows its
        // candidate lists, then uniqueness for all pairs
        if (prof) { RCN_HIP(hipEventRecord(ctx->ev_tail[evi][0], st)); ctx->ev_tail_on[evi] = true; }
        char *mw = ctx->mid_ws.as<char>();
        RCN_HIP(hipMemsetAsync(ccnt + 2, 0, 2 * sizeof(unsigned), st));      // fb2 count, work items
        MidArgs ma;
        memset(&ma, 0, sizeof(ma));
        ma.imgs = imgs; ma.pairs = pairs; ma.cand = nullptr; ma.fb_list = (const unsigned long long *)(mw + mo_def); ma.fb_count = ccnt + 7;
        ma.thr_in = (const float *)(mw + mo_defthr);
        ma.sorted = (unsigned long long *)(mw + mo_sorted); ma.thr = (float *)(mw + mo_thr); ma.ccount = (unsigned *)(mw + mo_cc);
        ma.clist = (int32_t *)(mw + mo_cl); ma.fb2_list = (unsigned long long *)(mw + mo_fb2);
        ma.hist = (unsigned *)(mw + mo_bins); ma.offs = ma.hist + (n_slots + 1); ma.cursor = ma.offs + (n_slots + 1); ma.ibase = ma.cursor + (n_slots + 1);
        ma.n_items = ccnt + 3; ma.fb2_count = ccnt + 2;
        ma.out = out_dev; ma.out_stride = out_stride; ma.sc = ra.sc; ma.n_slots = n_slots; ma.kq_stride = kq_stride; ma.D = ctx->D;
        ma.all_to_fallback = ra.all_to_fallback; ma.idx_mask = idx_mask; ma.ratio = ratio; ma.midrows = (uint32_t)mid_rows_cap;
        const int mb = ctx->prop.multiProcessorCount * 4;
        k_mid_hist<<<mb, 256, 0, st>>>(ma);
        k_mid_bins<<<1, 1024, 0, st>>>(ma);
        k_mid_scatter<<<mb, 256, 0, st>>>(ma);
        k_mid_eval<<<ctx->prop.multiProcessorCount * 4, 256, 0, st>>>(ma);
        k_mid_exact<<<mb, 256, 0, st>>>(ma);
        k_exact_rows_lds<<<ctx->prop.multiProcessorCount * 8, 64 * EX_WAVES, 0, st>>>(imgs, pairs, ma.fb2_list, ma.fb2_count, ctx->D, ratio, out_dev, out_stride);
        k_stats_acc_deferred<<<1, 64, 0, st>>>(ccnt);
        RCN_HIP(hipGetLastError());
        if (prof) RCN_HIP(hipEventRecord(ctx->ev_tail[evi][1], st));
        k_unique_pair<<<n_pairs, 256, 0, st>>>(imgs, pairs, out_dev, out_stride, counts_dev, 0);
        RCN_HIP(hipGetLastError());
        if (prof) RCN_HIP(hipEventRecord(ctx->ev_tail[evi][2], st));
    }
    if (prof) ctx->ev_n++;
    ctx->last_chunks = n_chunks;
    ctx->last_kq_stride = kq_stride; ctx->last_idx_mask = idx_mask; ctx->last_n_pairs = n_pairs;
    // stats: the fallback count is read back lazily in rcn_match_last_stats
    ctx->last_stats.rows_total = rows;
    ctx->last_stats.pair_distances = pd;
    ctx->last_stats.used_mfma_path = mfma ? 1 : 0;
    ctx->last_stats.rows_exact_fallback = -1;
    ctx->last_stats.err_bound_d2 = -1.0;      // from the scale constants, lazily (rcn_match_last_stats)
    return RCN_OK;
}

// =========================================================================================
extern "C" {

int rcn_desc_upload(rcn_ctx *ctx, int32_t img_id, const float *desc_host, int32_t K, int32_t D)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return upload_common(ctx, img_id, desc_host, false, K, D);
}

int rcn_desc_upload_device(rcn_ctx *ctx, int32_t img_id, const float *desc_dev, int32_t K, int32_t D)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return upload_common(ctx, img_id, desc_dev, true, K, D);
}

int rcn_desc_upload_batch_device(rcn_ctx *ctx, int32_t first_img_id, int32_t n_images,
                                 const float *desc_dev, int32_t K, int32_t D)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return upload_batch(ctx, first_img_id, n_images, desc_dev, K, D);
}

int rcn_desc_upload_batch(rcn_ctx *ctx, int32_t first_img_id, int32_t n_images, const float *const *rows_host,
                          const int32_t *K, int32_t D)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return upload_batch_host(ctx, first_img_id, n_images, rows_host, K, D);
}

int rcn_desc_remove(rcn_ctx *ctx, int32_t img_id)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->images.find(img_id);
    if (it == ctx->images.end()) return RCN_OK;
    RCN_HIP(hipSetDevice(ctx->device));
    RCN_HIP(hipStreamSynchronize(ctx->stream));      // a grid call may still read its rows
    free_image(it->second);
    ctx->images.erase(it);
    retire_unreferenced_slabs(ctx);
    ctx->prepared = false;
    if (ctx->images.empty()) ctx->D = ctx->DP = 0;   // nothing left to pin the descriptor length
    return RCN_OK;
}

int rcn_desc_clear(rcn_ctx *ctx)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_HIP(hipSetDevice(ctx->device));
    RCN_HIP(hipStreamSynchronize(ctx->stream));
    rcn_match_release(ctx);
    if (ctx->counters.p) RCN_HIP(hipMemsetAsync(ctx->counters.p, 0, RCN_COUNTER_BYTES, ctx->stream));
    ctx->hist_rows = 0;
    return RCN_OK;
}

int rcn_desc_count(const rcn_ctx *ctx)
{
    if (!ctx) return 0;
    int n = 0;
    for (const auto &kv : ctx->images)
        if (kv.first != INT32_MIN && kv.first != INT32_MIN + 1) ++n;   // rcn_match_pair's scratch images do not count
    return n;
}

int rcn_match_grid_device(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, float ratio,
                          int32_t *out_dev, int64_t out_stride, int32_t *counts_dev)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return rcn_int_match_grid(ctx, pairs_host, n_pairs, ratio, out_dev, out_stride, counts_dev);
}

int rcn_match_grid(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, float ratio,
                   int32_t *out_host, int64_t out_stride, int32_t *counts_host)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (n_pairs < 0 || (n_pairs > 0 && (!out_host || !counts_host)) || out_stride < 0) {
        ctx->set_error("rcn_match_grid: bad argument");
        return RCN_ERR_ARG;
    }
    if (n_pairs == 0) return RCN_OK;
    // out_stride == 0 is legal when every query image is empty: the device table still gets one
    // column, but nothing is copied into the caller's zero-width rows
    const int64_t stride = std::max<int64_t>(out_stride, 1);
    RCN_HIP(ctx->out_tmp.reserve((size_t)n_pairs * stride * sizeof(int32_t)));
    RCN_HIP(ctx->cnt_tmp.reserve((size_t)n_pairs * sizeof(int32_t)));
    int rc = rcn_int_match_grid(ctx, pairs_host, n_pairs, ratio, ctx->out_tmp.as<int32_t>(), stride,
                             ctx->cnt_tmp.as<int32_t>());
    if (rc) return rc;
    if (out_stride > 0)
        RCN_HIP(hipMemcpyAsync(out_host, ctx->out_tmp.p, (size_t)n_pairs * out_stride * sizeof(int32_t),
                               hipMemcpyDeviceToHost, ctx->stream));
    RCN_HIP(hipMemcpyAsync(counts_host, ctx->cnt_tmp.p, (size_t)n_pairs * sizeof(int32_t),
                           hipMemcpyDeviceToHost, ctx->stream));
    RCN_HIP(hipStreamSynchronize(ctx->stream));
    return RCN_OK;
}

// Lines :232-275 of the pair loop for a list of pairs: match, then (filter != 0) the epipolar filter on the table
// where it lies in HBM, then the dense table, the counts and the filter's verdict per pair back to the host.
int rcn_match_grid_filtered(rcn_ctx *ctx, const int32_t *pairs_host, int32_t n_pairs, float ratio, int32_t filter,
                            int32_t *out_host, int64_t out_stride, int32_t *counts_host, int32_t *status_host)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (n_pairs < 0 || (n_pairs > 0 && (!pairs_host || !out_host || !counts_host)) || out_stride < 1) {
        ctx->set_error("rcn_match_grid_filtered: bad argument");
        return RCN_ERR_ARG;
    }
    if (n_pairs == 0) return RCN_OK;
    RCN_HIP(hipSetDevice(ctx->device));
    RCN_HIP(ctx->out_tmp.reserve((size_t)n_pairs * out_stride * sizeof(int32_t)));
    RCN_HIP(ctx->cnt_tmp.reserve(2 * (size_t)n_pairs * sizeof(int32_t)));       // counts, then the verdicts
    int32_t *tab = ctx->out_tmp.as<int32_t>(), *cnt = ctx->cnt_tmp.as<int32_t>(), *ver = cnt + n_pairs;
    int rc = rcn_int_match_grid(ctx, pairs_host, n_pairs, ratio, tab, out_stride, cnt);
    if (rc) return rc;
    if (filter) {
        rc = rcn_int_table_filter(ctx, pairs_host, n_pairs, tab, out_stride, cnt, ver);
        if (rc) return rc;
    }
    RCN_HIP(hipMemcpyAsync(out_host, tab, (size_t)n_pairs * out_stride * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    RCN_HIP(hipMemcpyAsync(counts_host, cnt, (size_t)n_pairs * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (status_host && filter) RCN_HIP(hipMemcpyAsync(status_host, ver, (size_t)n_pairs * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    RCN_HIP(hipStreamSynchronize(ctx->stream));
    if (status_host && !filter)
        for (int p = 0; p < n_pairs; ++p) status_host[p] = -2;        // "not filtered"
    return RCN_OK;
}

int rcn_match_pair(rcn_ctx *ctx, const float *q_host, int32_t K1, const float *t_host, int32_t K2,
                   int32_t D, float ratio, int32_t *out_train_for_query, int32_t *out_count)
{
    if (!ctx) return RCN_ERR_ARG;
    if (K1 < 0 || K2 < 0 || D <= 0 || (K1 > 0 && (!q_host || !out_train_for_query)) || (K2 > 0 && !t_host)) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->set_error("rcn_match_pair: bad argument");
        return RCN_ERR_ARG;
    }
    // a private two-image context state would serialise callers anyway; reuse a scratch ctx
    // slot pair under the lock (ids INT32_MIN, INT32_MIN+1 are reserved for this call)
    std::lock_guard<std::mutex> lk(ctx->mu);
    { int rcd = drop_scratch_if_alone(ctx, D); if (rcd) return rcd; }
    if (!ctx->images.empty() && ctx->D != D) {
        ctx->set_error("rcn_match_pair: D differs from the resident images' D");
        return RCN_ERR_ARG;
    }
    const int32_t ida = INT32_MIN, idb = INT32_MIN + 1;
    int rc = upload_common(ctx, ida, q_host, false, K1, D);
    if (rc) return rc;
    rc = upload_common(ctx, idb, t_host, false, K2, D);
    if (rc) return rc;
    int32_t pr[2] = {ida, idb};
    int32_t cnt = 0;
    const int64_t stride = std::max(K1, 1);
    RCN_HIP(ctx->out_tmp.reserve((size_t)stride * sizeof(int32_t)));
    RCN_HIP(ctx->cnt_tmp.reserve(sizeof(int32_t)));
    rc = rcn_int_match_grid(ctx, pr, 1, ratio, ctx->out_tmp.as<int32_t>(), stride, ctx->cnt_tmp.as<int32_t>());
    if (rc == RCN_OK) {
        if (K1 > 0)
            RCN_HIP(hipMemcpyAsync(out_train_for_query, ctx->out_tmp.p, (size_t)K1 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        RCN_HIP(hipMemcpyAsync(&cnt, ctx->cnt_tmp.p, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        RCN_HIP(hipStreamSynchronize(ctx->stream));
        if (out_count) *out_count = cnt;
    }
    // the two scratch images stay resident: the next call reuses their allocations
    ctx->prepared = false;
    return rc;
}

int rcn_match_last_stats(const rcn_ctx *cctx, rcn_match_stats *out)
{
    rcn_ctx *ctx = const_cast<rcn_ctx *>(cctx);
    if (!ctx || !out) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->last_stats.rows_exact_fallback < 0 && ctx->counters.p) {
        unsigned n[3];
        RCN_HIP(hipMemcpyAsync(n, ctx->counters.as<unsigned>() + 8 + 4, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
        RCN_HIP(hipStreamSynchronize(ctx->stream));
        ctx->last_stats.rows_exact_fallback = n[0];
        ctx->last_stats.rows_reranked = n[1];
        ctx->last_stats.rows_brute_force = n[2];
    }
    if (ctx->last_stats.err_bound_d2 < 0.0 && ctx->scale_dev.p) {
        { int rcs = rcn_int_resolve_scale(ctx); if (rcs) return rcs; }
        ScaleDev m;
        fix_scale(1.f, 1.0, nullptr, ctx->DP ? ctx->DP : 32, &m);        // c_acc only depends on DP
        const double s2 = ctx->scale * ctx->scale, nq = ctx->max_norm, u = std::ldexp(1.0, -11);
        const double hn_max = 0.5 * s2 * nq * nq + ctx->bias;
        const double eps = (2 * u + u * u) * s2 * nq * nq + std::ldexp(1.0, -14) * std::sqrt((double)(ctx->DP ? ctx->DP : 32)) * ctx->scale * 2 * nq +
                           m.c_acc * (hn_max + s2 * nq * nq) + 6.0e-8 * hn_max;
        ctx->last_stats.err_bound_d2 = 2.0 * eps / s2;
    }
    if (ctx->ev_n > 0 && ctx->ev_made) {
        RCN_HIP(hipStreamSynchronize(ctx->stream));
        const int n = ctx->ev_n < 64 ? ctx->ev_n : 64;
        double c = 0, r = 0, u = 0;
        int launches = 0;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < ctx->ev_chunks[i]; ++k) {
                float ms = 0.f;
                RCN_HIP(hipEventElapsedTime(&ms, ctx->ev_c[i][k][0], ctx->ev_c[i][k][1])); c += ms;
                RCN_HIP(hipEventElapsedTime(&ms, ctx->ev_c[i][k][1], ctx->ev_c[i][k][2])); r += ms;
                RCN_HIP(hipEventElapsedTime(&ms, ctx->ev_c[i][k][2], ctx->ev_c[i][k][3])); u += ms;
                ++launches;
            }
        for (int i = 0; i < n; ++i)
            if (ctx->ev_tail_on[i]) {      // the deferred pass of the middle tier and the uniqueness of all pairs behind the last chunk
                float ms = 0.f;
                RCN_HIP(hipEventElapsedTime(&ms, ctx->ev_tail[i][0], ctx->ev_tail[i][1])); r += ms;
                RCN_HIP(hipEventElapsedTime(&ms, ctx->ev_tail[i][1], ctx->ev_tail[i][2])); u += ms;
            }
        ctx->last_stats.profiled_calls = n;
        ctx->last_stats.coarse_launches = launches;
        ctx->last_stats.coarse_ms = c; ctx->last_stats.rerank_ms = r; ctx->last_stats.unique_ms = u;
        ctx->ev_n = 0;
    }
    ctx->last_stats.chunks = ctx->last_chunks;
    *out = ctx->last_stats;
    return RCN_OK;
}

#ifdef RCN_DIAG
// Diagnostic build only (tools/librcn_diag.so; not declared in include/rcn.h): the packed (best, second)
// table the coarse pass left for the LAST grid call and the constants of its error model, so that a test can
// measure |MFMA accumulator - exact value| on the hardware against the bound the certificates rely on.
// cand_host: n_pairs x kq_stride x 2 words; model[8] = {s, BIAS, Nmax, idx_mask, kq_stride, DP, hn_max, n_pairs}.
int rcn_diag_coarse_table(rcn_ctx *ctx, uint32_t *cand_host, int64_t capacity_words, double *model)
{
    if (!ctx || !model) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->last_chunks != 1) { ctx->set_error("rcn_diag_coarse_table: the last grid ran in several pipeline chunks; the table holds only the last one"); return RCN_ERR_UNSUPPORTED; }
    const int64_t words = 2 * (int64_t)ctx->last_n_pairs * ctx->last_kq_stride;
    { int rcs = rcn_int_resolve_scale(ctx); if (rcs) return rcs; }
    model[0] = ctx->scale; model[1] = ctx->bias; model[2] = ctx->max_norm; model[3] = (double)ctx->last_idx_mask;
    model[4] = (double)ctx->last_kq_stride; model[5] = (double)ctx->DP;
    model[6] = 0.5 * ctx->scale * ctx->scale * ctx->max_norm * ctx->max_norm + ctx->bias; model[7] = (double)ctx->last_n_pairs;
    if (!cand_host) return RCN_OK;
    if (capacity_words < words || !ctx->cand.p) { ctx->set_error("rcn_diag_coarse_table: buffer too small / no grid call yet"); return RCN_ERR_ARG; }
    RCN_HIP(hipSetDevice(ctx->device));
    RCN_HIP(hipMemcpyAsync(cand_host, ctx->cand.p, (size_t)words * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    RCN_HIP(hipStreamSynchronize(ctx->stream));
    return RCN_OK;
}
#endif

// The workspace budget of the grid calls: query-row slots of the candidate table per pipeline chunk (8 bytes each, the row lists at most
// as long again).  A grid larger than that runs in consecutive chunks that reuse the workspace.
int rcn_match_set_workspace_rows(rcn_ctx *ctx, int64_t rows)
{
    if (!ctx || rows < 0) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->chunk_rows = rows == 0 ? RCN_CHUNK_ROWS : std::max<int64_t>(rows, 1);
    return RCN_OK;
}

int rcn_match_profile(rcn_ctx *ctx, int enable)
{
    if (!ctx) return RCN_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    RCN_HIP(hipSetDevice(ctx->device));
    if (enable && !ctx->ev_made) {
        for (auto &call : ctx->ev_c)
            for (auto &row : call)
                for (auto &e : row) RCN_HIP(hipEventCreate(&e));
        for (auto &row : ctx->ev_tail)
            for (auto &e : row) RCN_HIP(hipEventCreate(&e));
        ctx->ev_made = true;
    }
    ctx->profile = enable != 0;
    ctx->ev_n = 0;
    return RCN_OK;
}

}  // extern "C"
