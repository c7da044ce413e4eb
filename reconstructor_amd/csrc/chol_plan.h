// chol_plan.h -- the schedule of the dense blocked Cholesky (K7, ba.hip) as DATA: a list of tile operations in an order that is a
// correct SEQUENTIAL algorithm, each with the stream it runs on and the device counters it has to wait for.  Pure host code (no HIP):
// ba.hip launches the list, tests/test_chol_plan.py executes it in numpy -- in list order and in random orders that respect only the
// waits -- and checks that no two operations the waits leave unordered touch a common tile (one of them writing).
//
// The cross-stream waits are not written by hand: every operation declares the tiles it reads and writes (S = the system and its
// trailing updates, L = the factor's sub-diagonal tiles, Linv_k = the inverse of a diagonal block's factor, SI = the rows of a
// super-block's inverse), the builder keeps the last writer and the readers since of every tile and turns read-after-write,
// write-after-write and write-after-read pairs on different streams into waits.
//
// Two regimes (round 5):
//  * TWO-LEVEL super-steps of g panels while many tile rows remain.  The g x g super-diagonal block is factored by the chain
//    (DIAG / TRSM_Q / UPD_Q on its own tiles only) and inverted block row by block row behind it (SINV: W = L_JJ^-1, g x g tiles);
//    every row below is then ONE product with that inverse, L(i, J) = S(i, J) W' (PGEMM: a tile of column c is a K = 128 (c + 1)
//    pass without a C tile) -- no in-group column passes over the long columns at all -- and the trailing update applies the g
//    panels at once, K = 128 g: the C tile of the bulk kernel travels once per 128 g columns of k.  The next super-diagonal
//    block is brought up to date by a small K = 128 g launch of its own (the chain waits for nothing else), the bulk kernel
//    leads with the tiles the next super-step reads first and counts them out (two classes, two device counters).
//  * the right-looking steps of rounds 3/4 (critical tile first; pairs of panels per bulk update while >= pair_min rows
//    remain) for the rest, where the chain is what bounds the factorisation.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace chol {

enum Kind : int { DIAG = 0, TRSM_Q = 1, UPD_Q = 2, TRSM_PIPE = 3, UPD_PIPE = 4, SINV = 5, PGEMM = 6, PUBLISH = 7 };
enum : int { ST_A = 0, ST_B = 1, ST_C = 2, ST_D = 3, ST_E = 4, N_STREAMS = 5, CTR_SIG1 = 5, CTR_SIG2 = 6, N_CTR = 7 };      // ST_E: the resident diagonal-block workgroup

struct Wait { int ctr, val; };

struct Op {
    int kind = 0, stream = 0, ticket = 0;   // ticket: 1-based position on its stream; the operation publishes ctr[stream] = ticket - 1 when it starts
    int kb = 0;              // DIAG: the block; TRSM / UPD: the (first) panel; SINV / PGEMM: first block of the super-block
    int first = 0, m = 0;    // TRSM_Q / UPD_Q: tile rows kb + 1 + first .. kb + first + m
    int dj = 0;              // UPD_Q: the column that is updated is kb + dj; SINV / PGEMM: which of the two inverse buffers
    int nst = 0;             // pipe kernels: stages of 8 k (K = 8 nst = 128 panels); PGEMM: per tile, 16 (column + 1)
    int map_off = 0, map_n = 0;   // pipe kernels: slice of Plan::maps (row << 16 | class << 14 | column; ~0 = no tile), a multiple of 8 long
    int g = 0, pos = 0;      // SINV: block row pos of the inverse of super-block [kb, kb + g); PGEMM: g
    int nw = 0;
    Wait w[6] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
    int tl = 0;              // slot of the diagnostic build's device timeline
    int small = 0;           // UPD_PIPE / PGEMM: a handful of chain-critical tiles -- the latency form (k_gemm_qm: 16 small workgroups per tile, no LDS) instead of the pipelined one
    int fuse_with = -1;      // PGEMM on the bulk stream: index of the bulk update whose launch carries this product's tiles as its tail (-1: a launch of its own)
    int awaited = 0;         // somebody waits for the operation BEFORE this one on its stream: a pipe operation then needs its gate kernel even without waits of its own
};

struct Params {
    int nblk = 1;
    int tl_g = 4;         // panels per super-step of the two-level regime (0: none)
    int tl_min = 40;      // ... while at least this many tile rows remain below the super-block (below that the chain, not the bulk update, bounds a step)
    int pair = 1;         // right-looking regime: two panels per bulk update ...
    int pair_min = 24;    // ... while at least this many tile rows remain below the pair
    int pipe_min = 32;    // panel / column kernels go through the pipelined kernel from this many tiles on
    int diag_server = 0;  // 1 (tools/, measured and not kept: EXPERIMENTS.md): the diagonal blocks of the whole factorisation in ONE resident workgroup (stream E) that publishes its own ticket when a block is done
    int window = 0;       // two-level regime, measured and NOT shipped: the head rows and the next super-diagonal block through the per-step latency kernels (a 2 g-row
                          // window of the chain).  The chain then reads tiles of the bulk update ONE super-step back instead of two, and waits for it: 8.9 against 8.45 ms
    int tl_serial = 0;    // two-level regime: below this many tile rows under the super-block the chain bounds a step, not the bulk update -- the
                          // super-block's small operations then run ON the chain's stream, in order: no gate kernels, no cross-stream waits
    int head_small = 1;   // two-level regime: the head rows' panel product and the update of the next super-diagonal block through the latency kernel
    int fuse_tail = 1;    // two-level regime: the panel product for the rows below the head rides as the TAIL of the previous bulk update's launch
                          // (it fills that launch's drain; as a launch of its own beside the bulk update both ran 15 % and more slower)
    int bulk_behind = 0;  // right-looking regime, measured and NOT shipped: a bulk update listed BEHIND the next diagonal block, starting when that block's kernel has
                          // started (its first thread publishes the chain's progress).  Workgroups are dispatched launch by launch, and a one-workgroup diagonal
                          // kernel that arrives just behind a bulk launch waits 25-45 us for its several hundred workgroups to be placed (device timeline,
                          // round 5); with the order turned round that wait is gone -- and the critical tile's update waits as much longer for the bulk
                          // update's leading tiles, which now start later: 8.41-8.49 against 8.39-8.47 ms
    int carve_rows = 0;   // right-looking regime, measured and NOT shipped (8.41-8.60 against 8.41 ms; the diagonal kernel shares its CUs with more small kernels and
                          // the carved tiles wait for the WHOLE previous bulk update): from this many tile rows below the step on (0: never) the tiles the CHAIN reads next -- the next group's
                          // diagonal window out of the bulk update, row kb + 2 out of a step's column updates -- go through the latency kernel as
                          // operations of their own, so that the chain waits for a handful of small workgroups, not for a bulk launch's leading tiles
    int pg_stream = 0;    // two-level regime (first super-step, or fuse_tail = 0), 1: the product for the rows below the head runs on a stream of its own (D) -- on B it holds up the
                          // next super-step's in-block work, which the chain waits for.  With the products riding in the bulk launches only the first super-step is
                          // left, and that measures the same either way (8.42-8.51 against 8.36-8.48 ms): the shipping plan keeps to three streams -- a process's
                          // sixth stream is a slow one on this pool (EXPERIMENTS.md), and the library should not be the one that uses the fifth up
};

struct Plan {
    Params prm;
    Params prm_asked;             // what the caller asked for (prm.fuse_tail is 0 when the fused form could not be used)
    std::vector<Op> ops;
    std::vector<uint32_t> maps;
    int n_ops[N_STREAMS] = {0, 0, 0, 0, 0};
    int two_level_steps = 0;      // block steps covered by super-steps
};

inline uint32_t map_entry(int row, int col, int cls = 0) { return ((uint32_t)row << 16) | ((uint32_t)cls << 14) | (uint32_t)col; }

class Builder {
public:
    explicit Builder(const Params &p) : prm(p), nblk(p.nblk), N2((size_t)p.nblk * p.nblk)
    {
        plan.prm = p;
        cells.resize(2 * N2 + (size_t)nblk + 64);
    }
    bool fusion_failed() const { return fuse_bad; }
    Plan build()
    {
        int p = 0;
        const int g = prm.tl_g;
        while (g >= 2 && nblk - p - g >= prm.tl_min && nblk - p - g >= g) { superstep(p, g); p += g; }
        plan.two_level_steps = p;
        right_looking(p);
        // whoever is waited for must be followed by something that publishes its ticket
        for (int s = 0; s < N_STREAMS; ++s) {
            if (plan.n_ops[s] == 0 || s == ST_E) continue;      // (the resident diagonal workgroup publishes a block's ticket when the block is done)
            bool need = false;
            for (const Op &o : plan.ops)
                for (int i = 0; i < o.nw; ++i) need |= o.w[i].ctr == s && o.w[i].val == plan.n_ops[s];
            if (!need) continue;
            Op o; o.kind = PUBLISH; o.stream = s;
            add(o, {}, {});
        }
        // awaited: the operation whose start publishes the ticket somebody waits for
        std::vector<std::vector<char>> aw(N_STREAMS);
        for (int s = 0; s < N_STREAMS; ++s) aw[s].assign((size_t)plan.n_ops[s] + 2, 0);
        for (const Op &o : plan.ops)
            for (int i = 0; i < o.nw; ++i)
                if (o.w[i].ctr < N_STREAMS) aw[o.w[i].ctr][(size_t)o.w[i].val] = 1;
        for (Op &o : plan.ops) o.awaited = aw[o.stream][(size_t)o.ticket - 1];
        return std::move(plan);
    }

private:
    struct Cell { int wop = -1, wcls = 0; std::vector<int> readers; };
    struct Wr { int tile, cls; };
    Params prm;
    int nblk;
    size_t N2;
    Plan plan;
    std::vector<Cell> cells;
    std::vector<int> cum1, cum2;      // per operation: head tiles of class 1 / 2 counted out by the bulk stream up to and including it
    int sig_total[2] = {0, 0};
    int have[N_STREAMS][N_CTR] = {};
    int extra_need[N_CTR] = {-1, -1, -1, -1, -1, -1, -1};      // consumed by the next add()
    bool fuse_bad = false;
    int sib = 0;                      // which of the two super-block inverse buffers the current super-step uses
    int last_bulk = -1;               // the last bulk update of the two-level regime: the next super-step's panel product may ride in its launch

    int tS(int i, int j) const { return i * nblk + j; }
    int tL(int i, int j) const { return (int)N2 + i * nblk + j; }
    int tLinv(int k) const { return (int)(2 * N2) + k; }
    int tSI(int pos) const { return (int)(2 * N2) + nblk + 32 * sib + pos; }      // (two buffers, alternating by super-step: the inverse of the next super-block is built while the last one's product may still run)

    int add(Op o, const std::vector<int> &reads, const std::vector<Wr> &writes)
    {
        int need[N_CTR];
        for (int &v : need) v = -1;
        auto dep = [&](int y, int cls) {
            if (y < 0) return;
            const Op &oy = plan.ops[(size_t)y];
            if (oy.stream == o.stream && o.fuse_with == y && cls == 0) fuse_bad = true;      // a tail that needs an ordinary tile of its host cannot ride in its launch
            if (oy.stream == o.stream && !(o.fuse_with == y && cls > 0)) return;      // stream order (a tail inside its host's launch still waits for the host's leading tiles it reads)
            if (cls > 0) { int &v = need[CTR_SIG1 + cls - 1]; v = std::max(v, cls == 1 ? cum1[(size_t)y] : cum2[(size_t)y]); }
            else { int &v = need[oy.stream]; v = std::max(v, oy.ticket); }
        };
        for (int c = 0; c < N_CTR; ++c) { need[c] = std::max(need[c], extra_need[c]); extra_need[c] = -1; }      // (an ordering asked for on top of the tiles': bulk_behind)
        for (int t : reads) dep(cells[(size_t)t].wop, cells[(size_t)t].wcls);
        for (const Wr &w : writes) {
            Cell &c = cells[(size_t)w.tile];
            dep(c.wop, c.wcls);
            for (int r : c.readers) dep(r, 0);
        }
        // Whoever is waited for must be followed by something that publishes its ticket -- and SOON: the counter moves when the next
        // operation of that stream starts.  If the awaited operation is the last one queued on its stream so far, a publisher goes
        // into the list right here, in front of the waiter (the first version left it to the end of the list: the only two
        // operations of the fourth stream were published when the host had enqueued the whole factorisation, 0.5 ms late).
        if (o.kind != PUBLISH)
            for (int c = 0; c < N_STREAMS; ++c)
                if (c != o.stream && c != ST_E && need[c] > have[o.stream][c] && need[c] == plan.n_ops[c]) {
                    Op pb; pb.kind = PUBLISH; pb.stream = c;
                    add(pb, {}, {});
                }
        const int idx = (int)plan.ops.size();
        o.ticket = ++plan.n_ops[o.stream];
        // (what an earlier operation of this stream has waited for, this one has too: streams run in order, counters only grow)
        o.nw = 0;
        for (int c = 0; c < N_CTR; ++c)
            if (need[c] > have[o.stream][c]) { o.w[o.nw].ctr = c; o.w[o.nw].val = need[c]; ++o.nw; have[o.stream][c] = need[c]; }      // (at most 4 other streams + 2 head counters)
        int n1 = 0, n2 = 0;
        for (int t : reads) cells[(size_t)t].readers.push_back(idx);
        for (const Wr &w : writes) {
            Cell &c = cells[(size_t)w.tile];
            c.wop = idx; c.wcls = w.cls; c.readers.clear();
            n1 += w.cls == 1; n2 += w.cls == 2;
        }
        sig_total[0] += n1; sig_total[1] += n2;
        cum1.push_back(sig_total[0]); cum2.push_back(sig_total[1]);
        plan.ops.push_back(o);
        return idx;
    }

    // ---- single operations
    int diag(int k)
    {
        Op o; o.kind = DIAG; o.stream = prm.diag_server ? ST_E : ST_A; o.kb = k; o.tl = 8 * k + 0;
        return add(o, {tS(k, k)}, {{tLinv(k), 0}, {tS(k, k), 0}});
    }
    void trsm_q(int stream, int kb, int first, int m, int tl)
    {
        if (m <= 0) return;
        Op o; o.kind = TRSM_Q; o.stream = stream; o.kb = kb; o.first = first; o.m = m; o.tl = tl;
        std::vector<int> rd{tLinv(kb)};
        std::vector<Wr> wr;
        for (int r = kb + 1 + first; r <= kb + first + m; ++r) { rd.push_back(tS(r, kb)); wr.push_back({tL(r, kb), 0}); }
        add(o, rd, wr);
    }
    void upd_q(int stream, int kb, int first, int m, int dj, int tl)
    {
        if (m <= 0) return;
        Op o; o.kind = UPD_Q; o.stream = stream; o.kb = kb; o.first = first; o.m = m; o.dj = dj; o.tl = tl;
        std::vector<int> rd{tL(kb + dj, kb)};
        std::vector<Wr> wr;
        for (int r = kb + 1 + first; r <= kb + first + m; ++r) { rd.push_back(tL(r, kb)); wr.push_back({tS(r, kb + dj), 0}); }
        add(o, rd, wr);
    }
    // a map: entries dealt to the XCDs round-robin (workgroup b runs on XCD b % 8), padded with ~0 to a multiple of 8
    int put_map(const std::vector<uint32_t> (&per)[8], int *n_out)
    {
        size_t slots = 0;
        for (const auto &v : per) slots = std::max(slots, v.size());
        const int off = (int)plan.maps.size();
        for (size_t sl = 0; sl < slots; ++sl)
            for (int x = 0; x < 8; ++x) plan.maps.push_back(sl < per[x].size() ? per[x][sl] : ~0u);
        *n_out = (int)(8 * slots);
        return off;
    }
    int put_list(const std::vector<uint32_t> &tiles, int *n_out)
    {
        std::vector<uint32_t> per[8];
        for (size_t i = 0; i < tiles.size(); ++i) per[i & 7].push_back(tiles[i]);
        return put_map(per, n_out);
    }
    void trsm_pipe(int stream, int kb, int r0, int r1, int tl)      // rows r0 .. r1 - 1 of panel kb
    {
        if (r1 <= r0) return;
        Op o; o.kind = TRSM_PIPE; o.stream = stream; o.kb = kb; o.nst = 16; o.tl = tl;
        std::vector<uint32_t> tiles;
        std::vector<int> rd{tLinv(kb)};
        std::vector<Wr> wr;
        for (int r = r0; r < r1; ++r) { tiles.push_back(map_entry(r, 0)); rd.push_back(tS(r, kb)); wr.push_back({tL(r, kb), 0}); }
        o.map_off = put_list(tiles, &o.map_n);
        add(o, rd, wr);
    }
    // S(i, j) -= L(i, kb ..) L(j, kb ..)' over npan panels for the listed tiles (class in the entry)
    int upd_pipe(int stream, int kb, int npan, const std::vector<uint32_t> (&per)[8], int tl, int small = 0)
    {
        Op o; o.kind = UPD_PIPE; o.stream = stream; o.kb = kb; o.nst = 16 * npan; o.tl = tl; o.small = small;
        o.map_off = put_map(per, &o.map_n);
        if (o.map_n == 0) return -1;
        std::vector<int> rd;
        std::vector<Wr> wr;
        std::vector<char> seen((size_t)nblk, 0);
        for (const auto &v : per)
            for (uint32_t e : v) {
                const int i = (int)(e >> 16), j = (int)(e & 0x3fffu), cls = (int)((e >> 14) & 3u);
                wr.push_back({tS(i, j), cls});
                for (int r : {i, j})
                    if (!seen[(size_t)r]) { seen[(size_t)r] = 1; for (int q = 0; q < npan; ++q) rd.push_back(tL(r, kb + q)); }
            }
        return add(o, rd, wr);
    }
    void upd_pipe_list(int stream, int kb, int npan, const std::vector<uint32_t> &tiles, int tl, int small = 0)
    {
        std::vector<uint32_t> per[8];
        for (size_t i = 0; i < tiles.size(); ++i) per[i & 7].push_back(tiles[i]);
        upd_pipe(stream, kb, npan, per, tl, small);
    }
    void sinv(int stream, int p, int g, int pos, int tl)
    {
        Op o; o.kind = SINV; o.stream = stream; o.kb = p; o.g = g; o.pos = pos; o.tl = tl; o.dj = sib;
        std::vector<int> rd{tLinv(p + pos)};
        for (int r = 0; r < pos; ++r) { rd.push_back(tL(p + pos, p + r)); rd.push_back(tSI(r)); }
        add(o, rd, {{tSI(pos), 0}});
    }
    // L(i, p + c) = sum_{m <= c} S(i, p + m) SI[c][m]'  for rows r0 .. r1 - 1, columns 1 .. g - 1 (column 0 is a plain panel product)
    // (c_lo = 0: column 0 too -- as a 24-stage pass whose last 64 columns meet the zero block W[0][1]; a tail has no other kernel to send it to)
    void pgemm(int stream, int p, int g, int r0, int r1, int tl, int c_lo = 1, int fuse_with = -1, int small = 0)
    {
        if (r1 <= r0 || g < 2) return;
        Op o; o.kind = PGEMM; o.stream = stream; o.kb = p; o.g = g; o.tl = tl; o.fuse_with = fuse_with; o.small = small; o.dj = sib;
        std::vector<uint32_t> tiles;
        std::vector<int> rd;
        std::vector<Wr> wr;
        for (int c = g - 1; c >= c_lo; --c)              // longest passes first
            for (int r = r0; r < r1; ++r) { tiles.push_back(map_entry(r, c)); wr.push_back({tL(r, p + c), 0}); }
        for (int c = c_lo; c < g; ++c) rd.push_back(tSI(c));
        for (int r = r0; r < r1; ++r)
            for (int c = 0; c < g; ++c) rd.push_back(tS(r, p + c));
        o.map_off = put_list(tiles, &o.map_n);
        add(o, rd, wr);
    }

    // Tiles of a trailing region in launch order.  lead: tiles that go first, in the given order, rows dealt to the XCDs eight apart
    // (the tiles of a row share its panel rows in that XCD's L2); the rest of the lower triangle [c0, nblk) x [c0, nblk) -- minus
    // what `skip` names -- follows in WHOLE supertiles (4 x 4 tiles, 2 x 2 below 24 rows), heaviest first to the least loaded XCD.
    template <class Skip>
    void region_map(int c0, const std::vector<uint32_t> &lead, Skip skip, std::vector<uint32_t> (&per)[8])
    {
        for (uint32_t e : lead) per[(e >> 16) & 7u].push_back(e);
        const int mt = nblk - c0;
        const int SS = mt >= 24 ? 4 : 2, R = (mt + SS - 1) / SS;
        std::vector<std::vector<uint32_t>> st;
        for (int sr = 0; sr < R; ++sr)
            for (int sc = 0; sc <= sr; ++sc) {
                std::vector<uint32_t> tl;
                for (int r = sr * SS; r < std::min(mt, sr * SS + SS); ++r)
                    for (int c = sc * SS; c < sc * SS + SS; ++c)
                        if (c <= r && !skip(c0 + r, c0 + c)) tl.push_back(map_entry(c0 + r, c0 + c));
                if (!tl.empty()) st.push_back(std::move(tl));
            }
        std::stable_sort(st.begin(), st.end(), [](const std::vector<uint32_t> &a, const std::vector<uint32_t> &b) { return a.size() > b.size(); });
        for (auto &tl : st) {
            int x = 0;
            for (int i = 1; i < 8; ++i)
                if (per[i].size() < per[x].size()) x = i;
            per[x].insert(per[x].end(), tl.begin(), tl.end());
        }
    }

    // ---- a two-level super-step: panels p .. p + g - 1
    void superstep(int p, int g)
    {
        sib ^= 1;
        const int R0 = p + g, H1 = std::min(nblk, R0 + g);
        const int sb = nblk - R0 < prm.tl_serial ? ST_A : ST_B;      // where the super-block's small operations run
        const int kl = R0 - 1;            // the timeline files the super-step's own operations under its last block step
        if (prm.window) {
            // WINDOW form (round 5, second pass): the chain's window is the super-block AND the next super-diagonal block's rows, 2 g tile
            // rows.  Every step's panel product and update cover the window's tiles, K = 128, latency kernels: when the super-block's
            // last diagonal block is done, the head rows of the panel and the next super-diagonal block are one panel product and one
            // update away -- no inverse, no K = 128 g launches on the chain's critical path (the inverse is still built, behind the
            // chain, for the rows below the window).
            for (int pos = 0; pos < g; ++pos) {
                const int k = p + pos;
                diag(k);
                if (k + 1 < H1) {
                    trsm_q(ST_A, k, 0, 1, 8 * k + 1);
                    upd_q(ST_A, k, 0, 1, 1, 8 * k + 2);
                }
                if (k + 2 < H1) {
                    trsm_q(sb, k, 1, H1 - k - 2, 8 * k + 3);
                    upd_q(sb, k, 1, H1 - k - 2, 1, 8 * k + 4);            // column k + 1: the next critical tile waits for it
                    std::vector<uint32_t> tiles;
                    for (int j = k + 2; j < H1; ++j)
                        for (int i = j; i < H1; ++i) tiles.push_back(map_entry(i, j));
                    if (!tiles.empty()) upd_pipe_list(sb, k, 1, tiles, 8 * k + 5, 1);
                }
                sinv(sb, p, g, pos, 8 * k + 7);
            }
        } else {
        for (int pos = 0; pos < g; ++pos) {
            const int k = p + pos, nin = g - 1 - pos;      // rows of the super-block below block k
            diag(k);
            if (nin >= 1) {
                trsm_q(ST_A, k, 0, 1, 8 * k + 1);
                upd_q(ST_A, k, 0, 1, 1, 8 * k + 2);
            }
            if (nin >= 2) {
                trsm_q(sb, k, 1, nin - 1, 8 * k + 3);
                upd_q(sb, k, 1, nin - 1, 1, 8 * k + 4);
                // the other columns of the super-block, k + 2 .. p + g - 1
                int cnt = 0;
                for (int j = k + 2; j < R0; ++j) cnt += R0 - j;
                if (cnt <= 3) {
                    for (int j = k + 2; j < R0; ++j) upd_q(sb, k, j - k - 1, R0 - j, j - k, 8 * k + 5);
                } else {
                    std::vector<uint32_t> tiles;
                    for (int j = k + 2; j < R0; ++j)
                        for (int i = j; i < R0; ++i) tiles.push_back(map_entry(i, j));
                    upd_pipe_list(sb, k, 1, tiles, 8 * k + 5);
                }
            }
            sinv(sb, p, g, pos, 8 * k + 7);
        }
        // head rows: the next super-diagonal block's rows of this super-panel, then that block itself
        if (prm.head_small) pgemm(sb, p, g, R0, H1, 8 * kl + 4, 0, -1, 1);
        else {
            trsm_q(sb, p, g - 1, H1 - R0, 8 * kl + 3);
            pgemm(sb, p, g, R0, H1, 8 * kl + 4);
        }
        {
            std::vector<uint32_t> tiles;
            for (int i = R0; i < H1; ++i)
                for (int j = R0; j <= i; ++j) tiles.push_back(map_entry(i, j));
            upd_pipe_list(sb, p, g, tiles, 8 * kl + 5, prm.head_small);
        }
        }
        if (H1 >= nblk) return;
        // every row below
        if (prm.fuse_tail && last_bulk >= 0) {
            pgemm(ST_C, p, g, H1, nblk, 8 * kl + 2, 0, last_bulk);
        } else {
            const int sp = prm.pg_stream ? ST_D : ST_B;
            if (nblk - H1 >= 8) trsm_pipe(sp, p, H1, nblk, 8 * kl + 1);
            else trsm_q(sp, p, H1 - p - 1, nblk - H1, 8 * kl + 1);
            pgemm(sp, p, g, H1, nblk, 8 * kl + 2);
        }
        // the trailing update, K = 128 g: everything from column R0 on but the next super-diagonal block.  Class 1 (counted out
        // first): what the head of the NEXT super-step reads -- its head rows' tiles of its own panel columns and the super-diagonal
        // block behind them; class 2: the rest of those panel columns.
        {
            const int H2 = std::min(nblk, H1 + g);
            std::vector<uint32_t> lead;
            for (int i = H1; i < H2; ++i) {
                for (int j = R0; j < H1; ++j) lead.push_back(map_entry(i, j, 1));
                for (int j = H1; j <= i; ++j) lead.push_back(map_entry(i, j, 1));
            }
            for (int i = H2; i < nblk; ++i)
                for (int j = R0; j < H1; ++j) lead.push_back(map_entry(i, j, 2));
            std::vector<uint32_t> per[8];
            region_map(R0, lead, [&](int i, int j) { return (i < H1) || (j < H1) || (i < H2 && j < H2); }, per);
            last_bulk = upd_pipe(ST_C, p, g, per, 8 * kl + 6);
        }
    }

    // ---- right-looking steps from block p on (rounds 3 / 4: critical tile first, pairs of panels per bulk update)
    void right_looking(int p)
    {
        int g_size = 1, g_pos = 0, diag_listed = -1;
        for (int kb = p; kb < nblk; ++kb) {
            const int m = nblk - kb - 1;
            if (g_pos == 0) g_size = (prm.pair && m - 2 >= prm.pair_min && m >= 4) ? 2 : 1;
            const int ncols = g_size - g_pos;
            const bool last_of_group = g_pos == g_size - 1;
            if (diag_listed != kb) diag(kb);
            if (m <= 0) break;
            trsm_q(ST_A, kb, 0, 1, 8 * kb + 1);
            upd_q(ST_A, kb, 0, 1, 1, 8 * kb + 2);
            if (m <= 1) continue;
            const bool piped = m - 1 >= prm.pipe_min;
            const bool carve = prm.carve_rows > 0 && m <= prm.carve_rows;
            if (piped) trsm_pipe(ST_B, kb, kb + 2, nblk, 8 * kb + 3);
            else trsm_q(ST_B, kb, 1, m - 1, 8 * kb + 3);
            if (carve) {
                // row kb + 2 of the column updates first and by itself: what the next steps of the chain read
                std::vector<uint32_t> crit;
                for (int c = 1; c <= ncols && kb + c <= kb + 2; ++c) crit.push_back(map_entry(kb + 2, kb + c));
                upd_pipe_list(ST_B, kb, 1, crit, 8 * kb + 4, 1);
            }
            const int r_first = carve ? kb + 3 : kb + 2;
            if (piped) {
                std::vector<uint32_t> tiles;
                for (int r = r_first; r < nblk; ++r)
                    for (int c = 1; c <= ncols && kb + c <= r; ++c) tiles.push_back(map_entry(r, kb + c));
                if (!tiles.empty()) upd_pipe_list(ST_B, kb, 1, tiles, 8 * kb + 4);
            } else {
                for (int c = 1; c <= ncols; ++c) upd_q(ST_B, kb, r_first - kb - 1, nblk - r_first, c, 8 * kb + 3 + c);
            }
            if (!last_of_group) { ++g_pos; continue; }
            g_pos = 0;
            const int c0 = kb + 2, kfirst = kb - (g_size - 1);
            // (carve: the next group's diagonal window -- what its chain steps read of this update -- as an operation of its own)
            const int xw = carve ? std::min(nblk, c0 + g_size) : c0;
            if (carve) {
                std::vector<uint32_t> xt;
                for (int i = c0; i < xw; ++i)
                    for (int j = c0; j <= i; ++j) xt.push_back(map_entry(i, j));
                upd_pipe_list(ST_B, kfirst, g_size, xt, 8 * kb + 5, 1);
            }
            std::vector<uint32_t> lead;
            for (int i = c0; i < nblk; ++i)
                for (int j = c0; j < c0 + g_size && j <= i; ++j)
                    if (i >= xw) lead.push_back(map_entry(i, j, 1));
            std::vector<uint32_t> per[8];
            region_map(c0, lead, [&](int, int j) { return j < c0 + g_size; }, per);
            if (prm.bulk_behind && !prm.diag_server) {
                // the next diagonal block first (it needs the critical tile's update only), and the bulk update not before its kernel runs
                const Op &dn = plan.ops[(size_t)diag(kb + 1)];
                diag_listed = kb + 1;
                extra_need[dn.stream] = dn.ticket - 1;
            }
            upd_pipe(ST_C, kfirst, g_size, per, 8 * kb + 6);
            for (int &v : extra_need) v = -1;
        }
    }
};

inline Plan make_plan(const Params &p)
{
    Builder b(p);
    Plan pl = b.build();
    if (!b.fusion_failed()) return pl;
    Params q = p;
    q.fuse_tail = 0;
    return Builder(q).build();
}

}  // namespace chol
