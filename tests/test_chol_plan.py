"""The schedule of the dense factorisation inside rcn_ba_solve (csrc/chol_plan.h, exported by rcn_ba_factor_plan) is DATA: a list of
tile operations with streams and waits.  No GPU here -- the list is executed in numpy on a small tile size:

 * in list order it must BE a Cholesky factorisation (L L' = A);
 * in random orders that respect nothing but stream order and the declared waits it must give the same bits;
 * statically: any two operations that touch a common tile, one of them writing, must be ordered by stream order and waits
   (read-after-write, write-after-write and write-after-read), with the tiles derived HERE from the operations' parameters and
   maps, not taken from the builder.

This is what stands in for running the three-stream schedule on hardware in the CPU suite; the GPU suite runs the real thing
against the oracle (tests/test_ba_gpu.py) and twice for bit-reproducibility (tools/soak_ba_large.py)."""
import ctypes as C
import random

import numpy as np
import pytest

from reconstructor_amd import _lib

DIAG, TRSM_Q, UPD_Q, TRSM_PIPE, UPD_PIPE, SINV, PGEMM, PUBLISH = range(8)
OPW = 29
N_STREAMS = 5          # counters 0 .. 4: progress of the streams (4: the resident workgroup of the diagonal blocks); 5, 6: the two classes of leading tiles of the bulk updates
NONE = 0xFFFFFFFF


def get_plan(nblk, params=None):
    lib = _lib.load()
    fn = lib.rcn_ba_factor_plan
    fn.restype = C.c_int
    fn.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    n_ops, n_maps = C.c_int64(0), C.c_int64(0)
    prm = None
    if params is not None:
        params = tuple(params) + ((0,) if len(params) == 10 else ())
        params = params + ((0,) if len(params) == 11 else ())      # (eleven entries: bulk updates in front of the next diagonal block, the product's way)
        params = params + ((0,) if len(params) == 12 else ())      # (twelve entries: nothing carved out of the right-looking regime's updates)
        prm = np.asarray(params, dtype=np.int32)      # (ten entries: the diagonal blocks as launches on the chain's stream, the product's way)
    pp = prm.ctypes.data if prm is not None else None
    fn(nblk, pp, None, 0, None, 0, C.byref(n_ops), C.byref(n_maps))
    ops = np.zeros((max(n_ops.value, 1), OPW), dtype=np.int32)
    maps = np.zeros(max(n_maps.value, 1), dtype=np.uint32)
    rc = fn(nblk, pp, ops.ctypes.data, n_ops.value, maps.ctypes.data, n_maps.value, C.byref(n_ops), C.byref(n_maps))
    assert rc == 0
    out = []
    for r in ops[:n_ops.value]:
        o = dict(kind=int(r[0]), stream=int(r[1]), ticket=int(r[2]), kb=int(r[3]), first=int(r[4]), m=int(r[5]), dj=int(r[6]), nst=int(r[7]),
                 map_off=int(r[8]), map_n=int(r[9]), g=int(r[10]), pos=int(r[11]), waits=[(int(r[13 + 2 * i]), int(r[14 + 2 * i])) for i in range(int(r[12]))],
                 awaited=int(r[26]), fuse_with=int(r[27]), small=int(r[28]))
        o["tiles"] = [((int(e) >> 16), int(e) & 0x3FFF, (int(e) >> 14) & 3) for e in maps[o["map_off"]:o["map_off"] + o["map_n"]] if int(e) != NONE]
        out.append(o)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# what an operation reads and writes, from its parameters alone: ('S' | 'L', i, j), ('Linv', k), ('SI', pos); writes carry a class
def accesses(o):
    k, rd, wr = o["kb"], [], []
    rows = range(k + 1 + o["first"], k + 1 + o["first"] + o["m"])
    if o["kind"] == DIAG:
        rd = [("S", k, k)]
        wr = [(("Linv", k), 0), (("S", k, k), 0)]
    elif o["kind"] == TRSM_Q:
        rd = [("Linv", k)] + [("S", r, k) for r in rows]
        wr = [(("L", r, k), 0) for r in rows]
    elif o["kind"] == UPD_Q:
        j = k + o["dj"]
        rd = [("L", j, k)] + [("L", r, k) for r in rows]
        wr = [(("S", r, j), 0) for r in rows]
    elif o["kind"] == TRSM_PIPE:
        rd = [("Linv", k)] + [("S", i, k) for i, _, _ in o["tiles"]]
        wr = [(("L", i, k), 0) for i, _, _ in o["tiles"]]
    elif o["kind"] == UPD_PIPE:
        npan = o["nst"] // 16
        for i, j, cls in o["tiles"]:
            wr.append((("S", i, j), cls))
            for q in range(npan):
                rd += [("L", i, k + q), ("L", j, k + q)]
    elif o["kind"] == SINV:
        c = k + o["pos"]
        rd = [("Linv", c)] + [("L", c, k + r) for r in range(o["pos"])] + [("SI", o["dj"], r) for r in range(o["pos"])]
        wr = [(("SI", o["dj"], o["pos"]), 0)]
    elif o["kind"] == PGEMM:
        for i, col, _ in o["tiles"]:
            wr.append((("L", i, k + col), 0))
            # (a column-0 tile runs as a K = 192 pass: its last 64 columns are the next panel column's, met by a zero block of the inverse)
            rd += [("SI", o["dj"], col)] + [("S", i, k + q) for q in range(2 if (col == 0 and not o["small"]) else col + 1)]
    return rd, wr


class Machine:
    """numpy semantics of the operations on tiles of T x T"""

    def __init__(self, nblk, T, A):
        self.nblk, self.T = nblk, T
        self.S = A.copy()
        self.L = np.zeros_like(A)
        self.Linv = [None] * nblk
        self.Ld = [None] * nblk
        self.SI = {}

    def t(self, M, i, j):
        T = self.T
        return M[i * T:(i + 1) * T, j * T:(j + 1) * T]

    def run(self, o, part=None):
        """part: None = the whole operation; 0 / 1 / 2 = only its tiles of that class (a bulk launch's tiles finish in any order)"""
        k = o["kb"]
        rows = range(k + 1 + o["first"], k + 1 + o["first"] + o["m"])
        if o["kind"] == DIAG:
            Lk = np.linalg.cholesky(self.t(self.S, k, k))
            self.Ld[k] = Lk
            self.Linv[k] = np.linalg.inv(Lk)
        elif o["kind"] == TRSM_Q:
            for r in rows:
                self.t(self.L, r, k)[:] = self.t(self.S, r, k) @ self.Linv[k].T
        elif o["kind"] == UPD_Q:
            j = k + o["dj"]
            for r in rows:
                self.t(self.S, r, j)[:] -= self.t(self.L, r, k) @ self.t(self.L, j, k).T
        elif o["kind"] == TRSM_PIPE:
            for i, _, _ in o["tiles"]:
                self.t(self.L, i, k)[:] = self.t(self.S, i, k) @ self.Linv[k].T
        elif o["kind"] == UPD_PIPE:
            npan = o["nst"] // 16
            for i, j, cls in o["tiles"]:
                if part is not None and cls != part:
                    continue
                acc = np.zeros((self.T, self.T))
                for q in range(npan):
                    acc += self.t(self.L, i, k + q) @ self.t(self.L, j, k + q).T
                self.t(self.S, i, j)[:] -= acc
        elif o["kind"] == SINV:
            pos, c = o["pos"], k + o["pos"]
            b = o["dj"]
            self.SI[(b, pos, pos)] = self.Linv[c].copy()
            for m in range(pos):
                Y = np.zeros((self.T, self.T))
                for r in range(m, pos):
                    Y += self.t(self.L, c, k + r) @ self.SI[(b, r, m)]
                self.SI[(b, pos, m)] = -self.Linv[c] @ Y
        elif o["kind"] == PGEMM:
            for i, col, _ in o["tiles"]:
                acc = np.zeros((self.T, self.T))
                for m in range(col + 1):
                    acc += self.t(self.S, i, k + m) @ self.SI[(o["dj"], col, m)].T
                self.t(self.L, i, k + col)[:] = acc

    def factor(self):
        F = np.tril(self.L, -1).copy()
        T = self.T
        for k in range(self.nblk):
            F[k * T:(k + 1) * T, k * T:(k + 1) * T] = self.Ld[k]
        return F


def spd(n, seed):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    return M @ M.T + n * np.eye(n)


# ---------------------------------------------------------------------------------------------------------------------
# nodes of the happens-before graph: (op, part); a bulk launch with head classes has parts 0 (unclassed tiles), 1, 2
def graph(ops):
    parts = []
    for o in ops:
        cl = sorted({c for _, _, c in o["tiles"]}) if o["kind"] == UPD_PIPE and o["stream"] == 2 else [0]
        parts.append(cl if len(cl) > 1 or (cl and cl[0] != 0) else [0])
    node_id, nodes = {}, []
    for i, ps in enumerate(parts):
        for pt in ps:
            node_id[(i, pt)] = len(nodes)
            nodes.append((i, pt))
    by_ticket = {(o["stream"], o["ticket"]): i for i, o in enumerate(ops)}
    # cumulative head counts per class on the bulk stream
    cum = {1: [], 2: []}
    tot = {1: 0, 2: 0}
    for i, o in enumerate(ops):
        for c in (1, 2):
            n = sum(1 for _, _, cc in o["tiles"] if cc == c) if o["kind"] == UPD_PIPE and o["stream"] == 2 else 0
            tot[c] += n
            cum[c].append(tot[c])
    preds = [set() for _ in nodes]
    last_on_stream = {}
    prev_on_stream = {}
    for i, o in enumerate(ops):
        p = set()
        h = o.get("fuse_with", -1)
        if h >= 0:
            # the tail of bulk update h's launch: its workgroups follow h's in dispatch order but do not wait for them -- what holds them
            # is their own waits (h's leading tiles among them) and whatever preceded the launch on the stream
            assert ops[h]["kind"] == UPD_PIPE and ops[h]["stream"] == o["stream"] == 2 and last_on_stream[2] == h
            if h in prev_on_stream:
                j = prev_on_stream[h]
                p |= {node_id[(j, pt)] for pt in parts[j]}
        elif o["stream"] in last_on_stream:
            j = last_on_stream[o["stream"]]
            p |= {node_id[(j, pt)] for pt in parts[j]}
            hj = ops[j].get("fuse_with", -1)
            if hj >= 0:                       # behind a launch with a tail: behind both of its parts
                p |= {node_id[(hj, pt)] for pt in parts[hj]}
        if o["stream"] in last_on_stream:
            prev_on_stream[i] = last_on_stream[o["stream"]]
        for ctr, val in o["waits"]:
            if ctr < N_STREAMS:
                if val == 0:
                    continue
                j = by_ticket[(ctr, val)]
                assert j < i, "an operation waits for a later one"
                p |= {node_id[(j, pt)] for pt in parts[j]}
            else:
                c = ctr - N_STREAMS + 1
                js = [j for j in range(i) if cum[c][j] >= val and c in parts[j]]
                assert js, "a head count nobody reaches before operation %d" % i
                j = js[0]
                assert cum[c][j] == val, "a head count that is not the end of a launch's class"
                p.add(node_id[(j, c)])
        for pt in parts[i]:
            preds[node_id[(i, pt)]] = set(p)
        last_on_stream[o["stream"]] = i
    return parts, nodes, node_id, preds


def check_races(ops):
    parts, nodes, node_id, preds = graph(ops)
    anc = [0] * len(nodes)
    for v in range(len(nodes)):           # nodes are in list order: predecessors come first
        a = 0
        for u in preds[v]:
            a |= anc[u] | (1 << u)
        anc[v] = a
    last_w, readers = {}, {}
    n_pairs = 0
    for i, o in enumerate(ops):
        rd, wr = accesses(o)
        my_nodes = [node_id[(i, pt)] for pt in parts[i]]
        for t in rd:
            if t in last_w:
                u = last_w[t]
                if nodes[u][0] != i:
                    for v in my_nodes:
                        assert (anc[v] >> u) & 1, "operation %d %r reads %r without waiting for its writer %r" % (i, o, t, ops[nodes[u][0]])
                        n_pairs += 1
            readers.setdefault(t, []).append(my_nodes)
        for t, cls in wr:
            v = node_id[(i, cls)] if (i, cls) in node_id else my_nodes[0]
            if t in last_w and nodes[last_w[t]][0] != i:
                assert (anc[v] >> last_w[t]) & 1, "operation %d %r rewrites %r without waiting for its last writer" % (i, o, t)
                n_pairs += 1
            for rn in readers.get(t, []):
                for u in rn:
                    if nodes[u][0] != i:
                        assert (anc[v] >> u) & 1, "operation %d %r overwrites %r while %r may still read it" % (i, o, t, ops[nodes[u][0]])
                        n_pairs += 1
            last_w[t] = v
            readers[t] = []
    return n_pairs


def run_list_order(ops, nblk, T, A):
    mc = Machine(nblk, T, A)
    for o in ops:
        mc.run(o)
    return mc


def run_random_order(ops, nblk, T, A, seed):
    parts, nodes, node_id, preds = graph(ops)
    rnd = random.Random(seed)
    done = [False] * len(nodes)
    left = set(range(len(nodes)))
    mc = Machine(nblk, T, A)
    while left:
        ready = [v for v in left if all(done[u] for u in preds[v])]
        assert ready, "the waits form a cycle"
        v = rnd.choice(ready)
        i, pt = nodes[v]
        mc.run(ops[i], part=pt if parts[i] != [0] else None)
        done[v] = True
        left.discard(v)
    return mc


CASES = [
    (1, None), (2, None), (3, None), (6, None), (16, None),            # the reference's own sizes, cfg 4
    (40, None), (79, None),                                             # cfg 5: two-level super-steps, pairs, single steps
    (45, (8, 20, 1, 24, 32, 1, 1, 1, 30, 1)), (37, (4, 8, 0, 24, 6, 0, 1, 0, 0, 0)), (30, (2, 6, 1, 10, 4, 1, 0, 1, 99, 1)), (26, (0, 0, 1, 8, 4, 1, 1, 1, 0, 1)), (23, (0, 0, 0, 8, 64, 0, 0, 0, 0, 0)), (79, (4, 28, 1, 24, 32, 0, 0, 0, 0, 0)), (79, (4, 16, 1, 24, 32, 1, 1, 1, 44, 0)), (79, (4, 16, 1, 24, 32, 1, 1, 1, 0, 1)), (60, (4, 12, 1, 16, 8, 1, 1, 0, 30, 1)), (79, (4, 40, 1, 24, 32, 1, 1, 1, 0, 0, 0, 1)), (33, (0, 0, 1, 8, 4, 1, 1, 1, 0, 0, 0, 1)), (79, (4, 40, 1, 24, 32, 1, 1, 1, 0, 0, 0, 1, 34)), (41, (0, 0, 1, 8, 12, 1, 1, 1, 0, 0, 0, 1, 99)), (30, (0, 0, 1, 24, 64, 1, 1, 1, 0, 0, 0, 0, 99)), (7, (0, 0, 1, 2, 2, 1, 1, 1, 0, 0, 0, 1, 99)),      # (the last two: bulk updates listed in front of the next diagonal block, as in round 4)
    (79, (4, 40, 1, 24, 32, 1, 1, 1, 0, 0, 1)), (16, (4, 40, 1, 24, 32, 1, 1, 1, 0, 0, 1)), (37, (4, 8, 0, 24, 6, 0, 1, 0, 0, 0, 1)),      # the diagonal blocks in ONE resident workgroup (a fifth stream; tools/ only)
]


@pytest.mark.parametrize("nblk,params", CASES)
def test_list_order_is_a_cholesky_factorisation(nblk, params):
    ops = get_plan(nblk, params)
    T = 4
    A = spd(nblk * T, 11 + nblk)
    mc = run_list_order(ops, nblk, T, A)
    F = mc.factor()
    assert np.allclose(F @ F.T, A, rtol=1e-10, atol=1e-9)
    assert np.allclose(F, np.linalg.cholesky(A), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("nblk,params", CASES)
def test_waits_order_every_conflicting_pair(nblk, params):
    ops = get_plan(nblk, params)
    n = check_races(ops)
    if nblk > 2:
        assert n > 0
    # every wait names something that exists and comes earlier; a pipelined launch somebody waits behind has its gate
    per_stream = {}
    for o in ops:
        per_stream.setdefault(o["stream"], []).append(o["ticket"])
        assert len(o["waits"]) <= 6
        for ctr, val in o["waits"]:
            assert ctr != o["stream"]
    for s, tk in per_stream.items():
        assert tk == list(range(1, len(tk) + 1))
    waited = {(c, v) for o in ops for c, v in o["waits"] if c < N_STREAMS}
    for o in ops:
        assert o["awaited"] == (1 if (o["stream"], o["ticket"] - 1) in waited else 0)
    for c, v in waited:
        if c == 4:
            continue          # the resident workgroup publishes each block's ticket itself, when the block is done
        assert v < len(per_stream[c]), "the last operation of a stream is waited for and nothing publishes it"


@pytest.mark.parametrize("nblk,params", [(6, None), (16, None), (40, None), (79, None), (45, (8, 20, 1, 24, 32, 1, 1, 1, 30, 1)), (37, (4, 8, 0, 24, 6, 0, 1, 0, 0, 0)), (30, (2, 6, 1, 10, 4, 1, 0, 1, 99, 1)), (40, (4, 12, 1, 16, 8, 1, 1, 1, 0, 0, 1)), (40, (4, 24, 1, 8, 12, 1, 1, 1, 0, 0, 0, 1, 99))])
def test_any_order_the_waits_allow_gives_the_same_bits(nblk, params):
    ops = get_plan(nblk, params)
    T = 2 if nblk > 50 else 3
    A = spd(nblk * T, 5 + nblk)
    ref = run_list_order(ops, nblk, T, A)
    for seed in range(2 if nblk > 50 else 3):
        mc = run_random_order(ops, nblk, T, A, seed)
        assert np.array_equal(mc.L, ref.L) and np.array_equal(mc.S, ref.S)
        assert all(np.array_equal(a, b) for a, b in zip(mc.Linv, ref.Linv))


def test_the_shipping_schedule_at_cfg5_is_two_level():
    ops = get_plan(79)
    kinds = [o["kind"] for o in ops]
    assert kinds.count(SINV) >= 4 * 8 and kinds.count(PGEMM) >= 8
    bulk = [o for o in ops if o["kind"] == UPD_PIPE and o["stream"] == 2]
    assert max(o["nst"] for o in bulk) == 64                  # K = 512 per C-tile round trip while many rows remain
    assert sum(1 for o in bulk if o["nst"] == 64) >= 9
    # from the second super-step on, the panel product below the head rows rides in the previous bulk update's launch
    tails = [o for o in ops if o["fuse_with"] >= 0]
    assert len(tails) >= 8 and all(o["kind"] == PGEMM and ops[o["fuse_with"]]["nst"] == 64 for o in tails)
    assert not any(o["stream"] == 3 and o["kind"] != PUBLISH for o in ops[ops.index(tails[0]):]), "nothing is left for the fourth stream once the tails carry the products"
    # every tile is updated by every earlier panel exactly once -- except the rows below a super-step's WINDOW in its own columns, which
    # take the panels of their own super-step through the product with its inverse (PGEMM) instead
    nblk = 79
    first_of = list(range(nblk))
    for o in ops:
        if o["kind"] == SINV:
            for c in range(o["kb"], o["kb"] + o["g"]):
                first_of[c] = o["kb"]
    group_end = {p: max(c for c in range(nblk) if first_of[c] == p) + 1 for p in set(first_of)}
    window = any(o["kind"] == UPD_PIPE and o["small"] and o["nst"] == 16 for o in ops)      # (the 2 g-row window variant of the plan: not what ships)
    cnt = np.zeros((nblk, nblk), dtype=int)
    for o in ops:
        if o["kind"] == UPD_PIPE:
            for i, j, _ in o["tiles"]:
                cnt[i, j] += o["nst"] // 16
        elif o["kind"] == UPD_Q:
            for r in range(o["kb"] + 1 + o["first"], o["kb"] + 1 + o["first"] + o["m"]):
                cnt[r, o["kb"] + o["dj"]] += 1
    for i in range(nblk):
        for j in range(i + 1):
            p = first_of[j]
            # (the chain's window reaches one super-block further down: those rows take a super-step's panels one by one as well)
            in_window = i < group_end[p] + ((group_end[p] - p) if (window and group_end[p] - p > 1) else 0)
            want = j if in_window else p
            assert cnt[i, j] == want, (i, j, cnt[i, j], want)


def test_bulk_updates_behind_the_next_diagonal_block():
    """The plan option bulk_behind (measured, not shipped): a bulk update of the right-looking regime is listed behind the next diagonal
    block and waits for the chain's counter to say that block's kernel has started (the value the block publishes with its first thread)."""
    ops = get_plan(79, (4, 40, 1, 24, 32, 1, 1, 1, 0, 0, 0, 1))
    rl = [i for i, o in enumerate(ops) if o["kind"] == UPD_PIPE and o["stream"] == 2 and o["nst"] < 64]
    assert len(rl) >= 20
    for i in rl:
        prev_diag = max(j for j in range(i) if ops[j]["kind"] == DIAG)
        assert ops[prev_diag]["kb"] == ops[i]["kb"] + ops[i]["nst"] // 16, "the next diagonal block is listed in front of the bulk update"
        assert not any(ops[j]["stream"] == 0 for j in range(prev_diag + 1, i))
        assert any(c == 0 and v >= ops[prev_diag]["ticket"] - 1 for c, v in ops[i]["waits"])
