"""GPU suite: the sharded pair grid (rcn_shard_*, csrc/shard.hip) through the C ABI at world_size 1 -- the
exact code path every rank runs at N > 1 (RCCL communicators, all-reduce of the scale statistics, in-place
all-gathers, side-stream fp32 gather) -- and the host materialisation of a device match table."""
import ctypes as C

import numpy as np
import pytest

from oracle import orc
from reconstructor_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shard(gpu_ctx):
    from reconstructor_amd import pairgrid
    sh = pairgrid.Shard(gpu_ctx, 0, 1, pairgrid.unique_id())
    yield sh
    sh.close()


def _hip():
    return C.CDLL("libamdhip64.so")


def _run(shard, ims, via_slot):
    import torch
    n, (K, D) = len(ims), ims[0].shape
    shard.ctx.check(shard.ctx.lib.rcn_desc_clear(shard.ctx.h))      # whatever earlier tests left resident (another D)
    slot = shard.reserve(n, K, D)
    block = np.ascontiguousarray(np.stack(ims), np.float32)
    if via_slot:        # a producer writing its rows straight into the landing buffer
        assert _hip().hipMemcpy(C.c_void_p(slot), C.c_void_p(block.ctypes.data), C.c_size_t(block.nbytes), 1) == 0
        shard.exchange(None)
    else:
        dev = torch.from_numpy(block).cuda()
        torch.cuda.synchronize()
        shard.exchange(dev.data_ptr())
    P = n * (n - 1) // 2
    out = torch.full((P, K), -7, dtype=torch.int32, device="cuda")
    cnt = torch.full((P,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    shard.match(0.7, out.data_ptr(), K, cnt.data_ptr())
    shard.ctx.check(shard.ctx.lib.rcn_synchronize(shard.ctx.h))
    return out, cnt


@pytest.mark.parametrize("kind,K", [("superpoint", 300), ("sift", 700), ("orb", 130)])
def test_world1_equals_oracle(shard, kind, K):
    from reconstructor_amd.matcher import all_pairs
    ims = synth.descriptor_set(kind, 6, K, n_world=2 * K, seed=41)
    exp, ec = orc.match_grid(ims, all_pairs(6), threads=4)
    for via_slot in (False, True):
        out, cnt = _run(shard, ims, via_slot)
        assert np.array_equal(out.cpu().numpy(), exp) and np.array_equal(cnt.cpu().numpy(), ec)
    info = shard.info()
    assert info["n_pairs"] == 15 and info["world"] == 1 and info["exchange_bytes_f32"] == 6 * K * ims[0].shape[1] * 4


def test_every_exchange_sees_fresh_data(shard):
    """Back-to-back exchange + match steps on DIFFERENT data without a host synchronisation in between: the
    stream/event ordering between the gathers, the conversion and the readers of the previous batch is what
    keeps step k from reading rows of step k+1 (or the reverse)."""
    import torch
    from reconstructor_amd.matcher import all_pairs
    n, K, D = 5, 512, 256
    sets = [synth.descriptor_set("superpoint", n, K, n_world=1024, seed=100 + s) for s in range(4)]
    devs = [torch.from_numpy(np.stack(s)).cuda() for s in sets]
    P = n * (n - 1) // 2
    outs = [torch.empty((P, K), dtype=torch.int32, device="cuda") for _ in sets]
    cnts = [torch.empty((P,), dtype=torch.int32, device="cuda") for _ in sets]
    shard.ctx.check(shard.ctx.lib.rcn_desc_clear(shard.ctx.h))
    shard.reserve(n, K, D)
    torch.cuda.synchronize()
    for d, o, c in zip(devs, outs, cnts):
        shard.exchange(d.data_ptr())
        shard.match(0.7, o.data_ptr(), K, c.data_ptr())
    shard.ctx.check(shard.ctx.lib.rcn_synchronize(shard.ctx.h))
    for s, o, c in zip(sets, outs, cnts):
        exp, ec = orc.match_grid(s, all_pairs(n), threads=4)
        assert np.array_equal(o.cpu().numpy(), exp) and np.array_equal(c.cpu().numpy(), ec)


def test_exchange_twice_without_a_grid_call(shard):
    """Two exchanges in a row (the second with other rows), then one grid call: the fp32 gather of the first, still on
    the side stream, must not be overtaken by the second exchange's copy into the landing buffer."""
    import torch
    from reconstructor_amd.matcher import all_pairs
    n, K, D = 6, 2048, 256
    a = synth.descriptor_set("superpoint", n, K, n_world=4096, seed=7)
    b = synth.descriptor_set("superpoint", n, K, n_world=4096, seed=8)
    da, db = (torch.from_numpy(np.stack(x)).cuda() for x in (a, b))
    P = n * (n - 1) // 2
    out = torch.empty((P, K), dtype=torch.int32, device="cuda")
    cnt = torch.empty((P,), dtype=torch.int32, device="cuda")
    shard.ctx.check(shard.ctx.lib.rcn_desc_clear(shard.ctx.h))
    shard.reserve(n, K, D)
    torch.cuda.synchronize()
    shard.exchange(da.data_ptr())
    shard.exchange(db.data_ptr())
    shard.match(0.7, out.data_ptr(), K, cnt.data_ptr())
    shard.ctx.check(shard.ctx.lib.rcn_synchronize(shard.ctx.h))
    exp, ec = orc.match_grid(b, all_pairs(n), threads=8)
    assert np.array_equal(out.cpu().numpy(), exp) and np.array_equal(cnt.cpu().numpy(), ec)


def test_compact_to_host_lists(shard):
    """rcn_match_compact_*: per pair the (query, train) lists in ascending query order, offsets from the counts."""
    import torch
    lib, ctx = shard.ctx.lib, shard.ctx
    ims = synth.descriptor_set("superpoint", 7, 900, n_world=1500, seed=77)
    out, cnt = _run(shard, ims, False)
    P, K = out.shape
    table = out.cpu().numpy()
    for pinned in (True, False):
        cap = int(cnt.sum().item())
        offs = np.zeros(P + 1, np.int64)
        total = C.c_int64(0)
        if pinned:
            hp = C.c_void_p()
            assert lib.rcn_host_alloc(C.byref(hp), 8 * max(cap, 1)) == 0
            qt = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_int32)), shape=(max(cap, 1), 2))
        else:
            qt = np.zeros((max(cap, 1), 2), np.int32)
        qt[:] = -5
        ctx.check(lib.rcn_match_compact_begin(ctx.h, out.data_ptr(), K, cnt.data_ptr(), P, offs.ctypes.data, qt.ctypes.data, cap, C.byref(total)))
        ctx.check(lib.rcn_match_compact_wait(ctx.h))
        assert total.value == cap and offs[-1] == cap and np.array_equal(np.diff(offs), cnt.cpu().numpy())
        for p in range(P):
            q = np.nonzero(table[p] >= 0)[0]
            got = qt[offs[p]:offs[p + 1]]
            assert np.array_equal(got[:, 0], q) and np.array_equal(got[:, 1], table[p][q])
        if pinned:
            # too small a buffer is an argument error that still reports the total
            rc = lib.rcn_match_compact_begin(ctx.h, out.data_ptr(), K, cnt.data_ptr(), P, offs.ctypes.data, qt.ctypes.data, cap - 1, C.byref(total))
            assert rc == -1 and total.value == cap
            lib.rcn_host_free(hp)


def test_ragged_images_through_the_slots(shard):
    """Images of different sizes (the reference's SIFT keypoint counts vary per image): every image owns a slot of
    Kmax rows; rows past its own count are zero-filled and never matched.  Both ways in: host rows per image
    (rcn_shard_put_image) and a device block with the per-image counts."""
    import torch
    from reconstructor_amd.matcher import all_pairs
    Ks = [300, 64, 0, 513, 2, 97, 1]
    ims = synth.descriptor_set("sift", len(Ks), Ks, n_world=800, seed=5)
    n, Kmax, D = len(Ks), max(Ks), 128
    exp, ec = orc.match_grid(ims, all_pairs(n), threads=4)
    P = n * (n - 1) // 2
    for via_host in (True, False):
        shard.ctx.check(shard.ctx.lib.rcn_desc_clear(shard.ctx.h))
        shard.reserve(n, Kmax, D)
        if via_host:
            for i, im in enumerate(ims):
                shard.put_image(i, im)
            shard.exchange(None, None)
        else:
            block = np.full((n, Kmax, D), 7.0, np.float32)        # garbage in the tails: the exchange zeroes them
            for i, im in enumerate(ims):
                block[i, :len(im)] = im
            dev = torch.from_numpy(block).cuda()
            torch.cuda.synchronize()
            shard.exchange(dev.data_ptr(), Ks)
        out = torch.full((P, Kmax), -7, dtype=torch.int32, device="cuda")
        cnt = torch.full((P,), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        shard.match(0.7, out.data_ptr(), Kmax, cnt.data_ptr())
        shard.ctx.check(shard.ctx.lib.rcn_synchronize(shard.ctx.h))
        assert np.array_equal(out.cpu().numpy(), exp) and np.array_equal(cnt.cpu().numpy(), ec)
    # the ctx-owned tables + host lists
    shard.match(0.7)
    offs, qt = shard.lists()
    assert np.array_equal(np.diff(offs), ec)
    for p in range(P):
        q = np.nonzero(exp[p] >= 0)[0]
        assert np.array_equal(qt[offs[p]:offs[p + 1], 0], q) and np.array_equal(qt[offs[p]:offs[p + 1], 1], exp[p][q])


def test_local_failure_is_voted_and_every_entry_point_fails(shard):
    """Fault injection at world size 1 (every rank runs this code at N > 1): a failure only this rank saw -- a refused
    put_image, or one the host driver reports through rcn_shard_fail -- is remembered, travels in the status vote at
    the head of the next exchange, and that exchange fails BEFORE the statistics / payload collectives (where a peer
    would otherwise wait for ever); match and lists refuse to run on the abandoned exchange; the next exchange is clean."""
    import torch
    from reconstructor_amd import _lib
    from reconstructor_amd.matcher import all_pairs
    ims = synth.descriptor_set("sift", 4, 200, n_world=500, seed=77)
    exp, ec = orc.match_grid(ims, all_pairs(4), threads=4)
    out, cnt = _run(shard, ims, False)
    assert np.array_equal(out.cpu().numpy(), exp)
    dev = torch.from_numpy(np.ascontiguousarray(np.stack(ims))).cuda()
    torch.cuda.synchronize()
    shard.reserve(4, 200, 128)
    with pytest.raises(_lib.RcnError) as e:
        shard.put_image(0, np.zeros((201, 128), np.float32))              # does not fit the slot: local failure
    assert e.value.code == -1
    with pytest.raises(_lib.RcnError) as e:
        shard.exchange(dev.data_ptr())                                    # valid arguments, but the vote carries the failure
    assert e.value.code == -1 and "put_image" in str(e.value)
    with pytest.raises(_lib.RcnError) as e:
        shard.match(0.7)
    assert "no successful rcn_shard_exchange" in str(e.value)
    with pytest.raises(_lib.RcnError):
        shard.lists()
    out, cnt = _run(shard, ims, False)                                     # the vote cleared the status
    assert np.array_equal(out.cpu().numpy(), exp) and np.array_equal(cnt.cpu().numpy(), ec)
    shard.fail(-2)                                                         # a failure of the host driver's own
    with pytest.raises(_lib.RcnError) as e:
        shard.exchange(dev.data_ptr())
    assert e.value.code == -2
    with pytest.raises(_lib.RcnError):
        shard.match(0.7)
    out, cnt = _run(shard, ims, True)
    assert np.array_equal(out.cpu().numpy(), exp) and np.array_equal(cnt.cpu().numpy(), ec)


def test_phase_times_and_communicator_size(shard):
    ims = synth.descriptor_set("superpoint", 5, 256, n_world=600, seed=8)
    shard.profile(True)
    for _ in range(3):
        _run(shard, ims, False)
    t = shard.times()
    assert t["exchanges"] == 3 and t["matches"] == 3
    assert t["exchange_ms"] > 0 and t["match_ms"] > 0 and t["f32_gather_ms"] >= 0
    assert shard.times()["exchanges"] == 0                                 # read clears
    shard.profile(False)
    assert shard.info()["comm_ranks"] == 1


def test_device_side_scale_equals_the_host_side_scale(shard, gpu_ctx):
    """The exchange fixes the fp16 scale on the device (k_fix_scale behind the all-reduce, no host read); the plain
    upload path fixes it on the host.  Same constants: the packed candidate tables of the coarse pass -- which depend on
    scale and bias bit for bit -- lead to the same rows taking the same route, and the results are equal."""
    from reconstructor_amd.matcher import HipL2Matcher, all_pairs
    ims = synth.descriptor_set("sift", 5, 300, n_world=700, seed=21)
    ims[2] = ims[2] * np.float32(64.0)                                     # a scale that is not the default one
    out, cnt = _run(shard, ims, False)
    st_dev = HipL2Matcher(ctx=gpu_ctx).stats()
    m = HipL2Matcher(ctx=gpu_ctx)
    m.clear()
    for i, im in enumerate(ims):
        m.upload(i, im)
    tab, c2 = m.match_grid(all_pairs(5), 300)
    st_host = m.stats()
    m.clear()
    assert np.array_equal(out.cpu().numpy(), tab) and np.array_equal(cnt.cpu().numpy(), c2)
    for k in ("rows_reranked", "rows_exact_fallback", "err_bound_d2", "rows_total"):
        assert st_dev[k] == st_host[k], k


def test_gather_lists_at_world_one_equals_the_lists(shard):
    """rcn_shard_gather_lists is collective; at world size 1 the root receives nothing over RCCL, but every other step --
    totals + per-pair counts all-gathered, the bounded waits, the votes, the canonical interleave on the device, the one
    copy to the host -- is the code every rank runs at N > 1.  Own tables (ragged, some pairs empty) and caller's tables."""
    import torch
    from reconstructor_amd.matcher import all_pairs
    Ks = [300, 64, 0, 513, 2, 97]
    ims = synth.descriptor_set("sift", len(Ks), Ks, n_world=800, seed=15)
    n, Kmax, D = len(Ks), max(Ks), 128
    exp, ec = orc.match_grid(ims, all_pairs(n), threads=4)
    shard.ctx.check(shard.ctx.lib.rcn_desc_clear(shard.ctx.h))
    shard.reserve(n, Kmax, D)
    for i, im in enumerate(ims):
        shard.put_image(i, im)
    shard.exchange(None, None)
    shard.match(0.7)
    offs, qt = shard.lists()
    goffs, gqt = shard.gather_lists(0)
    assert np.array_equal(goffs, offs) and np.array_equal(gqt, qt) and np.array_equal(np.diff(goffs), ec)
    # a root buffer that is too small: an error with the total, and the shard stays usable
    total = C.c_int64(0)
    small = np.zeros((max(1, len(qt) - 1), 2), np.int32)
    rc = shard.ctx.lib.rcn_shard_gather_lists(shard.h, 0, None, 0, None, goffs.ctypes.data, small.ctypes.data, len(qt) - 1, C.byref(total))
    assert rc == -1 and total.value == len(qt)
    # the caller's own tables
    P = n * (n - 1) // 2
    out = torch.full((P, Kmax + 3), -7, dtype=torch.int32, device="cuda")
    cnt = torch.full((P,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    shard.match(0.7, out.data_ptr(), Kmax + 3, cnt.data_ptr())
    goffs2, gqt2 = shard.gather_lists(0, out.data_ptr(), Kmax + 3, cnt.data_ptr())
    assert np.array_equal(goffs2, offs) and np.array_equal(gqt2, qt)
    shard.set_timeout(600.0)


_FAULT = r"""
import sys, json, ctypes as C
import numpy as np
sys.path.insert(0, %r)
import torch
from reconstructor_amd import _lib, pairgrid, synth
from reconstructor_amd.matcher import all_pairs
L = _lib.load()
assert b"DIAGNOSTIC" in L.rcn_version()
L.rcn_diag_shard_fault.argtypes = [C.c_void_p, C.c_int]
ctx = _lib.Context(0)
sh = pairgrid.Shard(ctx, 0, 1, pairgrid.unique_id())
ims = synth.descriptor_set("sift", 4, 200, n_world=500, seed=77)
dev = torch.from_numpy(np.ascontiguousarray(np.stack(ims))).cuda()
torch.cuda.synchronize()
res = {}
def call(name, f):
    try:
        f(); res[name] = 0
    except _lib.RcnError as e:
        res[name] = e.code; res[name + "_text"] = str(e)
sh.reserve(4, 200, 128)
call("clean0", lambda: sh.exchange(dev.data_ptr()))
kind = int(sys.argv[1])
assert L.rcn_diag_shard_fault(sh.h, kind) == 0
call("faulty", lambda: sh.exchange(dev.data_ptr()))
call("match", lambda: sh.match(0.7))
call("lists", lambda: sh.lists())
call("gather", lambda: sh.gather_lists(0))
call("next", lambda: sh.exchange(dev.data_ptr()))          # kind 1: the vote carries last call's error; kind 2: the shard is dead
call("after", lambda: sh.exchange(dev.data_ptr()))
call("match2", lambda: sh.match(0.7))
call("reserve", lambda: sh.reserve(4, 200, 128))
sh.close()
print(json.dumps(res))
"""


@pytest.mark.parametrize("kind", [1, 2])
def test_failure_behind_the_vote_reaches_every_entry_point(kind):
    """an error AFTER the status vote must neither strand the peers inside a collective nor go unreported.
    The diagnostic build injects one (rcn_diag_shard_fault) in a child process.  kind 1, a HIP error behind the vote: the
    exchange still enters every remaining collective, returns the error, match / lists / gather refuse to run on it, the NEXT
    exchange's vote reports it once more (that is how the peers hear of it) and the one after is clean.  kind 2, an RCCL call
    that refuses to queue: the communicators are aborted and every later entry point returns RCN_ERR_COMM at once."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "tools", "librcn_diag.so")
    assert os.path.exists(diag), "tools/librcn_diag.so missing: run __graft_entry__.build()"
    r = subprocess.run([sys.executable, "-c", _FAULT % root, str(kind)], capture_output=True, text=True, timeout=300, env=dict(os.environ, RCN_LIB=diag))
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["clean0"] == 0
    if kind == 1:
        assert d["faulty"] == -2 and "behind the vote" in d["faulty_text"]                 # RCN_ERR_HIP, reported at once
        assert d["match"] != 0 and d["lists"] != 0 and d["gather"] != 0                      # nothing runs on the abandoned exchange
        assert d["next"] == -2                                                              # the vote of the next exchange carries it
        assert d["after"] == 0 and d["match2"] == 0 and d["reserve"] == 0                   # and then the shard is clean again
    else:
        assert d["faulty"] == -7 and "aborted" in d["faulty_text"]                          # RCN_ERR_COMM
        for k in ("match", "lists", "gather", "next", "after", "match2", "reserve"):
            assert d[k] == -7, (k, d)


# ---------------------------------------------------------------------------------------------------------------------
# Two ranks, two GPUs.  The driver's GPU box has ONE device, so these are skipped there; on a node with two they
# run each rank in a child process of its own (never re-exec'd; a rank that fails exits non-zero).
_TWO_RANKS = r"""
import sys, os, json, time
import numpy as np
sys.path.insert(0, %r)
rank, world, mode, uid_hex = int(sys.argv[1]), 2, sys.argv[2], sys.argv[3]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
from reconstructor_amd import _lib, pairgrid, synth
from reconstructor_amd.matcher import all_pairs
ctx = _lib.Context(rank)
sh = pairgrid.Shard(ctx, rank, world, bytes.fromhex(uid_hex))
n, K, D = 6, 256, 256
ims = synth.descriptor_set("superpoint", n, K, n_world=600, seed=11)
lo, hi = pairgrid.owned_images(n, world, rank)
sh.set_timeout(5.0)
sh.reserve(n, K, D)
for i in range(lo, hi):
    sh.put_image(i, ims[i])
out = {"rank": rank}
if mode == "dead_peer" and rank == 1:
    time.sleep(20)                       # never enters the collective: the peer must give up on its own
    print(json.dumps(out)); sys.exit(0)
t0 = time.time()
try:
    sh.exchange()
    sh.match(0.7)
    offs, qt = sh.gather_lists(root=0)
    if rank == 0:
        from oracle import orc
        exp, ec = orc.match_grid(ims, all_pairs(n), threads=2)
        ok = np.array_equal(np.diff(offs), ec) and all(exp[p][q] == t for p in range(len(ec)) for q, t in qt[offs[p]:offs[p + 1]])
        out["gather_equals_oracle"] = bool(ok)
    out["ok"] = True
except _lib.RcnError as e:
    out.update(ok=False, code=e.code, seconds=time.time() - t0)
print(json.dumps(out))
"""


def _two_rank_run(mode):
    import json
    import os
    import subprocess
    import sys
    from reconstructor_amd import _lib, pairgrid
    if _lib.load().rcn_device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    uid = pairgrid.unique_id().hex()
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_RANKS % root, str(r), mode, uid], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    return outs


def test_two_ranks_gather_lists_equals_the_oracle():
    """world = 2: the device path of rcn_shard_gather_lists (ncclSend / ncclRecv, interleave on the root) against the oracle's grid."""
    outs = _two_rank_run("gather")
    assert all(o["ok"] for o in outs) and outs[0]["gather_equals_oracle"]


def test_a_peer_that_never_joins_costs_a_timeout_not_a_hang():
    """One rank never enters the exchange: the other gives up after rcn_shard_set_timeout seconds with RCN_ERR_COMM (the read-backs in
    front of the bounded waits land in pinned memory, so the host reaches its polling loop)."""
    outs = _two_rank_run("dead_peer")
    r0 = [o for o in outs if o["rank"] == 0][0]
    assert r0["ok"] is False and r0["code"] == -7 and 4.0 <= r0["seconds"] <= 30.0, r0
