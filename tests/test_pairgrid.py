"""CPU suite: sharding of the pair grid -- the partition functions of the C ABI (rcn_shard_owned_images /
rcn_shard_pairs: pure host code, no GPU) and a world_size-2 gloo run that deals the grid with them."""
import os
import subprocess
import sys

import numpy as np

from reconstructor_amd import pairgrid
from reconstructor_amd.matcher import all_pairs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_partition_the_grid():
    for n in (2, 5, 100, 141, 1000):
        pairs = all_pairs(n)
        assert len(pairs) == n * (n - 1) // 2 and (pairs[:, 0] < pairs[:, 1]).all()
        for world in (1, 2, 4, 8):
            shards = [pairgrid.shard_pairs(n, world, r) for r in range(world)]
            assert all(np.array_equal(shards[r], pairs[r::world]) for r in range(world))      # pair number p -> rank p % world
            sizes = [len(s) for s in shards]
            assert max(sizes) - min(sizes) <= 1
            merged = pairgrid.merge_shards(shards, world)
            assert np.array_equal(merged, pairs)
            owned = [pairgrid.owned_images(n, world, r) for r in range(world)]
            assert owned[0][0] == 0 and owned[-1][1] == n
            assert all(owned[i][1] == owned[i + 1][0] for i in range(world - 1))


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from reconstructor_amd import pairgrid, synth
from reconstructor_amd.matcher import all_pairs
from oracle import orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n, K, D = 6, 48, 32
lo, hi = pairgrid.owned_images(n, world, rank)
pool = synth.world_pool("orb", 96, seed=7)
local = np.stack([synth.image_descriptors("orb", i, K, pool, seed=7) for i in range(lo, hi)])
gathered = torch.empty((n, K, D), dtype=torch.float32)
dist.all_gather_into_tensor(gathered.view(-1), torch.from_numpy(local).view(-1))      # the one collective
ims = [gathered[i].numpy() for i in range(n)]
mine = pairgrid.shard_pairs(n, world, rank)            # the C partition (rcn_shard_pairs)
out, counts = orc.match_grid(ims, mine, threads=1)          # stand-in for the per-rank GPU grid call
rows = [None] * world
dist.all_gather_object(rows, out)
def lists_of(table):                 # what rcn_shard_lists hands back for a table: offsets + (query, train) in ascending query order
    offs, qt = [0], []
    for row in table:
        q = np.nonzero(row >= 0)[0]
        qt.append(np.stack([q, row[q]], 1))
        offs.append(offs[-1] + len(q))
    return np.array(offs, np.int64), (np.concatenate(qt) if qt else np.zeros((0, 2))).astype(np.int32)
per_rank = [None] * world
dist.all_gather_object(per_rank, lists_of(out))      # the control plane carries the per-rank lists here; on GPUs rcn_shard_gather_lists does (RCCL)
if rank == 0:
    merged = pairgrid.merge_shards(rows, world)
    full = [synth.image_descriptors("orb", i, K, pool, seed=7) for i in range(n)]
    exp, _ = orc.match_grid(full, all_pairs(n), threads=1)
    assert np.array_equal(merged, exp)
    # the host half of the gather (rcn_shard_merge_lists): the single featureMatches map in canonical pair order
    goff, gqt = pairgrid.merge_lists(n, world, per_rank)
    eoff, eqt = lists_of(exp)
    assert np.array_equal(goff, eoff) and np.array_equal(gqt, eqt)
    print("OK", merged.shape, int((merged >= 0).sum()), len(gqt))
dist.destroy_process_group()
'''


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


def test_merge_lists_is_the_inverse_of_the_deal():
    """rcn_shard_merge_lists on random per-rank lists, worlds 1 .. 8, against a direct construction."""
    rng = np.random.default_rng(5)
    for n in (2, 3, 7, 20):
        P = n * (n - 1) // 2
        cnt = rng.integers(0, 5, P)
        cnt[rng.random(P) < 0.3] = 0
        full_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        full_qt = rng.integers(0, 1000, (int(full_off[-1]), 2)).astype(np.int32)
        for world in (1, 2, 3, 8):
            per = []
            for r in range(world):
                idx = np.arange(r, P, world)
                off = np.concatenate([[0], np.cumsum(cnt[idx])]).astype(np.int64)
                qt = np.concatenate([full_qt[full_off[p]:full_off[p + 1]] for p in idx]) if len(idx) else np.zeros((0, 2), np.int32)
                per.append((off, qt.astype(np.int32).reshape(-1, 2)))
            off, qt = pairgrid.merge_lists(n, world, per)
            assert np.array_equal(off, full_off) and np.array_equal(qt, full_qt)
