"""Host-side mirror (Python) of the sharded pair grid: include/rcn.h `rcn_shard_*`.

The reference treats image pairs as independent units (OpenMP collapse(2) over the N x N loop,
SequentialReconstructor.cpp:202-205).  The partition, the RCCL exchange and the per-rank grid call
all live in librcn.so (csrc/shard.hip); this module only marshals.  One `Shard` per GPU / process:
the 128-byte rendezvous id is drawn by one rank (`unique_id`) and handed to the others by whatever
the host has (bench.py: a torch.distributed broadcast over its control-plane process group).
"""
import ctypes as C

import numpy as np

from . import _lib


def owned_images(n_images, world, rank):
    """[lo, hi) of the image ids rank `rank` holds before the exchange (rcn_shard_owned_images)."""
    lo, cnt = C.c_int32(), C.c_int32()
    rc = _lib.load().rcn_shard_owned_images(n_images, world, rank, C.byref(lo), C.byref(cnt))
    if rc:
        raise _lib.RcnError(rc, "rcn_shard_owned_images")
    return lo.value, lo.value + cnt.value


def shard_pairs(n_images, world, rank):
    """This rank's share of the canonical i < j list (rcn_shard_pairs): pair number p -> rank p % world."""
    L = _lib.load()
    n = L.rcn_shard_pair_count(n_images, world, rank)
    if n < 0:
        raise _lib.RcnError(int(n), "rcn_shard_pair_count")
    out = np.zeros((int(n), 2), np.int32)
    rc = L.rcn_shard_pairs(n_images, world, rank, out.ctypes.data if n else None)
    if rc:
        raise _lib.RcnError(rc, "rcn_shard_pairs")
    return out


def merge_shards(shards, world):
    """Inverse of the round-robin deal for per-pair result rows: shards[r] = rows of rank r."""
    n = sum(len(s) for s in shards)
    first = next(s for s in shards if len(s))
    out = np.empty((n,) + first.shape[1:], first.dtype)
    for r, s in enumerate(shards):
        out[r::world] = s
    return out


def merge_lists(n_images, world, per_rank):
    """rcn_shard_merge_lists: per_rank[r] = (offsets, qt) as Shard.lists() returns them on rank r -> (offsets[P+1], qt[total, 2]) of
    the whole canonical grid (pair number p of the row-major i < j list came from rank p % world).  Pure host code."""
    L = _lib.load()
    P = n_images * (n_images - 1) // 2
    offs = [np.ascontiguousarray(o, np.int64) for o, _ in per_rank]
    qts = [np.ascontiguousarray(q, np.int32).reshape(-1, 2) for _, q in per_rank]
    op = (C.c_void_p * world)(*[o.ctypes.data for o in offs])
    qp = (C.c_void_p * world)(*[q.ctypes.data if len(q) else None for q in qts])
    out_off = np.zeros(P + 1, np.int64)
    total = C.c_int64(0)
    cap = int(sum(int(o[-1]) for o in offs))
    out_qt = np.zeros((max(cap, 1), 2), np.int32)
    rc = L.rcn_shard_merge_lists(n_images, world, op, qp, out_off.ctypes.data, out_qt.ctypes.data, cap, C.byref(total))
    if rc:
        raise _lib.RcnError(rc, "rcn_shard_merge_lists")
    return out_off, out_qt[:total.value]


def unique_id():
    """ncclGetUniqueId as 128 bytes (one rank draws it, every rank passes it to Shard)."""
    buf = (C.c_uint8 * _lib.SHARD_ID_BYTES)()
    rc = _lib.load().rcn_shard_unique_id(buf)
    if rc:
        raise _lib.RcnError(rc, "rcn_shard_unique_id")
    return bytes(buf)


class Shard:
    """One rank of the sharded grid (rcn_shard): a ctx plus its RCCL communicators."""

    def __init__(self, ctx, rank, world, uid):
        self.ctx, self.rank, self.world = ctx, rank, world
        h = C.c_void_p()
        buf = (C.c_uint8 * _lib.SHARD_ID_BYTES).from_buffer_copy(uid)
        ctx.check(ctx.lib.rcn_shard_create(ctx.h, rank, world, buf, C.byref(h)))
        self.h = h

    def reserve(self, n_images, K, D):
        """Returns the device address of this rank's [count][K][D] fp32 block of the landing buffer."""
        slot = C.c_void_p()
        self.ctx.check(self.ctx.lib.rcn_shard_reserve(self.h, n_images, K, D, C.byref(slot)))
        return slot.value

    def put_image(self, img_id, desc):
        """Host rows of one owned image into its slot (ragged K allowed: K_img <= reserved K)."""
        desc = np.ascontiguousarray(desc, np.float32)
        self.ctx.check(self.ctx.lib.rcn_shard_put_image(self.h, int(img_id), desc.ctypes.data if len(desc) else None, len(desc)))

    def exchange(self, local_dev_ptr=None, local_K=None):
        k = None if local_K is None else np.ascontiguousarray(local_K, np.int32)
        self.ctx.check(self.ctx.lib.rcn_shard_exchange(self.h, C.c_void_p(local_dev_ptr) if local_dev_ptr else None,
                                                       k.ctypes.data if k is not None else None))

    def match(self, ratio, out_dev_ptr=None, out_stride=0, counts_dev_ptr=None):
        """Both pointers None: the tables stay in the ctx's own buffers (then `lists()`)."""
        self.ctx.check(self.ctx.lib.rcn_shard_match(self.h, float(ratio), C.c_void_p(out_dev_ptr) if out_dev_ptr else None, out_stride,
                                                    C.c_void_p(counts_dev_ptr) if counts_dev_ptr else None))

    def lists(self):
        """(offsets[P+1] int64, qt[total, 2] int32) of the last match() into the ctx's own tables."""
        P = self.info()["n_pairs"]
        offs = np.zeros(P + 1, np.int64)
        total = C.c_int64(0)
        rc = self.ctx.lib.rcn_shard_lists(self.h, offs.ctypes.data, None, 0, C.byref(total))
        if rc != 0 and not (rc == -1 and total.value > 0):      # -1 with a total: the sizing call's "capacity too small"
            self.ctx.check(rc)
        qt = np.zeros((max(1, total.value), 2), np.int32)
        if total.value:
            self.ctx.check(self.ctx.lib.rcn_shard_lists(self.h, offs.ctypes.data, qt.ctypes.data, total.value, C.byref(total)))
        return offs, qt[:total.value]

    def gather_lists(self, root=0, table_dev_ptr=None, stride=0, counts_dev_ptr=None, capacity=None):
        """Collective (rcn_shard_gather_lists): the lists of every rank in canonical pair order on `root`, moved device to device
        over RCCL.  Returns (offsets, qt) on the root and (None, None) elsewhere.  capacity: entries the root's buffer holds
        (None: sized by a first call that reports the total -- every rank must then pass None as well)."""
        P = self.info()["n_images"] * (self.info()["n_images"] - 1) // 2
        is_root = self.rank == root
        t = C.c_void_p(table_dev_ptr) if table_dev_ptr else None
        c = C.c_void_p(counts_dev_ptr) if counts_dev_ptr else None
        offs = np.zeros(P + 1, np.int64) if is_root else None
        total = C.c_int64(0)
        if capacity is None:
            rc = self.ctx.lib.rcn_shard_gather_lists(self.h, root, t, stride, c, offs.ctypes.data if is_root else None, None, 0, C.byref(total))
            if rc == 0:                # nothing matched anywhere
                return (offs, np.zeros((0, 2), np.int32)) if is_root else (None, None)
            if not (rc == -1 and (not is_root or total.value > 0)):
                self.ctx.check(rc)
            capacity = total.value
        qt = np.zeros((max(1, capacity), 2), np.int32) if is_root else None
        self.ctx.check(self.ctx.lib.rcn_shard_gather_lists(self.h, root, t, stride, c, offs.ctypes.data if is_root else None,
                                                           qt.ctypes.data if is_root else None, capacity if is_root else 0, C.byref(total)))
        return (offs, qt[:total.value]) if is_root else (None, None)

    def set_timeout(self, seconds):
        self.ctx.check(self.ctx.lib.rcn_shard_set_timeout(self.h, float(seconds)))

    def info(self):
        s = _lib.ShardStats()
        self.ctx.check(self.ctx.lib.rcn_shard_info(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def fail(self, code=-1):
        """Report a failure of this rank's own host-side step: the next exchange() then fails on every rank together."""
        self.ctx.check(self.ctx.lib.rcn_shard_fail(self.h, int(code)))

    def profile(self, enable=True):
        self.ctx.check(self.ctx.lib.rcn_shard_profile(self.h, 1 if enable else 0))

    def times(self):
        """Phase times (HIP events) summed since the last call: exchange / fp32 gather / match, in ms."""
        t = _lib.ShardTimes()
        self.ctx.check(self.ctx.lib.rcn_shard_profile_read(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.rcn_shard_destroy(self.h)
            self.h = None
