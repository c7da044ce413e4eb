"""Sharding of the image-pair grid over the GPUs of one node (one process per GPU).

The reference treats image pairs as independent units (OpenMP collapse(2) over the N x N loop,
SequentialReconstructor.cpp:202-205).  Here: every rank owns a contiguous block of images
(the ones it "detected"), one RCCL all-gather replicates the descriptor blocks over xGMI,
and the canonical pair list is dealt round-robin to the ranks.  No other exchange: match
tables stay on the rank that computed them (or are gathered by the caller; they are small).
Pure functions here; the collective itself is torch.distributed (backend "nccl" == RCCL).
"""
import numpy as np


def owned_images(n_images, world, rank):
    """[lo, hi) of the images rank `rank` holds before the all-gather (equal blocks; the bench
    rounds n_images up to a multiple of world so one all_gather_into_tensor suffices)."""
    per = (n_images + world - 1) // world
    lo = min(n_images, rank * per)
    return lo, min(n_images, lo + per)


def shard_pairs(pairs, world, rank):
    """Round-robin deal of the canonical (i<j, row-major) pair list: balanced to within one
    pair, and consecutive pairs of a rank still share their query image (L2 reuse)."""
    pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
    return np.ascontiguousarray(pairs[rank::world])


def merge_shards(shards, world):
    """Inverse of shard_pairs for per-pair result rows: shards[r] = rows of rank r."""
    n = sum(len(s) for s in shards)
    first = next(s for s in shards if len(s))
    out = np.empty((n,) + first.shape[1:], first.dtype)
    for r, s in enumerate(shards):
        out[r::world] = s
    return out
