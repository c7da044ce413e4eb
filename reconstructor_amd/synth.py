"""Seeded synthetic inputs of the shapes BASELINE.json names (no dataset, no network).

Descriptors follow the output contract of the reference's feature stage:
  * "superpoint": D=256 fp32, L2-normalised rows (FeatureSuperPoint.cpp:183-211)
  * "sift":       D=128 non-negative integer-valued fp32 (FeatureDetector.cpp:19-24, CV_32F)
  * "orb":        D=32 integer-valued 0..255 fp32 (ORB bytes converted to float, same file)
Images share a pool of "world" descriptors so that a useful fraction of queries has a true
counterpart in the other image and survives the 0.7 ratio test.
Counter-based generator (Philox) keyed by (seed, image id): any rank can produce any image.
"""
import numpy as np


def _rng(seed, stream):
    return np.random.Generator(np.random.Philox(key=[int(seed), int(stream)]))


def world_pool(kind, n_world, seed=1234):
    r = _rng(seed, 0xFFFF_FFFF)
    if kind == "superpoint":
        w = r.standard_normal((n_world, 256), dtype=np.float32)
        w /= np.linalg.norm(w, axis=1, keepdims=True).astype(np.float32)
        return w
    if kind == "sift":
        return np.floor(r.gamma(0.6, 40.0, (n_world, 128)).clip(0, 255)).astype(np.float32)
    if kind == "orb":
        return r.integers(0, 256, (n_world, 32)).astype(np.float32)
    raise ValueError(kind)


def image_descriptors(kind, img_id, K, pool, seed=1234, sigma=None, return_pick=False):
    """K descriptors of image `img_id`: K distinct pool rows + noise, in the kind's format."""
    r = _rng(seed, img_id)
    n_world = pool.shape[0]
    if K <= n_world:
        pick = r.permutation(n_world)[:K]
    else:
        pick = r.integers(0, n_world, K)
    d = pool[pick].copy()
    if kind == "superpoint":
        s = 0.04 if sigma is None else sigma
        d += (s * r.standard_normal(d.shape, dtype=np.float32)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        d = np.ascontiguousarray(d, np.float32)
    else:
        s = 6.0 if sigma is None else sigma
        d = np.ascontiguousarray(np.rint(d + s * r.standard_normal(d.shape)).clip(0, 255), np.float32)
    return (d, pick) if return_pick else d


def scene_set(kind, n_images, K, n_world=None, seed=1234, noise_px=0.5, outlier_frac=0.05):
    """A descriptor set WITH geometry, the input of the whole pair loop (match + epipolar filter,
    SequentialReconstructor.cpp:199-279): every world descriptor belongs to a 3-d point, image i is a pinhole view of
    the scene (cameras on a ring looking at the origin, as synth_ba.make_scene) and a keypoint's integer pixel
    coordinates (Feature<int>::featCoord) are the projection of its point plus noise, truncated; a share of keypoints
    gets coordinates unrelated to its point (what the filter is there to reject).
    Returns (descriptors, coords, picks): per image (K_i, D) fp32, (K_i, 2) int32, (K_i,) world-point ids."""
    from . import synth_ba
    ks = K if hasattr(K, "__len__") else [K] * n_images
    if n_world is None:
        n_world = 4 * max(ks)
    pool = world_pool(kind, n_world, seed)
    sc = synth_ba.make_scene(n_images, n_world, obs_per_point=2, seed=seed, width=1024, height=768)
    ims, coords, picks = [], [], []
    for i in range(n_images):
        d, pick = image_descriptors(kind, i, ks[i], pool, seed, return_pick=True)
        r = _rng(seed, 0x4000_0000 + i)
        uv, _ = synth_ba.project(sc["poses_gt"][i][None, :], sc["intr_gt"][i][None, :], sc["points_gt"][pick])
        uv = uv + noise_px * r.standard_normal(uv.shape)
        bad = r.random(len(pick)) < outlier_frac
        uv[bad] = np.c_[r.uniform(0, 1024, bad.sum()), r.uniform(0, 768, bad.sum())]
        ims.append(d)
        coords.append(np.ascontiguousarray(np.trunc(uv).astype(np.int32)))
        picks.append(pick)
    return ims, coords, picks


def descriptor_set(kind, n_images, K, n_world=None, seed=1234, first_image=0, count=None):
    """List of per-image (K, D) fp32 arrays for images [first_image, first_image+count)."""
    if n_world is None:
        n_world = 4 * K
    pool = world_pool(kind, n_world, seed)
    count = n_images - first_image if count is None else count
    ks = K if hasattr(K, "__len__") else [K] * n_images
    return [image_descriptors(kind, i, ks[i], pool, seed) for i in range(first_image, first_image + count)]
