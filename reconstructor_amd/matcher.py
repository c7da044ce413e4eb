"""Host-side mirror (Python) of the reference's matcher interface over the C ABI.

Mirrors, name for name:
  FeatureMatcher::matchFeatures(features1, features2, matches, imgShape1, imgShape2)
      FeatureMatcher.h:18-22 -- per-pair plugin call; `matches` is the query->train map
  SequentialReconstructor::matchFeatures (pair loop only)   SequentialReconstructor.cpp:199-279
The C++ form of the same adapter is reconstructor_amd/host/HipFeatureMatcher.h.
All arithmetic happens in librcn.so on the GPU; nothing here computes a distance.
"""
import ctypes as C

import numpy as np

from . import _lib

RATIO_THRESH = np.float32(0.7)  # FeatureMatcher.h:45


class FeatureMatcher:
    """Abstract base, as FeatureMatcher.h:11-27 (the featNormalization flag is ignored there too)."""

    def __init__(self, feat_normalization=False):
        pass

    def match_features(self, features1, features2, matches, img_shape1=None, img_shape2=None):
        raise NotImplementedError


def _as_desc(features):
    """Dense row-major K x D fp32 (featDescToCV, FeatureMatcher.cpp:11-25).  Accepts an array
    or a sequence of objects with .featDesc.desc / .desc (the reference's Feature layout)."""
    if isinstance(features, np.ndarray):
        a = features
    elif len(features) == 0:
        return np.zeros((0, 0), np.float32)
    elif hasattr(features[0], "featDesc"):
        a = np.array([f.featDesc.desc for f in features], np.float32)
    elif hasattr(features[0], "desc"):
        a = np.array([f.desc for f in features], np.float32)
    else:
        a = np.asarray(features, np.float32)
    return np.ascontiguousarray(a, np.float32)


class HipL2Matcher(FeatureMatcher):
    """Exact brute-force L2 2-NN + ratio + uniqueness on MI355X: what FlannMatcher asks OpenCV
    for (FeatureMatcher.cpp:32-65), computed exactly."""

    def __init__(self, ctx=None, device=0, ratio=RATIO_THRESH):
        super().__init__()
        self.ctx = ctx if ctx is not None else _lib.Context(device)
        self.ratio = float(np.float32(ratio))
        self._D = 0     # descriptor length of the resident images (0 = nothing uploaded yet)

    # -- per-pair plugin boundary -------------------------------------------------------
    def match_features(self, features1, features2, matches, img_shape1=None, img_shape2=None):
        out = self.match_pair(_as_desc(features1), _as_desc(features2))
        for q in np.nonzero(out >= 0)[0]:
            matches[int(q)] = int(out[q])
        return matches

    def match_pair(self, q, t):
        """Dense result: out[i] = train row matched to query row i, or -1."""
        q = _as_desc(q)
        t = _as_desc(t)
        K1, K2 = q.shape[0], t.shape[0]
        if K1 == 0:
            return np.zeros(0, np.int32)
        D = q.shape[1]
        if K2 and t.shape[1] != D:
            raise ValueError("descriptor lengths differ")
        out = np.full(K1, -1, np.int32)
        cnt = C.c_int32(0)
        self.ctx.check(self.ctx.lib.rcn_match_pair(
            self.ctx.h, q.ctypes.data, K1, t.ctypes.data if K2 else None, K2, D, self.ratio,
            out.ctypes.data, C.byref(cnt)))
        return out

    # -- resident images + pair grid ----------------------------------------------------
    def upload(self, img_id, desc):
        desc = _as_desc(desc)
        K = desc.shape[0]
        D = desc.shape[1] if desc.ndim == 2 and desc.shape[1] else self._D
        if D <= 0:
            raise ValueError("the first upload must have a descriptor length (an empty set before any other image has no D)")
        self._D = D
        self.ctx.check(self.ctx.lib.rcn_desc_upload(self.ctx.h, int(img_id),
                                                    desc.ctypes.data if K else None, K, D))

    def upload_device(self, img_id, dev_ptr, K, D):
        self._D = D
        self.ctx.check(self.ctx.lib.rcn_desc_upload_device(self.ctx.h, int(img_id),
                                                           C.c_void_p(dev_ptr), K, D))

    def upload_batch_device(self, first_id, n_images, dev_ptr, K, D):
        """[n][K][D] fp32 in HBM, borrowed (zero copy); one stats + one conversion launch."""
        self._D = D
        self.ctx.check(self.ctx.lib.rcn_desc_upload_batch_device(
            self.ctx.h, int(first_id), int(n_images), C.c_void_p(dev_ptr), K, D))

    def upload_batch(self, first_id, images):
        """Ragged host images (a list of K_i x D arrays) as ids first_id ..: one call, one synchronisation
        (rcn_desc_upload_batch) instead of one synchronous upload per image."""
        arrs = [_as_desc(im) for im in images]
        D = next((a.shape[1] for a in arrs if a.ndim == 2 and a.shape[0] and a.shape[1]), self._D)
        if D <= 0:
            raise ValueError("no image of the batch has a descriptor length")
        if any(a.shape[0] and a.shape[1] != D for a in arrs):
            raise ValueError("descriptor lengths differ")
        self._D = D
        n = len(arrs)
        rows = (C.c_void_p * max(n, 1))(*[a.ctypes.data if a.shape[0] else None for a in arrs])
        Ks = np.array([a.shape[0] for a in arrs], np.int32)
        self.ctx.check(self.ctx.lib.rcn_desc_upload_batch(self.ctx.h, int(first_id), n, rows, Ks.ctypes.data, D))

    def remove(self, img_id):
        self.ctx.check(self.ctx.lib.rcn_desc_remove(self.ctx.h, int(img_id)))

    def clear(self):
        self.ctx.check(self.ctx.lib.rcn_desc_clear(self.ctx.h))
        self.ctx.check(self.ctx.lib.rcn_coords_clear(self.ctx.h))

    def match_grid(self, pairs, out_stride):
        """pairs: (P,2) (query image id, train image id).  Returns (out[P,out_stride], counts[P])."""
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        P = pairs.shape[0]
        out = np.full((P, max(1, out_stride)), -1, np.int32)
        counts = np.zeros(max(P, 1), np.int32)
        self.ctx.check(self.ctx.lib.rcn_match_grid(self.ctx.h, pairs.ctypes.data, P, self.ratio,
                                                   out.ctypes.data, out.shape[1], counts.ctypes.data))
        return out, counts[:P]

    def match_all_pairs(self, n_images, out_stride):
        """The reference's canonical grid (every i < j over the resident images, ascending ids): pairs = NULL."""
        P = n_images * (n_images - 1) // 2
        out = np.full((max(P, 1), max(1, out_stride)), -1, np.int32)
        counts = np.zeros(max(P, 1), np.int32)
        self.ctx.check(self.ctx.lib.rcn_match_grid(self.ctx.h, None, P, self.ratio, out.ctypes.data, out.shape[1], counts.ctypes.data))
        return out[:P], counts[:P]

    def match_grid_device(self, pairs, out_dev_ptr, out_stride, counts_dev_ptr):
        """Asynchronous on the ctx stream; results stay in HBM."""
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        self.ctx.check(self.ctx.lib.rcn_match_grid_device(
            self.ctx.h, pairs.ctypes.data, pairs.shape[0], self.ratio,
            C.c_void_p(out_dev_ptr), out_stride, C.c_void_p(counts_dev_ptr)))

    def upload_coords(self, img_id, xy):
        """Integer pixel coordinates of the image's keypoints (K x 2), for filter_table_device."""
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
        self.ctx.check(self.ctx.lib.rcn_coords_upload(self.ctx.h, int(img_id), xy.ctypes.data, len(xy)))

    def upload_coords_batch(self, first_id, coords):
        """Pixel coordinates of n images (a list of K_i x 2 arrays) in one call, one synchronisation."""
        arrs = [np.ascontiguousarray(c, np.int32).reshape(-1, 2) for c in coords]
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data if len(a) else None for a in arrs])
        Ks = np.array([len(a) for a in arrs], np.int32)
        self.ctx.check(self.ctx.lib.rcn_coords_upload_batch(self.ctx.h, int(first_id), n, ptrs, Ks.ctypes.data))

    def filter_table_device(self, pairs, table_dev_ptr, out_stride, counts_dev_ptr, status_dev_ptr=None):
        """Epipolar filter of a device match table in place (SequentialReconstructor.cpp:237-269 for every
        pair); asynchronous on the ctx stream."""
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        self.ctx.check(self.ctx.lib.rcn_match_table_filter_device(
            self.ctx.h, pairs.ctypes.data, pairs.shape[0], C.c_void_p(table_dev_ptr), out_stride,
            C.c_void_p(counts_dev_ptr), C.c_void_p(status_dev_ptr) if status_dev_ptr else None))

    def set_workspace_rows(self, rows=0):
        """Query-row slots of the candidate table per pipeline chunk (0 = the library's default, 2^27)."""
        self.ctx.lib.rcn_match_set_workspace_rows.argtypes = [C.c_void_p, C.c_int64]
        self.ctx.check(self.ctx.lib.rcn_match_set_workspace_rows(self.ctx.h, int(rows)))

    def profile(self, enable=True):
        self.ctx.check(self.ctx.lib.rcn_match_profile(self.ctx.h, 1 if enable else 0))

    def stats(self):
        s = _lib.MatchStats()
        self.ctx.check(self.ctx.lib.rcn_match_last_stats(self.ctx.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_ if k != "reserved"}


def all_pairs(n_images):
    """The grid FakeImgMatcher defines (ImageMatcher.cpp:6-23: every j != i), reduced as the
    reference's loop reduces it: (i,j) with i<j is matched with query=i, train=j; (j,i) is the
    inverted map (SequentialReconstructor.cpp:219-227)."""
    i, j = np.triu_indices(n_images, 1)
    return np.stack([i, j], 1).astype(np.int32)


def match_features_grid(matcher, images, pairs=None):
    """SequentialReconstructor::matchFeatures without the geometric filter: returns
    {(i,j): {query: train}} for every matched ordered pair, including the inverted (j,i)."""
    n = len(images)
    for i, im in enumerate(images):
        matcher.upload(i, im)
    pairs = all_pairs(n) if pairs is None else np.asarray(pairs, np.int32).reshape(-1, 2)
    kmax = max(1, max(im.shape[0] for im in images))
    out, counts = matcher.match_grid(pairs, kmax)
    feature_matches = {}
    for p, (a, b) in enumerate(pairs):
        row = out[p]
        qs = np.nonzero(row >= 0)[0]
        fwd = {int(q): int(row[q]) for q in qs}
        feature_matches[(int(a), int(b))] = fwd
        feature_matches[(int(b), int(a))] = {t: q for q, t in fwd.items()}
    return feature_matches
