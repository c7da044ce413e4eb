"""Host-side mirror (Python) of the store file API (include/rcn.h `rcn_store_*`, csrc/store.hip): the features /
matches cache the reference lists as a TODO (README.md:39).  Pure marshalling; the file format lives in the C code."""
import ctypes as C

import numpy as np

from . import _lib


def save(path, images, coords=None, pairs=None, offsets=None, qt=None, img_ids=None):
    """images: list of (K_i, D) fp32 arrays; coords: list of (K_i, 2) int32 or None;
    pairs (P, 2) int32, offsets (P+1,) int64, qt (total, 2) int32: the layout rcn_match_compact_begin produces."""
    L = _lib.load()
    n = len(images)
    ims = [np.ascontiguousarray(im, np.float32) for im in images]
    D = max([im.shape[1] for im in ims if im.ndim == 2 and im.shape[0]] + [0])
    ids = np.ascontiguousarray(np.arange(n) if img_ids is None else img_ids, np.int32)
    Ks = np.array([len(im) for im in ims], np.int32)
    dptr = (C.c_void_p * max(n, 1))(*[im.ctypes.data if len(im) else None for im in ims])
    cs = None if coords is None else [np.ascontiguousarray(c, np.int32).reshape(-1, 2) for c in coords]
    cptr = None if cs is None else (C.c_void_p * max(n, 1))(*[c.ctypes.data if len(c) else None for c in cs])
    c = _lib.StoreContents()
    c.n_images, c.D, c.has_coords = n, D, 0 if cs is None else 1
    c.img_ids, c.img_K = ids.ctypes.data, Ks.ctypes.data
    c.desc = C.cast(dptr, C.c_void_p)
    c.coords = C.cast(cptr, C.c_void_p) if cptr is not None else None
    if pairs is not None:
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        offsets = np.ascontiguousarray(offsets, np.int64)
        qt = np.ascontiguousarray(qt, np.int32).reshape(-1, 2)
        c.n_pairs = len(pairs)
        c.pairs, c.offsets, c.qt = pairs.ctypes.data, offsets.ctypes.data, qt.ctypes.data if len(qt) else None
    rc = L.rcn_store_save(str(path).encode(), C.byref(c))
    if rc:
        raise _lib.RcnError(rc, "rcn_store_save(%s)" % path)


class Store:
    """An opened (read and verified) store file."""

    def __init__(self, path):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.rcn_store_open(str(path).encode(), C.byref(h))
        if rc:
            raise _lib.RcnError(rc, "rcn_store_open(%s)" % path)
        self.h = h
        c = _lib.StoreContents()
        self.lib.rcn_store_contents_of(h, C.byref(c))
        self.n_images, self.D, self.has_coords, self.n_pairs = c.n_images, c.D, bool(c.has_coords), c.n_pairs
        n = c.n_images

        def arr(ptr, ctype, count, dtype):
            if not ptr or count == 0:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).copy()
        self.img_ids = arr(c.img_ids, C.c_int32, n, np.int32)
        self.img_K = arr(c.img_K, C.c_int32, n, np.int32)
        dp = arr(c.desc, C.c_uint64, n, np.uint64)          # arrays of pointers, read as 64-bit words
        cp = arr(c.coords, C.c_uint64, n, np.uint64) if self.has_coords else None
        self.images = [arr(int(dp[i]), C.c_float, int(self.img_K[i]) * self.D, np.float32).reshape(-1, max(self.D, 1))[:int(self.img_K[i])]
                       for i in range(n)]
        self.coords = None if cp is None else [arr(int(cp[i]), C.c_int32, 2 * int(self.img_K[i]), np.int32).reshape(-1, 2) for i in range(n)]
        self.pairs = arr(c.pairs, C.c_int32, 2 * c.n_pairs, np.int32).reshape(-1, 2)
        self.offsets = arr(c.offsets, C.c_int64, c.n_pairs + 1 if c.n_pairs else 0, np.int64)
        total = int(self.offsets[-1]) if c.n_pairs else 0
        self.qt = arr(c.qt, C.c_int32, 2 * total, np.int32).reshape(-1, 2)

    def upload(self, ctx):
        """Descriptors (and coordinates) of every stored image into the ctx: the resume point."""
        ctx.check(self.lib.rcn_store_upload(ctx.h, self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.rcn_store_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
