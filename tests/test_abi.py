"""CPU suite: the C-ABI library loads and exports every symbol include/rcn.h declares.
No compute call is made here (no GPU in this tier)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from reconstructor_amd import _build, _lib
    _build.build()
    return _lib.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "rcn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rcn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(lib):
    from reconstructor_amd import _lib
    declared = _declared()
    assert declared == sorted(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_version_and_default_options(lib):
    from reconstructor_amd import _lib
    assert b"gfx950" in lib.rcn_version()
    o = _lib.BaOptions()
    lib.rcn_ba_default_options(3, C.byref(o))
    assert (o.max_iterations, o.intrinsics_mode) == (150, 0)      # BundleAdjuster.cpp:112-115,135-137
    lib.rcn_ba_default_options(10, C.byref(o))
    assert (o.max_iterations, o.intrinsics_mode) == (50, 1)       # :117-121,139-141
    assert o.focal_upper_bound == 1000.0 and o.initial_trust_region_radius == 1e4


def test_no_device_fails_loudly(lib):
    """Without a GPU the product refuses to run: there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from reconstructor_amd import _lib
    with pytest.raises(_lib.RcnError):
        _lib.Context(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no file of the product may import / include / link it."""
    pkg = os.path.join(ROOT, "reconstructor_amd")
    bad = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.\.?oracle|#include\s+\".*oracle)|liborc|orc_", re.M)
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not bad.search(open(os.path.join(d, f)).read()), f


def test_shipping_library_reads_no_environment_switch(lib):
    """Ablations and alternative device paths are compiled out of the product (-DRCN_DIAG builds tools/librcn_diag.so
    only): the shipping binary does not even contain the names of the switches."""
    from reconstructor_amd import _lib
    blob = open(_lib.SO_PATH, "rb").read()
    for name in (b"RCN_COARSE_ABL", b"RCN_BA_SCHUR_ATOMICS", b"RCN_BA_TRSV_FWD", b"RCN_MATCH_CHUNKS", b"RCN_FORCE_EXACT",
                 b"RCN_MATCH_NO_ORDER", b"RCN_NO_CU_MASK"):
        assert name not in blob, name
    import re
    names = set(re.findall(rb"RCN_[A-Z][A-Z0-9_]{3,}", blob)) - {b"RCN_COUNTER_BYTES", b"RCN_HIST_BINS"}      # (two constants quoted in error messages)
    assert not names, "the shipping binary names an environment switch: %r" % names
    assert b"DIAGNOSTIC" not in lib.rcn_version()


def test_partition_functions_need_no_gpu(lib):
    """rcn_shard_owned_images / rcn_shard_pairs are pure host code: callable without a device."""
    import ctypes as C
    import numpy as np
    n, world = 11, 4
    seen = []
    for r in range(world):
        cnt = lib.rcn_shard_pair_count(n, world, r)
        buf = np.zeros((cnt, 2), np.int32)
        assert lib.rcn_shard_pairs(n, world, r, buf.ctypes.data) == 0
        seen.append(buf)
        lo, c = C.c_int32(), C.c_int32()
        assert lib.rcn_shard_owned_images(n, world, r, C.byref(lo), C.byref(c)) == 0 and lo.value == min(n, 3 * r)
    allp = np.concatenate(seen)
    assert len(allp) == n * (n - 1) // 2 and len({tuple(p) for p in allp}) == len(allp) and (allp[:, 0] < allp[:, 1]).all()
    assert lib.rcn_shard_pairs(n, 0, 0, None) == -1 and lib.rcn_shard_pair_count(n, 4, 4) == -1
