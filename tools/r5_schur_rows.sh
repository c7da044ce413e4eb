#!/bin/bash
# tools/r5_schur_rows.sh TAG -- round 5: variants of the Schur kernels against each other: tests, same bits on five scenes, kernel times
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5schur_${1:-a}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ba_gpu.py tests/test_ba_session_gpu.py -x -q > $O/test_ba_gpu.txt 2>&1 || { tail -40 $O/test_ba_gpu.txt; echo "BA tests failed"; exit 1; }
tail -2 $O/test_ba_gpu.txt
# same bits: the two forms on one scene
RCN_LIB=tools/librcn_diag.so python3 - > $O/same_bits.txt 2>&1 <<'PY'
import os, sys, subprocess, hashlib
code = '''
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
import torch
import numpy as np
from reconstructor_amd import _lib, ba, synth_ba
ctx = _lib.Context(0)
h = hashlib.sha256()
for nc, npts, opp in ((1000, 100000, 10), (200, 20000, 10), (25, 1500, 8), (6, 300, 6), (3, 40, 3)):
    sc = synth_ba.make_scene(nc, npts, obs_per_point=min(opp, nc), seed=77)
    P, I, X, s = ba.solve_scene(ctx, sc)
    for a in (P, I, X):
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(repr((s["iterations"], s["final_cost"])).encode())
print(h.hexdigest())
'''
outs = []
for smb in ("4", "1", "2", "3"):
    env = dict(os.environ, RCN_SCHUR_NBW=smb)
    outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1])
print(outs)
print("SAME BITS" if len(set(outs)) == 1 and len(outs[0]) == 64 else "DIFFERENT")
PY
tail -2 $O/same_bits.txt
bash tools/r5_stream_prof.sh schur_${1:-a} RCN_SCHUR_NBW=1 RCN_SCHUR_NBW=2 RCN_SCHUR_NBW=3 > $O/prof.txt 2>&1; grep -v "^W2026\|^E2026" $O/prof.txt | grep 'schur\|==\|^run\|streaming'
