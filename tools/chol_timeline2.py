"""tools/chol_timeline2.py <kernel_trace.csv> [nblk] -- the chain of the LAST Cholesky factorisation in a rocprofv3 kernel
trace, step by step: diagonal block, the two critical-tile kernels and the gaps between them; which queue each stream got."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 79
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', '?'), r.get('Queue_Id', '?'), int(r['Grid_Size_X'])) for r in rows)
diag = [i for i, k in enumerate(ks) if k[2].startswith('k_chol_diag')]
g = diag[-nb:]
t0, t1 = ks[g[0]][0], ks[g[-1]][1]
print("factorisation wall %.3f ms (%d diagonal blocks)" % ((t1 - t0) / 1e6, nb))
sel = ks[g[0]:g[-1] + 1]
qs = defaultdict(set)
for k in sel:
    qs[k[3]].add(k[4])
print("stream -> queues:", dict(qs))
chainA = [k for k in sel if k[3] == ks[g[0]][3]]
print("chain-A kernels per step (us from the step's diagonal start): name dur | gap to next")
step = []
out = []
for k in chainA:
    if k[2].startswith('k_chol_diag') and step:
        out.append(step); step = []
    step.append(k)
out.append(step)
for si, st in enumerate(out):
    if si % max(1, len(out) // 20) and si not in (0, len(out) - 1):
        continue
    base = st[0][0]
    nxt = out[si + 1][0][0] if si + 1 < len(out) else st[-1][1]
    desc = " ".join("%s[%d,%d..%d]" % (k[2].split('(')[0].replace('void ', '')[:14], k[5], (k[0] - base) // 1000, (k[1] - base) // 1000) for k in st)
    print("step %3d: %s | next diag at %d us" % (si, desc, (nxt - base) // 1000))
d2d = [(out[i + 1][0][0] - out[i][0][0]) / 1e3 for i in range(len(out) - 1)]
print("diag-to-diag us: min %.0f median %.0f max %.0f sum %.2f ms" % (min(d2d), sorted(d2d)[len(d2d) // 2], max(d2d), sum(d2d) / 1e3))
for name in ('k_chol_diag', 'void k_gemm_q<0>', 'void k_gemm_q<1>', 'void k_gemm_nt', 'k_ring_gate'):
    s2 = [k for k in sel if k[2].startswith(name)]
    if s2:
        print("%-18s n=%d sum %.2f ms avg %.1f us" % (name, len(s2), sum(k[1] - k[0] for k in s2) / 1e6, sum(k[1] - k[0] for k in s2) / 1e3 / len(s2)))
# the bulk stream: duration of every bulk update and the idle time in front of it
bulk_stream = max(set(k[3] for k in sel if k[2].startswith('void k_gemm_nt')), key=lambda q: sum(k[1] - k[0] for k in sel if k[3] == q and k[2].startswith('void k_gemm_nt')), default=None)
bulk = [k for k in sel if k[2].startswith('void k_gemm_nt') and k[3] == bulk_stream]
if bulk:
    print("bulk updates: start(us) dur(us) idle-before(us) grid")
    idle = 0.0
    for i, k in enumerate(bulk):
        gap = (k[0] - bulk[i - 1][1]) / 1e3 if i else (k[0] - t0) / 1e3
        idle += gap if i else 0.0
        if i % max(1, len(bulk) // 24) == 0 or i == len(bulk) - 1:
            print("  %8.0f %7.1f %7.1f %6d" % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, gap, k[5] // 256))
    print("bulk busy %.2f ms, idle between bulk kernels %.2f ms, first starts at %.0f us, last ends %.0f us before the end" % (
        sum(k[1] - k[0] for k in bulk) / 1e6, idle / 1e3, (bulk[0][0] - t0) / 1e3, (t1 - bulk[-1][1]) / 1e3))
# everything that ran during a few chosen steps, all streams (start..end us from the step's diagonal start, stream, grid in workgroups)
import os
for si in [int(x) for x in os.environ.get("TIMELINE_STEPS", "").split(",") if x]:
    if si + 1 >= len(out):
        continue
    a, b = out[si][0][0], out[si + 1][0][0]
    print("step %d, all streams:" % si)
    for k in sel:
        if k[1] > a and k[0] < b:
            print("   s%s %-22s grid %5d  %7.1f .. %7.1f" % (k[3], k[2].split('(')[0].replace('void ', '')[:22], k[5] // max(1, (256 if 'gemm' in k[2] else 64 if 'gate' in k[2] else 512)), (k[0] - a) / 1e3, (k[1] - a) / 1e3))
