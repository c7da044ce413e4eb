"""tools/chol_timeline.py <kernel_trace.csv> -- per-step timeline of the last Cholesky factorisation in a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:22], r.get('Stream_Id', '?')) for r in rows)
diag = [i for i, k in enumerate(ks) if k[2].startswith('k_chol_diag')]
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 79
g = diag[-nb:]
t0, t1 = ks[g[0]][0], ks[g[-1]][1]
print("factorisation wall %.2f ms" % ((t1 - t0) / 1e6))
ds = [ks[g[i + 1]][0] - ks[g[i]][0] for i in range(nb - 1)]
print("diag-to-diag us:", [round(x / 1e3) for x in ds])
for name in ('k_chol_diag', 'void k_gemm_q<0>', 'void k_gemm_q<1>', 'k_gemm_nt'):
    sel = [k for k in ks[g[0]:g[-1] + 1] if k[2].startswith(name)]
    print("%-18s n=%d sum %.2f ms" % (name, len(sel), sum(k[1] - k[0] for k in sel) / 1e6))
for i in range(g[-4], g[-1] + 1):
    k = ks[i]; print(round((k[0] - t0) / 1e3, 1), round((k[1] - k[0]) / 1e3, 1), k[2], k[3])
