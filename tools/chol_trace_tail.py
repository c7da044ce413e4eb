"""tools/chol_trace_tail.py TRACE.csv [WINDOW_US] -- the last WINDOW_US microseconds of the last factorisation in a rocprofv3 kernel trace:
one line per kernel (start, end relative to the backward substitution's start; queue; name; grid), in start order."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 700.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_trsv_bwd_chain" in r["Kernel_Name"]]
last = ends[-2] if len(ends) > 1 else ends[-1]
t_end = int(rows[last]["Start_Timestamp"])
qs = {}
for r in rows[:last + 1]:
    s, e = (int(r["Start_Timestamp"]) - t_end) / 1e3, (int(r["End_Timestamp"]) - t_end) / 1e3
    if e < -win:
        continue
    q = qs.setdefault(r["Queue_Id"], len(qs))
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    g = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    print("%9.1f %9.1f  %6.1f  q%-2d %-44s grid %s" % (s, e, e - s, q, name[:44], g))
