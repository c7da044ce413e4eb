"""tools/small_trace_seq.py KERNEL_TRACE.csv [MEMCOPY_TRACE.csv] -- the launches of the LAST solve in a rocprofv3 kernel trace of tools/ba_small_run.py,
in start order: start, end (us, relative to the solve's first launch), duration, gap to the previous end, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows]
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", r.get("Name", "?"))))
ev.sort()
# solves are separated by k_pair_count (lists are rebuilt per timed solve): take the last one with many launches
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_pair_count")]
# the timed solve is the one with the larger problem: its k_pair_count is every second one; take the last whose segment is longest
segs = [(starts[i], starts[i + 1] if i + 1 < len(starts) else len(ev)) for i in range(len(starts))]
a, b = max(segs[-4:], key=lambda s: s[1] - s[0])
# walk back to the solve's first launch: everything within 300 us before k_pair_count
t_pc = ev[a][0]
while a > 0 and t_pc - ev[a - 1][1] < 60000 and not ev[a - 1][2].startswith("k_ba_plus"):
    a -= 1
t0 = ev[a][0]
prev = t0
for s, e, n in ev[a:b]:
    print("%8.1f %8.1f  %6.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n[:60]))
    prev = max(prev, e)
