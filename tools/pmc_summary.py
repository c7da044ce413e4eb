"""tools/pmc_summary.py <dir> [<dir> ...] -- per-kernel mean of every counter in rocprofv3 --pmc output dirs."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-42s %-28s n=%d mean=%.4g sum=%.4g" % (k, c, len(v), sum(v) / len(v), sum(v)))
