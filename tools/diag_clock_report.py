"""tools/diag_clock_report.py <dir> -- per dispatch of a rocprofv3 --pmc run: duration, shader cycles (GRBM_GUI_ACTIVE / 8 XCDs), effective clock,
MFMA-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)."""
import collections
import csv
import glob
import os
import sys

for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
    print("==", f)
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0], "grid": r["Grid_Size"], "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    for i, d in rows.items():
        cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        print("%4d %-44s grid %9s  %9.1f us  cycles %.3e  clock %.2f GHz  mfma_busy %.3f  mfma %.3e  lds %.3e  wave_cyc %.3e" %
              (i, d["name"][:44], d["grid"], d["t"], cyc, cyc / max(d["t"], 1e-9) * 1e-3, busy / max(1024.0 * cyc, 1.0), d.get("SQ_INSTS_MFMA", 0.0), d.get("SQ_INSTS_LDS", 0.0), d.get("SQ_WAVE_CYCLES", 0.0)))
