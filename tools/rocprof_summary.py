#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, median, p10, p90 in us) of a rocprofv3 run.
ROCm 7 writes one sqlite database per process (`--kernel-trace --stats -d DIR`); this reads the
`kernels` view of every *.db under DIR and writes the CSV kept under profiles/.
    python tools/rocprof_summary.py gpurun_out/prof profiles/r1g_static_stack_kernel_stats.csv
"""
import csv
import glob
import os
import sqlite3
import sys

import numpy as np

src, dst = sys.argv[1], sys.argv[2]
durs = {}
for db in glob.glob(os.path.join(src, "**", "*.db"), recursive=True):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c.lower()][0]
    if "duration" in cols:
        q = "SELECT %s, duration FROM kernels" % name_col
    else:
        q = "SELECT %s, (end - start) FROM kernels" % name_col
    for name, d in con.execute(q):
        durs.setdefault(name, []).append(float(d))
    con.close()
if not durs:
    sys.exit("no kernel records under %s" % src)
total = sum(sum(v) for v in durs.values())
rows = []
for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    a = np.array(v) * 1e-3          # ns -> us
    rows.append([name, len(a), round(a.sum(), 3), round(a.mean(), 3), round(100.0 * a.sum() * 1e3 / total, 4),
                 round(float(np.median(a)), 3), round(float(np.percentile(a, 10)), 3),
                 round(float(np.percentile(a, 90)), 3)])
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDuration(us)", "AverageDuration(us)", "Percentage", "MedianDuration(us)",
                "P10(us)", "P90(us)"])
    w.writerows(rows)
print("wrote %s (%d kernels)" % (dst, len(rows)))
for r in rows[:3]:
    print("  %-70s calls %d avg %.2f us median %.2f us" % (r[0][:70], r[1], r[3], r[5]))
