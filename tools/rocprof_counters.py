#!/usr/bin/env python3
"""Mean per-dispatch value of the PMC counters of a rocprofv3 `--pmc` run (ROCm 7 sqlite output,
view `counters_collection`), per kernel.  FETCH_SIZE and WRITE_SIZE are collected in separate passes
(profiles/r1_traffic.json).
    python tools/rocprof_counters.py gpurun_out/pmc_fetch [kernel-name-substring]
"""
import glob
import os
import sqlite3
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = {}
for db in glob.glob(os.path.join(src, "**", "*.db"), recursive=True):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("PRAGMA table_info(counters_collection)")]
    kcol = next(c for c in cols if c in ("kernel_name", "name") or "kernel" in c.lower() and "name" in c.lower())
    ccol = next(c for c in cols if "counter" in c.lower() and "name" in c.lower())
    vcol = next(c for c in cols if c.lower() in ("value", "counter_value"))
    dcol = next((c for c in cols if "dispatch" in c.lower() and "id" in c.lower()), None)
    q = "SELECT %s, %s, %s%s FROM counters_collection" % (kcol, ccol, vcol, (", " + dcol) if dcol else "")
    for row in con.execute(q):
        if pat and pat not in row[0]:
            continue
        key = (row[0], row[1])
        d = acc.setdefault(key, {})
        disp = row[3] if dcol else len(d)
        d[disp] = d.get(disp, 0.0) + float(row[2])      # sum over XCDs / instances of one dispatch
    con.close()
for (kernel, counter), d in sorted(acc.items()):
    vals = list(d.values())
    print("%-24s mean per dispatch %.2f over %d dispatches   %s" % (counter, sum(vals) / len(vals), len(vals), kernel[:90]))
