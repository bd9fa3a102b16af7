#!/usr/bin/env python3
"""Timeline around one restart boundary from a rocprofv3 rocpd file: every kernel between the middle
krylov_cycle_end launch's predecessor iterations and the first iterations of the next cycle, with its duration and
the idle gap in front of it:  tools/rocpd_timeline.py run_results.db [before] [after] [which]
(which: index of the krylov_cycle_end launch to centre on; default: the middle one)"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
before = int(sys.argv[2]) if len(sys.argv) > 2 else 8
after = int(sys.argv[3]) if len(sys.argv) > 3 else 22
rows = db.execute("select name, start, end from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if "krylov_cycle_end" in r[0]]
if not ends:
    sys.exit("no krylov_cycle_end launch in the trace")
mid = ends[int(sys.argv[4])] if len(sys.argv) > 4 else ends[len(ends) // 2]
prev_end = rows[mid - before - 1][2]
t0 = rows[mid - before][1]
for name, st, en in rows[mid - before: mid + after]:
    short = name.split("(")[0].replace("void spk::k::", "").replace("spk::k::", "")[:44]
    print(f"{(st - t0) / 1e3:9.2f} us  gap {(st - prev_end) / 1e3:6.2f}  dur {(en - st) / 1e3:7.2f}  {short}")
    prev_end = en
