#!/usr/bin/env python3
"""Durations of the iteration's product launches in a rocprofv3 rocpd file, split by what ran before them:
inside a solve (behind iter_maxpy_uhead / maxpy_head) or in bench.py's batch of back-to-back repetitions.
usage: tools/rocpd_product.py run_results.db [kernel-substring]"""
import sqlite3, statistics, sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "spmv_dict2_kernel<true, true"
rows = db.execute("select name, start, end from kernels order by start").fetchall()
groups = {"in a solve": [], "batch": [], "other": []}
for i, (name, st, en) in enumerate(rows):
    if pat not in name:
        continue
    prev = rows[i - 1][0] if i else ""
    d = (en - st) / 1e3
    if "uhead" in prev or "maxpy_head" in prev:
        groups["in a solve"].append(d)
    elif pat in prev:
        groups["batch"].append(d)
    else:
        groups["other"].append(d)
for k, v in groups.items():
    if v:
        full = [x for x in v if x >= 0.5 * statistics.median(v)]
        print(f"{k:12s} n {len(v):4d}  mean {statistics.mean(v):7.2f} us  median {statistics.median(v):7.2f}  "
              f"mean without gated launches {statistics.mean(full):7.2f} (n {len(full)})  min {min(v):.2f} max {max(v):.2f}")
