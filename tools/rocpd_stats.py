#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel stats CSV
(the same columns as rocprofv3's *_kernel_stats.csv): name, calls, total/avg/min/max ns, %.

Persistent kernels launch with grid = #CUs whatever the problem size, so (name, grid) merges e.g. the encoder-sized
and decoder-sized calls of ffn_fused_kernel into one row.  When a row's durations are clearly bimodal (max > 2.5 x min)
it is split at sqrt(min * max) into `<name> #lo` and `<name> #hi` (the convention of tools/pmc_summary.py)."""
import csv
import math
import sqlite3
import sys


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, accum_vgpr_count, lds_size, duration "
                      "from kernels").fetchall()
    groups = {}
    for r in rows:
        groups.setdefault(r[:8], []).append(r[8])
    out = []
    for key, ds in groups.items():
        lo, hi = min(ds), max(ds)
        parts = [("", ds)]
        if len(ds) >= 4 and hi > 2.5 * lo:
            thr = math.sqrt(lo * hi)
            a, b = [d for d in ds if d < thr], [d for d in ds if d >= thr]
            if a and b and min(b) > 1.5 * max(a):
                parts = [(" #lo", a), (" #hi", b)]
        for tag, d in parts:
            out.append((key, tag, d))
    tot = sum(sum(d) for _, _, d in out) or 1
    out.sort(key=lambda x: -sum(x[2]))
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Grid", "Workgroup", "VGPR", "AGPR", "LDS", "Calls", "TotalDurationNs", "AverageNs", "MinNs",
                    "MaxNs", "Percentage"])
        for key, tag, d in out:
            w.writerow([key[0] + tag, f"{key[1]}x{key[2]}x{key[3]}", key[4], key[5], key[6], key[7], len(d), sum(d),
                        f"{sum(d) / len(d):.1f}", min(d), max(d), f"{100.0 * sum(d) / tot:.2f}"])
    print(f"{len(out)} kernel rows -> {out_csv}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
