#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel stats CSV
(the same columns as rocprofv3's *_kernel_stats.csv): name, calls, total/avg/min/max ns, %."""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, accum_vgpr_count, lds_size, "
                      "count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc").fetchall()
    tot = sum(r[9] for r in rows) or 1
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Grid", "Workgroup", "VGPR", "AGPR", "LDS", "Calls", "TotalDurationNs", "AverageNs", "MinNs",
                    "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], f"{r[1]}x{r[2]}x{r[3]}", r[4], r[5], r[6], r[7], r[8], r[9], f"{r[10]:.1f}", r[11], r[12],
                        f"{100.0 * r[9] / tot:.2f}"])
    print(f"{len(rows)} kernel rows -> {out_csv}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
