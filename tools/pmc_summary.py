#!/usr/bin/env python
"""Per-kernel averages of a rocprofv3 counter_collection.csv (one row per dispatch x counter)."""
import csv
import sys
from collections import defaultdict


def main(src, dst):
    rows = list(csv.DictReader(open(src)))
    if not rows:
        print("empty", src)
        return
    cols = rows[0].keys()
    kcol = "Kernel_Name" if "Kernel_Name" in cols else [c for c in cols if "ernel" in c and "ame" in c][0]
    gcol = "Grid_Size" if "Grid_Size" in cols else None
    # Persistent kernels launch the same grid for every problem size: when one (kernel, grid, counter)
    # group is clearly bimodal (max > 2.5 x min), its dispatches are reported as two rows, "#hi" / "#lo".
    vals = defaultdict(list)
    for r in rows:
        vals[(r[kcol][:90], r.get(gcol, "") if gcol else "", r["Counter_Name"])].append(float(r["Counter_Value"]))
    thr = {}
    for k, v in vals.items():
        lo, hi = min(v), max(v)
        if lo > 0 and hi > 2.5 * lo:
            thr[k] = (lo * hi) ** 0.5
    split = {k[:2] for k in thr}
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in rows:
        kern, grid, cname, v = r[kcol][:90], r.get(gcol, "") if gcol else "", r["Counter_Name"], float(r["Counter_Value"])
        if (kern, grid) in split:
            t = thr.get((kern, grid, cname))
            if t is None:      # this counter is not bimodal for the group: keep it on both rows
                for suf in (" #hi", " #lo"):
                    a = agg[(kern + suf, grid)][cname]; a[0] += v; a[1] += 1
                continue
            kern += " #hi" if v >= t else " #lo"
        a = agg[(kern, grid)][cname]
        a[0] += v
        a[1] += 1
    names = sorted({c for v in agg.values() for c in v})
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Grid", "Dispatches"] + names)
        for (k, g), v in sorted(agg.items(), key=lambda kv: -max(x[0] for x in kv[1].values())):
            n = max(x[1] for x in v.values())
            w.writerow([k, g, n] + [f"{v[c][0] / max(v[c][1], 1):.1f}" if c in v else "" for c in names])
    print(f"{len(agg)} kernels -> {dst}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
