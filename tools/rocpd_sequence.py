#!/usr/bin/env python
"""Print the kernels of the last `n` launches of a rocprofv3 rocpd database in launch order (name, duration us, gap to the previous end)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[-n:]
prev = None
for name, s, e in rows:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(e - s) / 1e3:9.1f} us  gap {gap:7.1f}  {name[:110]}")
    prev = e
