#!/usr/bin/env python
"""HBM bytes per launch from the FETCH_SIZE and WRITE_SIZE passes (gpurun_out/pmc/{fetch,write}.csv,
written by tools/gpu_pmc.sh + tools/pmc_summary.py) -> profiles/<name>.json.
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts a 128-B request as 64 B, so the read
side is doubled (MI355X_MICROARCH.md, HBM section)."""
import csv
import json
import sys


def load(path, col):
    out = {}
    for r in csv.DictReader(open(path)):
        if r.get(col):
            out[f"{r['Kernel']} grid={r['Grid']}"] = float(r[col]) * 1024.0
    return out


def main(fetch_csv, write_csv, dst, workload="bench.py --steps 2 --graph 0, B=64 T=500 C=6"):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0))):
        fb, wb = 2.0 * f.get(k, 0.0), w.get(k, 0.0)
        kernels[k] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
    note = ("rocprofv3 --pmc, separate passes (FETCH_SIZE alone, WRITE_SIZE alone), " + workload + "; per-launch averages; FETCH_SIZE doubled per MI355X_MICROARCH.md; KB -> bytes x1024; "
            "'#hi'/'#lo' = the large / small problem size of a persistent kernel (same grid for both)")
    json.dump({"note": note, "kernels": kernels}, open(dst, "w"), indent=1)
    print(f"{len(kernels)} kernels -> {dst}")


if __name__ == "__main__":
    main(*sys.argv[1:5])       # optional 4th argument: the workload that was profiled (goes into the note)
