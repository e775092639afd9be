#!/usr/bin/env python
"""Assertions on the JSON lines of a 2-rank rehearsal (tools/gpu_round.sh rehearse): what the driver's first N > 1 run must look
like -- n_gpus, a finite aggregate value, roofline from rank 0, cpu_baseline null with a reason, and (training) identical
parameters on every rank after the timed steps."""
import json
import math
import sys

ok = True
for path in sys.argv[1:]:
    try:
        line = [l for l in open(path).read().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as ex:                                   # noqa: BLE001
        print(f"{path}: no JSON line ({ex})"); ok = False; continue
    probs = []
    if d.get("n_gpus") != 2: probs.append(f"n_gpus = {d.get('n_gpus')}")
    if not (isinstance(d.get("value"), (int, float)) and math.isfinite(d["value"]) and d["value"] > 0): probs.append(f"value = {d.get('value')}")
    if d.get("scaling") != "weak": probs.append("scaling != weak")
    if "cpu_baseline" not in d or d["cpu_baseline"] is not None or not d.get("cpu_baseline_reason"): probs.append("cpu_baseline must be null with a reason at N > 1")
    if "train" in path:
        chk = d.get("dp_check")
        if not chk or not chk.get("identical_on_all_ranks"): probs.append(f"dp_check = {chk}")
        if not math.isfinite(d.get("final_loss", float("nan"))): probs.append("final_loss not finite")
    print(f"{path}: {'OK' if not probs else 'FAIL: ' + '; '.join(probs)}  (value {d.get('value'):.4g} {d.get('unit')}, {d.get('ms_per_step'):.3f} ms/step)")
    ok = ok and not probs
sys.exit(0 if ok else 1)
