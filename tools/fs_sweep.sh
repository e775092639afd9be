#!/bin/bash
# Batch-size and slot-count sweep of the FS-EEND model.test step on the GPU box (hipGraph replay, no side measurements).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {  # label, bench args...
  local label=$1; shift
  timeout 200 python bench.py "$@" --no-cpu-baseline --no-extras --no-breakdown 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$label', round(d['ms_per_step'], 4), 'ms', round(d['value'] / 1e6, 3), 'M frames/s')"
}
for b in 1 8 32 64 128 256 512; do run "B=$b C=6" --batch $b; done
for c in 3 4 10 12; do run "B=64 C=$c" --slots $c; done
