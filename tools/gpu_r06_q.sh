#!/bin/bash
# round 6, call Q: grouped weight gradient of the four retention projections: kernel test, LS tests, step timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_train_kernels.py -q -x -k "wgrad" -p no:cacheprovider 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_train_step_ls.py tests/test_ls_train_kernels.py -q -x -p no:cacheprovider 2>&1 | tail -4
for r in 1 2 3; do
  timeout 300 python bench.py --mode train --flavour ls --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LS train', d['value'], d['ms_per_step'])"
done
