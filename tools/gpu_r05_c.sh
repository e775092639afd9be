#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_train_step.py tests/test_train_kernels.py -q -x --timeout 600 -p no:cacheprovider > gpurun_out/r05_c_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_c_tests.log | cut -c1-200
for v in 1 0; do
  EEND_GEMM_BM128=$v timeout 600 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05_c_train_bm$v.json 2> gpurun_out/r05_c_train_bm$v.err; echo "train bm128=$v rc=$?"
  python - <<PY
import json
d=json.load(open('gpurun_out/r05_c_train_bm$v.json'))
print('ms_per_step', d['ms_per_step'])
for k in d.get('breakdown', d.get('kernel_breakdown', []))[:12]: print(k)
PY
done
