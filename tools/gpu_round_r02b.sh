#!/bin/bash
# GPU box: training tests (incl. dropout), rest of the suite, both bench modes, dropout on/off A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_train_kernels.py tests/test_train_step.py -m gpu -q --timeout 600 -p no:cacheprovider -s > gpurun_out/pytest_train.log 2>&1
echo "pytest train rc=$?" >> gpurun_out/pytest_train.log
grep -E "passed|failed|error|rc=" gpurun_out/pytest_train.log | tail -5
timeout 600 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/bench_train_drop.json 2> gpurun_out/bench_train_drop.err
tail -c 600 gpurun_out/bench_train_drop.json
EEND_TRAIN_DROPOUT=0 timeout 600 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/bench_train_nodrop.json 2> gpurun_out/bench_train_nodrop.err
tail -c 600 gpurun_out/bench_train_nodrop.json
timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider --ignore tests/test_train_kernels.py --ignore tests/test_train_step.py > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_infer.json 2> gpurun_out/bench_infer.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_infer.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d.get('roofline'))
PY
