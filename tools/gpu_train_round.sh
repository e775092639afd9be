#!/bin/bash
# GPU box: training-step kernel tests + whole-step parity, then the rest of the GPU suite (regression).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests/test_train_kernels.py tests/test_train_step.py -m gpu -q --timeout 600 -p no:cacheprovider -s ${PYTEST_ARGS:-} > gpurun_out/pytest_train.log 2>&1
echo "pytest train rc=$?" >> gpurun_out/pytest_train.log
grep -E "passed|failed|error" gpurun_out/pytest_train.log | tail -5
if [[ "${1:-all}" == all ]]; then
  timeout 1200 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider --ignore tests/test_train_kernels.py --ignore tests/test_train_step.py > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -15 gpurun_out/pytest_gpu.log
fi
