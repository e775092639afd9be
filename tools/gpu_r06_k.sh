#!/bin/bash
# round 6, call K: cheaper dropout hash (all sites): every training test
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 2400 python -m pytest tests/test_train_kernels.py tests/test_ls_train_kernels.py tests/test_train_step.py tests/test_train_step_ls.py tests/test_trainer_gpu.py -q -p no:cacheprovider 2>&1 | tail -12
