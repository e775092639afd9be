cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_train_kernels.py tests/test_train_step.py tests/test_train_step_ls.py tests/test_cabi.py -q -x -p no:cacheprovider 2>&1 | tail -3
