cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_train_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -k "attn" 2>&1 | tail -25
timeout 900 python -m pytest tests/test_fs_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x 2>&1 | grep -E "Error|assert|passed|failed" | head -20
timeout 300 python tools/ab_attn.py 2>&1 | tail -4
