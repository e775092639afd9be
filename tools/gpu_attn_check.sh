cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_attn_fused.py -m gpu -q --timeout 300 -p no:cacheprovider -s 2>&1 | tail -16
timeout 300 python tools/ab_attn.py 2>&1 | tail -10
