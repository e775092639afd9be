cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_attn_packed.py tests/test_cabi.py -q -p no:cacheprovider 2>&1 | tail -3
