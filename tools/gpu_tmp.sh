cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python tools/ab_attn_long.py 2>&1 | tail -12
