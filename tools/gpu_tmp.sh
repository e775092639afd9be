cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_fs_parity.py -q -s -p no:cacheprovider -k "long" 2>&1 | grep -v "^$" | tail -12
