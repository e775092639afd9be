cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for i in 1 2; do
EEND_LS_ENC_FFN=fused timeout 300 python tools/ls_prof.py 10 2>&1 | tail -1
EEND_LS_ENC_FFN=stream timeout 300 python tools/ls_prof.py 10 2>&1 | tail -1
done
timeout 1500 python -m pytest tests/test_ls_parity.py -q -x -p no:cacheprovider 2>&1 | tail -3
