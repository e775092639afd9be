#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for v in 1 0; do echo SPK_STREAM_LS=$v; EEND_SPK_STREAM_LS=$v timeout 400 python -m pytest tests/test_ls_parity.py -q -s -k golden -p no:cacheprovider 2>&1 | grep -E "RMS|max \|"; done
