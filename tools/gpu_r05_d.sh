#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_hip_ffn.py tests/test_hip_ffn_stream.py tests/test_hip_kernels.py -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -2
for v in 1 0; do echo OUT2_SPLIT=$v; EEND_LS_OUT2_SPLIT=$v timeout 400 python -m pytest tests/test_ls_parity.py -q -s -k golden -p no:cacheprovider 2>&1 | grep -E "max \|logits|passed|failed"; EEND_LS_OUT2_SPLIT=$v timeout 200 python tools/ls_breakdown.py 2>&1 | grep -E "frames/s|attnout_ffn_fused"; done
