#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_hip_spk_stream.py -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -2
for v in 1; do echo SPK_STREAM_LS=$v; EEND_SPK_STREAM_LS=$v timeout 400 python -m pytest tests/test_ls_parity.py -q -s -k golden -p no:cacheprovider 2>&1 | grep -E "max \|logits|passed|failed"; EEND_SPK_STREAM_LS=$v timeout 200 python tools/ls_breakdown.py 2>&1 | grep -E "frames/s|spk|ms/step"; done
