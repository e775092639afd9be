#!/bin/bash
# round 6, call I: FFN training stream kernels: tests, then A/B timing (+ study variants given as arguments)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_train_kernels.py -q -x -k "ffn_train_fused or ffn_bwd_data_fused or wgrad" -p no:cacheprovider 2>&1 | tail -8
timeout 600 python tools/ab_ffn_train.py 196608 32768 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_ab_ffn_train2.txt
