#!/bin/bash
# round 5: the fused retention (ret_stream.hip): kernel tests, LS goldens, breakdown A/B against the two-call form
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_hip_ret_stream.py -q -x --timeout 300 -p no:cacheprovider > gpurun_out/r05_ret_kernel.log 2>&1; echo "kernel rc=$?"; tail -25 gpurun_out/r05_ret_kernel.log
timeout 900 python -m pytest tests/test_ls_parity.py tests/test_ls_longform.py -q -s --timeout 600 -p no:cacheprovider > gpurun_out/r05_ret_parity.log 2>&1; echo "parity rc=$?"; grep -E "max \|logits|passed|failed|Error|error" gpurun_out/r05_ret_parity.log | tail -30
for v in "1 1" "1 0" "0 0"; do set -- $v
  EEND_RET_STREAM=$1 EEND_RET_XLO=$2 timeout 300 python tools/ls_breakdown.py > gpurun_out/r05_ls_breakdown_rs$1_lo$2.txt 2>&1; echo "breakdown RET_STREAM=$1 XLO=$2 rc=$?"; head -12 gpurun_out/r05_ls_breakdown_rs$1_lo$2.txt
done
