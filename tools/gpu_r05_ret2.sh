#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_hip_ret_stream.py -q --timeout 300 -p no:cacheprovider > gpurun_out/r05_ret_kernel.log 2>&1; echo "kernel rc=$?"; tail -12 gpurun_out/r05_ret_kernel.log | cut -c1-200
R=$PWD
rm -rf gpurun_out/prof_ls
(cd /tmp && EEND_RET_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_ls" -o ls -- python "$R/tools/ls_breakdown.py") > gpurun_out/r05_ls_breakdown_rs1.txt 2>&1
db=$(find gpurun_out/prof_ls -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/r05_ls_kernel_stats_rs1.csv && head -14 gpurun_out/r05_ls_kernel_stats_rs1.csv | cut -c1-150
rm -rf gpurun_out/prof_ls
grep -E "frames/s|retention|convert|attnout" gpurun_out/r05_ls_breakdown_rs1.txt | head
