#!/bin/bash
# GPU box: per-kernel breakdown of the LS-EEND batch path (rocprofv3 kernel trace -> tools/rocpd_stats.py).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD
timeout 300 python tools/ls_prof.py 5 | tail -1
rm -rf gpurun_out/prof_ls
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_ls" -o ls -- python "$R/tools/ls_prof.py" 3) > gpurun_out/prof_ls.log 2>&1
echo "prof rc=$?"; tail -2 gpurun_out/prof_ls.log
db=$(find gpurun_out/prof_ls -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/ls_kernel_stats.csv && head -60 gpurun_out/ls_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_ls
