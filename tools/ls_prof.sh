cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
rm -rf gpurun_out/prof_ls
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_ls" -o ls -- python "$OLDPWD/tools/ls_breakdown.py") > gpurun_out/prof_ls.log 2>&1
db=$(find gpurun_out/prof_ls -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/ls_kernel_stats.csv && head -22 gpurun_out/ls_kernel_stats.csv | cut -c1-170
rm -rf gpurun_out/prof_ls
