#!/bin/bash
# round 6, call M: LS decoder layer tail on the packed stream (LO form): kernel test, LS parity goldens, LS throughput + kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_hip_ffn.py tests/test_hip_ffn_stream.py -q -x -p no:cacheprovider 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_ls_parity.py tests/test_ls_streaming.py -q -x -p no:cacheprovider 2>&1 | tail -6
for r in 1 2 3; do timeout 300 python tools/ls_prof.py 10 2>&1 | tail -1; done
R=$PWD; rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o tr -- python "$R/tools/ls_prof.py" 5) > gpurun_out/prof_ls.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/r06_ls_kernel_stats_mid.csv && head -8 gpurun_out/r06_ls_kernel_stats_mid.csv | cut -c1-180
rm -rf gpurun_out/prof
