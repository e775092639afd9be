#!/bin/bash
# GPU box: train bench (+breakdown/roofline/cpu_baseline), its rocprofv3 kernel stats, the LS-EEND batch-path profile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD
timeout 600 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "bench train rc=$?"
tail -3 gpurun_out/bench_train.err
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o tr -- python "$R/bench.py" --mode train --steps 3 --warmup 2 --no-breakdown --no-cpu-baseline) > gpurun_out/prof_train.log 2>&1
echo "prof rc=$?"; tail -2 gpurun_out/prof_train.log
db=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/train_kernel_stats.csv && head -40 gpurun_out/train_kernel_stats.csv | cut -c1-170
rm -rf gpurun_out/prof
bash tools/gpu_ls_prof.sh
