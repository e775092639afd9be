#!/bin/bash
# round 6, call P: BatchNorm statistic / gradient kernels with four rows in flight + __umul24 mask function: training tests, step timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 2400 python -m pytest tests/test_train_kernels.py tests/test_ls_train_kernels.py tests/test_train_step.py tests/test_train_step_ls.py tests/test_trainer_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4
for r in 1 2; do
  timeout 300 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FS train', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --mode train --flavour ls --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LS train', d['value'], d['ms_per_step'])"
done
R=$PWD
for fl in fs ls; do
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o tr -- python "$R/bench.py" --mode train --flavour $fl --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown) > gpurun_out/prof_train_$fl.log 2>&1
  db=$(find gpurun_out/prof -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/r06_train_${fl}_kernel_stats_mid6.csv && grep -i "bn_\|ffn_train_stream_kernel<1" gpurun_out/r06_train_${fl}_kernel_stats_mid6.csv | cut -c1-200
  rm -rf gpurun_out/prof
done
