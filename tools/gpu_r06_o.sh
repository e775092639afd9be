#!/bin/bash
# round 6, call O: LayerNorm backward in the epilogue of the data-gradient GEMM: kernel test, step tests, step timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_train_kernels.py -q -x -k "gemm_acc or layernorm" -p no:cacheprovider 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_train_step.py tests/test_train_step_ls.py -q -x -p no:cacheprovider 2>&1 | tail -5
for r in 1 2; do
  timeout 300 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FS train', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --mode train --flavour ls --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LS train', d['value'], d['ms_per_step'])"
done
R=$PWD
for fl in fs ls; do
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o tr -- python "$R/bench.py" --mode train --flavour $fl --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown) > gpurun_out/prof_train_$fl.log 2>&1
  db=$(find gpurun_out/prof -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/r06_train_${fl}_kernel_stats_mid5.csv && grep -i "Li19E\|ln_bwd\|Li7ELi0" gpurun_out/r06_train_${fl}_kernel_stats_mid5.csv | cut -c1-200
  rm -rf gpurun_out/prof
done
