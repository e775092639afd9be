#!/bin/bash
# round 6, call A: transposing-LDS-read weight-gradient kernel (wgrad.hip): probe of the read, kernel tests, A/B against the round-5 library,
# training goldens, training-step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R05=$PWD/fs-eend_amd/csrc/variants/libeend_hip_r05.so
./tools/tr_probe.bin | tail -5
timeout 900 python -m pytest tests/test_train_kernels.py -q -x -k "wgrad or conv1d" -p no:cacheprovider 2>&1 | tail -5
echo "== new"; timeout 600 python tools/ab_wgrad.py 2>&1 | tee gpurun_out/r06_ab_wgrad_new.txt
echo "== r05"; EEND_HIP_LIB=$R05 timeout 600 python tools/ab_wgrad.py 2>&1 | tee gpurun_out/r06_ab_wgrad_r05.txt
timeout 1500 python -m pytest tests/test_train_step.py tests/test_train_step_ls.py tests/test_train_kernels.py tests/test_ls_train_kernels.py -q -x -p no:cacheprovider 2>&1 | tail -5
for r in 1 2; do for lib in new r05; do
  if [ $lib = r05 ]; then export EEND_HIP_LIB=$R05; else unset EEND_HIP_LIB; fi
  timeout 300 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib FS', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --mode train --flavour ls --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib LS', d['value'], d['ms_per_step'])"
done; done
unset EEND_HIP_LIB
