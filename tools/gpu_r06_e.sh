#!/bin/bash
# round 6, call E: fused attention / retention backward (attn_bwd_fused.hip): kernel tests, step goldens, A/B against the two-kernel form
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
V=$PWD/fs-eend_amd/csrc/variants
timeout 900 python -m pytest tests/test_train_kernels.py -q -x -k "attn_fwd_lse_and_bwd" -p no:cacheprovider 2>&1 | tail -12
timeout 900 python -m pytest tests/test_train_step_ls.py -q -x -k "retention_core" -p no:cacheprovider 2>&1 | tail -12
echo "== fused"; timeout 300 python tools/ab_attn_bwd.py
echo "== two kernels"; EEND_HIP_LIB=$V/libeend_hip_two.so timeout 300 python tools/ab_attn_bwd.py
timeout 1500 python -m pytest tests/test_train_step.py tests/test_train_step_ls.py -q -x -p no:cacheprovider 2>&1 | tail -15
for r in 1 2; do for lib in new two; do
  if [ $lib = new ]; then unset EEND_HIP_LIB; else export EEND_HIP_LIB=$V/libeend_hip_$lib.so; fi
  timeout 300 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib FS', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --mode train --flavour ls --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib LS', d['value'], d['ms_per_step'])"
done; done
unset EEND_HIP_LIB
