#!/bin/bash
# round 5: fused training FFN forward (ffn.hip MODE 3): kernel test, training goldens, A/B of the training steps
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_train_kernels.py -q -x -k "ffn" -p no:cacheprovider 2>&1 | tail -3
timeout 900 python -m pytest tests/test_train_step.py tests/test_train_step_ls.py -q -x -p no:cacheprovider 2>&1 | tail -3
for r in 1 2; do for v in 0 1; do
  echo "FUSED=$v"; EEND_TRAIN_FFN_FUSED=$v timeout 300 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FS', d['value'], d['ms_per_step'])"
  EEND_TRAIN_FFN_FUSED=$v timeout 300 python bench.py --mode train --flavour ls --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LS', d['value'], d['ms_per_step'])"
done; done
