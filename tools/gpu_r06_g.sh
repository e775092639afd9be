#!/bin/bash
# round 6, call G: after pruning the A/B switches / removed kernels: the whole GPU suite, the default bench line (FS inference + extras), training benches
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 2400 python -m pytest tests/ -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r06_pytest_gpu_mid.log
timeout 900 python bench.py > gpurun_out/r06_fs_bench_mid.json 2> gpurun_out/r06_fs_bench_mid.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_fs_bench_mid.json').read().strip().splitlines()[-1])
print('FS', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'])
ex=d.get('extras',{})
for k,v in ex.items():
    if isinstance(v,dict):
        print(' ', k, {kk:vv for kk,vv in v.items() if not isinstance(vv,(dict,list))})
    else: print(' ', k, v)
PY
for r in 1 2; do
  timeout 300 python bench.py --mode train --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FS train', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --mode train --flavour ls --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LS train', d['value'], d['ms_per_step'])"
done
