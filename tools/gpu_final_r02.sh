#!/bin/bash
# GPU box, end of round 2: full GPU suite, smoke, default bench, train bench, rocprofv3 kernel stats of both.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD
timeout 1800 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_infer.json 2> gpurun_out/bench_infer.err; echo "bench rc=$?"
timeout 900 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "bench train rc=$?"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o fs -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-breakdown --graph 0) > gpurun_out/prof_infer.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/fs_kernel_stats.csv | tail -1
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o tr -- python "$R/bench.py" --mode train --steps 3 --warmup 2 --no-breakdown --no-cpu-baseline) > gpurun_out/prof_train.log 2>&1
db=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/train_kernel_stats.csv | tail -1
rm -rf gpurun_out/prof
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_infer.json').read().strip().splitlines()[-1])
print('infer', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_attention']['mfma_frac'], d['cpu_baseline']['value'])
t = json.loads(open('gpurun_out/bench_train.json').read().strip().splitlines()[-1])
print('train', t['value'], t['ms_per_step'], t.get('roofline', {}).get('kernel'), t.get('roofline', {}).get('frac'), t.get('cpu_baseline', {}).get('value'))
PY
