cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_train_kernels.py tests/test_train_step.py -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -4
timeout 600 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_train.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for b in d['breakdown'][:8]: print(b['call'], b['shape'], round(b['avg_ms'],3), round(b['share'],3), b['tflops'])
PY
