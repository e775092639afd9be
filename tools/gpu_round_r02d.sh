#!/bin/bash
# GPU box: streaming tests (FS session), full suite, default bench (with LS roofline + FS streaming graph extras)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_fs_streaming.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -12
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_infer.json 2> gpurun_out/bench_infer.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_infer.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d.get('roofline', {}).get('frac'), d.get('roofline_attention', {}).get('mfma_frac'))
ex = d['extras']
print(ex['ls_eend_batch'].get('roofline'), ex['ls_eend_batch']['frames_per_s'])
print(ex.get('fs_eend_streaming'), ex.get('fs_eend_streaming_graph'))
PY
