#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_ls_streaming.py tests/test_hip_ffn_stream.py tests/test_feature_pins.py tests/test_hip_ret_stream.py -q -x --timeout 600 -p no:cacheprovider > gpurun_out/r05_b_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05_b_tests.log | cut -c1-200
timeout 900 python -m pytest tests/test_long_horizon.py -q -s -k "streams" --timeout 800 -p no:cacheprovider > gpurun_out/r05_b_hour.log 2>&1; echo "hour rc=$?"; grep -E "max \|d|against|passed|failed" gpurun_out/r05_b_hour.log | cut -c1-220
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/r05_b_bench.json 2> gpurun_out/r05_b_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r05_b_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_b_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'])
for k,v in d.get('extras',{}).items():
    print(k, json.dumps(v)[:300])
PY
