#!/bin/bash
# round 5: the f32-residual form of spk_stream in the LS decoder (A/B) + the new kernel-level tests + FS headline check (head-0 wait change)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_hip_spk_stream.py tests/test_hip_ffn.py tests/test_hip_kernels.py -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -3
for v in 1 0; do echo SPK_STREAM_LS=$v; EEND_SPK_STREAM_LS=$v timeout 400 python -m pytest tests/test_ls_parity.py -q -s -k golden -p no:cacheprovider 2>&1 | grep -E "max \|logits|passed|failed"; EEND_SPK_STREAM_LS=$v timeout 200 python tools/ls_breakdown.py 2>&1 | grep -E "frames/s|spk|linear_res_ln|ms/step|total"; done
timeout 300 python -m pytest tests/test_fs_parity.py tests/test_ls_longform.py -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_e_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_e_bench.json'))
print('FS', d['value'], d['ms_per_step'], d['roofline'])
print('LS', d['extras']['ls_eend_batch']['frames_per_s'], d['extras']['ls_eend_batch']['ms_per_step'])
print('LS64', d['extras']['ls_eend_streaming_graph_64streams'])
PY
