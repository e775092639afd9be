#!/bin/bash
# round 6, call F: PMC of the fused attention / retention backward kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
N_ITER=2 PMC_TARGET=cmd PMC_CMD="python tools/ab_ret_bwd.py" PMC_TAG=ret_ PMC_PASSES=sq,sq2 bash tools/gpu_pmc.sh > /tmp/pmc1.log 2>&1
for f in gpurun_out/pmc/ret_sq.csv gpurun_out/pmc/ret_sq2.csv; do echo "-- $f"; grep -i "Kernel\|attn_bwd_fused" $f | cut -c1-300; done
PMC_TARGET=cmd PMC_CMD="python tools/ab_attn_bwd.py" PMC_TAG=att_ PMC_PASSES=sq,sq2 bash tools/gpu_pmc.sh > /tmp/pmc2.log 2>&1
for f in gpurun_out/pmc/att_sq.csv gpurun_out/pmc/att_sq2.csv; do echo "-- $f"; grep -i "Kernel\|attn_bwd_fused" $f | cut -c1-300; done
