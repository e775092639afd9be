#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprofv3 kernel trace.
# Everything judged later is copied from gpurun_out/ into profiles/ by hand.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
WHAT="${1:-all}"
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
nproc >> gpurun_out/gpu_info.txt
if [[ "$WHAT" == all || "$WHAT" == *test* ]]; then
  timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -40 gpurun_out/pytest_gpu.log
fi
if [[ "$WHAT" == all || "$WHAT" == *smoke* ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log
fi
if [[ "$WHAT" == all || "$WHAT" == *bench* ]]; then
  timeout 600 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
  tail -5 gpurun_out/bench.err; cat gpurun_out/bench.json
fi
if [[ "$WHAT" == all || "$WHAT" == *prof* ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o fs -- \
      python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras --graph 0 ${BENCH_ARGS:-}) > gpurun_out/prof_bench.log 2>&1
  echo "prof rc=$?"; tail -2 gpurun_out/prof_bench.log
  db=$(find gpurun_out/prof -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/kernel_stats.csv && head -25 gpurun_out/kernel_stats.csv | cut -c1-200
  rm -rf gpurun_out/prof
fi
