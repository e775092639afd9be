#!/bin/bash
# GPU box: training bench (1 GPU), its rocprofv3 kernel stats, and a 2-process gloo rehearsal of the N > 1 paths.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD
timeout 600 python bench.py --mode train --steps 10 --warmup 3 ${BENCH_ARGS:-} > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "bench train rc=$?"
cat gpurun_out/bench_train.json | cut -c1-600; tail -3 gpurun_out/bench_train.err
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o tr -- python "$R/bench.py" --mode train --steps 3 --warmup 2 ${BENCH_ARGS:-}) > gpurun_out/prof_train.log 2>&1
echo "prof rc=$?"; tail -2 gpurun_out/prof_train.log
db=$(find gpurun_out/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/train_kernel_stats.csv && head -45 gpurun_out/train_kernel_stats.csv | cut -c1-170
rm -rf gpurun_out/prof
if [[ "${1:-all}" == all ]]; then
  export EEND_DIST_BACKEND=gloo
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --mode train --steps 3 --warmup 1 --batch 8 > gpurun_out/rehearse_train2.json 2> gpurun_out/rehearse_train2.err; echo "rehearse train rc=$?"
  cut -c1-300 gpurun_out/rehearse_train2.json; tail -3 gpurun_out/rehearse_train2.err
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 > gpurun_out/rehearse_infer2.json 2> gpurun_out/rehearse_infer2.err; echo "rehearse infer rc=$?"
  cut -c1-300 gpurun_out/rehearse_infer2.json; tail -3 gpurun_out/rehearse_infer2.err
fi
