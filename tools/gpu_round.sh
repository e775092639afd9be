#!/bin/bash
# Runs on the GPU box (via gpurun).  WHAT = space/comma separated subset of: test smoke bench train train_ls hour rehearse prof prof_train prof_train_ls
# Everything judged later is copied from gpurun_out/ into profiles/ by hand.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD
WHAT="${1:-test smoke bench}"
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
nproc >> gpurun_out/gpu_info.txt
has() { [[ " ${WHAT//,/ } " == *" $1 "* ]]; }
if has test; then
  timeout 1800 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -15 gpurun_out/pytest_gpu.log
fi
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log
fi
if has bench; then
  timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
  tail -3 gpurun_out/bench.err; cut -c1-600 gpurun_out/bench.json
fi
if has train; then
  timeout 900 python bench.py --mode train --steps 10 --warmup 3 ${TRAIN_ARGS:-} > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "bench train rc=$?"
  tail -3 gpurun_out/bench_train.err; cut -c1-600 gpurun_out/bench_train.json
fi
if has train_ls; then
  timeout 900 python bench.py --mode train --flavour ls --steps 10 --warmup 3 ${TRAIN_ARGS:-} > gpurun_out/bench_train_ls.json 2> gpurun_out/bench_train_ls.err; echo "bench train ls rc=$?"
  tail -3 gpurun_out/bench_train_ls.err; cut -c1-600 gpurun_out/bench_train_ls.json
fi
if has hour; then   # per-window error profile of the one-hour LS stream against the float64 recurrence
  timeout 600 python tools/ls_hour_profile.py > gpurun_out/ls_hour_stream_profile.txt 2>&1; echo "hour rc=$?"; tail -16 gpurun_out/ls_hour_stream_profile.txt
fi
if has rehearse; then   # the N > 1 code paths with two processes on this one GPU over gloo (no RCCL with a single device)
  export EEND_DIST_BACKEND=gloo
  port=29620
  for what in "infer" "train_fs --mode train --batch 8" "train_ls --mode train --flavour ls --batch 4"; do
    tag=${what%% *}; extra=${what#$tag}; port=$((port + 1))
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras $extra > gpurun_out/rehearse_$tag.json 2> gpurun_out/rehearse_$tag.err
    echo "rehearse $tag rc=$?"; cut -c1-260 gpurun_out/rehearse_$tag.json; tail -2 gpurun_out/rehearse_$tag.err
  done
  unset EEND_DIST_BACKEND
  python tools/check_rehearse.py gpurun_out/rehearse_infer.json gpurun_out/rehearse_train_fs.json gpurun_out/rehearse_train_ls.json | tee gpurun_out/rehearse_check.txt
fi
prof() {   # prof <tag> <bench args...>
  local tag=$1; shift
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o $tag -- python "$R/bench.py" "$@") > gpurun_out/prof_$tag.log 2>&1
  echo "prof $tag rc=$?"; tail -2 gpurun_out/prof_$tag.log
  db=$(find gpurun_out/prof -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/${tag}_kernel_stats.csv && head -12 gpurun_out/${tag}_kernel_stats.csv | cut -c1-180
  rm -rf gpurun_out/prof
}
if has prof; then prof fs --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-breakdown --graph 0 ${BENCH_ARGS:-}; fi
if has prof_train; then prof train --mode train --steps 3 --warmup 2 --no-breakdown --no-cpu-baseline; fi
if has prof_train_ls; then prof train_ls --mode train --flavour ls --steps 3 --warmup 2 --no-breakdown --no-cpu-baseline; fi
