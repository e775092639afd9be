#!/bin/bash
# round 6, call B: where does the transposing-read weight-gradient kernel spend its time?  kernel trace per variant / shape / stride
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
ROOT=$PWD; V=$ROOT/fs-eend_amd/csrc/variants
trace() {  # tag, lib, args of ab_wgrad_one.py
  tag=$1; lib=$2; shift 2
  rm -rf /tmp/tr_$tag
  (cd /tmp && EEND_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$tag -o t -- python $ROOT/tools/ab_wgrad_one.py "$@") > /tmp/tr_$tag.log 2>&1
  f=$(find /tmp/tr_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag [$*]"; python tools/kstats.py $f wgrad
}
NEW=$ROOT/fs-eend_amd/csrc/libeend_hip.so
for shape in "196608 2048 256" "196608 256 2048" "196608 256 256" "32768 256 256" "196608 256 256 2048 256" "196608 256 256 256 2048" "196608 256 256 2048 2048" "393216 256 256"; do
  trace new $NEW $shape
done
for v in noread nodma nst3 nst5 r05; do trace $v $V/libeend_hip_$v.so 196608 2048 256; done
for v in noread nodma nst3 nst5; do trace $v $V/libeend_hip_$v.so 196608 256 256; done
