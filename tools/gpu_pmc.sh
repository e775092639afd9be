#!/bin/bash
# PMC passes on the GPU box (separate passes: SQ counters, FETCH_SIZE, WRITE_SIZE; kernel-trace only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
ROOT=$PWD
# PMC_TARGET=fs (default: bench.py, FS-EEND model.test) | ls (tools/ls_prof.py: LS-EEND model.test 16 x T=2000, C=10) | train_ls
# PMC_TARGET=cmd: profile "$PMC_CMD" (relative to the repo root), outputs tagged ${PMC_TAG:-cmd_}
case "${PMC_TARGET:-fs}" in
  cmd) CMD="${PMC_CMD/#python /python $ROOT/}"; TAG=${PMC_TAG:-cmd_} ;;
  ls) CMD="python $ROOT/tools/ls_prof.py 2"; TAG=ls_ ;;
  train_ls) CMD="python $ROOT/bench.py --mode train --flavour ls --steps 1 --warmup 1 --no-cpu-baseline --no-breakdown"; TAG=train_ls_ ;;
  *) CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-breakdown --graph 0 ${BENCH_ARGS:-}"; TAG= ;;
esac
run_pass() {  # name, counters...
  name=$TAG$1; shift
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- $CMD) > gpurun_out/pmc/$name.log 2>&1
  echo "$name rc=$?"
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" gpurun_out/pmc/$name.csv
}
PASSES=${PMC_PASSES:-sq,sq2,fetch,tcc,write}
want() { [[ ",$PASSES," == *",$1,"* ]]; }
want sq && run_pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES
want sq2 && run_pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
want fetch && run_pass fetch FETCH_SIZE
want tcc && run_pass tcc TCC_HIT_sum TCC_MISS_sum
want write && run_pass write WRITE_SIZE
[ -f gpurun_out/pmc/${TAG}sq.csv ] && head -20 gpurun_out/pmc/${TAG}sq.csv | cut -c1-220
