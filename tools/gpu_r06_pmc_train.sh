#!/bin/bash
# round 6: PMC passes (FETCH_SIZE, WRITE_SIZE, SQ) of one FS and one LS training step -> gpurun_out/pmc/r06_train_{fs,ls}_*.csv + traffic JSONs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
PMC_TARGET=cmd PMC_TAG=r06_train_fs_ PMC_PASSES=fetch,write,sq PMC_CMD="python bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline --no-breakdown" bash tools/gpu_pmc.sh > gpurun_out/pmc/r06_train_fs.log 2>&1
PMC_TARGET=cmd PMC_TAG=r06_train_ls_ PMC_PASSES=fetch,write,sq PMC_CMD="python bench.py --mode train --flavour ls --steps 1 --warmup 1 --no-cpu-baseline --no-breakdown" bash tools/gpu_pmc.sh > gpurun_out/pmc/r06_train_ls.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc/r06_train_fs_fetch.csv gpurun_out/pmc/r06_train_fs_write.csv gpurun_out/r06_train_fs_pmc_traffic.json "bench.py --mode train --steps 1, FS B=64 T=500 4 speakers"
python tools/pmc_traffic.py gpurun_out/pmc/r06_train_ls_fetch.csv gpurun_out/pmc/r06_train_ls_write.csv gpurun_out/r06_train_ls_pmc_traffic.json "bench.py --mode train --flavour ls --steps 1, LS B=64 T=1000 4 speakers"
