#!/bin/bash
# Profiling recipe run on the GPU box (via gpurun); summaries are copied into profiles/ afterwards.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
W=${1:-C2}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$W -o $W -- python bench.py --workload $W --steps 100 --no-cpu-baseline > gpurun_out/prof_$W.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq_$W -o $W -- python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pmc_sq_$W.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$W -o $W -- python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pmc_fetch_$W.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$W -o $W -- python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pmc_write_$W.log 2>&1
find gpurun_out -name "*.csv" | head -30
