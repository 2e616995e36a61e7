#!/bin/bash
# Profiling recipe run on the GPU box (via gpurun); summaries are copied into profiles/ afterwards.
#   tools/profile_gpu.sh <workload> [trace] [sq] [mem]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
W=${1:-C2}; shift
WHAT="${*:-trace sq mem}"
B="python bench.py --workload $W --no-cpu-baseline --no-rollout --no-legs"
# the code the counters are measured on (bench.py flags figures from another tree as stale)
python -c 'from gym_pcgrl_amd import _lib; print(_lib.source_hash())' > gpurun_out/srchash_$W.txt
for what in $WHAT; do
  case $what in
    rtrace) rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${W}R -o ${W}R -- python bench.py --workload $W --no-cpu-baseline --no-legs --steps 200 > gpurun_out/prof_${W}R.log 2>&1 ;;   # with the pcgrl_rollout leg: k_step<..., true> / k_step_solver
    trace) rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$W -o $W -- $B --steps 100 > gpurun_out/prof_$W.log 2>&1 ;;
    sq) rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq_$W -o $W -- $B --steps 100 --warmup 20 > gpurun_out/pmc_sq_$W.log 2>&1
        rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/pmc_sq2_$W -o $W -- $B --steps 100 --warmup 20 > gpurun_out/pmc_sq2_$W.log 2>&1 ;;
    lds) # what the wavefronts of the step kernel wait on: LDS activity / waits / bank conflicts, the instruction mix, per-class busy cycles
        rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_lds_$W -o $W -- $B --steps 100 --warmup 20 > gpurun_out/pmc_lds_$W.log 2>&1
        rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_SENDMSG SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc_act_$W -o $W -- $B --steps 100 --warmup 20 > gpurun_out/pmc_act_$W.log 2>&1 ;;
    mem) rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$W -o $W -- $B --steps 100 --warmup 20 > gpurun_out/pmc_fetch_$W.log 2>&1
         rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$W -o $W -- $B --steps 100 --warmup 20 > gpurun_out/pmc_write_$W.log 2>&1 ;;
  esac
done
