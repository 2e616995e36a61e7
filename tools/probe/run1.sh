#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/probe1
O=gpurun_out/probe1
python tools/probe/stream_concurrency.py > $O/conc_default.txt 2>&1
for q in 8 16 32 64; do GPU_MAX_HW_QUEUES=$q python tools/probe/stream_concurrency.py > $O/conc_q$q.txt 2>&1; done
python tools/probe/async_slices.py C4 1 8 32 > $O/slices_default.txt 2>&1
GPU_MAX_HW_QUEUES=32 python tools/probe/async_slices.py C4 8 32 64 > $O/slices_q32.txt 2>&1
rocprofv3 -L > $O/counters.txt 2>&1
B="python bench.py --no-cpu-baseline --no-rollout --no-legs --steps 100 --warmup 20 --steady-warmup 0"
for w in C2 C3; do
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_lds_$w -o $w -- $B --workload $w > $O/pmc_lds_$w.log 2>&1
  rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_mix_$w -o $w -- $B --workload $w > $O/pmc_mix_$w.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/pmc_act_$w -o $w -- $B --workload $w > $O/pmc_act_$w.log 2>&1
done
ls -R $O | head -50
tail -5 $O/*.txt | head -150
