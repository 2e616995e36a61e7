mkdir -p gpurun_out/r6
bash tools/probe/ab_tuning.sh "C3" "step_prio=0 step_prio=4 step_prio=16 step_prio=20 step_prio=48 step_prio=1" 3 2>&1 | tee gpurun_out/r6/ab_step_prio_zelda.txt
