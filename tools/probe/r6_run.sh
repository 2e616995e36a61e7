mkdir -p gpurun_out/r6
bash tools/probe/ab_tuning.sh "C2" "step_prio=207 step_prio=223 step_prio=1231 step_prio=203 step_prio=206 step_prio=239" 3 2>&1 | tee gpurun_out/r6/ab_step_prio_around207.txt
