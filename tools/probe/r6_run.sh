mkdir -p gpurun_out/r6
bash tools/probe/ab_phases.sh "C2w" "gym_pcgrl_amd/lib/libpcgrl_hip.so gym_pcgrl_amd/lib/libexp_obs4t.so" 3 > gpurun_out/r6/ab_obs_tasks4.txt 2>&1; cat gpurun_out/r6/ab_obs_tasks4.txt
