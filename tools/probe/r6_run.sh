mkdir -p gpurun_out/r6
bash tools/probe/ab_phases.sh "C2 C3 C5" "gym_pcgrl_amd/lib/libpcgrl_hip.so gym_pcgrl_amd/lib/libexp_iterilp.so gym_pcgrl_amd/lib/libexp_trackers.so gym_pcgrl_amd/lib/libexp_nopostra.so gym_pcgrl_amd/lib/libexp_aasched.so" 3 2>&1 | cut -c1-84 | tee gpurun_out/r6/ab_compiler_sched4.txt
