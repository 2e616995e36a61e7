mkdir -p gpurun_out/r6
bash tools/probe/ab_phases.sh "C2 C3 C3w" "gym_pcgrl_amd/lib/libpcgrl_hip.so gym_pcgrl_amd/lib/libexp_w44.so gym_pcgrl_amd/lib/libexp_w45.so" 3 2>&1 | cut -c1-84 | tee gpurun_out/r6/ab_waves_per_eu.txt
