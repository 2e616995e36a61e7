mkdir -p gpurun_out/r6
bash tools/probe/ab_phases.sh "C2 C3 C5" "gym_pcgrl_amd/lib/libpcgrl_hip.so gym_pcgrl_amd/lib/libexp_os.so gym_pcgrl_amd/lib/libexp_nounroll.so gym_pcgrl_amd/lib/libexp_o2.so" 3 2>&1 | cut -c1-84 | tee gpurun_out/r6/ab_compiler_size.txt
