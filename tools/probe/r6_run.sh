mkdir -p gpurun_out/r6
bash tools/probe/ab_phases.sh "C2 C3 C5" "gym_pcgrl_amd/lib/libpcgrl_hip.so gym_pcgrl_amd/lib/libexp_ifcvt.so gym_pcgrl_amd/lib/libexp_wprio.so gym_pcgrl_amd/lib/libexp_kpre.so gym_pcgrl_amd/lib/libexp_noalign.so" 3 2>&1 | cut -c1-84 | tee gpurun_out/r6/ab_compiler_sched5.txt
