mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_liveness.py -x -q -k "round6 or tall or C5 or turtle_64 or split or 64x64 or wide or liveness" > gpurun_out/r6/t7.log 2>&1; tail -4 gpurun_out/r6/t7.log
bash tools/probe/ab_phases.sh "C5 C5b" "gym_pcgrl_amd/lib/libexp_head.so gym_pcgrl_amd/lib/libpcgrl_hip.so" 3 > gpurun_out/r6/ab_c5_waves16.txt 2>&1
cat gpurun_out/r6/ab_c5_waves16.txt
