mkdir -p gpurun_out/r6/fuzz_final2
timeout 900 python tools/fuzz_parity.py 120 6301 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/fuzz_final2/fuzz_all_6301.txt; tail -1 gpurun_out/r6/fuzz_final2/fuzz_all_6301.txt
timeout 600 python tools/fuzz_parity.py 40 6302 - goal 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/fuzz_final2/fuzz_goal_6302.txt; tail -1 gpurun_out/r6/fuzz_final2/fuzz_goal_6302.txt
timeout 900 python tools/fuzz_parity.py 30 6303 - big 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/fuzz_final2/fuzz_big_6303.txt; tail -1 gpurun_out/r6/fuzz_final2/fuzz_big_6303.txt
PCGRL_LIVENESS_REPEATS=150 timeout 1500 python -m pytest tests/test_gpu_liveness.py -x -q 2>&1 | tail -2
