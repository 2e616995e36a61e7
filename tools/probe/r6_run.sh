mkdir -p gpurun_out/r6/fuzz_final
timeout 900 python tools/fuzz_parity.py 90 6101 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/fuzz_final/fuzz_all_6101.txt; tail -2 gpurun_out/r6/fuzz_final/fuzz_all_6101.txt
timeout 500 python tools/fuzz_parity.py 40 6102 - goal 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/fuzz_final/fuzz_goal_6102.txt; tail -2 gpurun_out/r6/fuzz_final/fuzz_goal_6102.txt
timeout 500 python tools/fuzz_parity.py 30 6103 sokoban 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/fuzz_final/fuzz_sokoban_6103.txt; tail -2 gpurun_out/r6/fuzz_final/fuzz_sokoban_6103.txt
