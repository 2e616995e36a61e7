mkdir -p gpurun_out/r6
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6/t18.log 2>&1; tail -3 gpurun_out/r6/t18.log
bash tools/profile_r6.sh > gpurun_out/r6/profile_r6.log 2>&1; tail -3 gpurun_out/r6/profile_r6.log
