mkdir -p gpurun_out/r6
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6/t17.log 2>&1; tail -4 gpurun_out/r6/t17.log
