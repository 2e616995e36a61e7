set -x
cd /root/repo
mkdir -p gpurun_out/r5b
(timeout 1500 python -m pytest tests -x -q -m gpu) > gpurun_out/r5b/pytest_final.txt 2>&1
(timeout 700 python tools/fuzz_parity.py 60 515 binary big) > gpurun_out/r5b/fuzz_big_binary2.txt 2>&1
bash tools/profile_r5.sh > gpurun_out/r5b/profile_session.log 2>&1
tail -n 4 gpurun_out/r5b/pytest_final.txt gpurun_out/r5b/fuzz_big_binary2.txt; tail -n 3 gpurun_out/r5b/profile_session.log | cut -c1-300
