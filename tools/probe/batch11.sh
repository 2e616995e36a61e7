set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 1200 python tools/fuzz_parity.py 160 4242 binary big) > gpurun_out/r5b/fuzz_big_binary.txt 2>&1
(timeout 900 python tools/fuzz_parity.py 80 99 - big) > gpurun_out/r5b/fuzz_big_all.txt 2>&1
(timeout 600 python tools/fuzz_parity.py 60 7 smb) > gpurun_out/r5b/fuzz_smb.txt 2>&1
tail -n 3 gpurun_out/r5b/fuzz_big_binary.txt gpurun_out/r5b/fuzz_big_all.txt gpurun_out/r5b/fuzz_smb.txt
