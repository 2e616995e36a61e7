set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 300 python tools/probe/big_prof.py 50) > gpurun_out/r5b/big_prof5.txt 2>&1
cd /tmp && export TMPDIR=/tmp && (timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -o b1 -- python /root/repo/bench.py --workload B1 --steps 50 --warmup 10 --no-cpu-baseline --no-legs --steady-warmup 0 --no-rollout) > /root/repo/gpurun_out/r5b/rocprof_b1.txt 2>&1; find /tmp/prof_b1 -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r5b/B1_kernel_stats.csv \;
cd /root/repo
tail -n 8 gpurun_out/r5b/big_prof5.txt; head -8 gpurun_out/r5b/B1_kernel_stats.csv
