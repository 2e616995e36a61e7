set -x
mkdir -p gpurun_out/r5b
cd /root/repo
(timeout 60 tools/probe/dpp_cost) > gpurun_out/r5b/dpp_cost.txt 2>&1
(timeout 300 python tools/probe/step_multi_cost.py) > gpurun_out/r5b/step_multi_cost.txt 2>&1
(timeout 300 python tools/probe/big_prof.py 50) > gpurun_out/r5b/big_prof.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_async.py -x -q -k "smb_kept or node_driver_and_a_checkpoint or one_call_node") > gpurun_out/r5b/pytest_new.txt 2>&1
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "smb") > gpurun_out/r5b/pytest_smb.txt 2>&1
( time PCGRL_LIVENESS_REPEATS=300 timeout 600 python -m pytest tests/test_gpu_liveness.py -x -q -k "traj_binary_narrow" ) > gpurun_out/r5b/liveness_rate.txt 2>&1
(timeout 600 python bench.py --workload S1 --steps 20 --warmup 5 --no-cpu-baseline --no-legs --steady-warmup 45) > gpurun_out/r5b/bench_S1.json 2> gpurun_out/r5b/bench_S1.err
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > gpurun_out/r5b/smoke.txt 2>&1
tail -n 12 gpurun_out/r5b/*.txt; cat gpurun_out/r5b/bench_S1.json | head -c 1500
