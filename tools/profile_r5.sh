#!/bin/bash
# Round-5 profiling session on the GPU box (one gpurun call); results land in gpurun_out/.
#   tools/profile_r5.sh ; then: python tools/make_profile_summary.py r5_round5 C2 C3 C2w C3w C5 C4 C5b C3d M1 D1 S1 B1 K1 C2R C4R
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 700 tools/profile_gpu.sh C2 trace sq lds mem
timeout 600 tools/profile_gpu.sh C3 trace sq lds mem
for w in C2w C3w C5; do timeout 500 tools/profile_gpu.sh $w trace sq mem; done
timeout 900 tools/profile_gpu.sh C4 trace sq mem
for w in C5b C3d M1 D1 S1 B1 K1; do timeout 400 tools/profile_gpu.sh $w trace; done
timeout 300 tools/profile_gpu.sh C2 rtrace
timeout 400 tools/profile_gpu.sh C4 rtrace
for w in C2 C3 C3d C4 C5 C5b M1 D1 S1 C2w C3w B1 K1; do
  timeout 500 python bench.py --workload $w --no-legs > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  tail -c 200 gpurun_out/bench_$w.json; echo
done
( time timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default.time
# the asynchronous ticks: kernel traces + the phase profile of the search wavefronts
for w in C4 M1 D1; do tools/probe/trace_async.sh $w 64 > gpurun_out/trace_async_$w.txt 2>&1; done
python tools/probe/async_prof.py C4 64 > gpurun_out/async_prof_C4.txt 2>&1
python tools/probe/async_prof.py M1 64 > gpurun_out/async_prof_M1.txt 2>&1
python tools/timeline.py C2 > gpurun_out/timeline_C2.txt 2>&1
# cycle counters of the map path beyond 64 x 64 and of the smb searches (libraries built ahead on the CPU box: tools/probe/*.so)
python tools/probe/big_prof.py 50 > gpurun_out/big_prof_B1.txt 2>&1
python tools/smb_prof.py 16384 10 > gpurun_out/smb_prof_S1.txt 2>&1
python tools/probe/step_multi_cost.py > gpurun_out/step_multi_cost.txt 2>&1
# k_stats_wide block by block in the product's schedule (two blocks per compute unit), and what one BFS level costs a wavefront
python tools/probe/wide_blocks.py C5 800 3 2>&1 | grep -v amdgpu.ids > gpurun_out/wide_blocks_C5.txt
[ -x tools/probe/bfs_level_cost ] && tools/probe/bfs_level_cost > gpurun_out/bfs_level_cost.txt 2>&1
# gpurun copies at most 64 MiB back: the summaries are made here, the raw traces and counter files stay on the box
python tools/make_profile_summary.py r5_round5 C2 C3 C2w C3w C5 C4 C5b C3d M1 D1 S1 B1 K1 C2R C4R > /dev/null 2>&1
mkdir -p gpurun_out/r5_summary
cp -r profiles/r5_round5/* gpurun_out/r5_summary/
for w in C4 M1 D1; do rm -f gpurun_out/trace_async_$w/t_kernel_trace.csv gpurun_out/trace_async_$w/*agent_info.csv; done
rm -rf gpurun_out/prof_* gpurun_out/pmc_*
du -sh gpurun_out
tail -c 600 gpurun_out/bench_default.json; cat gpurun_out/bench_default.time
