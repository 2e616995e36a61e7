#!/bin/bash
# Full profiling session of a round on the GPU box (one gpurun call); results land in gpurun_out/.
#   tools/profile_all.sh ; then: python tools/make_profile_summary.py <name> C2 C3 C2w C3w C5 C4 C3d C5b M1 D1 S1 C2R C4R
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 tools/profile_gpu.sh C2 trace sq mem
timeout 400 tools/profile_gpu.sh C3 trace sq mem
timeout 400 tools/profile_gpu.sh C2w trace sq mem
timeout 400 tools/profile_gpu.sh C3w trace sq mem
timeout 400 tools/profile_gpu.sh C5 trace sq mem
timeout 300 tools/profile_gpu.sh C5b trace
timeout 300 tools/profile_gpu.sh C3d trace
timeout 900 tools/profile_gpu.sh C4 trace sq mem
timeout 300 tools/profile_gpu.sh C2 rtrace
timeout 400 tools/profile_gpu.sh C4 rtrace
timeout 400 tools/profile_gpu.sh M1 trace
timeout 400 tools/profile_gpu.sh D1 trace
timeout 400 tools/profile_gpu.sh S1 trace
timeout 300 tools/profile_gpu.sh B1 trace
timeout 300 tools/profile_gpu.sh K1 trace
for w in C2 C3 C3d C4 C5 C5b M1 D1 S1 C2w C3w B1 K1; do
  timeout 500 python bench.py --workload $w --no-legs > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  tail -c 300 gpurun_out/bench_$w.json
done
# the default line as the driver runs it (every config as a leg), and the store-stream microbenchmark of the image kernel
( time timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default.time
timeout 120 python tools/obs_bench.py > gpurun_out/obs_bench.txt 2>&1
