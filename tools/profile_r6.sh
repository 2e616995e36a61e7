#!/bin/bash
# Round-6 profiling session on the GPU box (one gpurun call); results land in gpurun_out/ (summaries: gpurun_out/r6_summary/).
#   tools/profile_r6.sh ; then copy gpurun_out/r6_summary/* into profiles/r6_round6/
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 700 tools/profile_gpu.sh C2 trace sq mem
timeout 600 tools/profile_gpu.sh C3 trace sq mem
for w in C2w C3w C5; do timeout 500 tools/profile_gpu.sh $w trace sq mem; done
timeout 900 tools/profile_gpu.sh C4 trace sq mem
for w in C5b C3d M1 D1 S1 B1 K1; do timeout 400 tools/profile_gpu.sh $w trace; done
timeout 300 tools/profile_gpu.sh C2 rtrace
for w in C2 C3 C3d C4 C5 C5b M1 D1 S1 C2w C3w B1 K1; do
  timeout 500 python bench.py --workload $w --no-legs > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  tail -c 200 gpurun_out/bench_$w.json; echo
done
( time timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default.time
# the wrapped steps with every image at the end of the launch (the form up to round 5) next to the default, same box
bash tools/probe/ab_tuning.sh "C2w C3w" "obs_at_end=1 obs_at_end=0" 2 > gpurun_out/ab_obs_in_flight.txt 2>&1
# sub-batches of one GPU
for w in C2 C3 C5; do python tools/probe/sub_batches.py $w 1 2 4 2>&1 | grep -v amdgpu.ids; done > gpurun_out/sub_batches.txt 2>&1
python tools/timeline.py C2 2>&1 | grep -v amdgpu.ids | cut -c1-1200 | head -60 > gpurun_out/timeline_C2.txt
python tools/timeline.py C3 2>&1 | grep -v amdgpu.ids | cut -c1-1200 | head -60 > gpurun_out/timeline_C3.txt
PCGRL_PROF_SO=gym_pcgrl_amd/lib/libexp_sokprof.so python tools/sok_prof.py 4000 2>&1 | grep -v amdgpu.ids > gpurun_out/sok_prof.txt
python tools/probe/step_multi_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/step_multi_cost.txt
# gpurun copies at most 64 MiB back: the summaries are made here, the raw traces and counter files stay on the box
python tools/make_profile_summary.py r6_round6 C2 C3 C2w C3w C5 C4 C5b C3d M1 D1 S1 B1 K1 C2R > /dev/null 2>&1
mkdir -p gpurun_out/r6_summary
cp profiles/r6_round6/*.csv profiles/r6_round6/*.json profiles/r6_round6/*.md gpurun_out/r6_summary/ 2>/dev/null
for f in bench_*.json bench_default.time ab_obs_in_flight.txt sub_batches.txt timeline_C2.txt timeline_C3.txt sok_prof.txt step_multi_cost.txt srchash_*.txt; do cp gpurun_out/$f gpurun_out/r6_summary/ 2>/dev/null; done
rm -rf gpurun_out/prof_*/ gpurun_out/pmc_*/
du -sh gpurun_out
tail -c 600 gpurun_out/bench_default.json; cat gpurun_out/bench_default.time
