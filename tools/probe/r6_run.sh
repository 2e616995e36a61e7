mkdir -p gpurun_out/r6
bash tools/probe/ab_lazy.sh "C4 M1 D1" > gpurun_out/r6/ab_lazy_draws.txt 2>&1; cat gpurun_out/r6/ab_lazy_draws.txt
timeout 900 tools/profile_gpu.sh C4 trace sq mem
python tools/make_profile_summary.py r6_lazy C4 > /dev/null 2>&1
cat profiles/r6_lazy/C4_traffic.json; grep "k_update\|k_reset\|k_stats" profiles/r6_lazy/SUMMARY.md | head -8
mkdir -p gpurun_out/r6/lazy_prof; cp profiles/r6_lazy/* gpurun_out/r6/lazy_prof/
rm -rf gpurun_out/pmc_*/ gpurun_out/prof_*/
