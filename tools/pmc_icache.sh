#!/bin/bash
# instruction-cache counters of one workload's kernels (GPU box): tools/pmc_icache.sh C2
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
W=${1:-C2}
mkdir -p gpurun_out
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d gpurun_out/pmc_ic_$W -o $W -- python bench.py --workload $W --no-cpu-baseline --no-rollout --steps 100 --warmup 20 --steady-warmup 0 > gpurun_out/pmc_ic_$W.log 2>&1
python - <<PY
import csv,glob,collections
for f in glob.glob("gpurun_out/pmc_ic_$W/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if "k_step" in k or "k_stats" in k or "k_smb" in k:
            print(k, {a:"%.4g"%(sum(b)/len(b)) for a,b in v.items()}, "rows", len(list(v.values())[0]))
PY
tail -3 gpurun_out/pmc_ic_$W.log | cut -c1-300
