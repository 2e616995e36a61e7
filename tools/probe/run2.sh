#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/probe2
O=gpurun_out/probe2
timeout 900 python -m pytest tests/test_gpu_async.py -q 2>&1 | tail -25
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_async -o C4a -- python tools/probe/async_bench.py C4 128 > $O/trace_async.log 2>&1
python - <<'PY'
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/probe2/trace_async/C4a_kernel_trace.csv")):
    d[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[len(v)//3:]
    print("%-62s n=%5d avg %.1f us (last two thirds %.1f, max %.1f)" % (k, len(v), sum(v)/len(v), sum(v2)/len(v2), max(v)))
PY
timeout 300 python tools/probe/async_bench.py M1 128 256 2>&1 | tail -3
timeout 300 python tools/probe/async_bench.py D1 128 256 2>&1 | tail -3
