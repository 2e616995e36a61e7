#!/bin/bash
# kernel trace of the asynchronous ticks: tools/probe/trace_async.sh <workload> <budget>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
W=${1:-C4}; Bd=${2:-128}
O=gpurun_out/trace_async_$W
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python tools/probe/async_bench.py $W $Bd > $O/log.txt 2>&1
python - <<PY
import csv
rows = [r for r in csv.DictReader(open("$O/t_kernel_trace.csv"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0][:44], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
idx = [i for i, k in enumerate(ks) if k[0].startswith("void k_update")]
for i0 in idx[300:302]:
    t0 = ks[i0][1]
    for k in ks[i0:i0 + 8]:
        print("%-46s start %7.1f dur %7.1f" % (k[0], (k[1] - t0) / 1e3, (k[2] - k[1]) / 1e3))
    print()
PY
tail -2 $O/log.txt
