#!/usr/bin/env python3
"""Summarise rocprofv3 CSVs under gpurun_out/ for one workload: per-kernel averages."""
import collections, csv, os, sys
W = sys.argv[1] if len(sys.argv) > 1 else "C2"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path): return d
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d
for sub in ("pmc_sq_", "pmc_sq2_", "pmc_lds_", "pmc_act_", "pmc_fetch_", "pmc_write_"):
    d = agg(os.path.join(root, sub + W, W + "_counter_collection.csv"))
    for k, v in d.items():
        if k.startswith(("void k_", "k_")) and len(next(iter(v.values()))) > 3:
            print(sub, k[:34], {c: round(sum(x) / len(x)) for c, x in v.items()}, "n=%d" % len(next(iter(v.values()))))
tr = os.path.join(root, "prof_" + W, W + "_kernel_trace.csv")
if os.path.exists(tr):
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(tr)))
    ks = [k for k in ks if k[2].startswith(("void k_", "k_"))][-400:]
    per = collections.defaultdict(list)
    seq = collections.defaultdict(int)
    for s, e, n in ks:
        per[n].append(e - s)
    for n, v in per.items():
        a, b = v[0::2], v[1::2]
        print("trace", n[:40], "avg us %.1f" % (sum(v) / len(v) / 1e3), "(alternating: %.1f / %.1f)" % (sum(a) / max(len(a), 1) / 1e3, sum(b) / max(len(b), 1) / 1e3), "n=%d" % len(v))
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    print("trace mean gap us %.2f" % (sum(gaps) / len(gaps) / 1e3))
