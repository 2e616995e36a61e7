#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of the last gpurun into profiles/<name>/ and write SUMMARY.md.
    python tools/make_profile_summary.py <name> C2 [C3 ...]"""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
name, workloads = sys.argv[1], sys.argv[2:]
dst = os.path.join(ROOT, "profiles", name)
os.makedirs(dst, exist_ok=True)
def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d
out = ["# %s (MI355X, rocprofv3)\n" % name,
       "Commands: `tools/profile_gpu.sh <workload> trace sq mem` = `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --workload W --steps 100 --no-cpu-baseline --no-rollout`,",
       "then separate `--pmc` passes (SQ_* counters; FETCH_SIZE; WRITE_SIZE).  Bench lines: `bench_<W>.json`; the default line as the driver runs it: "
       "`bench_default.json` (+ `.time`); cycles per pop of the two wavefronts of a capped A* search: `sok_prof.txt`, `md_prof_duo.txt` (`tools/sok_prof.py 4000`, "
       "`tools/sok_prof.py 1000 mdungeon`); the image kernel's store stream: `obs_bench.txt`.\n"]
for W in workloads:
    out.append("## %s\n" % W)
    if W.endswith("R"):
        out.append("`bench.py --workload %s --steps 200` *with* its pcgrl_rollout leg: the two calls of `k_step<..., true>` / `k_step_solver` are the 20-step "
                   "warm-up tape and the timed 200-step tape (max us / 200 = the `rollout.ms_per_step` of the bench line).\n" % W[:-1])
    ks = os.path.join(G, "prof_" + W, W + "_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(dst, W + "_kernel_stats.csv"))
        out.append("kernel-trace --stats (`%s_kernel_stats.csv`):\n\n| kernel | calls | avg us | min us | max us | %% |\n|---|---|---|---|---|---|" % W)
        for r in csv.DictReader(open(ks)):
            if r["Name"].startswith(("void k_", "k_")):
                out.append("| %s | %s | %.2f | %.2f | %.2f | %s |" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                    float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
        out.append("")
    sq = agg(os.path.join(G, "pmc_sq_" + W, W + "_counter_collection.csv"))
    sq2 = agg(os.path.join(G, "pmc_sq2_" + W, W + "_counter_collection.csv"))
    f = agg(os.path.join(G, "pmc_fetch_" + W, W + "_counter_collection.csv"))
    w = agg(os.path.join(G, "pmc_write_" + W, W + "_counter_collection.csv"))
    if sq:
        cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]
        out.append("PMC, average per dispatch:\n\n| kernel | " + " | ".join(cols) + " | SQ_WAIT_ANY | FETCH_SIZE KB | WRITE_SIZE KB |\n|" + "---|" * (len(cols) + 4))
        for k, v in sq.items():
            if not k.startswith(("void k_", "k_")) or len(v[cols[0]]) < 4:
                continue
            m = lambda d, c: (sum(d[k][c]) / len(d[k][c])) if k in d and c in d[k] else float("nan")
            out.append("| %s | " % k + " | ".join("%d" % m(sq, c) for c in cols) + " | %d | %.0f | %.0f |" % (m(sq2, "SQ_WAIT_ANY"), m(f, "FETCH_SIZE"), m(w, "WRITE_SIZE")))
        out.append("")
        lds, act = agg(os.path.join(G, "pmc_lds_" + W, W + "_counter_collection.csv")), agg(os.path.join(G, "pmc_act_" + W, W + "_counter_collection.csv"))
        extra = {}
        if lds or act:
            lcols = ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_INSTS_FLAT", "SQ_INSTS_BRANCH",
                     "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC", "SQ_INSTS_SENDMSG", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"]
            out.append("What the wavefronts wait on -- LDS and instruction-class counters (separate `--pmc` passes, `tools/profile_gpu.sh %s lds`), average per dispatch:\n\n| kernel | " % W +
                       " | ".join(lcols) + " |\n|" + "---|" * (len(lcols) + 1))
            for k in sorted(set(lds) | set(act)):
                if not k.startswith(("void k_", "k_")):
                    continue
                both = dict(lds.get(k, {})); both.update(act.get(k, {}))
                if not both or len(next(iter(both.values()))) < 4:
                    continue
                vals = {c: sum(both[c]) / len(both[c]) for c in lcols if c in both}
                extra[k.replace("void ", "").split("<")[0]] = vals
                out.append("| %s | " % k + " | ".join(("%d" % vals[c]) if c in vals else "-" for c in lcols) + " |")
            out.append("")
        # (the kernels of a lockstep step: the asynchronous ticks of the bench line's `async` leg -- k_search_async, not part of pcgrl_step --
        #  were counted in until round 5: C4's 207.9 MB of that round is 186.1 MB by this sum)
        in_step = lambda k: k.startswith("void k_") and not k.startswith("void k_search_async")
        tot_f = sum(sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) for k, v in f.items() if in_step(k) and len(v["FETCH_SIZE"]) > 4)
        tot_w = sum(sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) for k, v in w.items() if in_step(k) and len(v["WRITE_SIZE"]) > 4)
        out.append("HBM traffic per step (sum over the lockstep step's kernels; rocprofv3 KB): FETCH_SIZE %.1f MB + WRITE_SIZE %.1f MB as reported.  Calibrated "
                   "on this GPU (`traffic_calibration.md`, `tools/traffic_calib.hip`): FETCH_SIZE is half of the bytes of the 128-byte lines read -- for wide "
                   "coalesced and narrow scattered reads alike -- and WRITE_SIZE is exact for coalesced writes and counts a 32-byte sector per scattered "
                   "narrow write, so the step moves **%.1f MB** (2 x fetch + write; what `roofline.traffic` reports).\n" % (tot_f * 1024 / 1e6, tot_w * 1024 / 1e6, (2 * tot_f + tot_w) * 1024 / 1e6))
        hp = os.path.join(G, "srchash_%s.txt" % W)
        src_hash = open(hp).read().strip() if os.path.exists(hp) else None      # _lib.source_hash() of the tree the passes ran on
        json.dump({"workload": W, "fetch_bytes_per_step": tot_f * 1024, "write_bytes_per_step": tot_w * 1024, "csrc_hash": src_hash},
                  open(os.path.join(dst, W + "_traffic.json"), "w"))
        # per-kernel counter averages, read back by bench.py for the VALU-issue figure of the dominant kernel
        pm = {}
        for k, v in sq.items():
            if k.startswith(("void k_", "k_")) and len(v[cols[0]]) >= 4:
                kk = k.replace("void ", "").split("<")[0]
                pm[kk] = {c: sum(v[c]) / len(v[c]) for c in cols if c in v}
                if k in sq2 and "SQ_WAIT_ANY" in sq2[k]:
                    pm[kk]["SQ_WAIT_ANY"] = sum(sq2[k]["SQ_WAIT_ANY"]) / len(sq2[k]["SQ_WAIT_ANY"])
                pm[kk].update(extra.get(kk, {}))
        json.dump({"workload": W, "per_dispatch": pm, "csrc_hash": src_hash}, open(os.path.join(dst, W + "_pmc.json"), "w"))
    for cand in ("bench_%s.json" % W, "bench_%s.log" % W):
        b = os.path.join(G, cand)
        if os.path.exists(b) and os.path.getsize(b) > 10:
            line = open(b).read().strip().splitlines()[-1]
            try:
                d = json.loads(line)
            except ValueError:
                continue
            open(os.path.join(dst, "bench_%s.json" % W), "w").write(line + "\n")
            out.append("bench: **%.1f M env-steps/s**, %.1f us/step, roofline achieved %.0f GB/s (frac %.4f)%s%s\n" % (
                d["value"] / 1e6, d["ms_per_step"] * 1e3, d["roofline"]["achieved"], d["roofline"]["frac"],
                (", as one pcgrl_rollout call %.1f M env-steps/s (%.1f us/step)" % (d["rollout"]["value"] / 1e6, d["rollout"]["ms_per_step"] * 1e3)) if "rollout" in d else "",
                (", cpu_baseline %.2f M/s on %d threads (%.0f k/s single)" % (d["cpu_baseline"]["value"] / 1e6, d["cpu_baseline"]["cores"], d["cpu_baseline"]["single_core"] / 1e3)) if "cpu_baseline" in d else ""))
            break
cal = os.path.join(G, "traffic_calib.txt")
if os.path.exists(cal):
    open(os.path.join(dst, "traffic_calibration.md"), "w").write(
        "# FETCH_SIZE / WRITE_SIZE calibration (MI355X, rocprofv3, `tools/calibrate_traffic.sh`)\n\n"
        "Known access patterns over a 1 GiB buffer (beyond the 256 MiB Infinity Cache), third repetition:\n\n" + open(cal).read() +
        "\nReading: FETCH_SIZE = half of the bytes of the 128-byte lines touched, whatever the access width or order; WRITE_SIZE = bytes "
        "written for full-line coalesced stores, 32 bytes per store for isolated narrow stores.\n")
open(os.path.join(dst, "SUMMARY.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out)[:3000])
