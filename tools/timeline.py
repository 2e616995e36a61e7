#!/usr/bin/env python3
"""In-kernel timeline of one fused step (k_step) on the GPU box: builds a copy of the library with -DPCGRL_TIMELINE
(wavefront-private marks of the 100 MHz wall clock at the phase boundaries), runs the workload into its steady state and
prints where the time of a step goes: per-phase durations over the blocks and what the last blocks to finish were doing.
    python tools/timeline.py [workload] [envs] [warm-up steps]
(A mark reads the buffer pointer through the scalar cache: the first mark after an idle stretch can show ~2 us that are the mark's
own miss -- "lists ready" -> "ticket" of the waiting wavefronts in k_step is that, the product build has no such gap.)
"""
import ctypes as C, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
import _tuning_env; _tuning_env.apply()      # PCGRL_* environment variables -> the binding's tuning overrides (developer tools only)
# (`timeline.py build` on the CPU box: the library travels with the tree and the GPU call does not spend minutes compiling; A/B builds
#  with PCGRL_TL_FLAGS are compiled where they run)
so = os.path.join(ROOT, "tools", "probe", "libpcgrl_hip_tl.so") if not os.environ.get("PCGRL_TL_FLAGS") else "/tmp/libpcgrl_hip_tl.so"
if not os.path.exists(so) or os.environ.get("PCGRL_TL_FLAGS") or (len(sys.argv) > 1 and sys.argv[1] == "build"):
    subprocess.check_call([os.environ.get("HIPCC", "hipcc")] + _lib.HIPCC_FLAGS + ["-DPCGRL_TIMELINE"] + os.environ.get("PCGRL_TL_FLAGS", "").split() + _lib.SOURCES + ["-o", so])
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        sys.exit(0)
_lib.SO = so
import torch
import bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
prob, rep, calls, n_default, desc = bench.WORKLOADS[wl]
n = int(sys.argv[2]) if len(sys.argv) > 2 else n_default
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 800
env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, seed=0)
for kw in calls:
    env.adjust_param(**kw)
env.reset()
W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
acts = bench.make_actions(torch, rep, 256, n, W, H, nt, env.device, 1234)
for t in range(warm):
    env.step(acts[t % 256])
L = _lib.load()
L.pcgrl_debug_timeline.argtypes = [C.c_void_p]
EPB = int(os.environ.get("PCGRL_STEP_EPB", "256" if 192 * 256 <= n <= 256 * 256 else ("128" if n >= 192 * 128 else "64")))
SLOTS, WAVES = 48, EPB // 16
nblk = (n + EPB - 1) // EPB
names = {1: "start", 2: "update done", 3: "lists ready", 4: "task: certain reset", 5: "task: full", 6: "task: incremental", 7: "task end", 8: "wave end",
         9: "reset done", 10: "stats done", 11: "finalized", 12: "late reset", 13: "ring staged", 14: "map made", 15: "cursor drawn", 16: "reset stored", 17: "state staged", 18: "ticket"}
summ = []
for rep_i in range(5):
    buf = torch.zeros((nblk * WAVES * SLOTS,), dtype=torch.int64, device=env.device)
    _lib.check(L.pcgrl_debug_timeline(C.c_void_p(buf.data_ptr())), "timeline")
    torch.cuda.synchronize()
    env.step(acts[(warm + rep_i) % 256])
    torch.cuda.synchronize()
    _lib.check(L.pcgrl_debug_timeline(None), "timeline")
    a = buf.cpu().numpy().view(np.uint64).reshape(nblk, WAVES, SLOTS)
    tag = (a & np.uint64(255)).astype(np.int64)
    tm = (a >> np.uint64(8)).astype(np.float64) * 0.01          # us
    t0 = tm[tag == 1].min()
    tm = tm - t0
    start = np.where(tag == 1, tm, np.nan)
    end_w = np.nanmax(np.where(tag == 8, tm, np.nan), axis=2)          # [blk, wave]
    end_b = np.nanmax(end_w, axis=1)
    upd = np.nanmax(np.where(tag == 2, tm, np.nan), axis=(1, 2))
    lists = np.nanmin(np.where(tag == 3, tm, np.nan), axis=(1, 2))
    st_b = np.nanmin(start, axis=(1, 2))
    def pct(x):
        x = x[~np.isnan(x)]
        return "min %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (x.min(), np.percentile(x, 50), np.percentile(x, 90), np.percentile(x, 99), x.max())
    print("== step", rep_i, "blocks", nblk)
    print("block start      ", pct(st_b))
    print("update done      ", pct(upd), " (duration %s)" % pct(upd - st_b))
    print("lists ready      ", pct(lists))
    print("block end        ", pct(end_b))
    # task durations by kind
    for kind in (4, 5, 6):
        durs, resets = [], []
        for b, w in zip(*np.nonzero((tag == kind).any(axis=2))):
            tg, tt = tag[b, w], tm[b, w]
            for i in np.nonzero(tg == kind)[0]:
                j = i + 1
                while j < SLOTS and tg[j] not in (7, 0):
                    j += 1
                if j < SLOTS and tg[j] == 7:
                    durs.append(tt[j] - tt[i])
        if durs:
            print("%-20s n %5d  %s" % (names[kind], len(durs), pct(np.array(durs))))
    # inside the certain-reset tasks: time from the task's start to each mark of the chain (ring staged -> map made -> cursor drawn ->
    # reset stored -> planes (reset done) -> statistics done -> task end)
    chain = {13: [], 14: [], 15: [], 16: [], 9: [], 10: [], 7: []}
    fullc = {10: [], 7: []}
    for b, w in zip(*np.nonzero(((tag == 4) | (tag == 5)).any(axis=2))):
        tg, tt = tag[b, w], tm[b, w]
        for i in np.nonzero((tg == 4) | (tg == 5))[0]:
            dst = chain if tg[i] == 4 else fullc
            j = i + 1
            seen = set()
            while j < SLOTS and tg[j] != 0:
                g = int(tg[j])
                if g in dst and g not in seen:
                    dst[g].append(tt[j] - tt[i]); seen.add(g)
                if g == 7:
                    break
                j += 1
    print("certain reset chain (us from task start, p50 / p90 / max):", "  ".join(
        "%s %.1f/%.1f/%.1f" % (names[k], np.percentile(v, 50), np.percentile(v, 90), np.max(v)) for k, v in chain.items() if v))
    print("full task chain:", "  ".join("%s %.1f/%.1f/%.1f" % (names[k], np.percentile(v, 50), np.percentile(v, 90), np.max(v)) for k, v in fullc.items() if v))
    # the slowest blocks: their marks
    worst = np.argsort(end_b)[-3:]
    for b in worst:
        print("block", int(b), "end %.1f" % end_b[b])
        for w in range(WAVES):
            seq = ["%s@%.1f" % (names[int(g)].split(":")[-1].strip(), t) for g, t in zip(tag[b, w], tm[b, w]) if g]
            print("   wave", w, " ".join(seq))
    summ.append(float(end_b.max()))
print("kernel span (first start -> last end), us:", [round(x, 1) for x in summ])
