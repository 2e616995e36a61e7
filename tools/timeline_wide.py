#!/usr/bin/env python3
"""In-kernel timeline of k_stats_wide (tall binary maps, C5) on the GPU box: a copy of the library with -DPCGRL_TIMELINE, the
workload run into its steady state, then per wavefront the marks of one step: where the time of the longest items goes.
    python tools/timeline_wide.py [workload=C5] [warm-up steps=800]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
import _tuning_env; _tuning_env.apply()      # PCGRL_* environment variables -> the binding's tuning overrides (developer tools only)
so = "/tmp/libpcgrl_hip_tl.so"
subprocess.check_call([os.environ.get("HIPCC", "hipcc")] + _lib.HIPCC_FLAGS + ["-DPCGRL_TIMELINE"] + _lib.SOURCES + ["-o", so], stderr=subprocess.DEVNULL)
_lib.SO = so
import torch
import bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
wl = sys.argv[1] if len(sys.argv) > 1 else "C5"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 800
prob, rep, calls, n, desc = bench.WORKLOADS[wl]
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
SLOTS, WAVES, nblk = 48, 8, min(n, 16384)
names = {1: "start", 20: "stats>", 21: "tiny", 22: "seeds done", 23: "synced", 24: "lone>", 25: "map made", 26: "planes", 27: "items done", 28: "end",
         30: "comp", 31: "sweep>", 32: "sweep<"}
for rep_i in range(3):
    buf = torch.zeros((nblk * WAVES * SLOTS,), dtype=torch.int64, device=env.device)
    _lib.check(L.pcgrl_debug_timeline(C.c_void_p(buf.data_ptr())), "timeline")
    torch.cuda.synchronize()
    env.step(acts[(warm + rep_i) % 256])
    torch.cuda.synchronize()
    _lib.check(L.pcgrl_debug_timeline(None), "timeline")
    a = buf.cpu().numpy().view(np.uint64).reshape(nblk, WAVES, SLOTS)
    tag = (a & np.uint64(255)).astype(np.int64)
    tm = (a >> np.uint64(8)).astype(np.float64) * 0.01
    used = (tag == 1).any(axis=2).any(axis=1)
    t0 = tm[tag == 1].min()
    tm = tm - t0
    end_b = np.where(tag == 28, tm, np.nan)
    end_b = np.nanmax(end_b.reshape(nblk, -1), axis=1)
    items = np.where(tag == 27, tm, np.nan)
    items = np.nanmax(items.reshape(nblk, -1), axis=1)
    print("== step %d: blocks that ran %d; kernel span %.1f us; 'items done' p50 %.1f p99 %.1f max %.1f" % (
        rep_i, int(used.sum()), np.nanmax(end_b), np.nanpercentile(items, 50), np.nanpercentile(items, 99), np.nanmax(items)))
    ncomp = (tag == 30).sum(axis=2)
    nsw = (tag == 31).sum(axis=2)
    print("   components per wavefront (blocks with any): mean %.1f max %d; sweeps per wavefront mean %.2f max %d" % (
        ncomp[ncomp.sum(axis=1) > 0].mean(), ncomp.max(), nsw[ncomp.sum(axis=1) > 0].mean(), nsw.max()))
    busy = (tag == 20).any(axis=2).any(axis=1)                  # blocks that computed statistics of a map
    worst = np.argsort(np.where(busy, np.nan_to_num(items), -1.0))[-2:]
    lone = (tag == 24).any(axis=2).any(axis=1)
    for nm, sel in (("reset items", lone), ("full items", busy & ~lone)):
        if sel.any():
            st = np.nanmin(np.where(tag == 20, tm, np.nan).reshape(nblk, -1), axis=1)
            en = np.nanmax(np.where(tag == 23, tm, np.nan).reshape(nblk, -1), axis=1)
            d = (en - st)[sel]
            print("   %-12s n %4d  statistics phase: p50 %.1f  p90 %.1f  max %.1f us; ends: p50 %.1f max %.1f" % (nm, int(sel.sum()), np.nanpercentile(d, 50), np.nanpercentile(d, 90), np.nanmax(d), np.nanpercentile(items[sel], 50), np.nanmax(items[sel])))
    for b in worst:
        print("block", int(b), "items done %.1f" % items[b])
        for w in range(WAVES):
            seq = ["%s@%.1f" % (names.get(int(g), str(int(g))), t) for g, t in zip(tag[b, w], tm[b, w]) if g]
            print("   wave", w, " ".join(seq))
