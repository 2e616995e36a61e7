#!/usr/bin/env python3
"""Where do the microseconds of k_stats_wide (C5: binary-turtle 64x64) go, block by block, in the PRODUCT's schedule?  A copy of the library
with -DPCGRL_WIDE_TL: thread 0 of every block writes five wall-clock words (entry, lists read, item kind, item done, end) -- no extra
registers, so the kernel keeps its two blocks per compute unit (tools/timeline_wide.py's full marks cost it that: its picture of the
block starts is not the product's).
    python tools/probe/wide_blocks.py build            # on the CPU box: tools/probe/libpcgrl_hip_widetl.so (travels with the tree)
    python tools/probe/wide_blocks.py [workload=C5] [warm-up steps=800] [steps shown=3] [extra -D flags for an A/B build ...]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gym_pcgrl_amd import _lib
import _tuning_env; _tuning_env.apply()


def build(so, extra):
    # one translation unit (no -DPCGRL_PART): the marks' buffer pointer is a device variable, and every part of the split build would
    # have its own
    subprocess.check_call(["hipcc"] + _lib.HIPCC_FLAGS + ["-DPCGRL_WIDE_TL"] + extra + _lib.SOURCES + ["-o", so], stderr=subprocess.DEVNULL)


args = sys.argv[1:]
extra = [a for a in args if a.startswith("-D")]
args = [a for a in args if not a.startswith("-D")]
so = os.path.join(ROOT, "tools", "probe", "libpcgrl_hip_widetl%s.so" % "".join(f.replace("-D", "_") for f in extra))
if args and args[0] == "build":
    build(so, extra)
    sys.exit(0)
if not os.path.exists(so):
    build(so, extra)
_lib.SO = so
import torch
import bench
from gym_pcgrl_amd.envs import BatchedPcgrlEnv
wl = args[0] if args else "C5"
warm = int(args[1]) if len(args) > 1 else 800
shown = int(args[2]) if len(args) > 2 else 3
prob, rep, calls, n, desc = bench.WORKLOADS[wl]
env = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=n, seed=0)
for kw in calls:
    env.adjust_param(**kw)
env.reset()
W, H, nt = env._prob._width, env._prob._height, env.get_num_tiles()
acts = bench.make_actions(torch, rep, 284, n, W, H, nt, env.device, 1234)
for t in range(warm):
    env.step(acts[t % 284])
L = _lib.load()
L.pcgrl_debug_timeline.argtypes = [C.c_void_p]
NB = 8192
KIND = {1: "reset, old-map half", 2: "reset, new map", 3: "full item (many regions)", 4: "full items, two to a block (few regions)"}
print("%s after %d steps; us from the first block's entry (100 MHz wall clock)" % (desc, warm))
for i in range(shown):
    buf = torch.zeros((NB * 8,), dtype=torch.int64, device=env.device)
    _lib.check(L.pcgrl_debug_timeline(C.c_void_p(buf.data_ptr())), "timeline")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.step(acts[(warm + i) % 284]); e1.record()
    torch.cuda.synchronize()
    _lib.check(L.pcgrl_debug_timeline(None), "timeline")
    a = buf.cpu().numpy().view(np.uint64).reshape(NB, 8)
    ran = a[:, 0] != 0
    t0 = a[ran, 0].min()
    us = lambda col: (a[:, col].astype(np.float64) - float(t0)) * 0.01
    entry, lists, done, end = us(0), us(1), us(3), us(4)
    kind = (a[:, 2] & np.uint64(255)).astype(int)
    hdr = int(a[0, 5])
    n_items, n_rst, n_many, n_inc = hdr & 0xFFFF, (hdr >> 16) & 0xFFFF, (hdr >> 32) & 0xFFFF, hdr >> 48
    print("== step %d: %.1f us between events; blocks that ran %d; items %d (certain resets %d, many-region full items %d, paired blocks %d), incremental items %d; kernel span %.1f us"
          % (i, e0.elapsed_time(e1) * 1e3, ran.sum(), n_items, n_rst, n_many, n_items - 2 * n_rst - n_many, n_inc, end[ran].max()))
    pc = lambda v, q: np.percentile(v, q) if len(v) else float("nan")
    for k in (1, 2, 3, 4):
        m = ran & (kind == k)
        if not m.any():
            continue
        d = done[m] - lists[m]
        print("   %-42s n %4d  entry p50 %5.1f max %5.1f | lists read +%4.1f | item p50 %5.1f p90 %5.1f p99 %5.1f max %5.1f | done p50 %5.1f max %5.1f | end max %5.1f"
              % (KIND[k], m.sum(), pc(entry[m], 50), entry[m].max(), pc(lists[m] - entry[m], 50), pc(d, 50), pc(d, 90), pc(d, 99), d.max(), pc(done[m], 50), done[m].max(), end[m].max()))
    m = ran & (kind == 0)
    if m.any():
        print("   %-42s n %4d  entry p50 %5.1f max %5.1f | lists read +%4.1f | incremental items p50 %5.1f max %5.1f | end p50 %5.1f max %5.1f"
              % ("blocks without a full item", m.sum(), pc(entry[m], 50), entry[m].max(), pc(lists[m] - entry[m], 50), pc(end[m] - done[m], 50), (end[m] - done[m]).max(), pc(end[m], 50), end[m].max()))
    # the ten blocks that end last
    order = np.argsort(-np.where(ran, end, -1))[:10]
    for b in order:
        print("      block %4d %-28s entry %5.1f lists %5.1f item done %5.1f end %5.1f" % (b, KIND.get(kind[b], "-")[:28], entry[b], lists[b], done[b], end[b]))
