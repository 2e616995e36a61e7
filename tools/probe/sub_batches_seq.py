#!/usr/bin/env python3
"""GPU box: bench.sub_batch_leg several times in one process (does the order / the process history matter?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _tuning_env; print("tuning", _tuning_env.apply())
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for wl in sys.argv[1:]:
    w, k = wl.split(":")
    r = bench.sub_batch_leg(torch, dev, w, int(k))
    print("%s K=%s first %.2f us steady %.2f us (host %.2f)" % (w, k, r["ms_per_step"] * 1e3, r["steady_state"]["ms_per_step"] * 1e3, r["host_issue_ms_per_step"] * 1e3), flush=True)
