#!/usr/bin/env python3
"""Build a copy of the library with extra compiler flags, in the container (eight parts side by side), as
gym_pcgrl_amd/lib/libexp_<name>.so -- for tools/ab_so.sh / exp_build_bench.py so:<path> on the GPU box.
    python tools/exp_build_local.py "<flags>" <name>"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_pcgrl_amd import _lib
flags, name = sys.argv[1].split(), sys.argv[2]
objdir = "/tmp/exp_obj_" + name
os.makedirs(objdir, exist_ok=True)
base = [f for f in _lib.HIPCC_FLAGS if f != "-shared"] + flags
procs = [subprocess.Popen(["hipcc"] + base + ["-DPCGRL_PART=%d" % k, "-c", _lib.SOURCES[0], "-o", "%s/part%d.o" % (objdir, k)], stderr=subprocess.DEVNULL) for k in range(_lib.NPARTS)]
assert all(p.wait() == 0 for p in procs)
out = os.path.join(_lib.LIBDIR, "libexp_%s.so" % name)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + ["%s/part%d.o" % (objdir, k) for k in range(_lib.NPARTS)] + ["-o", out])
print(out)
