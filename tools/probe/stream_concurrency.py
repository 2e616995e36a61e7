#!/usr/bin/env python3
"""GPU box: K streams, one 1 ms single-block spin kernel each -> wall time (1 ms = all concurrent, K ms = serialized).
    [GPU_MAX_HW_QUEUES=n] python tools/probe/stream_concurrency.py"""
import ctypes as C, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libspin.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", os.path.join(HERE, "spin.hip"), "-o", so])
import torch
L = C.CDLL(so)
L.spin_launch.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
torch.zeros(1, device="cuda")
for mode in ("torch", "raw"):
  for K in (1, 2, 4, 8, 16, 32, 64):
    if mode == "torch":
        streams = [torch.cuda.Stream() for _ in range(K)]
        ptrs = [s.cuda_stream for s in streams]
    else:
        hip = C.CDLL("libamdhip64.so")
        ptrs = []
        for _ in range(K):
            p = C.c_void_p()
            assert hip.hipStreamCreateWithFlags(C.byref(p), 1) == 0
            ptrs.append(p.value)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for p in ptrs:
            assert L.spin_launch(C.c_void_p(p), 100000, 1, 64, 0) == 0       # 1 ms
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # chains: 10 x 0.1 ms kernels per stream
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        for p in ptrs:
            L.spin_launch(C.c_void_p(p), 10000, 1, 64, 0)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print("%s K=%2d: 1 x 1 ms per stream -> %.2f ms;  10 x 0.1 ms per stream -> %.2f ms" % (mode, K, dt * 1e3, dt2 * 1e3))
