"""node.MultiGpuPcgrlEnv.step with gather="list" -- the path a one-process trainer drives eight GPUs through -- is ONE call of the
library (pcgrl_step_multi) per step of the whole node and performs no host synchronisation.  CPU only: the shards, their streams and
the library are stand-ins that record what the driver does with them (the GPU counterpart, tests/test_gpu_round5.py, holds the same
path against per-shard stepping on real handles)."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import pytest
import torch

from gym_pcgrl_amd import node


class _Prob:
    packed_rows = False
    info_keys = ["regions", "path-length"]
    decode_rows = None


class _Rep:
    def action_width(self):
        return 1


class _Shard:
    _rep = _Rep()

    def __init__(self, i, n, lib):
        self._handle = C.c_void_p(1000 + i)
        self._lib = lib
        self.strict_actions = False
        self._needs_reset = False
        self.num_envs = n
        self._prob = _Prob()
        self._max_iterations, self._max_changes = 7644, 39
        self._bufs = {"reward": torch.zeros(n, dtype=torch.float64), "done": torch.zeros(n, dtype=torch.uint8), "info": torch.zeros((n, 10), dtype=torch.int32)}
        self._map = torch.zeros((n, 3, 3), dtype=torch.uint8)

    def _obs(self):
        return OrderedDict(map=self._map)

    def _as_actions(self, a):
        return a.to(torch.int32).contiguous()


class _Stream:
    def __init__(self, i, log):
        self.cuda_stream = 7000 + i
        self._log = log

    def wait_stream(self, other):
        self._log.append("wait_stream")

    def synchronize(self):
        raise AssertionError("a stream was synchronised")


class _Lib:
    def __init__(self):
        self.calls = []

    def pcgrl_step_multi(self, handles, actions, streams, count):
        self.calls.append(([handles[i] for i in range(count)], [actions[i] for i in range(count)], [streams[i] for i in range(count)], count))
        return 0

    def pcgrl_step(self, *a):
        raise AssertionError("per-shard pcgrl_step on the one-call path")


def _driver(G, n, sync_streams, log, lib):
    env = object.__new__(node.MultiGpuPcgrlEnv)
    env._torch = torch
    env.devices = [torch.device("cpu")] * G
    env.num_envs = G * n
    env.gather = "list"
    env.ranges = [(g * n, (g + 1) * n) for g in range(G)]
    env.shards = [_Shard(g, n, lib) for g in range(G)]
    env.streams = [_Stream(g, log) for g in range(G)]
    env.sync_streams = sync_streams
    env._multi, env._pinned, env._flip, env._pending, env._abufs = None, {}, 0, None, None
    return env


def test_list_gather_step_is_one_library_call_and_never_synchronises(monkeypatch):
    G, n = 8, 5
    log, lib = [], _Lib()
    env = _driver(G, n, False, log, lib)

    def boom(*a, **k):
        raise AssertionError("host synchronisation on the step path")

    for name in ("synchronize", "current_stream"):       # (sync_streams=False: not even the current stream is looked up)
        monkeypatch.setattr(torch.cuda, name, boom)
    for name in ("cpu", "item", "numpy", "tolist"):
        monkeypatch.setattr(torch.Tensor, name, boom)
    acts = torch.arange(G * n, dtype=torch.int32)
    out1 = env.step(acts)
    out2 = env.step(acts)
    assert len(lib.calls) == 2 and log == []
    handles, aptr, sptr, count = lib.calls[1]
    assert count == G and handles == [1000 + g for g in range(G)] and sptr == [7000 + g for g in range(G)]
    assert aptr == [a.data_ptr() for a in env._last_actions] and len(set(aptr)) == G
    # the outputs are the shards' live views, built once
    obs, rew, done, infos = out1
    assert out2 is out1 and isinstance(rew, node.ShardedTensor) and len(rew) == G * n and done[3].dtype == torch.bool
    assert obs["map"][2] is env.shards[2]._map and infos[5].table is env.shards[5]._bufs["info"]
    env.shards = []          # (nothing to close)


def test_action_buffers_are_stepped_without_looking_at_them(monkeypatch):
    """step(env.action_buffers()): after the first call the pointers are in place; the call touches no tensor at all."""
    G, n = 8, 4
    log, lib = [], _Lib()
    env = _driver(G, n, False, log, lib)
    bufs = env.action_buffers()
    assert len(bufs) == G and all(b.dtype == torch.int32 and tuple(b.shape) == (n,) for b in bufs) and env.action_buffers() is bufs
    env.step(bufs)
    want = [b.data_ptr() for b in bufs]

    def boom(*a, **k):
        raise AssertionError("the bound action buffers were inspected")

    for name in ("data_ptr", "is_contiguous", "to", "contiguous"):
        monkeypatch.setattr(torch.Tensor, name, boom)
    bufs[3].fill_(2)
    env.step(bufs)
    assert len(lib.calls) == 2 and lib.calls[1][1] == want
    monkeypatch.undo()
    other = [torch.ones(n, dtype=torch.int32) for _ in range(G)]          # any other list: looked at again, and the binding is dropped
    env.step(other)
    assert lib.calls[2][1] == [a.data_ptr() for a in other] and env._multi["bound"] is None
    env.step(bufs)
    assert lib.calls[3][1] == want and env._multi["bound"] is bufs
    env.shards = []


def test_sync_streams_orders_each_shard_stream_against_the_callers(monkeypatch):
    G, n = 4, 3
    log, lib = [], _Lib()
    env = _driver(G, n, True, log, lib)

    class Cur:
        def wait_stream(self, other):
            log.append("current.wait_stream")

    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: Cur())
    env.step(torch.zeros(G * n, dtype=torch.int32))
    assert log == ["wait_stream"] * G + ["current.wait_stream"] * G and len(lib.calls) == 1
    env.shards = []


def test_fast_path_takes_only_tensors_it_can_hand_to_the_kernels():
    """the per-shard list goes to the kernels as raw pointers, so the fast path checks device and element count and sends
    everything else through the shard's own conversion (which moves / reshapes / refuses) instead of launching on a bad pointer."""
    G, n = 4, 6
    log, lib = [], _Lib()
    env = _driver(G, n, False, log, lib)
    seen = []
    for sh in env.shards:
        sh._as_actions = (lambda a, _sh=sh: (seen.append(tuple(a.shape)), a.to(torch.int32).contiguous().reshape(-1)[:_sh.num_envs])[1])
    good = [torch.zeros(n, dtype=torch.int32) for _ in range(G)]
    env.step(good)
    assert seen == [] and lib.calls[-1][1] == [a.data_ptr() for a in good]
    short = [torch.zeros(n, dtype=torch.int32) for _ in range(G)]
    short[2] = torch.zeros(n - 1, dtype=torch.int32)                      # wrong length: not taken as it is
    env.step(short)
    assert len(seen) == G and seen[2] == (n - 1,)
    seen.clear()
    other_dev = [torch.zeros(n, dtype=torch.int32) for _ in range(G)]
    env._multi["want"][1] = (torch.device("meta"), n)                     # shard 1 lives on another device than the tensor offered for it
    env.step(other_dev)
    assert len(seen) == G
    env.shards = []


def test_cached_handles_are_dropped_when_a_shard_changes_behind_the_cache():
    """close() and shard-level re-allocation invalidate the pcgrl_step_multi cache (no destroyed handle reaches the library);
    strict_actions / a pending reset switch to the per-shard path."""
    G, n = 3, 4
    log, lib = [], _Lib()
    env = _driver(G, n, False, log, lib)
    acts = torch.zeros(G * n, dtype=torch.int32)
    env.step(acts)
    assert lib.calls[-1][0] == [1000, 1001, 1002]
    env.shards[1]._handle = C.c_void_p(5555)                              # re-allocated: a new handle value
    env.step(acts)
    assert lib.calls[-1][0] == [1000, 5555, 1002]
    stepped = []
    for sh in env.shards:
        sh.step = (lambda a, _sh=sh: (stepped.append(_sh._handle.value), (_sh._obs(), _sh._bufs["reward"], _sh._bufs["done"], None))[1])
    env._each = lambda f: [f(g, sh) for g, sh in enumerate(env.shards)]
    env.shards[0].strict_actions = True                                   # the shard wants to look at every action itself
    ncalls = len(lib.calls)
    env.step(acts)
    assert len(lib.calls) == ncalls and stepped == [1000, 5555, 1002] and env._multi is None
    env.shards[0].strict_actions = False
    env.step(acts)
    assert len(lib.calls) == ncalls + 1
    closed = []
    for sh in env.shards:
        sh.close = (lambda _sh=sh: closed.append(_sh))
    env.close()
    assert env._multi is None and len(closed) == G
    env.shards = []
