"""MultiGpuPcgrlEnv: ONE process drives the GPUs of a node -- the single-process counterpart of `sharding.ShardedPcgrlEnv`
(one process per GPU) and the replacement for the reference's `SubprocVecEnv` of `n_cpu` worker processes (utils.py:60-71) when the
trainer itself is one process.

SURVEY.md 8e: the environment axis is cut into contiguous shards, shard g lives on `devices[g]` with its own handle of the C ABI
(`pcgrl_create` ... -- a handle is bound to the device of its buffers, every launching entry point makes that device current for the
call) and its own HIP stream; `step()` issues step k on EVERY device before anything waits, there is no collective and no
cross-device traffic on the step path, and what comes back is either the list of per-device tensors (for a data-parallel learner that
keeps each shard where it is) or one concatenation on pinned host memory / on one device.  Environment i is seeded with
`seed + i` whatever the number of devices, so the results are bitwise independent of the sharding (the GPU tests hold
G in {1, 2, 4, 8} against each other).

`devices` may name the same GPU several times (two handles on two streams of one GPU): that is how the single-GPU test box
exercises the multi-handle path, and a legitimate way to overlap two half-batches on one device.
"""
from collections import OrderedDict

import numpy as np

from .sharding import shard_range


class ShardedTensor:
    """The per-device pieces of one logical [N, ...] tensor, in shard order."""

    def __init__(self, parts):
        self.parts = list(parts)

    def __len__(self):
        return sum(int(p.shape[0]) for p in self.parts)

    def __iter__(self):
        return iter(self.parts)

    def __getitem__(self, g):
        return self.parts[g]

    def to(self, device):
        """One tensor on `device` (peer copies; synchronises nothing on the host)."""
        import torch
        return torch.cat([p.to(device, non_blocking=True) for p in self.parts], 0)

    def cpu(self):
        import torch
        return torch.cat([p.cpu() for p in self.parts], 0)


def _pair_overlaps(torch, a, b, cycles=100000):
    """Do kernels of streams a and b (one device) run side by side?  A spinning kernel on each: next to each other they end together,
    on one hardware queue one after the other."""
    dev = a.device
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in ev:
        with torch.cuda.stream(a):
            e.record()
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(a):                      # one spin alone: the yardstick
        ev[0].record(); torch.cuda._sleep(cycles); ev[1].record()
    torch.cuda.synchronize(dev)
    alone = ev[0].elapsed_time(ev[1])
    with torch.cuda.stream(a):
        ev[2].record(); torch.cuda._sleep(cycles)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles); ev[3].record()
    torch.cuda.synchronize(dev)
    return ev[2].elapsed_time(ev[3]) < 1.5 * alone


def side_by_side_streams(torch, device, k, tries=None):
    """k streams of `device` whose kernels overlap pairwise (different hardware queues) at the time of the call, found by trying:
    streams come from torch's pool, a candidate that serialises with one already chosen is passed over.  That is best effort: HIP
    multiplexes streams onto a few hardware queues (four by default) and was seen to move them -- two streams that overlapped when
    they were chosen sat on one queue by the time an environment stepped on them (2 x 23 us instead of 25 us a C3 step), and apart
    again a few hundred launches later (tools/probe/sub_diag.py, profiles/r6_round6/probe/sub_diag_*.txt; a normal- and a
    high-priority stream always pass the test but twice in eleven environments stepped at 67 us: not used).  `streams_overlap()`
    of the environment says how things stand; correct either way, only the overlap is at stake."""
    chosen = []
    for _ in range(tries if tries is not None else 3 * k + 4):
        if len(chosen) == k:
            break
        c = torch.cuda.Stream(device=device)
        if all(c.cuda_stream != o.cuda_stream and _pair_overlaps(torch, o, c) for o in chosen):
            chosen.append(c)
    for _ in range(64):
        if len(chosen) == k:
            break
        c = torch.cuda.Stream(device=device)
        if all(c.cuda_stream != o.cuda_stream for o in chosen):
            chosen.append(c)
    return chosen


class MultiGpuPcgrlEnv:
    def __init__(self, prob="binary", rep="narrow", num_envs=1, devices=None, seed=0, auto_reset=True, gather="list", sync_streams=True):
        """gather: what reset()/step() return per output -- "list": a ShardedTensor of live per-device views (zero copy, no
        host sync: the default); "host": one pinned host tensor (the copies are issued per stream, then every stream is
        waited for -- the shape a central numpy/CPU consumer wants); a device string: one tensor on that device.
        With "host" the returned tensors are the driver's own pinned buffers, two sets used alternately: what step k returned
        stays valid through step k + 1 (observation and next observation of a transition can be held side by side) and is
        overwritten by step k + 2 -- copy what has to live longer.  ("list" views are live state: overwritten by the next step.)
        sync_streams: every step() makes each shard's stream wait (on the device) for what the caller's current stream of that
        device has queued -- the action tensors -- and the caller's stream wait for the shard's afterwards, so that the outputs
        can be consumed on the current stream like those of a single BatchedPcgrlEnv.  False: the caller orders the streams itself
        (per-shard policies that run on `streams[g]`): step() then issues nothing but the step."""
        import torch

        from .envs import BatchedPcgrlEnv
        self._torch = torch
        if devices is None:
            devices = ["cuda:%d" % i for i in range(max(torch.cuda.device_count(), 1))]
        self.devices = [torch.device(d) for d in devices]
        if not self.devices:
            raise ValueError("MultiGpuPcgrlEnv needs at least one device")
        for d in self.devices:
            if d.type != "cuda":
                raise RuntimeError("MultiGpuPcgrlEnv runs on AMD GPUs only (device=%r); there is no CPU fallback" % (d,))
        self.num_envs = int(num_envs)
        G = len(self.devices)
        if self.num_envs < G:
            raise ValueError("fewer environments (%d) than devices (%d)" % (self.num_envs, G))
        self.gather = gather
        self.ranges = [shard_range(self.num_envs, G, g) for g in range(G)]
        if seed is None:
            from . import seeding
            seed = seeding.create_seed(None, max_bytes=7)
        self.base_seed = int(seed) if np.ndim(seed) == 0 else None
        self.shards, self.streams = [], []
        for g, (lo, hi) in enumerate(self.ranges):
            s = (self.base_seed + lo) if self.base_seed is not None else [int(v) for v in seed[lo:hi]]
            self.shards.append(BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=hi - lo, device=self.devices[g], seed=s, auto_reset=auto_reset))
        # a stream per shard.  Shards that share a device (sub-batches of one GPU: the double-buffered rollout) only
        # overlap when their streams sit on different hardware queues -- HIP has four by default and hands them out as it sees fit:
        # two fresh streams were seen to serialise (2 x 23 us instead of 25 us a C3 step) -- so those are picked by trying them out
        by_dev = {}
        for g, d in enumerate(self.devices):
            by_dev.setdefault(str(d), []).append(g)
        self.streams = [None] * G
        for d, gs in by_dev.items():
            ss = [torch.cuda.Stream(device=d)] if len(gs) == 1 else side_by_side_streams(torch, d, len(gs))
            for g, st in zip(gs, ss):
                self.streams[g] = st
        self.sync_streams = bool(sync_streams)
        self._multi = None             # step() through pcgrl_step_multi: ctypes arrays + the cached live views (built by reset())
        self._pinned = {}
        self._flip = 0                 # which of the two pinned sets the current call fills (gather="host")
        self._pending = None
        self._abufs = None             # action_buffers()

    # ------------------------------------------------------------------ the surface of BatchedPcgrlEnv
    def __getattr__(self, name):          # spaces, get_border_tile, get_num_tiles, _prob, _rep, _max_changes ...: the same on every shard
        if name in ("shards", "_torch"):
            raise AttributeError(name)
        return getattr(self.shards[0], name)

    def _each(self, fn, fork=True):
        """fn(g, shard) on every shard's own stream, all issued before anything waits.  Each stream first waits (on the device)
        for what the caller's current stream of that device has queued -- its action tensors -- and the caller's stream then
        waits for the shard's: the outputs can be consumed on the current stream like those of a single BatchedPcgrlEnv."""
        torch = self._torch
        outs = []
        for g, sh in enumerate(self.shards):
            st = self.streams[g]
            cur = torch.cuda.current_stream(self.devices[g])
            if fork:
                st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(fn(g, sh))
        for g in range(len(self.shards)):
            torch.cuda.current_stream(self.devices[g]).wait_stream(self.streams[g])
        return outs

    def seed(self, seed=None):
        if seed is None:
            from . import seeding
            seed = seeding.create_seed(None, max_bytes=7)
        out = []
        for (lo, hi), sh in zip(self.ranges, self.shards):
            out += sh.seed(int(seed) + lo if np.ndim(seed) == 0 else [int(v) for v in seed[lo:hi]])
        return out

    def adjust_param(self, **kwargs):
        for sh in self.shards:
            sh.adjust_param(**kwargs)
        self._multi = None             # (the cached info views carry max_changes / max_iterations)

    def split(self, actions):
        """[N(, k)] actions -> the per-shard pieces, each on its shard's device (a list is taken as already split)."""
        torch = self._torch
        if isinstance(actions, (list, tuple)) and len(actions) == len(self.shards) and all(torch.is_tensor(a) for a in actions):
            return list(actions)
        if isinstance(actions, ShardedTensor):
            return actions.parts
        a = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions))
        return [a[lo:hi].to(self.devices[g], non_blocking=True) for g, (lo, hi) in enumerate(self.ranges)]

    def _collect(self, name, parts):
        """One output in the form `gather` asks for."""
        torch = self._torch
        if self.gather == "list":
            return ShardedTensor(parts)
        if self.gather == "host":
            key = (name, tuple(parts[0].shape[1:]), parts[0].dtype, self._flip)
            buf = self._pinned.get(key)
            if buf is None:
                buf = torch.empty((self.num_envs,) + tuple(parts[0].shape[1:]), dtype=parts[0].dtype).pin_memory()
                self._pinned[key] = buf
            for g, ((lo, hi), p) in enumerate(zip(self.ranges, parts)):
                with torch.cuda.stream(self.streams[g]):
                    buf[lo:hi].copy_(p, non_blocking=True)
            return buf
        return ShardedTensor(parts).to(self.gather)

    def _finish(self):
        if self.gather == "host":          # the pinned copies were issued on the shards' streams
            for st in self.streams:
                st.synchronize()

    def _obs(self, obs_list):
        o = OrderedDict()
        for k in obs_list[0]:
            o[k] = self._collect("obs_" + k, [ob[k] for ob in obs_list])
        return o

    def reset(self):
        self._flip ^= 1
        obs = self._each(lambda g, sh: sh.reset())
        out = self._obs(obs)
        self._finish()
        self._multi = None             # (a reset may have re-allocated a shard: new handles, new views)
        return out

    def _prepare_multi(self):
        """What step() needs to go through ONE call of the library (pcgrl_step_multi): the handles and streams as C arrays, and the
        outputs -- the shards' live views never move between resets, so the returned structures are built once."""
        import ctypes as C
        from .envs.batched_env import InfoBatch
        G = len(self.shards)
        VP = C.c_void_p * G
        res = []
        for sh in self.shards:
            b = sh._bufs
            decode = sh._prob.decode_rows if sh._prob.packed_rows else None
            res.append((sh._obs(), b["reward"], b["done"].view(self._torch.bool), InfoBatch(sh._prob.info_keys, b["info"], sh._max_iterations, sh._max_changes, decode)))
        want = []
        for g, sh in enumerate(self.shards):
            d = self._torch.device(self.devices[g])
            if d.type == "cuda" and d.index is None:          # (a tensor's device always carries its ordinal)
                d = self._torch.device("cuda", self._torch.cuda.current_device())
            want.append((d, sh.num_envs * sh._rep.action_width()))
        self._multi = dict(lib=self.shards[0]._lib, n=G, handles=VP(*[sh._handle.value for sh in self.shards]), actions=VP(), want=want,
                           hvals=[sh._handle.value for sh in self.shards],
                           streams=VP(*[st.cuda_stream for st in self.streams]), res=res, bound=None,
                           out=(self._obs([r[0] for r in res]), ShardedTensor([r[1] for r in res]), ShardedTensor([r[2] for r in res]), [r[3] for r in res]))

    def _multi_valid(self, M):
        """The cached handles are the shards' current ones and no shard asks for the per-shard path (strict actions, a pending
        reset): three attribute reads per shard and call, against a use-after-free of a destroyed handle."""
        for sh, hv in zip(self.shards, M["hvals"]):
            h = sh._handle
            if h is None or h.value != hv or sh.strict_actions or sh._needs_reset:
                return False
        return True

    def step(self, actions):
        """pcgrl_env.py:129-150 for every environment of every shard.  Returns (obs, reward, done, infos): obs / reward / done in
        the `gather` form, infos the list of the shards' InfoBatch objects (live device tables).  With gather="list" the whole
        node is stepped by one call of the library (pcgrl_step_multi) and the call performs no host synchronisation."""
        self._flip ^= 1
        torch = self._torch
        M = self._multi
        if M is not None and not self._multi_valid(M):
            M = self._multi = None             # (a shard was closed / re-allocated / switched to strict actions behind the cache's back)
        if M is None and self.gather == "list" and not any(sh.strict_actions or sh._needs_reset or sh._handle is None for sh in self.shards):
            self._prepare_multi()              # (dropped again by reset() / adjust_param(): the shards are then looked at anew)
            M = self._multi
        if M is not None:
            if actions is not M["bound"]:      # (action_buffers(): the pointers are in place already -- nothing to look at per call)
                i32 = torch.int32
                want = M["want"]               # per shard: (device, element count) -- a raw pointer goes to the kernels, so both are checked
                if (isinstance(actions, (list, tuple)) and len(actions) == M["n"] and
                        all(torch.is_tensor(a) and a.dtype is i32 and a.is_contiguous() and a.device == w[0] and a.numel() == w[1]
                            for a, w in zip(actions, want))):
                    acts = actions             # per-shard int32 tensors on their devices, as a per-shard policy produces them: taken as they are
                else:                          # anything else (host tensors, another GPU, another length): moved / reshaped / refused by the shard
                    acts = [sh._as_actions(p) for sh, p in zip(self.shards, self.split(actions))]
                self._last_actions = acts      # keep the buffers alive until the launches are done
                for g, a in enumerate(acts):
                    M["actions"][g] = a.data_ptr()
                M["bound"] = actions if (actions is self._abufs and acts is actions) else None
            if self.sync_streams:
                for g, st in enumerate(self.streams):
                    st.wait_stream(torch.cuda.current_stream(self.devices[g]))
            rc = M["lib"].pcgrl_step_multi(M["handles"], M["actions"], M["streams"], M["n"])
            if rc:
                from . import _lib
                _lib.check(rc, "pcgrl_step_multi")
            if self.sync_streams:
                for g, st in enumerate(self.streams):
                    torch.cuda.current_stream(self.devices[g]).wait_stream(st)
            return M["out"]
        parts = self.split(actions)
        res = self._each(lambda g, sh: sh.step(parts[g]))
        out = (self._obs([r[0] for r in res]), self._collect("reward", [r[1] for r in res]),
               self._collect("done", [r[2] for r in res]), [r[3] for r in res])
        self._finish()
        return out

    def action_buffers(self):
        """The driver's own per-shard action tensors (int32, [n_g] or [n_g, k] on the shard's device), for policies that write their
        actions in place (`torch.argmax(logits, 1, out=...)`, `buf.copy_(a)`): `step(env.action_buffers())` recognises the list by
        identity and issues the step without looking at the tensors -- the host cost of a step is then the one library call."""
        if self._abufs is None:
            torch = self._torch
            self._abufs = []
            for g, sh in enumerate(self.shards):
                k = sh._rep.action_width()
                self._abufs.append(torch.zeros((sh.num_envs,) if k == 1 else (sh.num_envs, k), dtype=torch.int32, device=self.devices[g]))
        return self._abufs

    def enable_async(self, nslots=1024):
        """BatchedPcgrlEnv.enable_async on every shard (`nslots` suspended searches per shard)."""
        return all([sh.enable_async(nslots) for sh in self.shards])

    def tick(self, actions, pop_budget=64):
        """One asynchronous tick of every shard (BatchedPcgrlEnv.tick: the search problems), each on its own stream, all issued before
        anything waits.  Returns (obs, reward, done, infos, pending) in the `gather` form."""
        self._flip ^= 1
        parts = self.split(actions)
        res = self._each(lambda g, sh: sh.tick(parts[g], pop_budget=pop_budget))
        out = (self._obs([r[0] for r in res]), self._collect("reward", [r[1] for r in res]), self._collect("done", [r[2] for r in res]),
               [r[3] for r in res], self._collect("pending", [r[4] for r in res]))
        self._finish()
        return out

    def step_async(self, actions):
        self._pending = self.step(actions)

    def step_wait(self):
        return self._pending

    def rollout(self, actions, want_info=True):
        """`T` steps on a tape [T, N(, k)] (or a list of per-shard tapes [T, n_g(, k)]): one pcgrl_rollout per device, all in
        flight together.  Returns (reward [T, N], done [T, N], list of per-shard InfoBatch or None) with reward / done gathered
        along the environment axis in the `gather` form ("list": ShardedTensor of [T, n_g] pieces)."""
        torch = self._torch
        if isinstance(actions, (list, tuple)):
            tapes = list(actions)
        else:
            a = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions))
            tapes = [a[:, lo:hi].to(self.devices[g], non_blocking=True).contiguous() for g, (lo, hi) in enumerate(self.ranges)]
        res = self._each(lambda g, sh: sh.rollout(tapes[g], want_info=want_info))
        rew, done = [r[0] for r in res], [r[1] for r in res]
        if self.gather == "list":
            out = (ShardedTensor(rew), ShardedTensor(done), [r[2] for r in res])
        else:
            dev = "cpu" if self.gather == "host" else self.gather
            out = (torch.cat([r.to(dev) for r in rew], 1), torch.cat([d.to(dev) for d in done], 1), [r[2] for r in res])
        return out

    def streams_overlap(self):
        """Do the streams of the shards that share a device run side by side right now (the spin test of side_by_side_streams)?
        HIP may have moved them onto one hardware queue since they were chosen; `repick_streams()` chooses again.  Synchronises."""
        by_dev = {}
        for g, d in enumerate(self.devices):
            by_dev.setdefault(str(d), []).append(g)
        return all(_pair_overlaps(self._torch, self.streams[a], self.streams[b]) for gs in by_dev.values() for i, a in enumerate(gs) for b in gs[i + 1:])

    def repick_streams(self):
        """New streams for the shards that share a device (everything queued on the old ones is waited for first)."""
        torch = self._torch
        self.synchronize()
        by_dev = {}
        for g, d in enumerate(self.devices):
            by_dev.setdefault(str(d), []).append(g)
        for d, gs in by_dev.items():
            if len(gs) > 1:
                for g, st in zip(gs, side_by_side_streams(torch, d, len(gs))):
                    self.streams[g] = st
        self._multi = None

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def check_status(self):
        return [sh.check_status() for sh in self.shards]

    def close(self):
        self._multi = None             # (the cached handle values die with the shards)
        for sh in self.shards:
            sh.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
