"""BatchedPcgrlEnv: N lockstep PcgrlEnv instances on one MI355X.

Mirrors the reference's PcgrlEnv (pcgrl_env.py) method for method -- seed / reset / step /
adjust_param / get_border_tile / get_num_tiles -- with a leading environment axis, vector-env
auto-reset (the reference leaves that to SubprocVecEnv, utils.py:60-71) and torch tensors that are
zero-copy views of the state the HIP kernels own.  All computation happens in
gym_pcgrl_amd/lib/libpcgrl_hip.so through the C ABI of include/pcgrl_hip.h; there is no CPU path.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from .. import _lib, seeding, spaces
from .problems import PROB_IDS, PROBLEMS
from .representations import REP_IDS, REPRESENTATIONS


class InfoBatch:
    """Struct-of-tensors view of the per-step info dicts (pcgrl_env.py:144-148)."""

    def __init__(self, keys, table, max_iterations, max_changes, decode=None):
        self.keys = list(keys)
        self.table = table            # int32 [N,10]: problem info (8 slots), iterations, changes
        self.max_iterations = max_iterations
        self.max_changes = max_changes
        self._decode = decode         # Problem.decode_rows for rows that pack several values into a slot (mdungeon)

    def __getitem__(self, key):
        if key == "iterations":
            return self.table[:, 8]
        if key == "changes":
            return self.table[:, 9]
        if key == "max_iterations":
            return self.max_iterations
        if key == "max_changes":
            return self.max_changes
        if self._decode is not None:
            return self._decode(self.table, [key])[:, 0]
        return self.table[:, self.keys.index(key)]

    def to_list(self):
        t = self.table.cpu().numpy()
        vals = self._decode(self.table, self.keys).cpu().numpy() if self._decode is not None else t
        out = []
        for row, v in zip(t, vals):
            d = {k: int(v[i]) for i, k in enumerate(self.keys)}
            d["iterations"] = int(row[8])
            d["changes"] = int(row[9])
            d["max_iterations"] = self.max_iterations
            d["max_changes"] = self.max_changes
            out.append(d)
        return out


class BatchedPcgrlEnv:
    metadata = {"render.modes": []}

    def __init__(self, prob="binary", rep="narrow", num_envs=1, device=None, seed=None, auto_reset=True, strict_actions=False, tuning=None):
        """strict_actions: check after every step() / rollout() whether an action was outside the action space and raise IndexError
        at the offending call, like the reference does (wide_rep.py:68-69 ...); costs a device synchronisation per step, so it is
        off by default -- the actions are then clamped and the status word reports it (check_status()).
        tuning: {switch: value} for the library's developer switches (include/pcgrl_hip.h pcgrl_tuning; A/B measurements, tests)."""
        import torch
        self._torch = torch
        self._lib = _lib.load()                      # fails loudly when the HIP library is absent
        self._prob = PROBLEMS[prob]()                # KeyError on unknown names, like pcgrl_env.py:28-29
        self._rep = REPRESENTATIONS[rep]()
        self.num_envs = int(num_envs)
        self.auto_reset = bool(auto_reset)
        self.strict_actions = bool(strict_actions)
        self._tuning = dict(tuning or {})
        if device is None:
            device = "cuda:0"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchedPcgrlEnv runs on an AMD GPU only (device=%r); there is no CPU fallback" % (device,))
        # pcgrl_env.py:33-34
        self._max_changes = max(int(0.2 * self._prob._width * self._prob._height), 1)
        self._max_iterations = self._max_changes * self._prob._width * self._prob._height
        self._handle = None
        self._bufs = None
        self._rng = None
        self._episode = None          # RNG tensors survive re-allocation on width/height changes
        self._needs_reset = True
        self._realloc = False
        self._alloc_dims = None
        self._probs_dirty = False
        self._obs_spec = None          # bind_observation(): (out tensor, h, w, centered, pad, onehot)
        self._async = None             # enable_async(): arena tensor + views (pending, counters)
        self._async_nslots = 0
        self._update_spaces()
        self.seed(seed)

    # ------------------------------------------------------------------ spaces / simple getters
    def _update_spaces(self):
        w, h, t = self._prob._width, self._prob._height, self.get_num_tiles()
        self.single_action_space = self._rep.get_action_space(w, h, t)
        self.single_observation_space = self._rep.get_observation_space(w, h, t)
        self.single_observation_space.spaces["heatmap"] = spaces.Box(low=0, high=self._max_changes, dtype=np.uint8, shape=(h, w))
        self.action_space = self.single_action_space
        self.observation_space = self.single_observation_space

    def get_border_tile(self):
        return self._prob.get_tile_types().index(self._prob._border_tile)

    def get_num_tiles(self):
        return len(self._prob.get_tile_types())

    # ------------------------------------------------------------------ config plumbing
    def _config(self, map_dims=None):
        """map_dims: (width, height) of the allocated maps when they are not the problem's -- adjust_param(width, height) without a
        reset(): the reference goes on stepping the old maps with the problem's new size in its formulas (pcgrl_env.py:106-115)."""
        c = _lib.Config()
        c.prob, c.rep, c.num_envs = PROB_IDS[self._prob.name], REP_IDS[self._rep.name], self.num_envs
        c.width, c.height = (int(self._prob._width), int(self._prob._height)) if map_dims is None else (int(map_dims[0]), int(map_dims[1]))
        c.prob_width, c.prob_height = int(self._prob._width), int(self._prob._height)
        # (pcgrl_env.py:33-34 computes max_iterations = max_changes * W * H with Python integers; the device field and the iteration
        #  counter it is compared with are 32-bit: beyond 2^31 - 1 -- e.g. 255 x 255 with change_percentage above 0.5 -- the limit is
        #  one no episode can reach either way, and the info dict keeps reporting the exact value from the host side)
        c.max_changes, c.max_iterations = int(self._max_changes), min(int(self._max_iterations), 2 ** 31 - 1)
        c.random_start, c.random_tile, c.warp, c.random_probs = 1, 1, 0, 0
        c.auto_reset = int(self.auto_reset)
        for k, v in list(self._prob.device_params().items()) + list(self._rep.device_params().items()):
            setattr(c, k, v)
        for i, t in enumerate(self._prob.tiles):
            c.tile_probs[i] = float(self._prob._prob[t])
        for i, k in enumerate(self._prob.reward_keys):
            c.rewards[i] = float(self._prob._rewards[k])
        return c

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _free(self):
        if self._handle is not None:
            self._torch.cuda.synchronize(self.device)
            self._lib.pcgrl_destroy(self._handle)
            self._handle = None
        self._bufs = None

    def _allocate(self):
        torch = self._torch
        self._free()
        cfg = self._config()
        lay = _lib.Layout()
        _lib.check(self._lib.pcgrl_query_layout(C.byref(cfg), C.byref(lay)), "pcgrl_query_layout")
        n, w, h = self.num_envs, cfg.width, cfg.height
        dev = self.device
        mask_dtype = torch.int32 if lay.mask_bytes == 4 else torch.int64
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        b = OrderedDict()
        b["map"] = z((n, h, w), torch.uint8)
        b["old_map"] = z((n, h, w), torch.uint8)
        b["heatmap"] = z((n, h, w), torch.int16)
        b["pos"] = z((n, 2), torch.uint8)
        b["planes"] = z((n, lay.group, lay.nplanes), mask_dtype) if lay.nplanes else z((2,), mask_dtype)   # smb keeps no bit planes
        b["counters"] = z((n, 2), torch.int32)
        b["stats"] = z((n, 8), torch.int32)
        b["start_stats"] = z((n, 8), torch.int32)
        b["info"] = z((n, 10), torch.int32)
        b["reward"] = z((n,), torch.float64)
        b["done"] = z((n,), torch.uint8)
        first = self._rng is None
        if first:
            self._rng = OrderedDict()
            self._rng["tile_p"] = z((n, 2), torch.float64)      # BinaryProblem._prob per environment
            self._rng["rng_rep"] = z((n, seeding.MT_N), torch.int32)
            self._rng["rng_prob"] = z((n, seeding.MT_N), torch.int32) if self._prob.name == "binary" else None
            self._rng["rng_cursor"] = z((n, 2), torch.int32)
            self._rng_seeded = False
        b.update(self._rng)
        b["scratch"] = z((int(lay.scratch),), torch.uint8)
        for name in _lib.BUFFER_NAMES:
            t = b[name]
            if t is not None:
                assert t.is_contiguous() and t.numel() * t.element_size() >= getattr(lay, name), name
        handle = C.c_void_p()
        _lib.check(self._lib.pcgrl_create(C.byref(cfg), C.byref(handle)), "pcgrl_create")
        tun = _lib.make_tuning(self._tuning)
        _lib.check(self._lib.pcgrl_set_tuning(handle, C.byref(tun)), "pcgrl_set_tuning")
        bufs = _lib.Buffers()
        for name in _lib.BUFFER_NAMES:
            setattr(bufs, name, b[name].data_ptr() if b[name] is not None else None)
        _lib.check(self._lib.pcgrl_bind(handle, C.byref(bufs), self._stream()), "pcgrl_bind")
        self._handle, self._bufs, self._layout = handle, b, lay
        self._alloc_dims = (w, h)
        if first or self._probs_dirty:
            _lib.check(self._lib.pcgrl_set_tile_probs(self._handle, self._stream()), "pcgrl_set_tile_probs")
            self._probs_dirty = False
        if not self._rng_seeded:
            self._upload_seeds()
        if self._episode is not None:      # rebinding after a reallocation keeps the running episodes' sums
            self._bind_episode_stats()
        self._apply_observation()
        self._async = None
        if self._async_nslots:
            self._bind_async()

    def _upload_seeds(self):
        words = np.ascontiguousarray(self._seed_keys, dtype=np.uint32)      # [N, 3]: the MT19937 states are made on the device
        _lib.check(self._lib.pcgrl_seed_words(self._handle, words.ctypes.data_as(C.c_void_p), 0, self.num_envs, self._stream()), "pcgrl_seed_words")
        self._rng_seeded = True

    # ------------------------------------------------------------------ reference surface
    def seed(self, seed=None):
        """Environment i is seeded with `seed + i` (pcgrl_env.py:54-57 per environment).  `seed` may
        also be a sequence of N seeds.  Returns the list of seeds used."""
        if seed is None:
            seed = seeding.create_seed(None, max_bytes=7)
        if np.ndim(seed) == 0:
            seeds = [int(seed) + i for i in range(self.num_envs)]
        else:
            seeds = [int(s) for s in seed]
            if len(seeds) != self.num_envs:
                raise ValueError("need %d seeds, got %d" % (self.num_envs, len(seeds)))
        for s in seeds:
            if s < 0:
                raise ValueError("Seed must be a non-negative integer or omitted, not %r" % (s,))
        self._seeds = seeds
        self._seed_keys = seeding.key_words_for_seeds(seeds)
        self._rng_seeded = False
        if self._handle is not None:
            self._upload_seeds()
        return seeds

    def adjust_param(self, **kwargs):
        """pcgrl_env.py:106-115, including the ordering quirk: max_changes is recomputed only when
        change_percentage is passed, and both limits use the width/height from *before* this call."""
        import copy
        if self._async is not None and self._handle is not None and not self._needs_reset:
            self.flush()           # suspended steps are finished under the parameters they were taken with, as in lockstep
        before = (copy.deepcopy(self._prob.__dict__), copy.deepcopy(self._rep.__dict__), self._max_changes, self._max_iterations)
        if "change_percentage" in kwargs:
            percentage = min(1, max(0, kwargs.get("change_percentage")))
            self._max_changes = max(int(percentage * self._prob._width * self._prob._height), 1)
        self._max_iterations = self._max_changes * self._prob._width * self._prob._height
        self._prob._probs_touched = False
        self._prob.adjust_param(**kwargs)
        self._rep.adjust_param(**kwargs)
        self._update_spaces()
        if self._handle is not None:
            # A new width / height takes effect on the maps at the next reset() (representation.py:40-45: only reset() makes a map of
            # the new size); until then the allocated maps go on being stepped, with the problem's new size in its formulas -- what
            # the reference does.  Everything else is applied to the live handle.
            stale = (self._prob._width, self._prob._height) != self._alloc_dims
            cfg = self._config(self._alloc_dims if stale else None)
            # a value the library refuses outright (solver_power < 1 or beyond 10^6, a size beyond its limits ...) is reported here, at
            # the call that passed it -- the layout query validates a configuration without touching the handle
            lay = _lib.Layout()
            if self._lib.pcgrl_query_layout(C.byref(self._config()), C.byref(lay)) != 0:
                self._prob.__dict__, self._rep.__dict__, self._max_changes, self._max_iterations = before      # nothing of the call sticks
                self._update_spaces()
                raise ValueError("adjust_param(%s): outside what the library takes (include/pcgrl_hip.h: map side <= 255 -- the wide representation, which "
                                 "has no cursor: any size whose row masks fit a compute unit's LDS --, max_changes <= 65535, search levels of at most "
                                 "16384 bordered cells, 1 <= solver_power <= 10^6; smb: width <= 250, height 3..32, solver_power <= 16383)" %
                                 ", ".join("%s=%r" % kv for kv in kwargs.items()))
            rc = self._lib.pcgrl_configure(self._handle, C.byref(cfg))
            if rc == _lib.PCGRL_EINVAL and not self._needs_reset:
                # a valid configuration the handle cannot take in place: a solver_power beyond what its search arena was sized for, or one
                # that moves the searches to the other kernel family.  The buffers are re-allocated by the next reset(), which has to
                # come first (the reference would let the new solver_power take effect in the middle of the episode)
                self._realloc = True
                self._needs_reset = True
                self._reset_reason = "adjust_param(%s) needs another search arena" % ", ".join("%s=%r" % kv for kv in kwargs.items())
            elif rc != _lib.PCGRL_EINVAL:
                _lib.check(rc, "pcgrl_configure")
            if rc == 0 and self._prob._probs_touched:
                _lib.check(self._lib.pcgrl_set_tile_probs(self._handle, self._stream()), "pcgrl_set_tile_probs")
                self._prob._probs_touched = False
        self._probs_dirty = self._probs_dirty or self._prob._probs_touched

    def _obs(self):
        b = self._bufs
        o = OrderedDict()
        if self._rep.has_pos:
            o["pos"] = b["pos"]
        o["map"] = b["map"]
        # the device counts changes per cell in 16 bits (pcgrl_env.py:35,137 keeps a float64): the view is int16 while a count cannot
        # pass 32 767 (max_changes, which bounds it, does not: every configuration of the reference's defaults) -- full operator
        # support in torch -- and uint16 beyond (maps of more than 32 767 cells with a change_percentage to match)
        o["heatmap"] = b["heatmap"] if self._max_changes <= 32767 else b["heatmap"].view(self._torch.uint16)
        return o

    def reset(self):
        """Reset every environment (pcgrl_env.py:66-76).  Returns the observation dict of tensors."""
        if self._handle is None or (self._prob._width, self._prob._height) != self._alloc_dims or self._realloc:
            self._allocate()
            self._realloc = False
        _lib.check(self._lib.pcgrl_reset(self._handle, self._stream()), "pcgrl_reset")
        self._needs_reset = False
        self._reset_reason = None
        return self._obs()

    def _as_actions(self, actions):
        torch = self._torch
        aw = self._rep.action_width()
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), device=self.device)
        if actions.device != self.device:
            actions = actions.to(self.device)
        if actions.dtype != torch.int32:
            actions = actions.to(torch.int32)
        actions = actions.reshape(self.num_envs, aw) if aw > 1 else actions.reshape(self.num_envs)
        return actions.contiguous()

    def step(self, actions):
        """pcgrl_env.py:129-150 for every environment.  actions: int [N] (narrow/turtle) or [N,3]
        (wide: x, y, tile).  Returns (obs, reward f64[N], done bool[N], InfoBatch); tensors are views
        of the live state and are overwritten by the next step."""
        if self._needs_reset:
            raise RuntimeError("reset() must be called before step()" + (": " + self._reset_reason if getattr(self, "_reset_reason", None) else ""))
        a = self._as_actions(actions)
        self._last_actions = a   # keep the buffer alive until the launches are done
        _lib.check(self._lib.pcgrl_step(self._handle, C.c_void_p(a.data_ptr()), self._stream()), "pcgrl_step")
        if self.strict_actions:
            self.check_status()
        b = self._bufs
        decode = self._prob.decode_rows if self._prob.packed_rows else None
        info = InfoBatch(self._prob.info_keys, b["info"], self._max_iterations, self._max_changes, decode)
        return self._obs(), b["reward"], b["done"].view(self._torch.bool), info

    def step_flat(self, flat, xyv):
        """ActionMap.step + step() for the wide representation in one call of the library (pcgrl_step_flat): `flat` int32 [N] device
        tensor of indices into (H, W, tiles), `xyv` int32 [N, 3] device scratch.  Same return value as step()."""
        if self._needs_reset:
            raise RuntimeError("reset() must be called before step()")
        self._last_actions = (flat, xyv)
        _lib.check(self._lib.pcgrl_step_flat(self._handle, C.c_void_p(flat.data_ptr()), C.c_void_p(xyv.data_ptr()), self._stream()), "pcgrl_step_flat")
        if self.strict_actions:
            self.check_status()
        b = self._bufs
        decode = self._prob.decode_rows if self._prob.packed_rows else None
        info = InfoBatch(self._prob.info_keys, b["info"], self._max_iterations, self._max_changes, decode)
        return self._obs(), b["reward"], b["done"].view(self._torch.bool), info

    def rollout(self, actions, want_info=True, out=None):
        """`T` consecutive steps on a tape of actions: int tensor [T, N] (narrow, turtle), [T, N, 3] (wide), [T, N, 2] /
        [T, N, 9] (cast / multi).  Returns (reward f64 [T, N], done bool [T, N], info) with `info` an InfoBatch over the
        [T*N, 10] table (rows in step-major order) or None.  The environments end up exactly where T calls of step()
        would leave them; where the whole step is one kernel (binary maps of at most 16 rows) the tape is ONE launch.
        `out`: optional preallocated (reward f64 [T,N], done u8 [T,N], info i32 [T,N,10] or None) on this device."""
        if self._needs_reset:
            raise RuntimeError("reset() must be called before rollout()")
        torch = self._torch
        a = torch.as_tensor(actions, device=self.device).to(torch.int32).contiguous()
        T = int(a.shape[0])
        aw = self._rep.action_width()
        if a.numel() != T * self.num_envs * aw:
            raise ValueError("actions must have shape [T, %d%s], got %s" % (self.num_envs, "" if aw == 1 else ", %d" % aw, tuple(a.shape)))
        if out is not None:
            rew, done, info = out
            want_info = info is not None
            for ten, dt, shape in ((rew, torch.float64, (T, self.num_envs)), (done, torch.uint8, (T, self.num_envs)),
                                   (info, torch.int32, (T, self.num_envs, 10))):
                if ten is not None and (ten.dtype != dt or tuple(ten.shape) != shape or not ten.is_contiguous() or ten.device != self.device):
                    raise ValueError("rollout(out=...): expected a contiguous %s tensor of shape %s on %s" % (dt, shape, self.device))
        else:
            rew = torch.empty((T, self.num_envs), dtype=torch.float64, device=self.device)
            done = torch.empty((T, self.num_envs), dtype=torch.uint8, device=self.device)
            info = torch.empty((T, self.num_envs, 10), dtype=torch.int32, device=self.device) if want_info else None
        self._last_actions = a
        _lib.check(self._lib.pcgrl_rollout(self._handle, C.c_void_p(a.data_ptr()), T, C.c_void_p(rew.data_ptr()),
                                           C.c_void_p(done.data_ptr()), C.c_void_p(info.data_ptr()) if want_info else None,
                                           self._stream()), "pcgrl_rollout")
        if self.strict_actions:
            self.check_status()
        ib = None
        if want_info:
            decode = self._prob.decode_rows if self._prob.packed_rows else None
            ib = InfoBatch(self._prob.info_keys, info.view(T * self.num_envs, 10), self._max_iterations, self._max_changes, decode)
        return rew, done.view(torch.bool), ib

    # ---- the wrapped observation (wrappers.py:215-248), written by the step itself
    def bind_observation(self, out_h, out_w, centered, pad_value, onehot, out=None, incremental=True):
        """From now on every reset() / step() / rollout() / set_maps() leaves the image the reference's composite wrappers
        would produce -- uint8 [N, out_h, out_w, D], D = 1 or the number of tiles (one-hot); centred on the cursor and padded
        with `pad_value`, or the map from its origin -- in the returned tensor (`out`, or a new one).  Where the step is one
        fused kernel it writes the image from its on-chip copy of the state; no extra launch.  `incremental` (default): the
        tensor is the environment's to maintain -- do not write into it -- so that a step may update it in place where that
        is cheaper (a window that does not follow a cursor changes in one cell per step); after set_observation_target() the
        next step writes a full image into the new tensor."""
        torch = self._torch
        depth = self.get_num_tiles() if onehot else 1
        shape = (self.num_envs, int(out_h), int(out_w), depth)
        if out is None:
            out = torch.empty(shape, dtype=torch.uint8, device=self.device)
        self._check_obs_target(out, shape)
        self._obs_spec = (out, int(out_h), int(out_w), int(bool(centered)), int(pad_value), int(bool(onehot)), int(bool(incremental)))
        self._apply_observation()
        return out

    def _check_obs_target(self, out, shape):
        torch = self._torch
        if out.dtype != torch.uint8 or tuple(out.shape) != tuple(shape) or not out.is_contiguous() or out.device != self.device or out.data_ptr() % 16:
            raise ValueError("observation target: expected a contiguous, 16-byte aligned uint8 tensor of shape %s on %s" % (tuple(shape), self.device))

    def set_observation_target(self, out):
        """Redirect the bound observation to another tensor of the same shape (e.g. row t+1 of a rollout buffer)."""
        old = self._obs_spec
        if old is None:
            raise RuntimeError("call bind_observation() first")
        self._check_obs_target(out, old[0].shape)
        self._obs_spec = (out,) + old[1:]
        self._apply_observation()

    def unbind_observation(self):
        self._obs_spec = None
        if self._handle is not None:
            _lib.check(self._lib.pcgrl_bind_observation(self._handle, None, 0, 0, 0, 0, 0, 0), "pcgrl_bind_observation")

    def _apply_observation(self):
        if self._obs_spec is None or self._handle is None:
            return
        out, h, w, centered, pad, onehot, inc = self._obs_spec
        _lib.check(self._lib.pcgrl_bind_observation(self._handle, C.c_void_p(out.data_ptr()), h, w, centered, pad, onehot, inc), "pcgrl_bind_observation")

    # gym.vector-style split call
    def step_async(self, actions):
        self._pending = self.step(actions)

    def step_wait(self):
        return self._pending

    # ---- asynchronous stepping of the search problems (pcgrl_step_async, csrc/kernels_search_async.h)
    def enable_async(self, nslots=1024):
        """Prepare tick(): room for `nslots` suspended searches (about 0.46 MB each at solver_power 5000).  Returns False when
        this configuration has no asynchronous form (binary, zelda, smb; levels or solver_power beyond the compact searches) --
        tick() is then step() with nothing ever pending."""
        self._async_nslots = int(nslots)
        if self._handle is None:
            self._allocate()
        elif self._async is None:
            self._bind_async()
        return self._async is not None

    def _bind_async(self):
        torch = self._torch
        cfg = self._config(self._alloc_dims)
        need = int(self._lib.pcgrl_async_bytes(C.byref(cfg), self._async_nslots))
        if need == 0:
            self._async = None
            return
        arena = torch.empty((need,), dtype=torch.uint8, device=self.device)          # (zeroed by the call)
        rc = self._lib.pcgrl_bind_async(self._handle, C.c_void_p(arena.data_ptr()), need, self._async_nslots, self._stream())
        if rc == _lib.PCGRL_EINVAL:
            # the handle's search arena was cut for another solver_power than the one in effect (an in-place adjust_param(solver_power=
            # smaller)): no asynchronous form on THIS allocation -- tick() steps in lockstep, nothing ever pending -- and the next
            # reset() re-allocates, after which the slots fit
            self._async = None
            self._realloc = True
            return
        _lib.check(rc, "pcgrl_bind_async")
        n = self.num_envs
        off = (n + 255) // 256 * 256
        self._async = dict(arena=arena, pending=arena[:n], counters=arena[off:off + 64 + 16 * 64].view(torch.int64))

    def tick(self, actions, pop_budget=256):
        """One tick of asynchronous stepping (include/pcgrl_hip.h pcgrl_step_async): every environment whose previous step is
        complete takes its action and steps; every search gets at most `pop_budget` pops, an environment whose search is not
        finished by then stays *pending* -- it ignores the actions of the following ticks until one of them completes its step.
        Returns (obs, reward, done, info, pending): the usual live views plus pending uint8 [N] (a live view too) -- where it is
        0 the outputs are those of the environment's last taken action and it takes the next one; elsewhere (1: a search is
        suspended, 2: its search ended the episode and the next tick resets it) the environment's rows are to be ignored.  Per environment the sequence of (taken action -> outputs) is bitwise that of
        step().  step() / rollout() / adjust_param() finish what is pending first (flush()); reset() and set_maps() drop it (every map is
        replaced: what was in flight is void)."""
        if self._needs_reset:
            raise RuntimeError("reset() must be called before tick()")
        if self._async is None:
            if self._async_nslots == 0:
                self.enable_async()
            if self._async is None:            # no asynchronous form: a lockstep step, nothing pending
                o, r, d, i = self.step(actions)
                if getattr(self, "_never_pending", None) is None:
                    self._never_pending = self._torch.zeros(self.num_envs, dtype=self._torch.uint8, device=self.device)
                return o, r, d, i, self._never_pending
        a = self._as_actions(actions)
        self._last_actions = a
        _lib.check(self._lib.pcgrl_step_async(self._handle, C.c_void_p(a.data_ptr()), int(pop_budget), self._stream()), "pcgrl_step_async")
        if self.strict_actions:
            self.check_status()
        b = self._bufs
        decode = self._prob.decode_rows if self._prob.packed_rows else None
        info = InfoBatch(self._prob.info_keys, b["info"], self._max_iterations, self._max_changes, decode)
        return self._obs(), b["reward"], b["done"].view(self._torch.bool), info, self._async["pending"]

    def flush(self):
        """Finish every pending step (searches with an unbounded budget)."""
        if self._handle is not None:
            _lib.check(self._lib.pcgrl_async_flush(self._handle, self._stream()), "pcgrl_async_flush")

    def async_counters(self):
        """{actions taken, searches suspended, jobs finished from a slot, slot overflows, pops of the resumable searches} since
        enable_async() (synchronises)."""
        if self._async is None:
            return None
        c = self._async["counters"].cpu().numpy()
        return dict(consumed=int(c[8::8].sum()), suspended=int(c[1]), late=int(c[2]), overflow=int(c[3]), pops=int(c[4]))

    # ---- episode statistics (stable-baselines Monitor as wrapped around the reference env, utils.py:13-29)
    def enable_episode_stats(self, enable=True):
        """Keep per-environment episode return/length on the device (updated by the step kernels).  After a step,
        `episode_stats()` gives the values of the episodes that just ended (where done is set)."""
        if enable:
            if self._handle is None:
                self._allocate()
            if self._episode is None:
                torch, n, dev = self._torch, self.num_envs, self.device
                self._episode = OrderedDict(
                    ep_return=torch.zeros(n, dtype=torch.float64, device=dev), ep_length=torch.zeros(n, dtype=torch.int32, device=dev),
                    last_return=torch.zeros(n, dtype=torch.float64, device=dev), last_length=torch.zeros(n, dtype=torch.int32, device=dev))
                self._bind_episode_stats(zero=True)
        elif self._episode is not None:
            self._episode = None
            if self._handle is not None:
                _lib.check(self._lib.pcgrl_bind_episode_stats(self._handle, None, None, None, None, self._stream()), "pcgrl_bind_episode_stats")

    def _bind_episode_stats(self, zero=False):
        e = self._episode
        keep = None if zero else {k: v.clone() for k, v in e.items()}
        _lib.check(self._lib.pcgrl_bind_episode_stats(self._handle, e["ep_return"].data_ptr(), e["ep_length"].data_ptr(),
                                                     e["last_return"].data_ptr(), e["last_length"].data_ptr(), self._stream()),
                   "pcgrl_bind_episode_stats")
        if keep is not None:           # the ABI call zeroes the buffers
            for k, v in keep.items():
                e[k].copy_(v)

    def episode_stats(self):
        """-> dict of device tensors: running `ep_return`/`ep_length`, and `last_return`/`last_length` of the most
        recently finished episode of every environment (valid where the last step returned done)."""
        if self._episode is None:
            raise RuntimeError("call enable_episode_stats() first")
        return self._episode

    def set_maps(self, maps):
        """Overwrite every map (uint8 [N,H,W]) and recompute the current stats on the device."""
        if self._needs_reset:
            raise RuntimeError("reset() must be called before set_maps()")
        torch = self._torch
        m = torch.as_tensor(maps, device=self.device).to(torch.uint8).contiguous()
        assert tuple(m.shape) == tuple(self._bufs["map"].shape)
        self._last_maps = m
        _lib.check(self._lib.pcgrl_set_maps(self._handle, C.c_void_p(m.data_ptr()), self._stream()), "pcgrl_set_maps")

    def check_status(self):
        """Raise if a kernel flagged an unsupported case (synchronises)."""
        st = C.c_int32()
        _lib.check(self._lib.pcgrl_status(self._handle, self._stream(), C.byref(st)), "pcgrl_status")
        if st.value and self.strict_actions:       # report this call's problem only once: the word is sticky otherwise
            _lib.check(self._lib.pcgrl_clear_status(self._handle, self._stream()), "pcgrl_clear_status")
        if st.value & 1:
            raise RuntimeError("a level was outside the limits of a search kernel (a Sokoban level with more crates than the search takes -- 32 in the "
                               "compact searches, 256 in the general ones -- or more than 255 tiles / collected things of one kind in a packed statistics row)")
        if st.value & 2:
            raise IndexError("an action outside the action space was passed to step()/rollout() (it was clamped into range; "
                             "the reference raises IndexError or writes the bad value)")
        if st.value & 4:
            raise ValueError("set_maps() was given a tile id >= get_num_tiles() (it was clamped)")
        return st.value

    def profile(self, enable=True):
        """Record HIP events around every phase of step() on the current stream."""
        _lib.check(self._lib.pcgrl_profile(self._handle, int(enable)), "pcgrl_profile")

    def profile_read(self):
        """-> ({phase: total ms}, steps) since profile(True); synchronises."""
        ms = (C.c_double * _lib.NPHASE)()
        steps = C.c_int32()
        _lib.check(self._lib.pcgrl_profile_read(self._handle, ms, C.byref(steps)), "pcgrl_profile_read")
        return dict(zip(_lib.PHASES, list(ms))), steps.value

    @property
    def stats(self):
        return self._prob.decode_rows(self._bufs["stats"])

    def state_dict(self):
        """Everything a batch of environments carries between steps (checkpoint / resume): maps, first maps,
        heatmaps, cursors, counters, current and start stats, last step outputs, binary tile probabilities, both MT19937
        rings with their cursors, and the episode statistics when they are on.  Device tensors (clones)."""
        self.flush()                       # (asynchronous ticks: a step in flight is finished first -- a checkpoint holds completed steps only)
        self._torch.cuda.synchronize(self.device)
        sd = {k: (v.clone() if v is not None else None) for k, v in self._bufs.items() if k not in ("scratch", "planes")}
        if self._episode is not None:
            sd.update({"episode_" + k: v.clone() for k, v in self._episode.items()})
        return sd

    def load_state_dict(self, sd):
        """Restore `state_dict()` of an environment batch with the same problem, representation, size and parameters.
        The derived state (row planes, cached champion components) is rebuilt from the maps on the device."""
        if self._handle is None:
            raise RuntimeError("call reset() once before load_state_dict() (buffers are allocated lazily)")
        for k, v in sd.items():
            if v is None or k.startswith("episode_"):
                continue
            dst = self._bufs[k]
            if tuple(dst.shape) != tuple(v.shape) or dst.dtype != v.dtype:
                raise ValueError("state_dict entry %r does not fit this environment: %s %s vs %s %s" % (k, tuple(v.shape), v.dtype, tuple(dst.shape), dst.dtype))
            dst.copy_(v)
        if any(k.startswith("episode_") for k in sd):
            self.enable_episode_stats()
            for k, v in self._episode.items():
                v.copy_(sd["episode_" + k])
        stats = self._bufs["stats"].clone()
        self.set_maps(self._bufs["map"].clone())       # planes + champion cache from the maps; recomputes the current stats ...
        self._torch.cuda.synchronize(self.device)
        n = len(self._prob.stat_keys)
        if not self._torch.equal(self._bufs["stats"][:, :n], stats[:, :n]):   # ... which must be the ones that were saved
            raise ValueError("state_dict is inconsistent: the saved stats are not the stats of the saved maps")

    def set_graphics(self, graphics):
        """Tile pictures for render(): None = the reference's grey fallback (the default), "drawn" = the package's own
        pictures (envs/tile_art.py), or a dict {tile name: tile_size x tile_size x 3 image} -- the reference's `_graphics`."""
        if graphics is None:
            self._graphics = None
            return
        if isinstance(graphics, str):
            if graphics != "drawn":
                raise ValueError("unknown graphics %r" % (graphics,))
            from .tile_art import make_graphics
            graphics = make_graphics(self._prob.tiles, int(self._prob._tile_size))
        missing = [t for t in self._prob.tiles if t not in graphics]
        if missing:
            raise KeyError("no picture for tiles %r" % (missing,))
        self._graphics = dict(graphics)

    def render(self, mode="rgb_array", index=0):
        """Image of environment `index` with the reference's layout (pcgrl_env.py:161-175): 16-pixel tiles, a one-tile
        border of the border tile (problem.py:142-156), a two-pixel red frame on the cursor cell for narrow/turtle
        (narrow_rep.py:128-142).  The tile pictures are the reference's *fallback* graphics -- grey level
        i*255/len(tiles), problem.py:136-140 -- its PNG assets are not part of this package.  Host side, one
        device->host map copy; returns a PIL image when PIL is importable (like the reference), else an ndarray."""
        if mode not in ("rgb_array", "human"):
            raise ValueError("unsupported render mode %r" % (mode,))
        tiles = self._prob.tiles
        ts = int(self._prob._tile_size)
        bx, by = self._prob._border_size
        m = self._bufs["map"][index].cpu().numpy()
        h, w = m.shape
        full = np.full((h + 2 * by, w + 2 * bx), tiles.index(self._prob._border_tile), dtype=np.int64)
        full[by:by + h, bx:bx + w] = m
        gfx = getattr(self, "_graphics", None)
        if gfx is None:
            grey = np.array([int(i * 255 / len(tiles)) for i in range(len(tiles))], dtype=np.uint8)
            img = np.repeat(np.repeat(grey[full], ts, axis=0), ts, axis=1)
            img = np.stack([img, img, img], -1)
        else:                                             # Problem.render with a `_graphics` dict (problem.py:128-156)
            pal = np.stack([np.asarray(gfx[t], dtype=np.uint8)[:ts, :ts, :3] for t in tiles])       # [tile, ts, ts, 3]
            img = pal[full].transpose(0, 2, 1, 3, 4).reshape(full.shape[0] * ts, full.shape[1] * ts, 3).copy()
        if self._rep.has_pos:
            x, y = [int(v) for v in self._bufs["pos"][index].cpu().numpy()]
            y0, x0 = (y + by) * ts, (x + bx) * ts
            cell = img[y0:y0 + ts, x0:x0 + ts]
            for sl in (np.s_[:2, :], np.s_[-2:, :], np.s_[:, :2], np.s_[:, -2:]):
                cell[sl] = (255, 0, 0)
        try:
            from PIL import Image
            return Image.fromarray(img, "RGB")
        except ImportError:
            return img

    def close(self):
        self._free()

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass
