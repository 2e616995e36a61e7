"""PcgrlEnv: the reference's single-environment gym.Env surface (pcgrl_env.py) as an N=1 view of
BatchedPcgrlEnv.  Same constructor, same old-style 4-tuple step(), numpy observations
(OrderedDict pos/map/heatmap with the reference's dtypes), no auto-reset -- so the reference's
wrappers (wrappers.py: they look for "PcgrlEnv" in str(type(env)), :11) and scripts drop in.

Every call crosses PCIe (a 1-environment launch plus a device->host copy), so this class is for
compatibility and parity tests; throughput work belongs on BatchedPcgrlEnv.
"""
from collections import OrderedDict

import numpy as np

from .. import gym_compat
from .batched_env import BatchedPcgrlEnv


class PcgrlEnv(gym_compat.env_base()):
    """A gym.Env when a gym with the four-tuple API is importable (pcgrl_env.py:14; gym_compat.find_gym), a plain class otherwise."""
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, prob="binary", rep="narrow", device=None):
        self._batched = BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=1, device=device, auto_reset=False)
        self._prob = self._batched._prob
        self._rep = self._batched._rep
        self.viewer = None
        self._pinned, self._action_buf = {}, None
        self._sync_spaces()

    def _sync_spaces(self):
        self.action_space = gym_compat.convert_space(self._batched.action_space)
        self.observation_space = gym_compat.convert_space(self._batched.observation_space)

    # attributes the reference exposes and scripts poke at
    _max_changes = property(lambda s: s._batched._max_changes)
    _max_iterations = property(lambda s: s._batched._max_iterations)

    def seed(self, seed=None):
        return [self._batched.seed(seed)[0]]

    def get_border_tile(self):
        return self._batched.get_border_tile()

    def get_num_tiles(self):
        return self._batched.get_num_tiles()

    def adjust_param(self, **kwargs):
        self._batched.adjust_param(**kwargs)
        self._sync_spaces()

    def _to_host(self, tensors):
        """The step's outputs in ONE host round trip: every tensor is copied into its own pinned host buffer on the launch
        stream (no wait in between), then the stream is waited for once.  (One `.cpu()` / `.item()` per output -- six of them --
        was six synchronisations per step: bench.py's `n1_facade` leg.)"""
        torch = self._batched._torch
        out = []
        for name, t in tensors:
            buf = self._pinned.get(name)
            if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                buf = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                self._pinned[name] = buf
            buf.copy_(t, non_blocking=True)
            out.append(buf)
        torch.cuda.current_stream(self._batched.device).synchronize()
        return [b.numpy() for b in out]

    def _np_obs(self, host):
        o = OrderedDict()
        if "pos" in host:
            o["pos"] = host["pos"][0].astype(np.uint8)
        o["map"] = host["map"][0].astype(np.uint8)                  # (astype copies: the pinned buffers are reused by the next step)
        o["heatmap"] = host["heatmap"][0].astype(np.float64)        # the reference's heatmap is float64
        return o

    def reset(self):
        obs = self._batched.reset()
        names = list(obs)
        return self._np_obs(dict(zip(names, self._to_host([(k, obs[k]) for k in names]))))

    def step(self, action):
        a = self._action_buf
        if a is None:        # the action goes up through a pinned staging buffer as well (no pageable-memory copy per step)
            torch = self._batched._torch
            a = self._action_buf = torch.empty((1, self._batched._rep.action_width()), dtype=torch.int32).pin_memory()
        a.numpy()[...] = np.asarray(action, dtype=np.int32).reshape(1, -1)
        obs, reward, done, info = self._batched.step(a.to(self._batched.device, non_blocking=True))
        names = list(obs)
        host = self._to_host([(k, obs[k]) for k in names] + [("reward", reward), ("done", done), ("info", info.table)])
        h = dict(zip(names + ["reward", "done", "info"], host))
        r = float(h["reward"][0])
        if self._prob.name != "sokoban" and r == int(r):
            r = int(r)    # binary/zelda rewards are ints in the reference, sokoban's is a float
        return self._np_obs(h), r, bool(h["done"][0]), self._info_dict(info, h["info"])

    def _info_dict(self, info, table):
        """InfoBatch.to_list()[0] from the host copy of the info table."""
        if info._decode is not None:
            torch = self._batched._torch
            vals = info._decode(torch.from_numpy(table), info.keys).numpy()
        else:
            vals = table
        d = {k: int(vals[0][i]) for i, k in enumerate(info.keys)}
        d["iterations"], d["changes"] = int(table[0][8]), int(table[0][9])
        d["max_iterations"], d["max_changes"] = info.max_iterations, info.max_changes
        return d

    def render(self, mode="human"):
        """pcgrl_env.py:161-175.  'rgb_array' returns the image; 'human' has no viewer here (gym's classic_control
        viewer is not a dependency) and returns the image as well."""
        return self._batched.render("rgb_array", 0)

    def set_graphics(self, graphics):
        """None (grey fallback), "drawn" (envs/tile_art.py) or the reference's {tile name: image} dict."""
        self._batched.set_graphics(graphics)

    def close(self):
        self._batched.close()
