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

    def _np_obs(self, obs):
        o = OrderedDict()
        if "pos" in obs:
            o["pos"] = obs["pos"][0].cpu().numpy().astype(np.uint8)
        o["map"] = obs["map"][0].cpu().numpy().astype(np.uint8)
        o["heatmap"] = obs["heatmap"][0].cpu().numpy().astype(np.float64)   # the reference's heatmap is float64
        return o

    def reset(self):
        return self._np_obs(self._batched.reset())

    def step(self, action):
        a = np.asarray(action, dtype=np.int32).reshape(1, -1)
        obs, reward, done, info = self._batched.step(a)
        r = float(reward[0].item())
        if self._prob.name != "sokoban" and r == int(r):
            r = int(r)    # binary/zelda rewards are ints in the reference, sokoban's is a float
        return self._np_obs(obs), r, bool(done[0].item()), info.to_list()[0]

    def render(self, mode="human"):
        """pcgrl_env.py:161-175.  'rgb_array' returns the image; 'human' has no viewer here (gym's classic_control
        viewer is not a dependency) and returns the image as well."""
        return self._batched.render("rgb_array", 0)

    def set_graphics(self, graphics):
        """None (grey fallback), "drawn" (envs/tile_art.py) or the reference's {tile name: image} dict."""
        self._batched.set_graphics(graphics)

    def close(self):
        self._batched.close()
