"""PcgrlVectorEnv: the batched environment behind the gym.vector.VectorEnv surface (gym <= 0.21, the reference's era:
`reset() -> obs`, `step(actions) -> (obs, rewards, dones, infos)` with 4-tuples, `step_async/step_wait`,
`reset_async/reset_wait`, `seed`, `close`, `num_envs`, `single_observation_space`, `single_action_space`,
batched `observation_space` / `action_space`).  A subclass of gym.vector.VectorEnv when gym / gymnasium provides one.

Observations are dicts of arrays with a leading environment axis -- numpy copies by default (the gym.vector contract), or
the live device tensors with `to_numpy=False` (zero copy; overwritten by the next step).  A done environment is reset
inside step() and the observation returned for it is the first one of its next episode, as gym.vector does; `infos` is
the reference's list of per-environment dicts (pcgrl_env.py:144-148), built lazily (`infos` supports len / [] / iter) from a
host snapshot of the step's info table (with `to_numpy=False`: from the live device table, valid until the next step).
"""
from collections import OrderedDict

import numpy as np

from . import gym_compat, spaces


def _batch_space(space, n):
    if isinstance(space, spaces.Dict):
        return spaces.Dict(OrderedDict((k, _batch_space(s, n)) for k, s in space.spaces.items()))
    if isinstance(space, spaces.Box):
        return spaces.Box(low=np.broadcast_to(space.low, (n,) + space.shape), high=np.broadcast_to(space.high, (n,) + space.shape), dtype=space.dtype)
    if isinstance(space, spaces.Discrete):
        return spaces.MultiDiscrete(np.full((n,), space.n, dtype=np.int64))
    if isinstance(space, spaces.MultiDiscrete):
        return spaces.MultiDiscrete(np.broadcast_to(space.nvec, (n,) + space.nvec.shape).copy())
    return space


class LazyInfos:
    """The per-environment info dicts of a step, materialised on first use (one device -> host copy)."""

    def __init__(self, batch):
        self.batch, self._list = batch, None

    def _get(self):
        if self._list is None:
            self._list = self.batch.to_list()
        return self._list

    def __len__(self):
        return self.batch.table.shape[0]

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())


class PcgrlVectorEnv(gym_compat.vector_env_base()):
    def __init__(self, env_id_or_env, num_envs=None, seed=None, device=None, to_numpy=True, **adjust):
        from . import make_batched
        if isinstance(env_id_or_env, str):
            if num_envs is None:
                raise ValueError("num_envs is required with an environment id")
            self.env = make_batched(env_id_or_env, num_envs=num_envs, seed=seed, **({"device": device} if device else {}))
        else:
            self.env = env_id_or_env
        if adjust:
            self.env.adjust_param(**adjust)
        self.to_numpy = bool(to_numpy)
        self.num_envs = self.env.num_envs
        self._sync_spaces()
        self.closed = False
        self._actions = None

    def _sync_spaces(self):
        e = self.env
        self.single_observation_space = gym_compat.convert_space(e.single_observation_space)
        self.single_action_space = gym_compat.convert_space(e.single_action_space)
        self.observation_space = gym_compat.convert_space(_batch_space(e.single_observation_space, self.num_envs))
        self.action_space = gym_compat.convert_space(_batch_space(e.single_action_space, self.num_envs))

    def adjust_param(self, **kwargs):
        self.env.adjust_param(**kwargs)
        self._sync_spaces()

    def get_border_tile(self):
        return self.env.get_border_tile()

    def get_num_tiles(self):
        return self.env.get_num_tiles()

    def seed(self, seeds=None):
        return self.env.seed(seeds)

    def _obs(self, obs):
        if not self.to_numpy:
            return obs
        out = OrderedDict()
        for k, v in obs.items():
            a = v.cpu().numpy()
            out[k] = a.astype(np.float64) if k == "heatmap" else a      # the reference's heatmap is float64 (pcgrl_env.py:35)
        return out

    def reset_async(self):
        pass

    def reset_wait(self, **kwargs):
        return self._obs(self.env.reset())

    def reset(self):
        self.reset_async()
        return self.reset_wait()

    def step_async(self, actions):
        self._actions = actions

    def step_wait(self, **kwargs):
        obs, rew, done, info = self.env.step(self._actions)
        self._actions = None
        if self.to_numpy:
            # host copies all round: the info table is snapshotted now (the dicts are still built on first use), so an `infos`
            # kept across the next step() / reset() still describes THIS step
            from .envs.batched_env import InfoBatch
            snap = InfoBatch(info.keys, info.table.cpu(), info.max_iterations, info.max_changes, info._decode)
            return self._obs(obs), rew.cpu().numpy(), done.cpu().numpy(), LazyInfos(snap)
        return obs, rew, done, LazyInfos(info)      # zero copy: a live view, overwritten by the next step

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close_extras(self, **kwargs):
        self.env.close()

    def close(self, **kwargs):
        if not self.closed:
            self.close_extras(**kwargs)
            self.closed = True

    def __del__(self):
        pass
