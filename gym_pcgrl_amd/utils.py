"""`make_vec_envs` with the reference's call shape (utils.py:60-71): where the reference builds `n_cpu`
SubprocVecEnv workers of wrapped single environments, this returns ONE batched, wrapped environment of
`n_cpu` lockstep environments on the GPU.  It exposes the VecEnv surface a trainer uses: `num_envs`,
`observation_space`, `action_space`, `reset()`, `step(actions)`, `step_async/step_wait`, `close()`;
observations are one uint8 image tensor [n_cpu, h, w, depth] (the reference's wrapped observation), done
environments are reset inside `step` like SubprocVecEnv does.

The reference wraps every worker in RenderMonitor (a stable-baselines Monitor, utils.py:13-29) whose product is
`info["episode"] = {"r": return, "l": length}` on the step that ends an episode.  Here the step kernels keep those
sums per environment (`BatchedPcgrlEnv.enable_episode_stats`); `monitor=True` (the default when `log_dir` is given,
as in the reference) adds the same `episode` entry to the infos of finished environments and `episode_stats()`
returns them as device tensors without any host round trip.  `render` and `max_step` are accepted and ignored
(rendering is outside the accelerated path; `max_step` is unused by the reference wrapper too, utils.py:21-29).
"""
import numpy as np

from . import spaces
from .wrappers import ActionMapImagePCGRLWrapper, CroppedImagePCGRLWrapper


class BatchedVecEnv:
    """`async_ticks` (a pop budget per search and call, e.g. 64; sokoban / mdungeon / ddave): step() is an asynchronous TICK
    (BatchedPcgrlEnv.tick, include/pcgrl_hip.h pcgrl_step_async) -- the reference's deployment runs every environment in a worker
    of its own (utils.py:60-71), so one long A* never holds the others up; here an environment whose search is not finished within
    the budget sits the following calls out (its action is ignored) until its step completes.  The call still returns the usual
    four values; `infos.took` / `infos.fresh` (bool [N] device tensors) say which environments took their action in this call and
    which completed a step in it -- reward is 0 and done False where `fresh` is not set, so sums over calls stay right, and per
    environment the (took action -> fresh outcome) pairs are bitwise the lockstep transitions.  A problem without an asynchronous
    form steps in lockstep with both masks all true."""

    def __init__(self, wrapped, image_shape, n_actions, monitor=False, async_ticks=None):
        self.env = wrapped
        self.num_envs = wrapped.num_envs
        self.monitor = bool(monitor)
        if self.monitor:
            wrapped.pcgrl_env.enable_episode_stats()
        self.observation_space = spaces.Box(low=0, high=255, shape=image_shape, dtype=np.uint8)
        self.action_space = spaces.Discrete(n_actions) if np.ndim(n_actions) == 0 else spaces.MultiDiscrete(n_actions)
        self._pending = None
        self.async_ticks = int(async_ticks) if async_ticks else None
        if self.async_ticks is not None and self.async_ticks < 1:
            raise ValueError("async_ticks: a pop budget >= 1")
        self._sitting_out = None           # bool [N]: environments whose search is suspended (they ignore the next action)

    def reset(self):
        self._sitting_out = None           # (reset() drops what was pending)
        return self.env.reset()

    def _tick(self, actions):
        e = self.env.pcgrl_env
        torch = e._torch
        if self._sitting_out is None:
            self._sitting_out = torch.zeros(self.num_envs, dtype=torch.bool, device=e.device)
        took = ~self._sitting_out
        obs, rew, done, infos, pend = self.env.tick(actions, pop_budget=self.async_ticks)
        self._sitting_out = pend != 0
        fresh = ~self._sitting_out
        return obs, torch.where(fresh, rew, torch.zeros_like(rew)), done & fresh, _TickInfos(infos, took, fresh)

    def step(self, actions):
        out = self._tick(actions) if self.async_ticks is not None else self.env.step(actions)
        if not self.monitor:
            return out
        obs, rew, done, infos = out
        idx = done.nonzero().flatten()
        if idx.numel():                      # Monitor.step: ep_info = {"r": round(sum(rewards), 6), "l": len(rewards)}
            st = self.env.pcgrl_env.episode_stats()
            r = st["last_return"][idx].cpu().numpy()
            l = st["last_length"][idx].cpu().numpy()
            infos = _EpisodeInfos(infos, {int(i): {"r": round(float(a), 6), "l": int(b)} for i, a, b in zip(idx.cpu().numpy(), r, l)})
        return obs, rew, done, infos

    def episode_stats(self):
        """Device tensors of the in-kernel episode statistics (see BatchedPcgrlEnv.episode_stats)."""
        return self.env.pcgrl_env.episode_stats()

    def step_async(self, actions):
        self._pending = self.step(actions)

    def step_wait(self):
        return self._pending

    def seed(self, seed=None):
        return self.env.seed(seed)

    def close(self):
        self.env.close()


class _TickInfos:
    """The batched info object of an asynchronous tick plus the masks `took` / `fresh` (BatchedVecEnv, async_ticks)."""

    def __init__(self, base, took, fresh):
        self._base, self.took, self.fresh = base, took, fresh

    def __getitem__(self, key):
        return self._base[key]

    def __getattr__(self, name):
        return getattr(self._base, name)


class _EpisodeInfos:
    """The batched info object of the step plus Monitor's per-environment `episode` entries."""

    def __init__(self, base, episodes):
        self._base, self.episodes = base, episodes

    def __getitem__(self, key):
        return self._base[key]

    def __getattr__(self, name):
        return getattr(self._base, name)

    def to_list(self):
        out = self._base.to_list()
        for i, ep in self.episodes.items():
            out[i]["episode"] = ep
        return out


def make_vec_envs(env_name, representation, log_dir=None, n_cpu=1, seed=None, device=None, monitor=None, async_ticks=None, **kwargs):
    """utils.py:60-71.  `async_ticks=<pop budget>`: the returned environment steps the search problems asynchronously (BatchedVecEnv)."""
    kwargs = dict(kwargs)
    monitor = (log_dir is not None) if monitor is None else monitor
    kwargs.pop("render", None)
    kwargs.pop("max_step", None)
    crop_size = kwargs.pop("cropped_size", 28)
    if representation == "wide":          # utils.py:49-50
        w = ActionMapImagePCGRLWrapper(env_name, num_envs=n_cpu, seed=seed, device=device, **kwargs)
        p = w.pcgrl_env._prob
        h_, w_, d_ = int(p._height), int(p._width), (w.pcgrl_env.get_num_tiles() if w.one_hot else 1)
        return BatchedVecEnv(w, (h_, w_, d_), h_ * w_ * w.pcgrl_env.get_num_tiles(), monitor, async_ticks)
    w = CroppedImagePCGRLWrapper(env_name, crop_size, num_envs=n_cpu, seed=seed, device=device, **kwargs)   # utils.py:51-53
    d_ = w.pcgrl_env.get_num_tiles() if w.one_hot else 1
    a = w.pcgrl_env.action_space
    return BatchedVecEnv(w, (crop_size, crop_size, d_), a.n if hasattr(a, "n") else [int(v) for v in a.nvec], monitor, async_ticks)
