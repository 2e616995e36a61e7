"""`make_vec_envs` with the reference's call shape (utils.py:60-71): where the reference builds `n_cpu`
SubprocVecEnv workers of wrapped single environments, this returns ONE batched, wrapped environment of
`n_cpu` lockstep environments on the GPU.  It exposes the VecEnv surface a trainer uses: `num_envs`,
`observation_space`, `action_space`, `reset()`, `step(actions)`, `step_async/step_wait`, `close()`;
observations are one uint8 image tensor [n_cpu, h, w, depth] (the reference's wrapped observation), done
environments are reset inside `step` like SubprocVecEnv does.

`log_dir`, `render` and `max_step` belong to the trainer's RenderMonitor (utils.py:13-29) and are accepted but
ignored: monitoring/rendering is outside the accelerated path.
"""
import numpy as np

from . import spaces
from .wrappers import ActionMapImagePCGRLWrapper, CroppedImagePCGRLWrapper


class BatchedVecEnv:
    def __init__(self, wrapped, image_shape, n_actions):
        self.env = wrapped
        self.num_envs = wrapped.num_envs
        self.observation_space = spaces.Box(low=0, high=255, shape=image_shape, dtype=np.uint8)
        self.action_space = spaces.Discrete(n_actions) if np.ndim(n_actions) == 0 else spaces.MultiDiscrete(n_actions)
        self._pending = None

    def reset(self):
        return self.env.reset()

    def step(self, actions):
        return self.env.step(actions)

    def step_async(self, actions):
        self._pending = self.env.step(actions)

    def step_wait(self):
        return self._pending

    def seed(self, seed=None):
        return self.env.seed(seed)

    def close(self):
        self.env.close()


def make_vec_envs(env_name, representation, log_dir=None, n_cpu=1, seed=None, device=None, **kwargs):
    kwargs = dict(kwargs)
    kwargs.pop("render", None)
    kwargs.pop("max_step", None)
    crop_size = kwargs.pop("cropped_size", 28)
    if representation == "wide":          # utils.py:49-50
        w = ActionMapImagePCGRLWrapper(env_name, num_envs=n_cpu, seed=seed, device=device, **kwargs)
        p = w.pcgrl_env._prob
        h_, w_, d_ = int(p._height), int(p._width), (w.pcgrl_env.get_num_tiles() if w.one_hot else 1)
        return BatchedVecEnv(w, (h_, w_, d_), h_ * w_ * w.pcgrl_env.get_num_tiles())
    w = CroppedImagePCGRLWrapper(env_name, crop_size, num_envs=n_cpu, seed=seed, device=device, **kwargs)   # utils.py:51-53
    d_ = w.pcgrl_env.get_num_tiles() if w.one_hot else 1
    a = w.pcgrl_env.action_space
    return BatchedVecEnv(w, (crop_size, crop_size, d_), a.n if hasattr(a, "n") else [int(v) for v in a.nvec])
