"""The new gym API (gym >= 0.26, gymnasium) over the same environments: `reset(seed=None, options=None) -> (obs, info)` and
`step(a) -> (obs, reward, terminated, truncated, info)`.

The reference is written against the old API (pcgrl_env.py:66-76,129-150: `reset() -> obs`, four-tuple `step`), and
`gym_compat` integrates with a gym of that era only.  Code that lives on gymnasium gets the two adapters below; they wrap the
old-API classes of this package (`PcgrlEnv`, `PcgrlVectorEnv`) and change nothing but the call shapes:

  * `done` is split by the rule gym's own `step_api_compatibility` shim uses for time limits: the reference's
    `done = episode_over or changes >= max_changes or iterations >= max_iterations` (pcgrl_env.py:143) becomes
    `truncated = done and (changes >= max_changes or iterations >= max_iterations)` -- the budget ran out --
    and `terminated = done and not truncated` -- the problem's own goal (`get_episode_over`) ended the episode.  When both
    hold on the same step the reference cannot tell them apart either; the budget wins (`truncated`), so that a learner that
    bootstraps on truncation never treats a budget end as a goal.
  * `reset(seed=s)` seeds before resetting (`PcgrlEnv.seed`, pcgrl_env.py:54-57); `options` may carry `adjust_param` kwargs
    under the key "adjust_param".
  * the vector adapter keeps gym.vector's autoreset contract of the old class (the observation returned for a finished
    environment is the first one of its next episode).

With gymnasium (or gym >= 0.26) importable, `register_all` registers '{prob}-{rep}-v0' there with entry point
`gym_pcgrl_amd.gymnasium_compat:NewApiPcgrlEnv`; neither is on the MI355X image, so the tests run it under a stand-in module.
"""
import importlib
import sys

import numpy as np


def find_gymnasium():
    """gymnasium if importable (or already in sys.modules: the test stand-in), else a gym with the new API, else None."""
    for name in ("gymnasium", "gym"):
        g = sys.modules.get(name)
        if g is None:
            try:
                g = importlib.import_module(name)
            except ImportError:
                continue
        if name == "gymnasium":
            return g
        from .gym_compat import _old_api
        if not _old_api(g):
            return g
    return None


def split_done(done, changes, iterations, max_changes, max_iterations):
    """(terminated, truncated) from the reference's `done` and the counters of the step's info (pcgrl_env.py:143-148).
    Works on Python scalars, numpy arrays and torch tensors alike."""
    budget = (changes >= max_changes) | (iterations >= max_iterations)
    truncated = done & budget
    terminated = done & ~truncated if hasattr(done, "dtype") else (done and not truncated)
    return terminated, truncated


def _env_base():
    g = find_gymnasium()
    return g.Env if g is not None and hasattr(g, "Env") else object


class NewApiPcgrlEnv(_env_base()):
    """`PcgrlEnv` (envs/pcgrl_env.py) behind the five-tuple API.  The class name still contains 'PcgrlEnv'
    (wrappers.py:11 finds the environment that way)."""
    metadata = {"render_modes": ["human", "rgb_array"], "render.modes": ["human", "rgb_array"]}

    def __init__(self, prob="binary", rep="narrow", device=None, render_mode=None):
        from .envs import PcgrlEnv
        self._env = PcgrlEnv(prob=prob, rep=rep, device=device)
        self.render_mode = render_mode
        self._sync_spaces()

    def _sync_spaces(self):
        self.action_space = _convert_space(self._env._batched.action_space)
        self.observation_space = _convert_space(self._env._batched.observation_space)

    # the reference surface that is not part of either gym API
    def adjust_param(self, **kwargs):
        self._env.adjust_param(**kwargs)
        self._sync_spaces()

    def get_border_tile(self):
        return self._env.get_border_tile()

    def get_num_tiles(self):
        return self._env.get_num_tiles()

    _prob = property(lambda s: s._env._prob)
    _rep = property(lambda s: s._env._rep)
    _max_changes = property(lambda s: s._env._max_changes)
    _max_iterations = property(lambda s: s._env._max_iterations)

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._env.seed(seed)
        if options and options.get("adjust_param"):
            self.adjust_param(**options["adjust_param"])
        obs = self._env.reset()
        b = self._env._batched
        return obs, {"iterations": 0, "changes": 0, "max_iterations": b._max_iterations, "max_changes": b._max_changes}

    def step(self, action):
        obs, reward, done, info = self._env.step(action)
        terminated, truncated = split_done(bool(done), info["changes"], info["iterations"], info["max_changes"], info["max_iterations"])
        return obs, reward, bool(terminated), bool(truncated), info

    def render(self):
        return self._env.render(self.render_mode or "rgb_array")

    def close(self):
        self._env.close()


def _vector_base():
    g = find_gymnasium()
    vec = getattr(g, "vector", None) if g is not None else None
    return getattr(vec, "VectorEnv", object) if vec is not None else object


class NewApiPcgrlVectorEnv(_vector_base()):
    """`PcgrlVectorEnv` (vector.py) behind gymnasium.vector's call shapes: `reset(seed=None, options=None) -> (obs, infos)`,
    `step(actions) -> (obs, rewards, terminations, truncations, infos)`; `infos` is a dict of arrays (gymnasium's convention) with
    the reference's info keys (pcgrl_env.py:144-148) -- device tensors with `to_numpy=False`."""

    def __init__(self, env_id_or_env, num_envs=None, seed=None, device=None, to_numpy=True, **adjust):
        from .vector import PcgrlVectorEnv
        # the batch resets a finished environment inside the step that finished it (the observation returned with done is the first
        # one of the next episode): gymnasium >= 1.0 wants that declared
        g = find_gymnasium()
        mode = getattr(getattr(getattr(g, "vector", None), "AutoresetMode", None), "SAME_STEP", None) if g is not None else None
        self.metadata = dict(getattr(type(self), "metadata", None) or {})
        if mode is not None:
            self.metadata["autoreset_mode"] = mode
        self._v = PcgrlVectorEnv(env_id_or_env, num_envs=num_envs, seed=seed, device=device, to_numpy=to_numpy, **adjust)
        self.num_envs = self._v.num_envs
        self.to_numpy = self._v.to_numpy
        self.closed = False
        self._sync_spaces()

    def _sync_spaces(self):
        v = self._v
        self.single_observation_space = _convert_space(v.env.single_observation_space)
        self.single_action_space = _convert_space(v.env.single_action_space)
        from .vector import _batch_space
        self.observation_space = _convert_space(_batch_space(v.env.single_observation_space, self.num_envs))
        self.action_space = _convert_space(_batch_space(v.env.single_action_space, self.num_envs))

    def adjust_param(self, **kwargs):
        self._v.adjust_param(**kwargs)
        self._sync_spaces()

    def get_border_tile(self):
        return self._v.get_border_tile()

    def get_num_tiles(self):
        return self._v.get_num_tiles()

    def _infos(self, batch):
        keys = list(batch.keys) + ["iterations", "changes"]
        conv = (lambda t: t.cpu().numpy()) if self.to_numpy else (lambda t: t)
        out = {k: conv(batch[k]) for k in keys}
        out["max_iterations"], out["max_changes"] = batch.max_iterations, batch.max_changes
        return out

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._v.seed(seed)
        if options and options.get("adjust_param"):
            self.adjust_param(**options["adjust_param"])
        obs = self._v.reset()
        return obs, {}

    def step(self, actions):
        obs, rew, done, infos = self._v.step(actions)
        b = infos.batch
        it, ch = b["iterations"], b["changes"]
        if self.to_numpy:
            it, ch = it.cpu().numpy(), ch.cpu().numpy()
            done = np.asarray(done, dtype=bool)
        terminated, truncated = split_done(done, ch, it, b.max_changes, b.max_iterations)
        return obs, rew, terminated, truncated, self._infos(b)

    def close(self, **kwargs):
        if not self.closed:
            self._v.close()
            self.closed = True


def _convert_space(space):
    """gym_pcgrl_amd.spaces descriptor -> the new-API library's own space class (identity without one)."""
    from collections import OrderedDict

    from . import spaces as S
    g = find_gymnasium()
    gs = getattr(g, "spaces", None) if g is not None else None
    if gs is None or gs is S:
        return space
    if isinstance(space, S.Discrete):
        return gs.Discrete(space.n)
    if isinstance(space, S.MultiDiscrete):
        return gs.MultiDiscrete(space.nvec)
    if isinstance(space, S.Box):
        return gs.Box(low=space.low, high=space.high, dtype=space.dtype.type)
    if isinstance(space, S.Dict):
        return gs.Dict(OrderedDict((k, _convert_space(s)) for k, s in space.spaces.items()))
    return space


_registered = []


def register_all(ids):
    """Register {env id: (prob, rep)} with gymnasium / gym >= 0.26 (entry point NewApiPcgrlEnv).  Ids the registry already
    holds are left alone.  Returns the ids registered by this call; [] without such a library."""
    g = find_gymnasium()
    if g is None:
        return []
    register = getattr(g, "register", None)
    if register is None:
        try:
            register = importlib.import_module(g.__name__ + ".envs.registration").register
        except (ImportError, AttributeError):
            return []
    registry = getattr(getattr(g, "envs", None), "registry", None)
    if registry is None:
        registry = getattr(g, "registry", None)
    done = []
    for env_id, (prob, rep) in sorted(ids.items()):
        if env_id in _registered or (registry is not None and env_id in registry):
            continue
        register(id=env_id, entry_point="gym_pcgrl_amd.gymnasium_compat:NewApiPcgrlEnv", kwargs={"prob": prob, "rep": rep})
        _registered.append(env_id)
        done.append(env_id)
    return done
