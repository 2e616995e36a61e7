"""Throw-away `gym` stand-in so the *unmodified* reference imports in this container.

Used ONLY by tests/golden/make_golden.py (fixture generation, run where /root/reference
exists).  Nothing in the product, the `-m gpu` tests, smoke() or bench.py imports this.

The reference needs exactly: gym.Env, gym.Wrapper, gym.spaces.{Box,Discrete,MultiDiscrete,Dict},
gym.utils.seeding.np_random, gym.envs.registration.register, gym.make.
`seeding.np_random` is the gym<=0.21 algorithm (sha512 hash -> init_by_array), restated in
gym_pcgrl_amd/seeding.py.
"""
import importlib
import sys
import types

from gym_pcgrl_amd import seeding as _seeding
from gym_pcgrl_amd import spaces as _spaces


class Env:
    metadata = {}
    action_space = None
    observation_space = None

    def seed(self, seed=None):
        return [seed]

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = getattr(env, "action_space", None)
        self.observation_space = getattr(env, "observation_space", None)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kw):
        return self.env.reset(**kw)


_REGISTRY = {}


def register(id, entry_point=None, kwargs=None, **_):
    _REGISTRY[id] = (entry_point, kwargs or {})


def make(id, **kw):
    entry_point, kwargs = _REGISTRY[id]
    mod, cls = entry_point.split(":")
    args = dict(kwargs)
    args.update(kw)
    return getattr(importlib.import_module(mod), cls)(**args)


def install(reference_root="/root/reference"):
    """Put the stub in sys.modules and the reference on sys.path (no bytecode is written)."""
    sys.dont_write_bytecode = True
    gym = types.ModuleType("gym")
    gym.Env, gym.Wrapper, gym.make = Env, Wrapper, make
    gym.spaces = _spaces
    utils = types.ModuleType("gym.utils")
    utils.seeding = _seeding
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")
    registration.register = register
    envs.registration = registration
    gym.utils, gym.envs = utils, envs
    sys.modules.update({
        "gym": gym, "gym.spaces": _spaces, "gym.utils": utils, "gym.utils.seeding": _seeding,
        "gym.envs": envs, "gym.envs.registration": registration,
    })
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    return gym
