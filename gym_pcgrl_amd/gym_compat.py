"""gym integration, active only when a gym with the reference's API is importable (none is on the MI355X image).

"The reference's API" is gym <= 0.25: `reset() -> obs`, `step() -> (obs, reward, done, info)` (pcgrl_env.py:66-76,129-150).
gym >= 0.26 and gymnasium changed both (`(obs, info)`, five-tuple steps) and wrap every `make()` in checkers that reject the old
shapes, so this package does not register itself there -- it would only produce ids that fail at the first reset().

The reference registers '{prob}-{rep}-v0' for every problem x representation when it is imported
(gym_pcgrl/__init__.py:6-12), its PcgrlEnv is a gym.Env (envs/pcgrl_env.py:14), and its wrappers start from
`gym.make(game)` (wrappers.py:21-24).  With gym present `import gym_pcgrl_amd` does the same: the ids are registered with
entry point `gym_pcgrl_amd.envs:PcgrlEnv` (a gym.Env subclass then, spaces converted to the library's own classes), so the
reference's wrappers.py and scripts run on this package by importing it in place of `gym_pcgrl`.  `PcgrlVectorEnv` is the
batched environment behind the gym.vector.VectorEnv surface.
"""
import importlib
import sys


def _old_api(g):
    """True for a gym with four-tuple steps: version below 0.26 (a module without a version -- the test shim -- counts)."""
    v = getattr(g, "__version__", None)
    if v is None:
        return True
    try:
        major, minor = (int(x) for x in str(v).split(".")[:2])
    except ValueError:
        return False
    return (major, minor) < (0, 26)


def find_gym():
    """The gym module to integrate with: an already imported `gym` (this is how a test shim is found too), else an
    importable one; None when there is none or it has the new API (see the module docstring)."""
    g = sys.modules.get("gym")
    if g is None:
        try:
            g = importlib.import_module("gym")
        except ImportError:
            return None
    return g if _old_api(g) else None


def env_base():
    g = find_gym()
    return g.Env if g is not None and hasattr(g, "Env") else object


def vector_env_base():
    g = find_gym()
    vec = getattr(g, "vector", None) if g is not None else None
    return getattr(vec, "VectorEnv", object) if vec is not None else object


def convert_space(space):
    """Descriptor of gym_pcgrl_amd.spaces -> the gym library's own space class (identity without gym, and under a gym
    whose `spaces` module IS gym_pcgrl_amd.spaces, as in the test shim)."""
    from collections import OrderedDict

    from . import spaces as S
    g = find_gym()
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
        return gs.Dict(OrderedDict((k, convert_space(s)) for k, s in space.spaces.items()))
    return space


_registered = []


def _registered_ids(g):
    """The ids gym's registry already holds (set), or None when this gym does not expose them in a way we know."""
    try:
        reg = importlib.import_module(g.__name__ + ".envs.registration").registry
    except (ImportError, AttributeError):
        reg = getattr(getattr(g, "envs", None), "registry", None)
    if reg is None:
        return None
    specs = getattr(reg, "env_specs", reg)          # gym <= 0.21: EnvRegistry.env_specs; 0.22-0.25: a dict
    try:
        return set(specs.keys())
    except AttributeError:
        return None


def register_all(ids):
    """gym_pcgrl/__init__.py:6-12 for this package.  `ids`: {env id: (prob, rep)}.  Ids that something else registered
    already (the reference itself, or an earlier import) are left alone.  Returns the ids registered by this call."""
    g = find_gym()
    if g is None:
        return []
    try:
        register = importlib.import_module(g.__name__ + ".envs.registration").register
    except (ImportError, AttributeError):
        register = getattr(g, "register", None)
    if register is None:
        return []
    done = []
    taken = _registered_ids(g)
    gym_error = getattr(getattr(g, "error", None), "Error", None)
    for env_id, (prob, rep) in sorted(ids.items()):
        if env_id in _registered or (taken is not None and env_id in taken):
            continue                 # something else (the reference itself, an earlier import) owns the id: leave it alone
        try:
            register(id=env_id, entry_point="gym_pcgrl_amd.envs:PcgrlEnv", kwargs={"prob": prob, "rep": rep})
        except Exception as ex:
            # gym.error.Error("Cannot re-register id: ...") from a registry whose contents could not be listed above: keep the
            # existing registration.  Anything that is not gym's own error class is a bug and propagates.
            if gym_error is None or not isinstance(ex, gym_error):
                raise
            continue
        _registered.append(env_id)
        done.append(env_id)
    return done
