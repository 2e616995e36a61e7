"""gym_pcgrl_amd: MI355X-native batched PCGRL environment (the step()/reset() hot path of
amidos2006/gym-pcgrl as hand-written HIP kernels for gfx950, behind the reference's env surface).

    import gym_pcgrl_amd
    env = gym_pcgrl_amd.make("binary-narrow-v0")                       # reference-style single env
    venv = gym_pcgrl_amd.make_batched("binary-narrow-v0", num_envs=65536)  # one GPU, lockstep batch

Ids follow gym_pcgrl/__init__.py:6-12: '{prob}-{rep}-v0' for the in-scope problems
(binary, zelda, sokoban) x all six representations (narrow, narrowcast, narrowmulti, wide, turtle, turtlecast).
"""
__version__ = "0.1.0"

_ENV_IDS = {}


def _register_all():
    from .envs.problems import PROBLEMS
    from .envs.representations import REPRESENTATIONS
    for prob in PROBLEMS:
        for rep in REPRESENTATIONS:
            _ENV_IDS["%s-%s-v0" % (prob, rep)] = (prob, rep)


def registered_ids():
    if not _ENV_IDS:
        _register_all()
    return sorted(_ENV_IDS)


def register_with_gym():
    """Register every id with gym when one with the reference's four-tuple API is importable (gym_pcgrl/__init__.py:6-12 does this on
    import; so does this package, below).  Returns the ids registered by this call."""
    from . import gym_compat
    if gym_compat.find_gym() is None:
        return []
    if not _ENV_IDS:
        _register_all()
    return gym_compat.register_all(_ENV_IDS)


def register_with_gymnasium():
    """The same for the new API (gymnasium, or gym >= 0.26): the ids are registered there with the five-tuple adapter
    `gymnasium_compat.NewApiPcgrlEnv` as entry point.  Returns the ids registered by this call."""
    from . import gymnasium_compat
    if gymnasium_compat.find_gymnasium() is None:
        return []
    if not _ENV_IDS:
        _register_all()
    return gymnasium_compat.register_all(_ENV_IDS)


def _lookup(env_id):
    if not _ENV_IDS:
        _register_all()
    if env_id not in _ENV_IDS:
        raise KeyError("unknown environment id %r; available: %s" % (env_id, ", ".join(sorted(_ENV_IDS))))
    return _ENV_IDS[env_id]


def make(env_id, **kwargs):
    """gym.make(id) replacement: a single reference-style PcgrlEnv backed by the HIP kernels."""
    from .envs import PcgrlEnv
    prob, rep = _lookup(env_id)
    return PcgrlEnv(prob=prob, rep=rep, **kwargs)


def make_batched(env_id, num_envs, **kwargs):
    from .envs import BatchedPcgrlEnv
    prob, rep = _lookup(env_id)
    return BatchedPcgrlEnv(prob=prob, rep=rep, num_envs=num_envs, **kwargs)



register_with_gym()      # no-op without a gym of the reference's era (none is on the MI355X image)
register_with_gymnasium()   # likewise for gymnasium / gym >= 0.26 (five-tuple adapter)
