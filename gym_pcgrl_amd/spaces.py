"""Minimal observation/action space descriptors.

The reference builds its spaces from `gym.spaces` (narrow_rep.py:45-46,60-64,
wide_rep.py:28-29,42-45, turtle_rep.py:58-59,73-77, pcgrl_env.py:42).  `gym` is
not installed on the MI355X image, so the batched env ships these small
stand-ins with the attributes the reference's wrappers actually read
(`.n`, `.nvec`, `.shape`, `.low`, `.high`, `.dtype`, `.spaces[...]`,
wrappers.py:26-60,118-131,170-193).  When a real `gym`/`gymnasium` is importable
`to_gym()` converts a descriptor into the library's own class.
"""
from collections import OrderedDict

import numpy as np


class Space:
    shape = None
    dtype = None

    def sample(self, rng=None):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def sample(self, rng=None):
        rng = rng or np.random
        return int(rng.randint(self.n))

    def contains(self, x):
        try:
            v = int(x)
        except (TypeError, ValueError):
            return False
        return 0 <= v < self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n

    def __eq__(self, other):
        return isinstance(other, Discrete) and other.n == self.n


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(np.int64)

    def sample(self, rng=None):
        rng = rng or np.random
        return (rng.random_sample(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(((0 <= x) & (x < self.nvec)).all())

    def __repr__(self):
        return "MultiDiscrete(%s)" % (self.nvec.tolist(),)

    def __eq__(self, other):
        return isinstance(other, MultiDiscrete) and np.array_equal(other.nvec, self.nvec)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            low = np.asarray(low)
            high = np.asarray(high)
            shape = low.shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low), self.shape).astype(self.dtype)
        self.high = np.broadcast_to(np.asarray(high), self.shape).astype(self.dtype)

    def sample(self, rng=None):
        rng = rng or np.random
        span = self.high.astype(np.float64) - self.low.astype(np.float64)
        v = self.low + rng.random_sample(self.shape) * (span + (self.dtype.kind in "iu"))
        return np.floor(v).astype(self.dtype) if self.dtype.kind in "iu" else v.astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)

    def __eq__(self, other):
        return (isinstance(other, Box) and other.shape == self.shape and other.dtype == self.dtype
                and np.array_equal(other.low, self.low) and np.array_equal(other.high, self.high))


class Dict(Space):
    def __init__(self, spaces=None, **kw):
        if spaces is None:
            spaces = kw
        if isinstance(spaces, dict) and not isinstance(spaces, OrderedDict):
            spaces = OrderedDict(sorted(spaces.items()))  # gym<=0.21 sorts plain dicts by key
        self.spaces = OrderedDict(spaces)

    def sample(self, rng=None):
        return OrderedDict((k, s.sample(rng)) for k, s in self.spaces.items())

    def contains(self, x):
        return isinstance(x, dict) and all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

    def __getitem__(self, k):
        return self.spaces[k]

    def __repr__(self):
        return "Dict(%s)" % ", ".join("%s:%r" % kv for kv in self.spaces.items())

    def __eq__(self, other):
        return isinstance(other, Dict) and other.spaces == self.spaces


def to_gym(space):
    """Convert a descriptor into a real gym/gymnasium space if one is importable."""
    try:
        import gymnasium as g  # pragma: no cover - not on the image
    except ImportError:
        try:
            import gym as g  # pragma: no cover
        except ImportError:
            return space
    if isinstance(space, Discrete):
        return g.spaces.Discrete(space.n)
    if isinstance(space, MultiDiscrete):
        return g.spaces.MultiDiscrete(space.nvec)
    if isinstance(space, Box):
        return g.spaces.Box(low=space.low, high=space.high, dtype=space.dtype.type)
    if isinstance(space, Dict):
        return g.spaces.Dict(OrderedDict((k, to_gym(s)) for k, s in space.spaces.items()))
    return space
