"""Batched GPU versions of the two composite wrappers the reference's trainer puts on every environment
(gym_pcgrl/wrappers.py:215-248, used by utils.make_env :42-58):

  CroppedImagePCGRLWrapper(game, crop_size)   narrow / turtle: map padded with the border tile and cropped to a
      crop_size window centred on the cursor (Cropped :163-206), one-hot unless the game is binary
      (OneHotEncoding :67-104), as an image [size, size, depth] (ToImage :18-60)
  ActionMapImagePCGRLWrapper(game)            wide: action = flat index into (H, W, tiles) (ActionMap :111-154),
      observation = the (one-hot) map image

Here they wrap a BatchedPcgrlEnv: observations are one uint8 tensor [N, h, w, depth] that the step itself keeps up to date
(`pcgrl_bind_observation`: the fused step kernel writes the image from its on-chip copy of the state, other pipelines add one
store-stream kernel), actions are decoded on the device (`pcgrl_action_map`).  The tensor returned by reset() / step() is
overwritten by the next step, like every other output of the batched environment.  Same names, same constructor
arguments plus `num_envs`/`seed`/`device`; values equal the reference's wrappers element for element.
"""
import ctypes as C

from . import _lib, make_batched


class _ImageWrapper:
    def __init__(self, game, num_envs, seed, device, kwargs):
        if isinstance(game, str):
            self.pcgrl_env = make_batched(game, num_envs=num_envs, seed=seed, **({"device": device} if device else {}))
            self.game = game
        else:
            self.pcgrl_env = game
            self.game = "%s-%s-v0" % (game._prob.name, game._rep.name)
        self.pcgrl_env.adjust_param(**kwargs)
        self.env = self.pcgrl_env
        self.num_envs = self.pcgrl_env.num_envs
        self.one_hot = "binary" not in self.game           # wrappers.py:222-224 / :244-246
        self._obs = None
        self._bound = None
        self._external = False        # set_observation_target(): the image lives in a caller's tensor

    def _window(self):
        raise NotImplementedError

    def _bind(self):
        """(Re)bind the image to the environment when its shape changed (first use, adjust_param(width/height))."""
        h, w, centered, pad = self._window()
        key = (h, w, centered, pad, self.one_hot)
        if self._bound != key:
            if self._external:
                # a caller's tensor is installed (set_observation_target) and the image's shape changed under it
                # (adjust_param(width/height)): allocating a fresh one here would leave the caller reading a tensor that is
                # no longer written -- it has to hand over a target of the new shape first
                raise RuntimeError("the observation's shape changed to %s while an external observation target is installed; "
                                   "call set_observation_target() with a tensor of the new shape (or release_observation_target()) "
                                   "before the next reset()" % (key[:2],))
            self._obs = self.pcgrl_env.bind_observation(h, w, centered, pad, self.one_hot)
            self._bound = key
        return self._obs

    def set_observation_target(self, out):
        """The next images go to `out` (same shape; e.g. a row of a rollout buffer)."""
        h, w, centered, pad = self._window()
        key = (h, w, centered, pad, self.one_hot)
        if self._bound != key:              # first use, or a new map size: bind with the caller's tensor right away
            self.pcgrl_env.bind_observation(h, w, centered, pad, self.one_hot, out=out)
            self._bound = key
        else:
            self.pcgrl_env.set_observation_target(out)
        self._obs = out
        self._external = True

    def release_observation_target(self):
        """Back to a tensor of the wrapper's own (allocated by the next reset() / _bind())."""
        self._external = False
        self._bound = None

    def reset(self):
        self._bind()
        self.pcgrl_env.reset()
        return self._obs

    def seed(self, seed=None):
        return self.pcgrl_env.seed(seed)

    def adjust_param(self, **kwargs):
        self.pcgrl_env.adjust_param(**kwargs)          # (a new map size: the image is bound again by the next reset())

    def close(self):
        self.pcgrl_env.close()


class CroppedImagePCGRLWrapper(_ImageWrapper):
    def __init__(self, game, crop_size, num_envs=1, seed=None, device=None, **kwargs):
        super().__init__(game, num_envs, seed, device, kwargs)
        if not self.pcgrl_env._rep.has_pos:
            raise AssertionError("This wrapper only works for representations thave have a position")   # wrappers.py:170
        self.size = int(crop_size)
        self.pad_value = self.pcgrl_env.get_border_tile()

    def _window(self):
        return self.size, self.size, True, self.pad_value

    def step(self, actions):
        _, reward, done, info = self.pcgrl_env.step(actions)          # the step wrote the image
        return self._obs, reward, done, info

    def tick(self, actions, pop_budget=64):
        """One asynchronous tick (BatchedPcgrlEnv.tick; the search problems): -> (image, reward, done, info, pending)."""
        _, reward, done, info, pending = self.pcgrl_env.tick(actions, pop_budget=pop_budget)
        return self._obs, reward, done, info, pending


class ActionMapImagePCGRLWrapper(_ImageWrapper):
    def __init__(self, game, num_envs=1, seed=None, device=None, **kwargs):
        super().__init__(game, num_envs, seed, device, kwargs)
        if self.pcgrl_env._rep.name != "wide":
            raise NotImplementedError("this composite is the trainer's wide-representation wrapper (utils.py:49-50); for a representation with "
                                      "a cursor compose ToImage(OneHotEncoding(ActionMap(env), 'map'), ['map']) from the classes below")
        self._xyv = None

    def _window(self):
        p = self.pcgrl_env._prob
        return int(p._height), int(p._width), False, 0

    def step(self, actions):
        e = self.pcgrl_env
        torch = e._torch
        a = actions if torch.is_tensor(actions) else torch.as_tensor(actions)
        a = a.to(device=e.device, dtype=torch.int32).reshape(self.num_envs).contiguous()
        if self._xyv is None:
            self._xyv = torch.empty((self.num_envs, 3), dtype=torch.int32, device=e.device)
        _, reward, done, info = e.step_flat(a, self._xyv)          # decode + step: one call, one launch where the step is fused
        return self._obs, reward, done, info

    def tick(self, actions, pop_budget=64):
        """One asynchronous tick on flat ActionMap indices (the search problems): -> (image, reward, done, info, pending)."""
        import ctypes as C
        from . import _lib
        e = self.pcgrl_env
        torch = e._torch
        a = actions if torch.is_tensor(actions) else torch.as_tensor(actions)
        a = a.to(device=e.device, dtype=torch.int32).reshape(self.num_envs).contiguous()
        if self._xyv is None:
            self._xyv = torch.empty((self.num_envs, 3), dtype=torch.int32, device=e.device)
        _lib.check(e._lib.pcgrl_action_map(e._handle, C.c_void_p(a.data_ptr()), C.c_void_p(self._xyv.data_ptr()), e._stream()), "pcgrl_action_map")
        _, reward, done, info, pending = e.tick(self._xyv, pop_budget=pop_budget)
        return self._obs, reward, done, info, pending


# ------------------------------------------------------------------------------------------------------------------------
# The reference's single-purpose wrappers as separately composable classes (gym_pcgrl/wrappers.py:18-206), batched: they wrap a
# BatchedPcgrlEnv (or one another) and transform its dict-of-tensors observation with a few torch calls on the device.  The two
# composites above are what a trainer should use (one fused image per step); these exist so that code written against the
# reference's building blocks -- e.g. ToImage(OneHotEncoding(Cropped(env, 22, pad, 'map'), 'map'), ['map', 'heatmap']) -- runs
# unchanged on the batch.  Values equal the reference's element for element (one-hot planes are uint8 0/1 where the
# reference's np.eye gives float64 0/1).
def _pcgrl_env_of(env):
    while not hasattr(env, "_bufs"):
        env = env.env
    return env


class _ObsWrapper:
    def __init__(self, game, num_envs, seed, device, kwargs):
        if isinstance(game, str):
            game = make_batched(game, num_envs=num_envs, seed=seed, **({"device": device} if device else {}))
        self.env = game
        self.pcgrl_env = _pcgrl_env_of(game)
        self.pcgrl_env.adjust_param(**kwargs)
        self.num_envs = self.pcgrl_env.num_envs
        self.action_space = getattr(game, "action_space", None)
        self.observation_space = getattr(game, "observation_space", None)

    def transform(self, obs):
        return obs

    def reset(self):
        return self.transform(self.env.reset())

    def step(self, actions):
        obs, reward, done, info = self.env.step(actions)
        return self.transform(obs), reward, done, info

    def seed(self, seed=None):
        return self.env.seed(seed)

    def adjust_param(self, **kwargs):
        self.pcgrl_env.adjust_param(**kwargs)

    def get_border_tile(self):
        return self.pcgrl_env.get_border_tile()

    def get_num_tiles(self):
        return self.pcgrl_env.get_num_tiles()

    def close(self):
        self.env.close()

    def _spaces_copy(self):
        from collections import OrderedDict

        from . import spaces
        return spaces.Dict(OrderedDict(self.env.observation_space.spaces.items()))


class Cropped(_ObsWrapper):
    """wrappers.py:163-206: obs[name] [N,H,W] -> the crop_size window centred on the cursor, padded with pad_value."""

    def __init__(self, game, crop_size, pad_value, name, num_envs=1, seed=None, device=None, **kwargs):
        super().__init__(game, num_envs, seed, device, kwargs)
        import numpy as np

        from . import spaces
        sp = self.env.observation_space.spaces
        assert "pos" in sp, "This wrapper only works for representations thave have a position"
        assert name in sp, "This wrapper only works if you have a {} key".format(name)
        assert len(sp[name].shape) == 2, "This wrapper only works on 2D arrays."
        self.name, self.size, self.pad, self.pad_value = name, int(crop_size), int(crop_size) // 2, int(pad_value)
        self.observation_space = self._spaces_copy()
        self.observation_space.spaces[name] = spaces.Box(low=0, high=int(np.max(sp[name].high)), shape=(self.size, self.size), dtype=np.uint8)

    def transform(self, obs):
        torch = self.pcgrl_env._torch
        m = obs[self.name]
        n, h, w = m.shape
        padded = torch.nn.functional.pad(m, (self.pad, self.pad, self.pad, self.pad), value=self.pad_value)
        pos = obs["pos"].to(torch.long)                       # [N, 2] = (x, y)
        ar = torch.arange(self.size, device=m.device)
        rows = (pos[:, 1:2] + ar[None, :])[:, :, None].expand(n, self.size, self.size)
        cols = (pos[:, 0:1] + ar[None, :])[:, None, :].expand(n, self.size, self.size)
        out = type(obs)(obs)
        out[self.name] = padded[torch.arange(n, device=m.device)[:, None, None], rows, cols]
        return out


class OneHotEncoding(_ObsWrapper):
    """wrappers.py:67-104: obs[name] -> np.eye(dim)[obs[name]] (a trailing axis of `dim` planes)."""

    def __init__(self, game, name, num_envs=1, seed=None, device=None, **kwargs):
        super().__init__(game, num_envs, seed, device, kwargs)
        import numpy as np

        from . import spaces
        sp = self.env.observation_space.spaces
        assert name in sp, "This wrapper only works for representations thave have a {} key".format(name)
        self.name = name
        self.dim = int(np.max(sp[name].high)) - int(np.min(sp[name].low)) + 1
        self.observation_space = self._spaces_copy()
        self.observation_space.spaces[name] = spaces.Box(low=0, high=1, shape=tuple(sp[name].shape) + (self.dim,), dtype=np.uint8)

    def transform(self, obs):
        torch = self.pcgrl_env._torch
        out = type(obs)(obs)
        out[self.name] = torch.nn.functional.one_hot(obs[self.name].to(torch.long), self.dim).to(torch.uint8)
        return out


class ToImage(_ObsWrapper):
    """wrappers.py:18-60: the named entries stacked along a trailing axis into one [N, h, w, depth] tensor (the last layer)."""

    def __init__(self, game, names, num_envs=1, seed=None, device=None, **kwargs):
        super().__init__(game, num_envs, seed, device, kwargs)
        import numpy as np

        from . import spaces
        sp = self.env.observation_space.spaces
        self.shape, depth, max_value = None, 0, 0
        for n in names:
            assert n in sp, "This wrapper only works if your observation_space is spaces.Dict with the input names."
            if self.shape is None:
                self.shape = sp[n].shape
            new_shape = sp[n].shape
            depth += 1 if len(new_shape) <= 2 else new_shape[2]
            assert self.shape[0] == new_shape[0] and self.shape[1] == new_shape[1], "This wrapper only works when all objects have same width and height"
            max_value = max(max_value, int(np.max(sp[n].high)))
        self.names = list(names)
        self.observation_space = spaces.Box(low=0, high=max_value, shape=(self.shape[0], self.shape[1], depth), dtype=np.uint8)

    def transform(self, obs):
        torch = self.pcgrl_env._torch
        parts = [obs[n].reshape(obs[n].shape[0], self.shape[0], self.shape[1], -1) for n in self.names]
        dt = parts[0].dtype
        for p in parts[1:]:
            dt = torch.promote_types(dt, p.dtype)             # (np.append promotes the same way; the values are what counts)
        return torch.cat([p.to(dt) for p in parts], dim=3) if len(parts) > 1 else parts[0]


class ActionMap(_ObsWrapper):
    """wrappers.py:111-160: the action is a flat index into (h, w, tiles).  Without a cursor it becomes the wide representation's
    [x, y, tile]; with one (narrow / turtle) the reference's rule is kept literally: where the cursor stands on (x, y) the inner
    environment is stepped with `tile`, elsewhere with the tile value that is already under the cursor."""

    def __init__(self, game, num_envs=1, seed=None, device=None, **kwargs):
        super().__init__(game, num_envs, seed, device, kwargs)
        from . import spaces
        sp = self.env.observation_space.spaces
        assert "map" in sp, "This wrapper only works if you have a map key"
        self.one_hot = len(sp["map"].shape) > 2
        self.h, self.w = int(sp["map"].shape[0]), int(sp["map"].shape[1])
        self.dim = self.pcgrl_env.get_num_tiles()
        self.action_space = spaces.Discrete(self.h * self.w * self.dim)
        self.old_obs = None

    def reset(self):
        self.old_obs = self.env.reset()
        return self.old_obs

    def step(self, actions):
        torch = self.pcgrl_env._torch
        a = torch.as_tensor(actions, device=self.pcgrl_env.device).to(torch.long).reshape(self.num_envs)
        v = a % self.dim
        x = (a // self.dim) % self.w
        y = a // (self.dim * self.w)
        if "pos" in self.old_obs:
            pos = self.old_obs["pos"].to(torch.long)
            m = self.old_obs["map"]
            o_v = m[torch.arange(self.num_envs, device=m.device), pos[:, 1], pos[:, 0]]
            if self.one_hot:
                o_v = o_v.argmax(-1)
            inner = torch.where((pos[:, 0] == x) & (pos[:, 1] == y), v, o_v.to(torch.long))
            out = self.env.step(inner.to(torch.int32))
        else:
            out = self.env.step(torch.stack([x, y, v], 1).to(torch.int32))
        self.old_obs = out[0]
        return out
