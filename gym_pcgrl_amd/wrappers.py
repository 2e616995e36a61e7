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

    def _window(self):
        raise NotImplementedError

    def _bind(self):
        """(Re)bind the image to the environment when its shape changed (first use, adjust_param(width/height))."""
        h, w, centered, pad = self._window()
        key = (h, w, centered, pad, self.one_hot)
        if self._bound != key:
            self._obs = self.pcgrl_env.bind_observation(h, w, centered, pad, self.one_hot)
            self._bound = key
        return self._obs

    def set_observation_target(self, out):
        """The next images go to `out` (same shape; e.g. a row of a rollout buffer)."""
        self._bind()
        self.pcgrl_env.set_observation_target(out)
        self._obs = out

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


class ActionMapImagePCGRLWrapper(_ImageWrapper):
    def __init__(self, game, num_envs=1, seed=None, device=None, **kwargs):
        super().__init__(game, num_envs, seed, device, kwargs)
        if self.pcgrl_env._rep.name != "wide":
            raise NotImplementedError("the batched ActionMap is provided for the wide representation (the reference's trainer "
                                      "only uses it there, utils.py:49-50)")
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
        _lib.check(e._lib.pcgrl_action_map(e._handle, C.c_void_p(a.data_ptr()), C.c_void_p(self._xyv.data_ptr()), e._stream()), "pcgrl_action_map")
        _, reward, done, info = e.step(self._xyv)
        return self._obs, reward, done, info
