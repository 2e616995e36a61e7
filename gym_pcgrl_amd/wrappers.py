"""Batched GPU versions of the two composite wrappers the reference's trainer puts on every environment
(gym_pcgrl/wrappers.py:215-248, used by utils.make_env :42-58):

  CroppedImagePCGRLWrapper(game, crop_size)   narrow / turtle: map padded with the border tile and cropped to a
      crop_size window centred on the cursor (Cropped :163-206), one-hot unless the game is binary
      (OneHotEncoding :67-104), as an image [size, size, depth] (ToImage :18-60)
  ActionMapImagePCGRLWrapper(game)            wide: action = flat index into (H, W, tiles) (ActionMap :111-154),
      observation = the (one-hot) map image

Here they wrap a BatchedPcgrlEnv: observations are one uint8 tensor [N, h, w, depth] written by a HIP kernel
(`pcgrl_observe`), actions are decoded on the device (`pcgrl_action_map`).  Same names, same constructor
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

    def _alloc(self, h, w):
        torch = self.pcgrl_env._torch
        depth = self.pcgrl_env.get_num_tiles() if self.one_hot else 1
        if self._obs is None or tuple(self._obs.shape) != (self.num_envs, h, w, depth):
            self._obs = torch.empty((self.num_envs, h, w, depth), dtype=torch.uint8, device=self.pcgrl_env.device)
        return self._obs

    def _observe(self, h, w, centered, pad_value):
        e = self.pcgrl_env
        out = self._alloc(h, w)
        _lib.check(e._lib.pcgrl_observe(e._handle, C.c_void_p(out.data_ptr()), h, w, int(centered), int(pad_value),
                                        int(self.one_hot), e._stream()), "pcgrl_observe")
        return out

    def seed(self, seed=None):
        return self.pcgrl_env.seed(seed)

    def adjust_param(self, **kwargs):
        self.pcgrl_env.adjust_param(**kwargs)

    def close(self):
        self.pcgrl_env.close()


class CroppedImagePCGRLWrapper(_ImageWrapper):
    def __init__(self, game, crop_size, num_envs=1, seed=None, device=None, **kwargs):
        super().__init__(game, num_envs, seed, device, kwargs)
        if not self.pcgrl_env._rep.has_pos:
            raise AssertionError("This wrapper only works for representations thave have a position")   # wrappers.py:170
        self.size = int(crop_size)
        self.pad_value = self.pcgrl_env.get_border_tile()

    def _image(self):
        return self._observe(self.size, self.size, True, self.pad_value)

    def reset(self):
        self.pcgrl_env.reset()
        return self._image()

    def step(self, actions):
        _, reward, done, info = self.pcgrl_env.step(actions)
        return self._image(), reward, done, info


class ActionMapImagePCGRLWrapper(_ImageWrapper):
    def __init__(self, game, num_envs=1, seed=None, device=None, **kwargs):
        super().__init__(game, num_envs, seed, device, kwargs)
        if self.pcgrl_env._rep.name != "wide":
            raise NotImplementedError("the batched ActionMap is provided for the wide representation (the reference's trainer "
                                      "only uses it there, utils.py:49-50)")
        self._xyv = None

    def _image(self):
        p = self.pcgrl_env._prob
        return self._observe(int(p._height), int(p._width), False, 0)

    def reset(self):
        self.pcgrl_env.reset()
        return self._image()

    def step(self, actions):
        e = self.pcgrl_env
        torch = e._torch
        a = actions if torch.is_tensor(actions) else torch.as_tensor(actions)
        a = a.to(device=e.device, dtype=torch.int32).reshape(self.num_envs).contiguous()
        if self._xyv is None:
            self._xyv = torch.empty((self.num_envs, 3), dtype=torch.int32, device=e.device)
        _lib.check(e._lib.pcgrl_action_map(e._handle, C.c_void_p(a.data_ptr()), C.c_void_p(self._xyv.data_ptr()), e._stream()), "pcgrl_action_map")
        _, reward, done, info = e.step(self._xyv)
        return self._image(), reward, done, info
