"""Host-side mirrors of the reference's Representation classes: parameters and spaces.

`update()` itself runs in the k_update HIP kernel.  Spaces follow narrow_rep.py:45-46,60-64,
wide_rep.py:28-29,42-45, turtle_rep.py:58-59,73-77; adjust_param follows representation.py:53-54,
narrow_rep.py:86-88, turtle_rep.py:42-44.
"""
from collections import OrderedDict

import numpy as np

from .. import spaces

REP_IDS = {"narrow": 0, "wide": 1, "turtle": 2, "narrowcast": 3, "narrowmulti": 4, "turtlecast": 5}


class Representation:
    name = None
    has_pos = True

    def __init__(self):
        self._random_start = True

    def adjust_param(self, **kwargs):
        self._random_start = kwargs.get("random_start", self._random_start)

    def _pos_map_space(self, width, height, num_tiles):
        return spaces.Dict(OrderedDict([
            ("pos", spaces.Box(low=np.array([0, 0]), high=np.array([width - 1, height - 1]), dtype=np.uint8)),
            ("map", spaces.Box(low=0, high=num_tiles - 1, dtype=np.uint8, shape=(height, width))),
        ]))

    def device_params(self):
        return dict(random_start=int(bool(self._random_start)))


class NarrowRepresentation(Representation):
    name = "narrow"

    def __init__(self):
        super().__init__()
        self._random_tile = True

    def adjust_param(self, **kwargs):
        super().adjust_param(**kwargs)
        self._random_tile = kwargs.get("random_tile", self._random_tile)

    def get_action_space(self, width, height, num_tiles):
        return spaces.Discrete(num_tiles + 1)

    def get_observation_space(self, width, height, num_tiles):
        return self._pos_map_space(width, height, num_tiles)

    def action_width(self):
        return 1

    def device_params(self):
        d = super().device_params()
        d["random_tile"] = int(bool(self._random_tile))
        return d


class WideRepresentation(Representation):
    name = "wide"
    has_pos = False

    def get_action_space(self, width, height, num_tiles):
        return spaces.MultiDiscrete([width, height, num_tiles])

    def get_observation_space(self, width, height, num_tiles):
        return spaces.Dict(OrderedDict([
            ("map", spaces.Box(low=0, high=num_tiles - 1, dtype=np.uint8, shape=(height, width))),
        ]))

    def action_width(self):
        return 3


class TurtleRepresentation(Representation):
    name = "turtle"

    def __init__(self):
        super().__init__()
        self._dirs = [(-1, 0), (1, 0), (0, -1), (0, 1)]
        self._warp = False

    def adjust_param(self, **kwargs):
        super().adjust_param(**kwargs)
        self._warp = kwargs.get("warp", self._warp)

    def get_action_space(self, width, height, num_tiles):
        return spaces.Discrete(len(self._dirs) + num_tiles)

    def get_observation_space(self, width, height, num_tiles):
        return self._pos_map_space(width, height, num_tiles)

    def action_width(self):
        return 1

    def device_params(self):
        d = super().device_params()
        d["warp"] = int(bool(self._warp))
        return d


class NarrowCastRepresentation(NarrowRepresentation):
    """narrow_cast_rep.py: action (type, tile): 0 nothing, 1 this cell, 2 the 3x3 block around the cursor."""
    name = "narrowcast"

    def get_action_space(self, width, height, num_tiles):
        return spaces.MultiDiscrete([3, num_tiles])

    def action_width(self):
        return 2


class NarrowMultiRepresentation(NarrowRepresentation):
    """narrow_multi_rep.py: nine values, one per cell of the 3x3 block (0 keeps the cell, v writes tile v-1)."""
    name = "narrowmulti"

    def get_action_space(self, width, height, num_tiles):
        return spaces.MultiDiscrete([num_tiles + 1] * 9)

    def action_width(self):
        return 9


class TurtleCastRepresentation(TurtleRepresentation):
    """turtle_cast_rep.py: action (type, tile): 0-3 move, 4 write this cell, 5 write the 3x3 block."""
    name = "turtlecast"

    def get_action_space(self, width, height, num_tiles):
        return spaces.MultiDiscrete([len(self._dirs) + 2, num_tiles])

    def action_width(self):
        return 2


REPRESENTATIONS = {"narrow": NarrowRepresentation, "wide": WideRepresentation, "turtle": TurtleRepresentation,
                   "narrowcast": NarrowCastRepresentation, "narrowmulti": NarrowMultiRepresentation,
                   "turtlecast": TurtleCastRepresentation}
