"""Host-side mirrors of the reference's Problem classes: parameters only.

The statistics, rewards and episode-over tests run on the GPU (csrc/pcgrl_algos.h); these
classes hold what the reference keeps as Python attributes (probs/problem.py:10-20,
binary_prob.py:14-27, zelda_prob.py:17-37, sokoban_prob.py:15-36, mdungeon_prob.py:16-41, ddave_prob.py:15-40) and reproduce
`adjust_param` (problem.py:66-72 and the subclasses) including its quirks: `probs` only
overrides existing keys, sokoban `max_targets` overwrites `max_crates`, the sokoban kwarg is
`min_solution`.
"""
from collections import OrderedDict

PROB_IDS = {"binary": 0, "zelda": 1, "sokoban": 2, "mdungeon": 3, "ddave": 4, "smb": 5}


class Problem:
    name = None
    tiles = ()
    stat_keys = ()     # order of the device stats row
    info_keys = ()     # get_debug_info keys, in the reference's order
    reward_keys = ()   # order of `_rewards` (the device weights row)

    def __init__(self):
        self._width = 9
        self._height = 9
        self._border_tile = self.tiles[1]
        self._border_size = (1, 1)
        self._tile_size = 16
        self._prob = OrderedDict()
        self._rewards = OrderedDict()
        self._probs_touched = False

    def get_tile_types(self):
        return list(self.tiles)

    def adjust_param(self, **kwargs):
        self._width, self._height = kwargs.get("width", self._width), kwargs.get("height", self._height)
        prob = kwargs.get("probs")
        if prob is not None:
            for t in prob:
                if t in self._prob:
                    self._prob[t] = prob[t]
                    self._probs_touched = True
        rewards = kwargs.get("rewards")
        if rewards is not None:
            for t in rewards:
                if t in self._rewards:
                    self._rewards[t] = rewards[t]

    # scalars shipped to the device config
    def device_params(self):
        return {}

    packed_rows = False      # True: the device row packs several values into a slot (decode_rows spreads them out)

    def decode_rows(self, table, keys=None):
        """Device stats / info rows (torch int32 [N, >=8]) -> columns in the order of `keys` (default `stat_keys`).
        Plain slices for the problems whose rows hold one value per slot."""
        keys = list(keys or self.stat_keys)
        idx = [self.stat_keys.index(k) for k in keys]
        if idx == list(range(len(idx))):
            return table[:, :len(idx)]
        return table[:, idx]


class BinaryProblem(Problem):
    name = "binary"
    tiles = ("empty", "solid")
    stat_keys = ("regions", "path-length")
    info_keys = ("regions", "path-length", "path-imp")
    reward_keys = ("regions", "path-length")

    def __init__(self):
        super().__init__()
        self._width, self._height = 14, 14
        self._prob = OrderedDict([("empty", 0.5), ("solid", 0.5)])
        self._border_tile = "solid"
        self._target_path = 20
        self._random_probs = True
        self._rewards = OrderedDict([("regions", 5), ("path-length", 1)])

    def adjust_param(self, **kwargs):
        super().adjust_param(**kwargs)
        self._target_path = kwargs.get("target_path", self._target_path)
        self._random_probs = kwargs.get("random_probs", self._random_probs)

    def device_params(self):
        return dict(target_path=int(self._target_path), random_probs=int(bool(self._random_probs)))


class ZeldaProblem(Problem):
    name = "zelda"
    tiles = ("empty", "solid", "player", "key", "door", "bat", "scorpion", "spider")
    stat_keys = ("player", "key", "door", "enemies", "regions", "nearest-enemy", "path-length")
    info_keys = stat_keys
    reward_keys = ("player", "key", "door", "regions", "enemies", "nearest-enemy", "path-length")

    def __init__(self):
        super().__init__()
        self._width, self._height = 11, 7
        self._prob = OrderedDict([("empty", 0.58), ("solid", 0.3), ("player", 0.02), ("key", 0.02), ("door", 0.02),
                                  ("bat", 0.02), ("scorpion", 0.02), ("spider", 0.02)])
        self._border_tile = "solid"
        self._max_enemies = 5
        self._target_enemy_dist = 4
        self._target_path = 16
        self._rewards = OrderedDict([("player", 3), ("key", 3), ("door", 3), ("regions", 5), ("enemies", 1),
                                     ("nearest-enemy", 2), ("path-length", 1)])

    def adjust_param(self, **kwargs):
        super().adjust_param(**kwargs)
        self._max_enemies = kwargs.get("max_enemies", self._max_enemies)
        self._target_enemy_dist = kwargs.get("target_enemy_dist", self._target_enemy_dist)
        self._target_path = kwargs.get("target_path", self._target_path)

    def device_params(self):
        return dict(target_path=int(self._target_path), max_enemies=int(self._max_enemies),
                    target_enemy_dist=int(self._target_enemy_dist))


class SokobanProblem(Problem):
    name = "sokoban"
    tiles = ("empty", "solid", "player", "crate", "target")
    stat_keys = ("player", "crate", "target", "regions", "dist-win", "sol-length")
    info_keys = stat_keys
    reward_keys = ("player", "crate", "target", "regions", "ratio", "dist-win", "sol-length")

    def __init__(self):
        super().__init__()
        self._width, self._height = 5, 5
        self._prob = OrderedDict([("empty", 0.45), ("solid", 0.4), ("player", 0.05), ("crate", 0.05), ("target", 0.05)])
        self._border_tile = "solid"
        self._solver_power = 5000
        self._max_crates = 3
        self._target_solution = 18
        self._rewards = OrderedDict([("player", 3), ("crate", 2), ("target", 2), ("regions", 5), ("ratio", 2),
                                     ("dist-win", 0.0), ("sol-length", 1)])

    def adjust_param(self, **kwargs):
        super().adjust_param(**kwargs)
        self._solver_power = kwargs.get("solver_power", self._solver_power)
        self._max_crates = kwargs.get("max_crates", self._max_crates)
        self._max_crates = kwargs.get("max_targets", self._max_crates)
        self._target_solution = kwargs.get("min_solution", self._target_solution)

    def device_params(self):
        return dict(max_crates=int(self._max_crates), target_solution=int(self._target_solution),
                    solver_power=int(self._solver_power))


class MDungeonProblem(Problem):
    """probs/mdungeon_prob.py.  The reference's eleven statistics travel in the eight slots of a device row (see
    include/pcgrl_hip.h, pcgrl_layout.stats); `decode_rows` spreads them out again."""
    name = "mdungeon"
    tiles = ("empty", "solid", "player", "exit", "potion", "treasure", "goblin", "ogre")
    stat_keys = ("player", "exit", "potions", "treasures", "enemies", "regions", "col-potions", "col-treasures",
                 "col-enemies", "dist-win", "sol-length")
    info_keys = stat_keys
    reward_keys = ("player", "exit", "potions", "treasures", "enemies", "regions", "col-enemies", "dist-win", "sol-length")

    def __init__(self):
        super().__init__()
        self._width, self._height = 7, 11
        self._prob = OrderedDict([("empty", 0.4), ("solid", 0.4), ("player", 0.02), ("exit", 0.02), ("potion", 0.03),
                                  ("treasure", 0.03), ("goblin", 0.05), ("ogre", 0.05)])
        self._border_tile = "solid"
        self._solver_power = 5000
        self._max_enemies = 6
        self._max_potions = 2
        self._max_treasures = 3
        self._target_col_enemies = 0.5
        self._target_solution = 20
        self._rewards = OrderedDict([("player", 3), ("exit", 3), ("potions", 1), ("treasures", 1), ("enemies", 2),
                                     ("regions", 5), ("col-enemies", 2), ("dist-win", 0.1), ("sol-length", 1)])

    def adjust_param(self, **kwargs):
        super().adjust_param(**kwargs)
        self._solver_power = kwargs.get("solver_power", self._solver_power)
        self._max_enemies = kwargs.get("max_enemies", self._max_enemies)
        self._max_potions = kwargs.get("max_potions", self._max_potions)
        self._max_treasures = kwargs.get("max_treasures", self._max_treasures)
        self._target_col_enemies = kwargs.get("target_col_enemies", self._target_col_enemies)
        self._target_solution = kwargs.get("target_solution", self._target_solution)

    def device_params(self):
        return dict(solver_power=int(self._solver_power), max_enemies=int(self._max_enemies),
                    max_potions=int(self._max_potions), max_treasures=int(self._max_treasures),
                    target_col_enemies=float(self._target_col_enemies), target_solution=int(self._target_solution))

    packed_rows = True

    def decode_rows(self, table, keys=None):
        import torch
        won = (table[:, 7] >> 24) & 1
        zero = torch.zeros_like(won)
        cols = {k: table[:, i] for i, k in enumerate(self.stat_keys[:6])}
        cols.update({"col-potions": table[:, 7] & 255, "col-treasures": (table[:, 7] >> 8) & 255,
                     "col-enemies": (table[:, 7] >> 16) & 255,
                     "dist-win": torch.where(won == 1, zero, table[:, 6]), "sol-length": torch.where(won == 1, table[:, 6], zero)})
        return torch.stack([cols[k] for k in (keys or self.stat_keys)], 1)


class DDaveProblem(Problem):
    """probs/ddave_prob.py.  Eleven statistics in the eight slots of a device row (include/pcgrl_hip.h,
    pcgrl_layout.stats); note that get_debug_info leaves `dist-floor` out and swaps two keys (ddave_prob.py:232-245)."""
    name = "ddave"
    tiles = ("empty", "solid", "player", "exit", "diamond", "key", "spike")
    stat_keys = ("player", "dist-floor", "exit", "diamonds", "key", "spikes", "regions", "num-jumps", "col-diamonds",
                 "dist-win", "sol-length")
    info_keys = ("player", "exit", "diamonds", "key", "spikes", "regions", "col-diamonds", "num-jumps", "dist-win", "sol-length")
    reward_keys = ("player", "dist-floor", "exit", "diamonds", "key", "spikes", "regions", "num-jumps", "dist-win", "sol-length")
    packed_rows = True

    def __init__(self):
        super().__init__()
        self._width, self._height = 11, 7
        self._prob = OrderedDict([("empty", 0.5), ("solid", 0.3), ("player", 0.02), ("exit", 0.02), ("diamond", 0.04),
                                  ("key", 0.02), ("spike", 0.1)])
        self._border_tile = "solid"
        self._solver_power = 5000
        self._max_diamonds = 3
        self._min_spikes = 10
        self._target_jumps = 2
        self._target_solution = 20
        self._rewards = OrderedDict([("player", 3), ("dist-floor", 2), ("exit", 3), ("diamonds", 1), ("key", 3), ("spikes", 1),
                                     ("regions", 5), ("num-jumps", 3), ("dist-win", 0.1), ("sol-length", 1)])

    def adjust_param(self, **kwargs):
        super().adjust_param(**kwargs)
        self._solver_power = kwargs.get("solver_power", self._solver_power)
        self._max_diamonds = kwargs.get("max_diamonds", self._max_diamonds)
        self._min_spikes = kwargs.get("min_spikes", self._min_spikes)
        self._target_jumps = kwargs.get("target_jumps", self._target_jumps)
        self._target_solution = kwargs.get("target_solution", self._target_solution)

    def device_params(self):
        return dict(solver_power=int(self._solver_power), max_diamonds=int(self._max_diamonds), min_spikes=int(self._min_spikes),
                    target_jumps=int(self._target_jumps), target_solution=int(self._target_solution))

    def decode_rows(self, table, keys=None):
        import torch
        won = (table[:, 7] >> 24) & 1
        zero = torch.zeros_like(won)
        cols = {"player": table[:, 0] & 255, "exit": (table[:, 0] >> 8) & 255, "key": (table[:, 0] >> 16) & 255,
                "dist-floor": table[:, 1], "diamonds": table[:, 2], "spikes": table[:, 3], "regions": table[:, 4],
                "num-jumps": table[:, 5], "col-diamonds": table[:, 7] & 255,
                "dist-win": torch.where(won == 1, zero, table[:, 6]), "sol-length": torch.where(won == 1, table[:, 6], zero)}
        return torch.stack([cols[k] for k in (keys or self.stat_keys)], 1)


class SMBProblem(Problem):
    """probs/smb_prob.py: a 114 x 14 platformer level; every changed map is played by two A* agents (smb/engine.py).
    `solver_power` is an attribute, not an adjust_param key (smb_prob.py:40-53 does not read it)."""
    name = "smb"
    tiles = ("empty", "solid", "enemy", "brick", "question", "coin", "tube")
    stat_keys = ("dist-floor", "disjoint-tubes", "enemies", "empty", "noise", "jumps", "jumps-dist", "dist-win")
    info_keys = stat_keys
    reward_keys = stat_keys

    def __init__(self):
        super().__init__()
        self._width, self._height = 114, 14
        self._prob = OrderedDict([("empty", 0.75), ("solid", 0.1), ("enemy", 0.01), ("brick", 0.04), ("question", 0.01),
                                  ("coin", 0.02), ("tube", 0.02)])
        self._border_tile = "solid"
        self._border_size = (3, 0)
        self._solver_power = 10000
        self._min_empty = 900
        self._min_enemies = 10
        self._max_enemies = 30
        self._min_jumps = 20
        self._rewards = OrderedDict([("dist-floor", 2), ("disjoint-tubes", 1), ("enemies", 1), ("empty", 1), ("noise", 4),
                                     ("jumps", 2), ("jumps-dist", 2), ("dist-win", 5)])

    def adjust_param(self, **kwargs):
        super().adjust_param(**kwargs)
        self._min_empty = kwargs.get("min_empty", self._min_empty)
        self._min_enemies = kwargs.get("min_enemies", self._min_enemies)
        self._max_enemies = kwargs.get("max_enemies", self._max_enemies)
        self._min_jumps = kwargs.get("min_jumps", self._min_jumps)

    def device_params(self):
        return dict(solver_power=int(self._solver_power), min_empty=int(self._min_empty), min_enemies=int(self._min_enemies),
                    max_enemies=int(self._max_enemies), min_jumps=int(self._min_jumps))


PROBLEMS = {"binary": BinaryProblem, "zelda": ZeldaProblem, "sokoban": SokobanProblem, "mdungeon": MDungeonProblem,
            "ddave": DDaveProblem, "smb": SMBProblem}
