"""Tile pictures for render(): drawn here, from the tile's name, with a few lines of numpy.

The reference draws its levels with per-problem PNG sprites (probs/<problem>/*.png, loaded in each Problem.render,
e.g. probs/binary_prob.py:150-165) and falls back to grey levels when a problem has none (probs/problem.py:136-140).
The sprites are reference data and are not part of this package; render() uses the grey fallback by default and these
pictures when asked to (BatchedPcgrlEnv.set_graphics("drawn") / PcgrlEnv.set_graphics): a background colour per tile
kind plus a simple shape, enough to tell a door from a key at a glance.  A caller who owns sprite images passes them as
a dict {tile name: HxWx3 uint8 array or PIL image} instead, exactly as the reference's `_graphics` dict is keyed."""
import numpy as np

_BG = (236, 232, 220)
# kind -> (background, shape, shape colour)
_STYLE = {
    "empty": (_BG, None, None),
    "solid": ((96, 72, 56), "bricks", (70, 50, 40)),
    "player": (_BG, "disc", (40, 90, 200)),
    "key": (_BG, "key", (220, 170, 30)),
    "door": (_BG, "door", (120, 70, 30)),
    "exit": (_BG, "door", (30, 140, 70)),
    "bat": (_BG, "diamond", (110, 40, 140)),
    "scorpion": (_BG, "diamond", (190, 60, 40)),
    "spider": (_BG, "diamond", (40, 40, 40)),
    "goblin": (_BG, "diamond", (60, 140, 60)),
    "ogre": (_BG, "diamond", (150, 40, 40)),
    "enemy": (_BG, "diamond", (190, 60, 40)),
    "crate": (_BG, "box", (170, 120, 60)),
    "target": (_BG, "ring", (200, 60, 60)),
    "potion": (_BG, "disc", (200, 60, 160)),
    "treasure": (_BG, "box", (220, 180, 40)),
    "diamond": (_BG, "diamond", (60, 170, 210)),
    "spike": (_BG, "spike", (90, 90, 100)),
    "brick": ((170, 96, 60), "bricks", (120, 64, 40)),
    "question": ((230, 170, 50), "ring", (120, 64, 40)),
    "coin": (_BG, "disc", (235, 195, 40)),
    "tube": ((60, 160, 70), "box", (40, 110, 50)),
}


def draw_tile(name, size=16):
    """A size x size x 3 uint8 picture for a tile name (unknown names: a grey square)."""
    bg, shape, col = _STYLE.get(name, ((128, 128, 128), None, None))
    img = np.empty((size, size, 3), np.uint8)
    img[:] = bg
    if shape is None:
        return img
    yy, xx = np.mgrid[0:size, 0:size]
    cy = cx = (size - 1) / 2.0
    r = size / 2.0
    if shape == "disc":
        m = (yy - cy) ** 2 + (xx - cx) ** 2 <= (0.62 * r) ** 2
    elif shape == "ring":
        d2 = (yy - cy) ** 2 + (xx - cx) ** 2
        m = (d2 <= (0.7 * r) ** 2) & (d2 >= (0.38 * r) ** 2)
    elif shape == "diamond":
        m = np.abs(yy - cy) + np.abs(xx - cx) <= 0.75 * r
    elif shape == "box":
        m = (np.abs(yy - cy) <= 0.6 * r) & (np.abs(xx - cx) <= 0.6 * r)
    elif shape == "door":
        m = (np.abs(xx - cx) <= 0.45 * r) & (yy >= 0.2 * size)
    elif shape == "key":
        m = ((yy - 0.3 * size) ** 2 + (xx - cx) ** 2 <= (0.28 * r) ** 2) | ((np.abs(xx - cx) <= 0.08 * size) & (yy >= 0.3 * size) & (yy <= 0.85 * size))
    elif shape == "spike":
        m = (yy >= size - 1 - 2 * np.minimum(xx % (size // 2), (size // 2 - 1) - xx % (size // 2))) & (yy >= size // 3)
    elif shape == "bricks":
        m = (yy % (size // 4) == 0) | (((xx + (size // 4) * ((yy // (size // 4)) % 2)) % (size // 2)) == 0)
    else:
        m = np.zeros((size, size), bool)
    img[m] = col
    return img


def make_graphics(tiles, size=16):
    """{tile name: picture} for a problem's tile list."""
    return {t: draw_tile(t, size) for t in tiles}
