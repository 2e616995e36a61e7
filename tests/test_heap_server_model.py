"""The heap server's two primitives (gym_pcgrl_amd/csrc/sokoban_fast.h: sok_duo_repair, sok_duo_append) restated lane for lane
in Python and held against CPython's heapq: the same pops and the same array after every operation.  (On the GPU the HIP code
itself goes through the same comparison: tests/test_gpu_parity.py::test_heap_server_primitives_match_heapq.)

What the restatement pins is the argument the kernels rest on: heappop's repair -- _siftup to a leaf, then _siftdown -- equals
ONE top-down walk that stops at the first node whose smaller child is strictly greater than the displaced entry, and which nodes
lie on that walk follows from the direction / stop bits of all 63 nodes of a six-level subtree at once; heappush's climb is the
run of ancestors the item beats, counted from the parent."""
import heapq

import numpy as np
import pytest


def lt(a, b):                                     # Node.__lt__: priority only (sokoban/engine.py:49-50); sok_lt
    return (a >> 16) < (b >> 16)


class W:
    __slots__ = ("w",)

    def __init__(self, w):
        self.w = w

    def __lt__(self, other):
        return lt(self.w, other.w)


def lanes():
    """sok_duo_lanes: per lane its level / offset in the subtree and the ancestor masks A (ancestors) and D (those it hangs under
    as a right child)."""
    T = []
    for lane in range(64):
        j1 = lane + 1
        lj = j1.bit_length() - 1
        A = D = 0
        for k in range(1, lj + 1):
            a = (j1 >> k) - 1
            A |= 1 << a
            if (j1 >> (k - 1)) & 1:
                D |= 1 << a
        T.append((lj, j1 - (1 << lj) - 1, A, D))
    return T


LANES = lanes()


def repair(heap, n, item):
    """sok_duo_repair: heap[0..n) with the root vacant, `item` goes in; every `for lane` loop is one wavefront instruction."""
    pos = 0
    while True:
        q = [((pos + 1) << LANES[l][0]) + LANES[l][1] for l in range(64)]
        has = [l < 63 and 2 * q[l] + 1 < n for l in range(64)]
        a = [heap[2 * q[l] + 1] if has[l] else 0 for l in range(64)]
        b = [heap[2 * q[l] + 2] if has[l] and 2 * q[l] + 2 <= n else 0 for l in range(64)]      # (index n: stale, never chosen)
        r = [has[l] and 2 * q[l] + 2 < n and not lt(a[l], b[l]) for l in range(64)]
        m = [b[l] if r[l] else a[l] for l in range(64)]
        stop = [(not has[l]) or lt(item, m[l]) for l in range(64)]
        R = sum(1 << l for l in range(64) if r[l])
        S = sum(1 << l for l in range(64) if stop[l])
        on = [l < 63 and (R & LANES[l][2]) == LANES[l][3] and (S & LANES[l][2]) == 0 for l in range(64)]
        OP = sum(1 << l for l in range(64) if on[l])
        for l in range(64):
            if on[l] and not stop[l]:
                heap[q[l]] = m[l]
        ST = OP & S
        if ST:
            assert bin(ST).count("1") == 1
            t = (ST & -ST).bit_length() - 1
            heap[q[t]] = item
            return
        d = OP.bit_length() - 1
        assert LANES[d][0] == 5                       # a node of the sixth level
        pos = 2 * q[d] + 1 + ((R >> d) & 1)


def append(heap, p, item):
    """sok_duo_append: `item` to the new leaf p."""
    up = [(p + 1) >> (l + 1) if l < 16 else 0 for l in range(64)]
    v = [heap[up[l] - 1] if up[l] else 0 for l in range(64)]
    beats = sum(1 << l for l in range(64) if up[l] and lt(item, v[l]))
    c = (~beats & (beats + 1)).bit_length() - 1     # trailing ones
    for l in range(c):
        heap[((p + 1) >> l) - 1] = v[l]
    heap[((p + 1) >> c) - 1] = item


@pytest.mark.parametrize("seed,spread,n_ops,p_push", [(0, 1, 4000, 0.75), (1, 2, 6000, 0.7), (2, 3, 6000, 0.62), (3, 40, 6000, 0.6),
                                                      (4, 400, 5000, 0.8), (5, 2, 1500, 0.5)])
def test_lane_model_matches_heapq(seed, spread, n_ops, p_push):
    rs = np.random.RandomState(seed)
    ref = []
    heap = [0] * 8192
    n = 0
    idx = 0
    for i in range(n_ops):
        drain = i > 0.8 * n_ops
        if rs.rand() < (0.3 if drain else p_push) and n < 8000:
            w = (min(0xFFF0, i // 100 + int(rs.randint(spread))) << 16) | (idx & 0xFFFF)
            idx += 1
            heapq.heappush(ref, W(w))
            append(heap, n, w)
            n += 1
        elif n:
            top = heapq.heappop(ref).w
            assert heap[0] == top
            n -= 1
            last = heap[n]
            if n:
                repair(heap, n, last)
        if i % 64 == 0 or i == n_ops - 1:
            assert heap[:n] == [x.w for x in ref], i
    assert n == len(ref)
