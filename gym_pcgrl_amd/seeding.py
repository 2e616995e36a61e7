"""Seed -> MT19937 state, the way the reference's RNGs are seeded.

Reference call sites: representation.py:28-30 and problem.py:34-36 both call
`gym.utils.seeding.np_random(seed)`; pcgrl_env.py:54-57 feeds the *same* seed to
both, so the representation stream and the problem stream start identical.

`gym` itself is not vendored in the reference (setup.py:8, unpinned; era gym<=0.21
because `RandomState.randint/.random` are used).  Its published algorithm:

    seed  -> create_seed: int % 2**64
          -> hash_seed:   first 8 bytes of sha512(str(seed)) read as little-endian u32 words
                          (zero padded; bigint = sum(word_i << 32 i))
          -> _int_list_from_bigint: base-2**32 digits, least significant first ([0] for 0)
          -> numpy.random.RandomState().seed(list)   == MT19937 init_by_array(list)

Host side only; the device consumes the resulting 624-word state.  The device keeps the
state as a *lazy circular buffer* (see csrc/mt19937.h): right after seeding every slot is
"previous generation" and the cursor is 0, which is exactly numpy's (key, pos=624).
"""
import hashlib
import os
import struct

import numpy as np

MT_N = 624


def create_seed(a=None, max_bytes=8):
    if a is None:
        a = int.from_bytes(os.urandom(max_bytes), "little")
    elif isinstance(a, (int, np.integer)):
        a = int(a) % 2 ** (8 * max_bytes)
    else:
        raise TypeError("Invalid type for seed: %r" % (a,))
    return a


def hash_seed_words(seed):
    """u32 key list handed to init_by_array for an (already reduced) integer seed."""
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()[:8]
    w0, w1 = struct.unpack("<2I", digest)
    big = w0 + (w1 << 32)
    if big == 0:
        return [0]
    words = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        words.append(mod)
    return words


def init_by_array(key):
    """MT19937 init_by_array (Matsumoto & Nishimura 2002), vectorised over nothing: 624 words."""
    mt = np.empty(MT_N, dtype=np.uint64)
    mt[0] = 19650218
    for i in range(1, MT_N):
        mt[i] = (1812433253 * (int(mt[i - 1]) ^ (int(mt[i - 1]) >> 30)) + i) & 0xFFFFFFFF
    mt = [int(v) for v in mt]
    i, j = 1, 0
    klen = len(key)
    for _ in range(max(MT_N, klen)):
        mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525)) + key[j] + j) & 0xFFFFFFFF
        i += 1
        j += 1
        if i >= MT_N:
            mt[0] = mt[MT_N - 1]
            i = 1
        if j >= klen:
            j = 0
    for _ in range(MT_N - 1):
        mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941)) - i) & 0xFFFFFFFF
        i += 1
        if i >= MT_N:
            mt[0] = mt[MT_N - 1]
            i = 1
    mt[0] = 0x80000000
    return np.asarray(mt, dtype=np.uint32)


def mt_state_for_seed(seed):
    """624-word MT19937 key for `np_random(seed)`; uses numpy's own init_by_array."""
    words = hash_seed_words(create_seed(seed))
    rs = np.random.RandomState()
    rs.seed(words)
    st = rs.get_state()
    assert st[0] == "MT19937" and st[2] == MT_N
    return np.asarray(st[1], dtype=np.uint32)


def init_by_array_batch(keys):
    """init_by_array for many keys of one length at once: keys [B, klen] uint -> [B, 624] uint32.

    Same recurrence as `init_by_array`, with the B independent generators as the numpy vector
    axis (the 2*624 sequential steps stay a Python loop)."""
    keys = np.asarray(keys, dtype=np.uint64)
    B, klen = keys.shape
    M32 = np.uint64(0xFFFFFFFF)
    base = np.empty(MT_N, dtype=np.uint64)
    base[0] = 19650218
    for i in range(1, MT_N):
        base[i] = (1812433253 * (int(base[i - 1]) ^ (int(base[i - 1]) >> 30)) + i) & 0xFFFFFFFF
    mt = np.tile(base, (B, 1)).T.copy()          # [624, B], rows contiguous
    i, j = 1, 0
    for _ in range(max(MT_N, klen)):
        prev = mt[i - 1]
        mt[i] = ((mt[i] ^ (((prev ^ (prev >> np.uint64(30))) * np.uint64(1664525)) & M32)) + keys[:, j] + np.uint64(j)) & M32
        i += 1
        j += 1
        if i >= MT_N:
            mt[0] = mt[MT_N - 1]
            i = 1
        if j >= klen:
            j = 0
    for _ in range(MT_N - 1):
        prev = mt[i - 1]
        mt[i] = ((mt[i] ^ (((prev ^ (prev >> np.uint64(30))) * np.uint64(1566083941)) & M32)) + (np.uint64(1 << 32) - np.uint64(i))) & M32
        i += 1
        if i >= MT_N:
            mt[0] = mt[MT_N - 1]
            i = 1
    mt[0] = 0x80000000
    return np.ascontiguousarray(mt.T).astype(np.uint32)


def mt_states_for_seeds(seeds):
    """[len(seeds), 624] uint32 MT19937 keys for `np_random(seed)` of every seed."""
    words = [hash_seed_words(create_seed(int(s))) for s in seeds]
    out = np.empty((len(words), MT_N), dtype=np.uint32)
    by_len = {}
    for k, w in enumerate(words):
        by_len.setdefault(len(w), []).append(k)
    for klen, idx in by_len.items():
        keys = np.array([words[k] for k in idx], dtype=np.uint64).reshape(len(idx), klen)
        out[idx] = init_by_array_batch(keys)
    return out


def key_words_for_seeds(seeds):
    """[len(seeds), 3] uint32: (key word 0, key word 1, number of key words) of `np_random(seed)` for every seed -- the
    input of the device-side init_by_array (pcgrl_seed_words)."""
    out = np.zeros((len(seeds), 3), dtype=np.uint32)
    for k, sd in enumerate(seeds):
        w = hash_seed_words(create_seed(int(sd)))
        out[k, :len(w)] = w
        out[k, 2] = len(w)
    return out


def np_random(seed=None):
    """Drop-in for gym<=0.21 `seeding.np_random`: (RandomState, seed)."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not %r" % (seed,))
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(hash_seed_words(seed))
    return rng, seed
