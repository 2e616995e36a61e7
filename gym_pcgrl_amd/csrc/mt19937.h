// MT19937 as a *lazy circular buffer*, plus numpy's legacy draw rules.
//
// The reference draws from numpy.random.RandomState (MT19937): `randint(n)` at
// reps/narrow_rep.py:30-31,105-106 and reps/turtle_rep.py:32-33, `choice(p)` at helper.py:311,
// `random()` at probs/binary_prob.py:71.
//
// The textbook generator regenerates all 624 words at once every 624 draws.  In a
// thread-per-environment kernel that is a 624-iteration serial loop hitting a different lane
// every step.  The recurrence itself is x[k+624] = x[k+397] ^ twist(x[k], x[k+1]), so we keep
// the 624 most recent words in a ring and produce exactly one new word per draw:
//
//     slot s = cursor;  y = x[s+397] ^ twist(x[s], x[s+1]);  x[s] = y;  cursor = s+1 (mod 624)
//
// The output stream is identical to the bulk form; right after seeding (numpy: pos = 624) the
// ring is the init_by_array key and the cursor is 0.  Because a new word only depends on words
// at distance 0, 1 and 397, up to 227 consecutive words can be produced in parallel by the lanes
// of a wavefront (used by the reset kernel).
#pragma once
#include "pcgrl_common.h"

PCGRL_HD uint32_t mt_twist(uint32_t x0, uint32_t x1, uint32_t xm) {
    uint32_t y = (x0 & 0x80000000u) | (x1 & 0x7fffffffu);
    return xm ^ (y >> 1) ^ ((x1 & 1u) ? 0x9908b0dfu : 0u);
}
PCGRL_HD uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
PCGRL_HD int mt_wrap(int s) { return s >= PCGRL_MT_N ? s - PCGRL_MT_N : s; }

// One draw from a ring living in any addressable memory (global or LDS).  `cursor` in [0,624).
template <class Ptr>
PCGRL_HD uint32_t mt_draw(Ptr ring, int& cursor) {
    int s = cursor;
    uint32_t y = mt_twist(ring[s], ring[mt_wrap(s + 1)], ring[mt_wrap(s + PCGRL_MT_M)]);
    ring[s] = y;
    cursor = mt_wrap(s + 1);
    return mt_temper(y);
}

// numpy legacy RandomState.randint(n) for 0 < n <= 2^32: masked rejection on 32-bit draws;
// n == 1 consumes nothing.
template <class Ptr>
PCGRL_HD int mt_randint(Ptr ring, int& cursor, int n) {
    uint32_t rng = (uint32_t)(n - 1);
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = mt_draw(ring, cursor) & mask; } while (v > rng);
    return (int)v;
}

// numpy legacy double in [0,1): 53 bits from two consecutive 32-bit outputs a, b.
PCGRL_HD double mt_to_double(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}
template <class Ptr>
PCGRL_HD double mt_random(Ptr ring, int& cursor) {
    uint32_t a = mt_draw(ring, cursor);
    uint32_t b = mt_draw(ring, cursor);
    return mt_to_double(a, b);
}

// numpy choice(): cdf = cumsum(p/sum(p)); cdf /= cdf[-1]  (helper.py:343-352 + RandomState.choice).
// All in IEEE fp64, no contraction (the library is built with -ffp-contract=off).
PCGRL_HD void pcgrl_build_cdf(const double* prob, int n, double* cdf) {
    double total = 0.0;
    for (int i = 0; i < n; i++) total += prob[i];
    double acc = 0.0;
    for (int i = 0; i < n; i++) {
        double p = prob[i] / total;
        acc = (i == 0) ? p : acc + p;
        cdf[i] = acc;
    }
    double last = cdf[n - 1];
    for (int i = 0; i < n; i++) cdf[i] /= last;
}
// searchsorted(cdf, u, side='right') with the number of tiles known at compile time (cdf stays in registers)
template <int N>
PCGRL_HD int pcgrl_pick_tile_c(const double* cdf, double u) {
    int idx = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < N; i++) idx += (cdf[i] <= u) ? 1 : 0;
    return idx < N ? idx : N - 1;
}
// searchsorted(cdf, u, side='right')
PCGRL_HD int pcgrl_pick_tile(const double* cdf, int n, double u) {
    int idx = 0;
    for (int i = 0; i < n; i++) idx += (cdf[i] <= u) ? 1 : 0;
    return idx < n ? idx : n - 1;
}
