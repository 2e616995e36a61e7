// The level builders of the three search problems by the 64 lanes of a wavefront (round 5).
//
// sok_build_level / md_build_level / dd_build_level (+ sok_init_deadlocks, mdf_level, ddf_level) walk the bordered level cell by
// cell on ONE lane: 50..120 cells x a few dozen dependent instructions at the 8-9 cycles a lone lane gets = ~40 us per level
// (tools/probe/async_prof.py: 41 % of a tick of pcgrl_step_async on C4, where 1 650 levels are built per tick).  Here lane i takes
// cell 64 r + i: the cell's tile, one ballot per tile class gives 64 bits of every mask at once, row-major lists (crates,
// targets, things on the floor) are filled through the rank of a lane among the set bits below it.  Same structs, same contents
// as the one-lane builders (sokoban_prob.py:85-102, engine.py:135-184, 203-246; mdungeon_prob.py:92-108; ddave_prob.py:93-109) --
// every field, including which player / door / key wins when a map holds several (the last in row-major order).
// Part of the single translation unit pcgrl_abi.hip; device only.  `m`: the map bytes (global or LDS), L / root / F in LDS;
// every lane of the wavefront calls, the results are visible to all of them on return.
#pragma once
#if defined(__HIPCC__)

// cell p = 64 r + lane of a bordered (W + 2) x (H + 2) level: coordinates and tile (border cells are solid = 1; p >= cells: -1)
struct WaveCell { int p, x, y, t; };
__device__ __forceinline__ WaveCell wave_cell(const uint8_t* m, int W, int H, int r, int lane) {
    const int w = W + 2, cells = w * (H + 2);
    WaveCell c;
    c.p = r * 64 + lane;
    c.y = c.p / w;                       // (an integer division per cell, all lanes at once)
    c.x = c.p - c.y * w;
    c.t = -1;
    if (c.p < cells) {
        const bool border = c.x == 0 || c.y == 0 || c.x == w - 1 || c.y == H + 1;
        c.t = border ? 1 : (int)m[(c.y - 1) * W + (c.x - 1)];
    }
    return c;
}
__device__ __forceinline__ int wave_last_bit(uint64_t mask) { return 63 - __builtin_clzll(mask); }
// bit p of a four-word set held in registers (selects, not a dynamically indexed private array)
__device__ __forceinline__ bool wave_bit4(const uint64_t* m, int p) {
    const int k = p >> 6;
    const uint64_t v = k == 0 ? m[0] : (k == 1 ? m[1] : (k == 2 ? m[2] : m[3]));
    return (v >> (p & 63)) & 1ull;
}
__device__ __forceinline__ void wave_set4(uint64_t* m, int p) {
    const int k = p >> 6;
    const uint64_t b = 1ull << (p & 63);
    m[0] |= k == 0 ? b : 0ull; m[1] |= k == 1 ? b : 0ull; m[2] |= k == 2 ? b : 0ull; m[3] |= k == 3 ? b : 0ull;
}

// sok_build_level: returns the number of crates found (may exceed SOK_MAXC; the lists are then truncated)
__device__ __forceinline__ int sok_build_level_wave(const uint8_t* m, int W, int H, SokLevel& L, SokNode& root, int lane) {
    const int w = W + 2, h = H + 2, cells = w * h;
    if (lane < SOK_MAXC) { root.crate[lane] = 0; L.target[lane] = 0; }
    __builtin_amdgcn_wave_barrier();
    int ncr = 0, nt = 0, player = 0;
    uint64_t solid[4] = {0, 0, 0, 0}, tmask[4] = {0, 0, 0, 0};
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r * 64 >= cells) break;
        const WaveCell c = wave_cell(m, W, H, r, lane);
        if (c.t >= 0) { L.cx[c.p] = (uint8_t)c.x; L.cy[c.p] = (uint8_t)c.y; }
        const uint64_t mp = __ballot(c.t == 2), mc = __ballot(c.t == 3), mt = __ballot(c.t == 4);
        solid[r] = __ballot(c.t == 1); tmask[r] = mt;
        if (mp) player = r * 64 + wave_last_bit(mp);
        if (c.t == 3) { const int k = ncr + __popcll(mc & below); if (k < SOK_MAXC) root.crate[k] = (uint8_t)c.p; }
        if (c.t == 4) { const int k = nt + __popcll(mt & below); if (k < SOK_MAXC) L.target[k] = (uint8_t)c.p; }
        ncr += __popcll(mc); nt += __popcll(mt);
    }
    if (lane == 0) {
        L.w = w; L.h = h; L.cells = cells; L.nc = ncr < SOK_MAXC ? ncr : SOK_MAXC;
        L.dirs[0] = -1; L.dirs[1] = 1; L.dirs[2] = -w; L.dirs[3] = w;
        for (int k = 0; k < 4; k++) { L.solid[k] = solid[k]; L.targetmask[k] = tmask[k]; L.dead[k] = 0; }
        root.player = (uint8_t)player; root.pad = 0; root.pad2 = 0; root.depth = 0; root.h = 0;
    }
    __threadfence_block();
    return ncr;
}

// sok_init_deadlocks (engine.py:203-246): corner cells that are not targets, and the wall-hugging runs between two of them.
// `corners`: 64 bytes of LDS scratch.
__device__ __forceinline__ void sok_init_deadlocks_wave(SokLevel& L, uint8_t* corners, int lane) {
    const int w = L.w, cells = L.cells;
    uint64_t solid[4], tmask[4];
    for (int k = 0; k < 4; k++) { solid[k] = L.solid[k]; tmask[k] = L.targetmask[k]; }
    uint64_t dead[4] = {0, 0, 0, 0};
    const uint64_t below = (1ull << lane) - 1ull;
    int nc = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r * 64 >= cells) break;
        const int p = r * 64 + lane;
        bool corner = false;
        if (p < cells) {
            const int y = (int)L.cy[p], x = (int)L.cx[p];
            if (x >= 1 && x < w - 1 && y >= 1 && y < L.h - 1 && !wave_bit4(solid, p)) {
                const bool up = wave_bit4(solid, p - w), dn = wave_bit4(solid, p + w), lf = wave_bit4(solid, p - 1), rt = wave_bit4(solid, p + 1);
                corner = ((up && lf) || (up && rt) || (dn && lf) || (dn && rt)) && !wave_bit4(tmask, p);
            }
        }
        const uint64_t mc = __ballot(corner);
        dead[r] = mc;
        if (corner) { const int k = nc + __popcll(mc & below); if (k < 64) corners[k] = (uint8_t)p; }     // (the one-lane builder keeps the first 64)
        nc += __popcll(mc);
    }
    if (nc > 64) nc = 64;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // pairs (a, b): lane b, a in a uniform loop
    const int pb = lane < nc ? (int)corners[lane] : 0;
    const int bx = (int)L.cx[pb], by = (int)L.cy[pb];
    uint64_t mine[4] = {0, 0, 0, 0};
    for (int a = 0; a < nc; a++) {
        const int pa = (int)corners[a];
        const int ax = (int)L.cx[pa], ay = (int)L.cy[pa];
        if (lane >= nc) continue;
        const int dx = (ax > bx) - (ax < bx), dy = (ay > by) - (ay < by);
        if ((dx == 0 && dy == 0) || (dx != 0 && dy != 0)) continue;
        bool ok = true;
        if (dx != 0) {
            for (int x = bx + dx; x != ax; x += dx) {
                const int p = by * w + x;
                if (wave_bit4(tmask, p) || wave_bit4(solid, p) || (!wave_bit4(solid, p - w) && !wave_bit4(solid, p + w))) { ok = false; break; }
            }
            if (ok) for (int x = bx + dx; x != ax; x += dx) wave_set4(mine, by * w + x);
        } else {
            for (int y = by + dy; y != ay; y += dy) {
                const int p = y * w + bx;
                if (wave_bit4(tmask, p) || wave_bit4(solid, p) || (!wave_bit4(solid, p - 1) && !wave_bit4(solid, p + 1))) { ok = false; break; }
            }
            if (ok) for (int y = by + dy; y != ay; y += dy) wave_set4(mine, y * w + bx);
        }
    }
    // OR over the lanes (a butterfly of DPP-free shuffles: this runs once per level)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t v = mine[k];
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
            v |= ((uint64_t)hi << 32) | lo;
        }
        dead[k] |= v;
    }
    if (lane == 0) for (int k = 0; k < 4; k++) L.dead[k] = dead[k];
    __threadfence_block();
}

// md_build_level (mdungeon_prob.py:92-108 + engine.py:143-181) and mdf_level (the things on the floor numbered in row-major order)
__device__ __forceinline__ int md_build_level_wave(const uint8_t* m, int W, int H, MdLevel& L, MdNode& root, MdFastLevel& F, int lane) {
    const int w = W + 2, h = H + 2, cells = w * h;
    int player = 0, door = 0, n = 0;
    uint64_t solid[4] = {0, 0, 0, 0}, potion[4] = {0, 0, 0, 0}, treasure[4] = {0, 0, 0, 0}, goblin[4] = {0, 0, 0, 0}, ogre[4] = {0, 0, 0, 0}, alive[4] = {0, 0, 0, 0};
    uint64_t f_potion = 0, f_treasure = 0, f_ogre = 0;
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r * 64 >= cells) break;
        const WaveCell c = wave_cell(m, W, H, r, lane);
        if (c.t >= 0) { L.cx[c.p] = (uint8_t)c.x; L.cy[c.p] = (uint8_t)c.y; }
        const uint64_t mp = __ballot(c.t == 2), md = __ballot(c.t == 3), ma = __ballot(c.t >= 4);
        solid[r] = __ballot(c.t == 1); potion[r] = __ballot(c.t == 4); treasure[r] = __ballot(c.t == 5); goblin[r] = __ballot(c.t == 6); ogre[r] = __ballot(c.t == 7);
        alive[r] = ma;
        if (mp) player = r * 64 + wave_last_bit(mp);
        if (md) door = r * 64 + wave_last_bit(md);
        // mdf_level: thing number = rank among the cells with something on them; only the first MDF_MAXI get a number and a class bit
        const int k = n + __popcll(ma & below);
        const bool numbered = c.t >= 4 && k < MDF_MAXI;
        if (c.t >= 0) F.item[c.p] = numbered ? (uint8_t)k : (uint8_t)255;
        // class masks over thing numbers: bit k for this lane's thing -- OR over the lanes of this round
        const uint64_t bit = numbered ? (1ull << k) : 0ull;
        uint64_t bp = c.t == 4 ? bit : 0ull, bt = c.t == 5 ? bit : 0ull, bo = c.t == 7 ? bit : 0ull;
        for (int o = 32; o > 0; o >>= 1) {
            bp |= ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(bp >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)bp, o, 64);
            bt |= ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(bt >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)bt, o, 64);
            bo |= ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(bo >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)bo, o, 64);
        }
        f_potion |= bp; f_treasure |= bt; f_ogre |= bo;
        n += __popcll(ma);
    }
    if (lane == 0) {
        L.w = w; L.h = h; L.cells = cells; L.door = door;
        L.dirs[0] = -1; L.dirs[1] = 1; L.dirs[2] = -w; L.dirs[3] = w;
        for (int k = 0; k < 4; k++) { L.solid[k] = solid[k]; L.potion[k] = potion[k]; L.treasure[k] = treasure[k]; L.goblin[k] = goblin[k]; L.ogre[k] = ogre[k]; root.alive[k] = alive[k]; }
        root.player = (uint8_t)player; root.health = 5; root.depth = 0; root.treasures = 0; root.pad = 0;
        F.potion_m = f_potion; F.treasure_m = f_treasure; F.ogre_m = f_ogre;
        F.nitems = n;
        F.alive0 = n >= 64 ? ~0ull : ((1ull << n) - 1);
    }
    __threadfence_block();
    if (lane == 0) root.h = (int16_t)md_heuristic(L, player, 5, 0);
    __threadfence_block();
    return n;
}

// dd_build_level (ddave_prob.py:93-109 + engine.py:141-190) and ddf_level (the diamonds numbered in row-major order)
__device__ __forceinline__ int dd_build_level_wave(const uint8_t* m, int W, int H, DdLevel& L, DdNode& root, DdFastLevel& F, int lane) {
    const int w = W + 2, h = H + 2, cells = w * h;
    int player = 0, door = 0, keycell = 0, n = 0;
    bool key_there = false;
    uint64_t solid[4] = {0, 0, 0, 0}, spike[4] = {0, 0, 0, 0}, diamond[4] = {0, 0, 0, 0};
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r * 64 >= cells) break;
        const WaveCell c = wave_cell(m, W, H, r, lane);
        if (c.t >= 0) { L.cx[c.p] = (uint8_t)c.x; L.cy[c.p] = (uint8_t)c.y; }
        const uint64_t mp = __ballot(c.t == 2), md = __ballot(c.t == 3), mk = __ballot(c.t == 5), mdi = __ballot(c.t == 4);
        solid[r] = __ballot(c.t == 1); spike[r] = __ballot(c.t == 6); diamond[r] = mdi;
        if (mp) player = r * 64 + wave_last_bit(mp);
        if (md) door = r * 64 + wave_last_bit(md);
        if (mk) { keycell = r * 64 + wave_last_bit(mk); key_there = true; }
        const int k = n + __popcll(mdi & below);
        if (c.t >= 0) F.item[c.p] = (c.t == 4 && k < DDF_MAXD) ? (uint8_t)k : (uint8_t)255;
        n += __popcll(mdi);
    }
    if (lane == 0) {
        L.w = w; L.h = h; L.cells = cells; L.door = door; L.keycell = keycell;
        for (int k = 0; k < 4; k++) { L.solid[k] = solid[k]; L.spike[k] = spike[k]; L.diamond0[k] = diamond[k]; root.alive[k] = diamond[k]; }
        root.player = (uint8_t)player; root.flags = (uint8_t)(DD_F_HEALTH | (key_there ? DD_F_KEY_THERE : 0)); root.depth = 0; root.jumps = 0;
        F.ndiamonds = n;
        F.alive0 = n >= 64 ? ~0ull : ((1ull << n) - 1);
    }
    __threadfence_block();
    if (lane == 0) root.h = (int16_t)dd_heuristic(L, player, key_there, 0);
    __threadfence_block();
    return n;
}
#endif
