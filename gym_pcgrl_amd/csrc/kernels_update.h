// k_update / k_update_block: Representation.update for every environment (thread per environment).
// Part of the single translation unit pcgrl_abi.hip (see its header comment for the overall picture).
#pragma once
// ------------------------------------------------------------------------------------------
// k_update: thread per environment
// The kernel is a chain of dependent scattered loads per thread at one wavefront per SIMD, so it is
// written to keep that chain at three round trips: (1) action, counters, cursor, old stats; (2) the map
// cell, its plane words and the MT19937 ring words of up to PCGRL_SPEC_DRAWS speculative draws (every
// operand of draw i is an *old* word: distance 397); (3) the heatmap cell of the new cursor.
#define PCGRL_SPEC_DRAWS 6
// What Representation.update + the bookkeeping of PcgrlEnv.step did to one environment, for the routing of its statistics.
struct UpdateOut {
    bool chg;          // a tile changed: the statistics have to be recomputed (or updated)
    bool rst;          // nothing changed and the episode ended (auto_reset): reset only
    bool cheap;        // binary: incremental update possible; zelda: the cell's passability did not change
    bool touch;        // binary: the change is in or next to the champion and the bound on the other components is known (binary_touch)
    bool sure_done;    // a tile changed and the episode ends whatever the new statistics are
    int bucket;        // difficulty bucket (binary)
    int inc_item;      // packed (environment, cell, passability change) for the incremental routes
    // FIFO form (fused step kernel): the cursor draws came out of the environment's draw cache; `k` of its words were
    // consumed (cursor before: cur0) and still have to be written to the ring, and the cache refilled (fifo_refill)
    int k, cur0;
};
// One environment of k_update (thread per environment; also the first phase of the fused step kernel k_step).
// FIFO (k_step only): B's per-environment state pointers lead into the block's LDS copy; cursor draws come from the draw
// cache; an environment that is certain to be reset writes nothing to the byte map and the heatmap (the reset rewrites both).
// SPLIT (k_step): the decision part only -- the tile write, the counters, what the statistics will have to do, the unchanged
// environments finished -- so that the block's task lists are complete before the narrow representation's cursor draws (the
// longest piece of an update: up to eight tempered words and a rejection loop) and the heat-map increment are done; those follow
// in update_env_cursor, behind the block's list barrier, while the other wavefronts already compute.  `mid` carries what they need.
struct UpdateMid { int x, y, hx, hy, cur; bool chg, dead; };
template <int REP, class MaskT, bool FIFO = false, bool SPLIT = false>
__device__ __forceinline__ UpdateOut update_env(const PcgrlParams& P, const DevBufs& B, const int32_t* __restrict__ actions, int e, UpdateMid* mid = nullptr) {
    bool chg = false, rst = false, cheap = false, sure_done = false, touch_item = false;
    int bucket = 0, inc_item = 0, k_used = 0, cur0 = 0;
    uint32_t fw[PCGRL_FIFO_N];
#pragma unroll
    for (int i = 0; i < PCGRL_FIFO_N; i++) fw[i] = 0;
    {
        const int W = P.width, H = P.height, G = P.group, NPL = P.nplanes;
        // ---- round trip 1
        const int2 c = reinterpret_cast<const int2*>(B.counters)[e];
        int a0_ = 0, a1_ = 0, a2_ = 0;
        if (REP == PCGRL_REP_WIDE) { a0_ = actions[3 * e + 0]; a1_ = actions[3 * e + 1]; a2_ = actions[3 * e + 2]; }
        else a0_ = actions[e];
        uchar2 p0 = make_uchar2(0, 0);
        if (REP != PCGRL_REP_WIDE) p0 = reinterpret_cast<const uchar2*>(B.pos)[e];
        const bool draws = REP == PCGRL_REP_NARROW && P.random_tile;
        int cur = 0;
        if (draws) cur = B.rng_cur[2 * e];
        const int4* sp = reinterpret_cast<const int4*>(B.stats + (size_t)e * 8);
        const int4* tp = reinterpret_cast<const int4*>(B.start_stats + (size_t)e * 8);
        const int4 s0 = sp[0], s1 = sp[1], t0 = tp[0], t1 = tp[1];

        bucket = difficulty_bucket(P, s0, s1, B.wide_few);
        const int iter = c.x + 1;
        int changes = c.y;
        int x = p0.x, y = p0.y;
        int tile = -1, wx = 0, wy = 0, hx = 0, hy = 0;
        // An action outside the action space is clamped into it (the reference raises IndexError or writes the bad value)
        // and reported through the sticky status word: PCGRL_STATUS_BAD_ACTION, BatchedPcgrlEnv.check_status().
        bool bad = false;
        if (REP == PCGRL_REP_NARROW) {
            const int a = clampi(a0_, 0, P.ntiles);
            bad = a != a0_;
            if (a > 0) tile = a - 1;
            wx = x; wy = y;
        } else if (REP == PCGRL_REP_WIDE) {
            wx = clampi(a0_, 0, W - 1);
            wy = clampi(a1_, 0, H - 1);
            tile = clampi(a2_, 0, P.ntiles - 1);
            bad = wx != a0_ || wy != a1_ || tile != a2_;
            hx = wx; hy = wy;
        } else {
            const int a = clampi(a0_, 0, P.ntiles + 3);
            bad = a != a0_;
            if (a < 4) {   // turtle_rep.py:18,103-125: L,R,U,D with clamp or warp on both axes
                const int dx = (a == 0) ? -1 : (a == 1 ? 1 : 0), dy = (a == 2) ? -1 : (a == 3 ? 1 : 0);
                x += dx;
                if (x < 0) x = P.warp ? x + W : 0;
                if (x >= W) x = P.warp ? x - W : W - 1;
                y += dy;
                if (y < 0) y = P.warp ? y + H : 0;
                if (y >= H) y = P.warp ? y - H : H - 1;
            } else {
                tile = a - 4;
            }
            wx = x; wy = y; hx = x; hy = y;
        }
        // ---- round trip 2: everything addressed by the cursor cell and by the ring cursor
        uint8_t* cell = B.map + ((size_t)e * H + wy) * W + wx;
        MaskT* pl = reinterpret_cast<MaskT*>(B.planes) + ((size_t)e * G + wy) * NPL;   // the planes of one row are adjacent
        // the old tile: from the plane word when there is a single plane (one scattered read less), else from the byte map
        // (nothing of the cell is needed when the action writes no tile: a third of the narrow actions, the moves of turtle)
        const bool writes = tile >= 0;
        // binary, 16-row maps: the three champion rows around the cell, for the routing decision below
        const bool inc_on = P.prob == PCGRL_PROB_BINARY && B.champ != nullptr;
        MaskT ch0 = 0, chu = 0, chd = 0;
        bool big_touch = false;
        if (inc_on && writes && P.big) {
            // maps beyond 64 x 64 (bigmap.h): the champion's rows are [H][KW] 64-bit words per environment; is the cell in it or next to it?
            const int KW = (W + 63) >> 6;
            const uint64_t* ch = reinterpret_cast<const uint64_t*>(B.champ) + (size_t)e * H * KW;
            const int xs[5] = {wx, wx - 1, wx + 1, wx, wx}, ys[5] = {wy, wy, wy, wy - 1, wy + 1};
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const int xx = xs[q], yy = ys[q];
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) big_touch = big_touch || ((ch[yy * KW + (xx >> 6)] >> (xx & 63)) & 1ull) != 0ull;
            }
        } else if (inc_on && writes) {
            const MaskT* ch = reinterpret_cast<const MaskT*>(B.champ) + (size_t)e * G;
            ch0 = ch[wy];
            chu = ch[wy > 0 ? wy - 1 : wy];
            chd = ch[wy < G - 1 ? wy + 1 : wy];
        }
        MaskT m0 = 0, m1 = 0, m2 = 0;
        int old_byte = 0;
        if (writes) {
            if (NPL == 0) old_byte = (int)*cell;          // smb keeps no bit planes (114 columns): the old tile comes from the byte map
            else {
                m0 = pl[0];
                if (NPL > 1) { m1 = pl[1]; m2 = pl[2]; }
            }
        }
        const int old = NPL == 0 ? old_byte : ((int)((m0 >> wx) & 1) | ((NPL > 1) ? (int)(((m1 >> wx) & 1) << 1) | (int)(((m2 >> wx) & 1) << 2) : 0));
        uint32_t* ring = B.rng_rep + (size_t)e * PCGRL_MT_N;
        uint32_t xa[PCGRL_SPEC_DRAWS + 1], xb[PCGRL_SPEC_DRAWS];
        cur0 = cur;
        if (draws && FIFO && !SPLIT) {
            const int tag = B.fifo_tag[e];
            const uint4* fp = reinterpret_cast<const uint4*>(B.fifo + (size_t)e * PCGRL_FIFO_N);
            const uint4 fa = fp[0], fb = fp[1];
            fw[0] = fa.x; fw[1] = fa.y; fw[2] = fa.z; fw[3] = fa.w; fw[4] = fb.x; fw[5] = fb.y; fw[6] = fb.z; fw[7] = fb.w;
            if (tag != cur) {     // not made for this cursor (right after seeding, or another pipeline drew from the ring): make it now
#pragma unroll
                for (int i = 0; i < PCGRL_FIFO_N; i++) {
                    fw[i] = mt_twist(ring[mt_wrap(cur + i)], ring[mt_wrap(cur + i + 1)], ring[mt_wrap(mt_wrap(cur + PCGRL_MT_M) + i)]);
                    B.fifo[(size_t)e * PCGRL_FIFO_N + i] = fw[i];
                }
            }
        }
        if (draws && !FIFO) {
#pragma unroll
            for (int i = 0; i <= PCGRL_SPEC_DRAWS; i++) xa[i] = ring[mt_wrap(cur + i)];
#pragma unroll
            for (int i = 0; i < PCGRL_SPEC_DRAWS; i++) xb[i] = ring[mt_wrap(mt_wrap(cur + PCGRL_MT_M) + i)];
        }
        if (tile >= 0 && old != tile) {
            chg = true;
            if (inc_on && s0.z != 0 && P.big) {
                cheap = !big_touch;                           // (big_incremental, bigmap.h)
                inc_item = wl_incbig_pack(e, wx, wy, tile == 0);
            } else if (inc_on && s0.z != 0) {
                // the statistics can be updated from the previous ones when the cell is neither in the champion component
                // nor next to it (binary_incremental); s0.z = "there is a champion"
                const MaskT bit = (MaskT)1 << wx;
                const MaskT touch = ((ch0 | chu | chd) & bit) | (ch0 & ((bit << 1) | (bit >> 1)));
                cheap = touch == 0;
                touch_item = touch != 0 && s0.w != 0;         // s0.w = bound on the other components + 1, 0 = not known
                inc_item = wl_inc_pack(G, e, wy, wx, tile == 0 ? 1u : 0u);
            }
            if (P.prob == PCGRL_PROB_SMB && B.champ != nullptr) {
                // smb: the play-through reads the level only as "blocked / free" (kernels_smb.h: solid, brick, question and tube block),
                // and only in the cells its searches looked at (DevBufs::champ: a row mask per map column, written by k_smb).  A change
                // that keeps the cell's kind, or a cell no search read, leaves jumps / jumps-dist / dist-win what they were.
                const bool so = (0x5Au >> old) & 1u, sn = (0x5Au >> tile) & 1u;
                const uint32_t seen = reinterpret_cast<const uint32_t*>(B.champ)[(size_t)e * W + wx];
                cheap = so == sn || ((seen >> wy) & 1u) == 0u;
                inc_item = e | SMB_KEEP_PLAY;
            }
            if (B.zelda_inc) {
                // zelda: what the write does to the cell's passability for the region count (zelda_prob.py:93: everything
                // but solid = 1 and door = 4), so that k_stats can update the count instead of recounting
                const bool po = old != 1 && old != 4, pn = tile != 1 && tile != 4;
                inc_item = wl_inc_pack(16, e, wy, wx, po == pn ? 0u : (pn ? 1u : 2u));
                cheap = po == pn;
            }
            // (the fused step kernel leaves the byte map and the heatmap of an environment that is certain to be reset to the
            //  reset, which rewrites both: nothing of wavefront 0 is then in flight for it when the reset starts)
            const bool dead_writes = FIFO && P.auto_reset && B.inline_reset && (c.y + 1 >= P.max_changes || iter >= P.max_iterations);
            if (!dead_writes) *cell = (uint8_t)tile;
            const MaskT bit = (MaskT)1 << wx;
            if (NPL > 0) pl[0] = (tile & 1) ? (m0 | bit) : (m0 & ~bit);
            if (NPL > 1) {
                pl[1] = (tile & 2) ? (m1 | bit) : (m1 & ~bit);
                pl[2] = (tile & 4) ? (m2 | bit) : (m2 & ~bit);
            }
        }
        if (REP == PCGRL_REP_NARROW && !SPLIT) {   // the cursor moves on every step (narrow_rep.py:104-113)
            if (draws && FIFO) {
                // numpy randint(W) then randint(H) (masked rejection) on the words of the draw cache
                const uint32_t rx = (uint32_t)(W - 1), ry = (uint32_t)(H - 1);
                uint32_t mx = rx, my = ry;
                mx |= mx >> 1; mx |= mx >> 2; mx |= mx >> 4; mx |= mx >> 8; mx |= mx >> 16;
                my |= my >> 1; my |= my >> 2; my |= my >> 4; my |= my >> 8; my |= my >> 16;
                int stage = 0, used = 0;
                if (rx == 0) { x = 0; stage = 1; }
                if (stage == 1 && ry == 0) { y = 0; stage = 2; }
#pragma unroll
                for (int i = 0; i < PCGRL_FIFO_N; i++) {
                    if (stage < 2) {
                        used = i + 1;
                        const uint32_t v = mt_temper(fw[i]);
                        if (stage == 0) {
                            if ((v & mx) <= rx) { x = (int)(v & mx); stage = 1; if (ry == 0) { y = 0; stage = 2; } }
                        } else {
                            if ((v & my) <= ry) { y = (int)(v & my); stage = 2; }
                        }
                    }
                }
                if (stage < 2) {
                    // every cached word rejected ((1/8)^k tail): put them into the ring and go on there; the cache is rebuilt
                    // by the next step
#pragma unroll
                    for (int i = 0; i < PCGRL_FIFO_N; i++) ring[mt_wrap(cur + i)] = fw[i];
                    cur = mt_wrap(cur + PCGRL_FIFO_N);
                    if (stage == 0) { x = mt_randint(ring, cur, W); y = mt_randint(ring, cur, H); }
                    else y = mt_randint(ring, cur, H);
                    B.fifo_tag[e] = -1;
                    // a reset of this environment later in the launch stages the ring from memory, possibly on another wavefront
                    // behind an LDS-only barrier: the ring words have to have left this wavefront before it gets there (a
                    // workgroup-scope fence alone omits the vmcnt wait outside threadgroup-split mode)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __threadfence_block();
                    cur0 = cur;
                } else {
                    cur = mt_wrap(cur + used);
                    k_used = used;
                }
                B.rng_cur[2 * e] = cur;
            } else if (draws) {
                if (B.fifo_tag) B.fifo_tag[e] = -1;     // the draws below bypass the cache
                // numpy randint(W) then randint(H): masked rejection, consumed in order from the speculative words
                const uint32_t rx = (uint32_t)(W - 1), ry = (uint32_t)(H - 1);
                uint32_t mx = rx, my = ry;
                mx |= mx >> 1; mx |= mx >> 2; mx |= mx >> 4; mx |= mx >> 8; mx |= mx >> 16;
                my |= my >> 1; my |= my >> 2; my |= my >> 4; my |= my >> 8; my |= my >> 16;
                int stage = 0, used = 0;             // stage 0: drawing x, 1: drawing y, 2: done
                if (rx == 0) { x = 0; stage = 1; }   // randint(1) draws nothing
                if (stage == 1 && ry == 0) { y = 0; stage = 2; }
#pragma unroll
                for (int i = 0; i < PCGRL_SPEC_DRAWS; i++) {
                    if (stage < 2) {
                        const uint32_t yv = mt_twist(xa[i], xa[i + 1], xb[i]);
                        ring[mt_wrap(cur + i)] = yv;
                        used = i + 1;
                        const uint32_t v = mt_temper(yv);
                        if (stage == 0) {
                            if ((v & mx) <= rx) { x = (int)(v & mx); stage = 1; if (ry == 0) { y = 0; stage = 2; } }
                        } else {
                            if ((v & my) <= ry) { y = (int)(v & my); stage = 2; }
                        }
                    }
                }
                cur = mt_wrap(cur + used);
                if (stage < 2) {                      // (1/8)^k tail: finish with ordinary draws
                    if (stage == 0) { x = mt_randint(ring, cur, W); y = mt_randint(ring, cur, H); }
                    else y = mt_randint(ring, cur, H);
                }
                B.rng_cur[2 * e] = cur;
            } else {
                x += 1;
                if (x >= W) { x = 0; y += 1; if (y >= H) y = 0; }
            }
            hx = x; hy = y;   // pcgrl_env.py:137 marks the *new* cursor cell
        }
        // ---- round trip 3
        bool dead_heat = false;
        if (chg) {
            changes += 1;
            const bool dead_writes = FIFO && P.auto_reset && B.inline_reset && (changes >= P.max_changes || iter >= P.max_iterations);
            dead_heat = dead_writes;
            if (!dead_writes && !(SPLIT && REP == PCGRL_REP_NARROW)) heat_increment(B, B.heat + ((size_t)e * H + hy) * W + hx);
        }
        reinterpret_cast<int2*>(B.counters)[e] = make_int2(iter, changes);
        if (bad) atomicOr(B.status, PCGRL_STATUS_BAD_ACTION);
        if (REP != PCGRL_REP_WIDE && !(SPLIT && REP == PCGRL_REP_NARROW)) reinterpret_cast<uchar2*>(B.pos)[e] = make_uchar2((unsigned char)x, (unsigned char)y);
        if (SPLIT && mid) { mid->x = x; mid->y = y; mid->hx = hx; mid->hy = hy; mid->cur = cur; mid->chg = chg; mid->dead = dead_heat; }
        // the episode ends whatever the new statistics are (pcgrl_env.py:143): the reset is certain
        sure_done = chg && P.auto_reset && B.inline_reset && (changes >= P.max_changes || iter >= P.max_iterations);
        if (sure_done) { cheap = false; touch_item = false; }
        if (!chg) {
            // new_stats is old_stats (pcgrl_env.py:132-142): reward 0, done/info from the current stats
            int32_t s[PCGRL_MAX_STATS], st[PCGRL_MAX_STATS];
            s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w; s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
            st[0] = t0.x; st[1] = t0.y; st[2] = t0.z; st[3] = t0.w; st[4] = t1.x; st[5] = t1.y; st[6] = t1.z; st[7] = t1.w;
            const bool d = episode_over(P, s, st) || changes >= P.max_changes || iter >= P.max_iterations;
            B.reward[e] = 0.0;
            B.done[e] = d ? 1 : 0;
            episode_account(B, e, 0.0, d);
            int32_t* inf = B.info + (size_t)e * 10;
            inf[0] = s0.x; inf[1] = s0.y; inf[2] = s0.z; inf[3] = s0.w;
            inf[4] = s1.x; inf[5] = s1.y; inf[6] = s1.z; inf[7] = s1.w;
            if (P.prob == PCGRL_PROB_BINARY) { inf[2] = s0.y - t0.y; inf[3] = 0; }   // path-imp (binary_prob.py:137); slots 2 and 3 of the stats row are the library's own
            inf[8] = iter; inf[9] = changes;
            rst = d && P.auto_reset;
        }
    }
    UpdateOut o;
    o.chg = chg; o.rst = rst; o.cheap = cheap; o.touch = touch_item; o.sure_done = sure_done; o.bucket = bucket; o.inc_item = inc_item;
    o.k = k_used; o.cur0 = cur0;
    return o;
}

// The second part of a SPLIT update (k_step, narrow representation): the cursor move of the step -- numpy randint(W) then randint(H),
// masked rejection, on the words of the environment's draw cache (narrow_rep.py:104-113) --, the cursor and the heat-map cell it
// marks (pcgrl_env.py:137: the NEW cursor cell).  Not called for environments that are certain to be reset: their reset consumes the
// step's draws itself from the staged ring (wave_reset_env, step_draws) and rewrites cursor and heat map.  Returns the number of
// cache words consumed (to be written to the ring by fifo_refill; 0 when the draws went to the ring directly) and the cursor before.
template <int REP, class MaskT>
__device__ __forceinline__ void update_env_cursor(const PcgrlParams& P, const DevBufs& B, int e, const UpdateMid& mid, int& k_used, int& cur0) {
    const int W = P.width, H = P.height;
    int x = mid.x, y = mid.y, cur = mid.cur;
    k_used = 0; cur0 = cur;
    if (REP != PCGRL_REP_NARROW) return;
    if (P.random_tile) {
        uint32_t* ring = B.rng_rep + (size_t)e * PCGRL_MT_N;
        uint32_t fw[PCGRL_FIFO_N];
        const int tag = B.fifo_tag[e];
        const uint4* fp = reinterpret_cast<const uint4*>(B.fifo + (size_t)e * PCGRL_FIFO_N);
        const uint4 fa = fp[0], fb = fp[1];
        fw[0] = fa.x; fw[1] = fa.y; fw[2] = fa.z; fw[3] = fa.w; fw[4] = fb.x; fw[5] = fb.y; fw[6] = fb.z; fw[7] = fb.w;
        if (tag != cur) {     // not made for this cursor (right after seeding, or another pipeline drew from the ring): make it now
#pragma unroll
            for (int i = 0; i < PCGRL_FIFO_N; i++) {
                fw[i] = mt_twist(ring[mt_wrap(cur + i)], ring[mt_wrap(cur + i + 1)], ring[mt_wrap(mt_wrap(cur + PCGRL_MT_M) + i)]);
                B.fifo[(size_t)e * PCGRL_FIFO_N + i] = fw[i];
            }
        }
        const uint32_t rx = (uint32_t)(W - 1), ry = (uint32_t)(H - 1);
        uint32_t mx = rx, my = ry;
        mx |= mx >> 1; mx |= mx >> 2; mx |= mx >> 4; mx |= mx >> 8; mx |= mx >> 16;
        my |= my >> 1; my |= my >> 2; my |= my >> 4; my |= my >> 8; my |= my >> 16;
        int stage = 0, used = 0;
        if (rx == 0) { x = 0; stage = 1; }
        if (stage == 1 && ry == 0) { y = 0; stage = 2; }
#pragma unroll
        for (int i = 0; i < PCGRL_FIFO_N; i++) {
            if (stage < 2) {
                used = i + 1;
                const uint32_t v = mt_temper(fw[i]);
                if (stage == 0) {
                    if ((v & mx) <= rx) { x = (int)(v & mx); stage = 1; if (ry == 0) { y = 0; stage = 2; } }
                } else {
                    if ((v & my) <= ry) { y = (int)(v & my); stage = 2; }
                }
            }
        }
        if (stage < 2) {
            // every cached word rejected ((1/8)^k tail): put them into the ring and go on there; the cache is rebuilt by the next step
#pragma unroll
            for (int i = 0; i < PCGRL_FIFO_N; i++) ring[mt_wrap(cur + i)] = fw[i];
            cur = mt_wrap(cur + PCGRL_FIFO_N);
            if (stage == 0) { x = mt_randint(ring, cur, W); y = mt_randint(ring, cur, H); }
            else y = mt_randint(ring, cur, H);
            B.fifo_tag[e] = -1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (a later reset of this environment stages the ring from memory)
            __threadfence_block();
            cur0 = cur;
        } else {
            cur = mt_wrap(cur + used);
            k_used = used;
        }
        B.rng_cur[2 * e] = cur;
    } else {
        x += 1;
        if (x >= W) { x = 0; y += 1; if (y >= H) y = 0; }
    }
    if (mid.chg && !mid.dead) heat_increment(B, B.heat + ((size_t)e * H + y) * W + x);
    reinterpret_cast<uchar2*>(B.pos)[e] = make_uchar2((unsigned char)x, (unsigned char)y);
}

template <int REP, class MaskT>
__global__ __launch_bounds__(PCGRL_BLOCK) void k_update(PcgrlParams P, DevBufs B, const int32_t* __restrict__ actions, int parity) {
    __shared__ int s_cnt[3][4];
    __shared__ int s_base[3];
    __shared__ int s_hist[WL_NSHARD + 1], s_gbase[WL_NSHARD + 1];
    const int e = blockIdx.x * PCGRL_BLOCK + threadIdx.x;
    UpdateOut u = {};
    bool live = e < P.num_envs;
    if (B.pending) {
        // a tick of pcgrl_step_async: an environment whose step is still in flight sits this one out; one whose search ended its
        // episode in the last tick (kernels_search_async.h ASYNC_PEND_RESET) is reset now: it goes on the reset list like an
        // environment whose episode the update itself ends, and takes no action
        const int pv = live ? (int)B.pending[e] : 1;
        live = pv == 0;
        const int cnt = __popcll(__ballot(live));
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(B.async_stats + 8 + 8 * (blockIdx.x & 15), (unsigned long long)cnt);
        if (pv == 2) { u.rst = true; B.pending[e] = 0; }      // (a search of the new map that is cut short marks it pending again)
        async_list_runnable(B, e);
    }
    if (live) u = update_env<REP, MaskT>(P, B, actions, e);
    const bool chg = u.chg, rst = u.rst, cheap = u.cheap, sure_done = u.sure_done;
    int bucket = u.bucket;
    const int inc_item = u.inc_item;
    // bucketing pays where four maps share a wavefront and their cost varies a lot (binary); elsewhere the
    // plain per-block append is cheaper (kernel-uniform branch)
    // an unchanged environment whose episode ended rides the changed list flagged "reset only": k_stats resets
    // in-kernel (every problem but Sokoban, whose resets wait for the solver and go through k_reset)
    const bool inl = rst && B.inline_reset;
    const int val = inl ? (e | WL_RESET_ONLY) : e;
    if (P.prob == PCGRL_PROB_BINARY && P.group == 16 && !P.big) {
        // bucket 0 = environments that k_stats is certain to reset (k_stats starts those first, a wavefront each)
        bucket = (inl || sure_done) ? 0 : (bucket < 1 ? 1 : bucket);
        block_append_bucketed((chg && !cheap) || inl, bucket, val, B, parity, WL_CHG, s_hist, s_gbase, cheap, inc_item, B.champ != nullptr ? WL_INC : -1);
        if (!B.inline_reset) block_append(rst, e, B, parity, WL_RST, s_cnt[1], &s_base[1]);
    } else {
        // same idea without buckets: the certain resets go to their own list, which k_stats works through first (without the
        // in-kernel reset WL_RST is simply the reset list of k_reset).  Zelda: changed environments are split by what their
        // statistics will cost -- the region count has to be updated (WL_CHG) or not (WL_INC) -- and carry the cell.
        const bool first = B.inline_reset ? (inl || sure_done) : rst;
        int dest = -1, v = e;
        if (first) { dest = 2; v = val; }
        else if (chg) { dest = cheap ? 1 : 0; v = (cheap || B.zelda_inc) ? inc_item : e; }
        if (P.prob == PCGRL_PROB_BINARY && P.group == 64 && !P.big) {
            // tall maps: the full recomputations ranked by what they are going to cost (difficulty_bucket), dearest first -- a launch
            // of k_stats_wide ends with its last item, and a dear item that starts late is what it ends with
            block_append_bucketed(dest == 0, bucket, v, B, parity, WL_CHG, s_hist, s_gbase, dest == 1, v, WL_INC);
            block_append(dest == 2, v, B, parity, WL_RST, s_cnt[1], &s_base[1]);
            return;
        }
        // smb: a level whose last play-through was long goes on the list k_smb starts with (WL_INC, which smb has no other use for)
        // (an smb change that keeps the play-through -- `cheap`, update_env -- is a short job on the plain list, flagged)
        if (P.prob == PCGRL_PROB_SMB && dest == 1) dest = 0;
        else if (P.prob == PCGRL_PROB_SMB && dest == 0 && B.sok_cnt[e] >= SMB_LONG_POPS) dest = 1;
        block_append3(dest, v, B, parity, s_cnt, s_base);
    }
}

// ------------------------------------------------------------------------------------------
// k_update_block: the 3x3 "cast" / "multi" representations (narrow_cast_rep.py:36-59, narrow_multi_rep.py:39-59,
// turtle_cast_rep.py:38-76).  Same contract as k_update; up to nine tiles change per step and `change`
// counts them (pcgrl_env.py:136 adds it to _changes; the heatmap still gets +1).
template <int REP, class MaskT>
__global__ __launch_bounds__(PCGRL_BLOCK) void k_update_block(PcgrlParams P, DevBufs B, const int32_t* __restrict__ actions, int parity) {
    __shared__ int s_cnt[2][4];
    __shared__ int s_base[2];
    const int e = blockIdx.x * PCGRL_BLOCK + threadIdx.x;
    bool act = e < P.num_envs;
    bool chg = false, rst = false, sure_done = false;
    if (B.pending) {            // (see k_update)
        const int pv = act ? (int)B.pending[e] : 1;
        act = pv == 0;
        const int cnt = __popcll(__ballot(act));
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(B.async_stats + 8 + 8 * (blockIdx.x & 15), (unsigned long long)cnt);
        if (pv == 2) { rst = true; B.pending[e] = 0; }
        async_list_runnable(B, e);
    }
    if (act) {
        const int W = P.width, H = P.height, G = P.group, NPL = P.nplanes, NT = P.ntiles;
        const int2 c = reinterpret_cast<const int2*>(B.counters)[e];
        const uchar2 p0 = reinterpret_cast<const uchar2*>(B.pos)[e];
        const int iter = c.x + 1;
        int changes = c.y, x = p0.x, y = p0.y;
        int vals[9];
#pragma unroll
        for (int i = 0; i < 9; i++) vals[i] = -1;
        bool bad = false;     // out-of-range actions are clamped and reported (see update_env)
        if (REP == PCGRL_REP_NARROW_MULTI) {
#pragma unroll
            for (int i = 0; i < 9; i++) { const int raw = actions[9 * e + i], a = clampi(raw, 0, NT); bad = bad || a != raw; vals[i] = a - 1; }
        } else {
            const int type = actions[2 * e], value = clampi(actions[2 * e + 1], 0, NT - 1);
            bad = value != actions[2 * e + 1] || type < 0 || type > (REP == PCGRL_REP_NARROW_CAST ? 2 : 5);
            if (REP == PCGRL_REP_NARROW_CAST) {
                const int t = clampi(type, 0, 2);
                if (t == 1) vals[4] = value;
                if (t == 2) { for (int i = 0; i < 9; i++) vals[i] = value; }
            } else {
                const int t = clampi(type, 0, 5);
                if (t < 4) {   // turtle move (turtle_rep.py:103-125 semantics)
                    const int dx = (t == 0) ? -1 : (t == 1 ? 1 : 0), dy = (t == 2) ? -1 : (t == 3 ? 1 : 0);
                    x += dx;
                    if (x < 0) x = P.warp ? x + W : 0;
                    if (x >= W) x = P.warp ? x - W : W - 1;
                    y += dy;
                    if (y < 0) y = P.warp ? y + H : 0;
                    if (y >= H) y = P.warp ? y - H : H - 1;
                }
                if (t == 4) vals[4] = value;
                if (t == 5) { for (int i = 0; i < 9; i++) vals[i] = value; }
            }
        }
        int change = 0;
        uint8_t* map_e = B.map + (size_t)e * H * W;
        MaskT* pl_e = reinterpret_cast<MaskT*>(B.planes) + (size_t)e * NPL * G;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            if (vals[(dy + 1) * 3] < 0 && vals[(dy + 1) * 3 + 1] < 0 && vals[(dy + 1) * 3 + 2] < 0) continue;
            MaskT m0 = NPL > 0 ? pl_e[yy * NPL] : (MaskT)0, m1 = NPL > 1 ? pl_e[yy * NPL + 1] : (MaskT)0, m2 = NPL > 1 ? pl_e[yy * NPL + 2] : (MaskT)0;
            bool touched = false;
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const int xx = x + dx, v = vals[(dy + 1) * 3 + dx + 1];
                if (xx < 0 || xx >= W || v < 0) continue;
                uint8_t* cell = map_e + yy * W + xx;
                if (*cell != v) {
                    change++;
                    touched = true;
                    *cell = (uint8_t)v;
                    const MaskT bit = (MaskT)1 << xx;
                    m0 = (v & 1) ? (m0 | bit) : (m0 & ~bit);
                    m1 = (v & 2) ? (m1 | bit) : (m1 & ~bit);
                    m2 = (v & 4) ? (m2 | bit) : (m2 & ~bit);
                }
            }
            if (touched && NPL > 0) { pl_e[yy * NPL] = m0; if (NPL > 1) { pl_e[yy * NPL + 1] = m1; pl_e[yy * NPL + 2] = m2; } }
        }
        if (REP != PCGRL_REP_TURTLE_CAST) {   // narrow cursor move (narrow_rep.py:104-113), after the write
            if (P.random_tile) {
                uint32_t* ring = B.rng_rep + (size_t)e * PCGRL_MT_N;
                int cur = B.rng_cur[2 * e];
                x = mt_randint(ring, cur, W);
                y = mt_randint(ring, cur, H);
                B.rng_cur[2 * e] = cur;
            } else {
                x += 1;
                if (x >= W) { x = 0; y += 1; if (y >= H) y = 0; }
            }
        }
        if (change > 0) {
            chg = true;
            changes += change;
            heat_increment(B, B.heat + ((size_t)e * H + y) * W + x);
        }
        reinterpret_cast<int2*>(B.counters)[e] = make_int2(iter, changes);
        reinterpret_cast<uchar2*>(B.pos)[e] = make_uchar2((unsigned char)x, (unsigned char)y);
        if (bad) atomicOr(B.status, PCGRL_STATUS_BAD_ACTION);
        if (!chg) {
            int32_t s[PCGRL_MAX_STATS], st[PCGRL_MAX_STATS];
            int32_t* inf = B.info + (size_t)e * 10;
            for (int k = 0; k < 8; k++) { s[k] = B.stats[(size_t)e * 8 + k]; st[k] = B.start_stats[(size_t)e * 8 + k]; inf[k] = s[k]; }
            const bool d = episode_over(P, s, st) || changes >= P.max_changes || iter >= P.max_iterations;
            B.reward[e] = 0.0;
            B.done[e] = d ? 1 : 0;
            episode_account(B, e, 0.0, d);
            if (P.prob == PCGRL_PROB_BINARY) inf[2] = s[1] - st[1];
            inf[8] = iter; inf[9] = changes;
            rst = d && P.auto_reset;
        }
    }
    const bool inl = rst && B.inline_reset;   // see k_update
    if (B.inline_reset) {
        const bool first = inl || sure_done;
        block_append2(chg && !first, e, WL_CHG, first, inl ? (e | WL_RESET_ONLY) : e, WL_RST, B, parity, s_cnt, s_base);
    } else {
        block_append2(chg, e, WL_CHG, rst, e, WL_RST, B, parity, s_cnt, s_base);
    }
}
