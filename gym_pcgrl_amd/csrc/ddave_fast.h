// Compact Dangerous Dave planner search: the same agents as dd_search (ddave_solver.h) for levels with at most DDF_MAXD
// diamonds, built like mdungeon_fast.h:
//   * what State.getKey (engine.py:283-294) distinguishes is ONE 64-bit key: which diamonds are left (a bit per diamond,
//     numbered row-major) | player cell << 48 | key still lying there << 56 (a live node always has health 1); the air
//     time and the jump counter travel in the node but are not part of the key, exactly as in the engine,
//   * the visited set is LDS open addressing on that key; a child that is dead (spike) or whose key is already visited
//     when it is made is queued as a flagged entry without a pool node -- it keeps its real priority, is popped, counted
//     and dropped like in the reference (a visited key stays visited, whatever the air time of the later copy),
//   * 16-byte pool nodes, fetch-ahead and the four-entry child cache, two-level heapq sifts (sokoban_fast.h), the four
//     children of a pop made by four lanes on the device.
#pragma once
#include "ddave_solver.h"
#include "mdungeon_fast.h"

#define DDF_MAXD 48
#define DDF_KEY_THERE (1ull << 56)

struct alignas(16) DdFastNode { uint64_t key; uint32_t hd; uint32_t aj; };   // hd = (h + DD_PRIO_BIAS) | depth << 16; aj = air | jumps << 2

struct DdFastLevel {
    uint64_t alive0;          // every diamond of the level
    uint8_t item[256];        // bordered cell -> diamond number, 255 = none
    int ndiamonds;
};
// Diamonds numbered in row-major order.  Returns their number (the compact search needs <= DDF_MAXD).
PCGRL_D int ddf_level(const DdLevel& L, DdFastLevel& F) {
    int n = 0;
    for (int p = 0; p < L.cells; p++) {
        F.item[p] = 255;
        if (!sok_bit(L.diamond0, p)) continue;
        if (n < DDF_MAXD) F.item[p] = (uint8_t)n;
        n++;
    }
    F.ndiamonds = n;
    F.alive0 = n >= 64 ? ~0ull : ((1ull << n) - 1);
    return n;
}
PCGRL_D int ddf_heuristic(const DdLevel& L, const DdFastLevel& F, int player, bool key_there, uint64_t alive) {
    return dd_heuristic(L, player, key_there, md_popcount(F.alive0 & ~alive));
}

// Child d (0..3 = stay, left, right, jump) of a live state.
struct DdChild { uint64_t key; int h; int aj; int drop; };
template <class TP>
PCGRL_D DdChild ddf_child(const DdLevel& L, const DdFastLevel& F, TP table, int table_mask, uint64_t key, int aj, bool ground, bool ceiling, int d) {
    const uint64_t alive = key & MDF_ALIVE_MASK;
    int np = (int)((key >> 48) & 0xFF), air = aj & 3, jumps = aj >> 2;
    bool key_there = (key & DDF_KEY_THERE) != 0, dead = false;
    if (d == 1) { if (!sok_bit(L.solid, np - 1)) np -= 1; }
    else if (d == 2) { if (!sok_bit(L.solid, np + 1)) np += 1; }
    else if (d == 3) { if (ground && !ceiling) { air = 3; jumps += 1; } }
    if (air > 1) {
        air -= 1;
        if (!sok_bit(L.solid, np - L.w)) np -= L.w; else air = 1;
    } else if (air == 1) {
        air = 0;
    } else {
        if (!sok_bit(L.solid, np + L.w)) np += L.w;
    }
    uint64_t al = alive;
    const int it = F.item[np];
    if (it != 255 && ((al >> it) & 1ull)) al &= ~(1ull << it);
    else if (sok_bit(L.spike, np)) dead = true;
    else if (key_there && np == L.keycell) key_there = false;
    DdChild c;
    c.key = al | ((uint64_t)np << 48) | (key_there ? DDF_KEY_THERE : 0ull);
    c.h = ddf_heuristic(L, F, np, key_there, al);
    c.aj = air | (jumps << 2);
    uint32_t cslot;
    c.drop = (dead || mdf_lookup(table, table_mask, c.key, cslot)) ? 1 : 0;
    return c;
}
struct DdKidsSerial {     // one lane makes the four children one after the other (host build, tests)
    template <class TP>
    PCGRL_D void operator()(const DdLevel& L, const DdFastLevel& F, TP table, int table_mask, uint64_t key, int aj, bool ground, bool ceiling,
                            DdChild* out) const {
        for (int d = 0; d < 4; d++) out[d] = ddf_child(L, F, table, table_mask, key, aj, ground, ceiling, d);
    }
};

// One search.  `table` (64-bit slots) must be all zeros; `cache` is room for four nodes.  ret_* describe the returned node.
template <class HP, class TP, class Hook, class Kids, class RSP = SokNoResume>
PCGRL_D bool dd_search_fast(const DdLevel& L, const DdFastLevel& F, DdFastNode* pool, HP heap, TP table, int table_mask, DdFastNode* cache,
                            const DdNode& root, int k, int power, uint64_t& ret_key, int& ret_h, int& ret_depth, int& ret_jumps, int& out_iters,
                            bool& out_exhausted, Hook hook, Kids kids, SokDuoBox* duo = nullptr, RSP rsp = RSP()) {
    constexpr bool RS = SokRs<RSP>::on;          // suspend / resume: sokoban_fast.h SokResume
    SokResume* const rst = sok_rs_state(rsp);
    const int rs_limit = sok_rs_limit(rsp);
    const bool resumed = RS && rst->iterations > 0;
    bool suspended = false;
    int npool = 0, head = 0, heapn = 0, iterations = 0, best_h = 0, best_depth = 0, best_aj = 0;
    bool have_best = false, aborted = false, win = false;
    uint64_t best_key = 0;
    DdFastNode n0;
    n0.key = F.alive0 | ((uint64_t)root.player << 48) | ((root.flags & DD_F_KEY_THERE) ? DDF_KEY_THERE : 0ull);
    n0.hd = (uint32_t)(root.h + DD_PRIO_BIAS); n0.aj = 0;
    DdFastNode ahead = n0;
    int ahead_idx = 0, cache_base = 0, cache_n = 0;
    if (!resumed) {
        pool[0] = n0;
        npool = 1;
        heap[0] = (k >= 0) ? ((uint32_t)(2 * root.h + DD_PRIO_BIAS) << 16) : 0u;
        heapn = 1;                                 // BFS: entries [head, heapn) of the same array are the queue
    } else {
        npool = rst->npool; head = rst->head; heapn = rst->heapn; iterations = rst->iterations;
        best_h = rst->best_h; best_depth = rst->best_depth; have_best = rst->have_best != 0; best_key = rst->best_key; best_aj = rst->best_aux;
        ahead_idx = -1;
    }
    ret_key = n0.key; ret_h = root.h; ret_depth = 0; ret_jumps = 0;
#if defined(__HIPCC__)
    if (duo && k >= 0) {
        // the search wavefront of a two-wavefront A* search (sokoban_fast.h, SokDuoBox; the same split as in mdungeon_fast.h):
        // the block's heap server owns the heap.  Same operations in the same order as the loop below.
        uint32_t cur_word = (uint32_t)(2 * root.h + DD_PRIO_BIAS) << 16;      // the root's word: pool index 0, not flagged
        int hn = 0;                                      // (RS) the heap's entries after the server's removal for the pending pop
        if (resumed) { cur_word = rst->cur_word; hn = heapn; duo->resume_n = heapn; duo->resume_aw = rst->aw; }
        duo->session = resumed ? 2 : 1;
        sok_duo_sync();                                  // (0)
        bool empty = false;
        int turn = 1;                                    // the pop the coming barrier (A) belongs to (SokDuoBox: its parity selects the set)
        for (;;) {
            if (SOK_UNI(cur_word == SOK_DUO_NONE)) { empty = true; break; }
            if (iterations >= power) break;
            if (RS && iterations >= rs_limit) { suspended = true; break; }
            iterations++;
            if ((iterations & SOK_POLL_MASK) == 0 && SOK_UNI(hook(iterations))) { aborted = true; break; }     // (the A* hooks poll at that rate)
            const uint32_t ent = cur_word & 0xFFFFu;
            int npush = 0;
            uint32_t cmin = SOK_DUO_NONE;              // the first child with the smallest priority (sok_duo_first_smallest)
            if (SOK_UNI(!(ent & MDF_FLAG))) {
                const int cur = ent & 0x7FFF;
                DdFastNode nd = ahead;
                if (SOK_UNI(cur != ahead_idx)) {
                    if (SOK_UNI((unsigned)(cur - cache_base) < (unsigned)cache_n)) nd = cache[cur - cache_base];
                    else nd = pool[cur];
                }
                const uint64_t key = nd.key;
                const int node_player = (int)((key >> 48) & 0xFF);
                const int node_h = (int)(nd.hd & 0xFFFFu) - DD_PRIO_BIAS, node_depth = (int)(nd.hd >> 16), node_aj = (int)nd.aj;
                if (SOK_UNI(!(key & DDF_KEY_THERE) && node_player == L.door)) {   // checkWin
                    win = true; ret_key = key; ret_h = node_h; ret_depth = node_depth; ret_jumps = node_aj >> 2; break;
                }
                uint32_t slot;
                if (!mdf_lookup_uni(table, table_mask, key, slot)) {
                    table[slot] = key;
                    cache_base = npool; cache_n = 0;
                    const bool better = !have_best || node_h < best_h || (node_h == best_h && node_depth < best_depth);
                    have_best = true; best_h = better ? node_h : best_h; best_depth = better ? node_depth : best_depth;
                    best_key = better ? key : best_key; best_aj = better ? node_aj : best_aj;
                    const bool ground = sok_bit(L.solid, node_player + L.w), ceiling = sok_bit(L.solid, node_player - L.w);
                    // stay, left, right, jump -- always four.  Lane d makes child d and files it itself (mdungeon_fast.h)
                    const DdChild mine = kids.mine(L, F, table, table_mask, key, node_aj, ground, ceiling);
                    const uint32_t keepm = (uint32_t)__builtin_amdgcn_ballot_w64(!mine.drop) & 15u;
                    const int rank = __builtin_popcount(keepm & ((1u << (kids.lane & 3)) - 1u));
                    uint32_t ent_c = MDF_FLAG;
                    if (!mine.drop) {
                        DdFastNode c;
                        c.key = mine.key; c.hd = (uint32_t)(mine.h + DD_PRIO_BIAS) | ((uint32_t)(node_depth + 1) << 16); c.aj = (uint32_t)mine.aj;
                        pool[npool + rank] = c;
                        cache[rank] = c;
                        ent_c = (uint32_t)(npool + rank);
                    }
                    const uint32_t word = ((uint32_t)(2 * mine.h + k * (node_depth + 1) + DD_PRIO_BIAS) << 16) | ent_c;
                    duo->push[turn & 1][kids.lane & 3] = word;
                    cache_n = __builtin_popcount(keepm); npool += cache_n; npush = 4;
                    cmin = sok_duo_first_smallest(true, word, kids.lane);
                }
            }
            duo->npush[turn & 1] = npush;
            if (RS) { hn += npush; hn -= hn > 0 ? 1 : 0; }
            sok_duo_sync();                              // (A) children one way, the top the repair left the other
            const uint32_t aw = SOK_SCALAR(duo->ahead_word[turn & 1]);
            turn++;
            uint32_t nxt = aw;                           // the next pop: that top, unless a child is strictly smaller (then the first smallest)
if (cmin != SOK_DUO_NONE && (nxt == SOK_DUO_NONE || sok_lt(cmin, nxt))) nxt = cmin;
            cur_word = nxt;
            ahead_idx = -1;
            if (SOK_UNI(aw != SOK_DUO_NONE && nxt == aw && !(aw & MDF_FLAG))) { ahead_idx = (int)(aw & 0x7FFFu); ahead = pool[ahead_idx]; }
        }
        duo->npush[turn & 1] = -1;                       // the server leaves the search
        sok_duo_sync();                                  // (A)
        if (RS) {
            rst->suspended = suspended ? 1 : 0;
            rst->iterations = iterations; rst->npool = npool; rst->head = 0; rst->heapn = hn;
            rst->cur_word = cur_word; rst->aw = SOK_SCALAR(duo->ahead_word[turn & 1]);
            rst->best_h = best_h; rst->best_depth = best_depth; rst->have_best = have_best ? 1 : 0; rst->best_key = best_key; rst->best_aux = best_aj;
        }
        if (!win && have_best) { ret_key = best_key; ret_h = best_h; ret_depth = best_depth; ret_jumps = best_aj >> 2; }
        out_iters = iterations;
        out_exhausted = !win && !aborted && empty;
        return win;
    }
#endif
    while (iterations < power && (k >= 0 ? heapn > 0 : head < heapn)) {
        if (RS && iterations >= rs_limit) { suspended = true; break; }
        iterations++;
        if (hook(iterations)) { aborted = true; break; }
        uint32_t ent;
        DdFastNode nd = ahead;
        if (k >= 0) {
            ent = heap[0];
            const uint32_t last = heap[--heapn];
            const int cur = (int)(ent & 0x7FFFu);
            if (!(ent & MDF_FLAG) && cur != ahead_idx) {
                if ((unsigned)(cur - cache_base) < (unsigned)cache_n) nd = cache[cur - cache_base];
                else nd = pool[cur];
            }
            if (heapn > 0) { heap[0] = last; sokf_siftup_root(heap, heapn); }
            ahead_idx = -1;
            if (heapn > 0) {
                const uint32_t top = heap[0];
                if (!(top & MDF_FLAG)) { ahead_idx = (int)(top & 0x7FFFu); ahead = pool[ahead_idx]; }
            }
        } else {
            ent = heap[head++];
            const int cur = (int)(ent & 0x7FFFu);
            if (!(ent & MDF_FLAG) && cur != ahead_idx) nd = pool[cur];
            ahead_idx = -1;
            if (head < heapn) {
                const uint32_t nxt = heap[head];
                if (!(nxt & MDF_FLAG)) { ahead_idx = (int)(nxt & 0x7FFFu); ahead = pool[ahead_idx]; }
            }
        }
        if (ent & MDF_FLAG) continue;                    // dead, or a key that was visited before it was queued
        const uint64_t key = nd.key;
        const int node_player = (int)((key >> 48) & 0xFF);
        const int node_h = (int)(nd.hd & 0xFFFFu) - DD_PRIO_BIAS, node_depth = (int)(nd.hd >> 16), node_aj = (int)nd.aj;
        if (!(key & DDF_KEY_THERE) && node_player == L.door) {   // checkWin
            win = true; ret_key = key; ret_h = node_h; ret_depth = node_depth; ret_jumps = node_aj >> 2; break;
        }
        uint32_t slot;
        if (mdf_lookup(table, table_mask, key, slot)) continue;
        table[slot] = key;
        cache_base = npool; cache_n = 0;
        if (!have_best || node_h < best_h || (node_h == best_h && node_depth < best_depth)) {
            have_best = true; best_h = node_h; best_depth = node_depth; best_key = key; best_aj = node_aj;
        }
        const bool ground = sok_bit(L.solid, node_player + L.w), ceiling = sok_bit(L.solid, node_player - L.w);
        DdChild kid[4];                         // stay, left, right, jump -- always four
        kids(L, F, table, table_mask, key, node_aj, ground, ceiling, kid);
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int d = 0; d < 4; d++) {
            uint32_t ent_c = MDF_FLAG;
            if (!kid[d].drop) {
                DdFastNode c;
                c.key = kid[d].key; c.hd = (uint32_t)(kid[d].h + DD_PRIO_BIAS) | ((uint32_t)(node_depth + 1) << 16); c.aj = (uint32_t)kid[d].aj;
                pool[npool] = c;
                if (k >= 0) cache[cache_n++] = c;
                ent_c = (uint32_t)npool;
                npool++;
            }
            if (k >= 0) {
                heap[heapn] = ((uint32_t)(2 * kid[d].h + k * (node_depth + 1) + DD_PRIO_BIAS) << 16) | ent_c;
                heapn++;
                sokf_siftdown(heap, heapn - 1);
            } else {
                heap[heapn++] = ent_c;
            }
        }
    }
    if (RS) {
        rst->suspended = suspended ? 1 : 0;
        rst->iterations = iterations; rst->npool = npool; rst->head = head; rst->heapn = heapn;
        rst->best_h = best_h; rst->best_depth = best_depth; rst->have_best = have_best ? 1 : 0; rst->best_key = best_key; rst->best_aux = best_aj;
    }
    if (!win && have_best) { ret_key = best_key; ret_h = best_h; ret_depth = best_depth; ret_jumps = best_aj >> 2; }
    out_iters = iterations;
    out_exhausted = !win && !aborted && !suspended && !(k >= 0 ? heapn > 0 : head < heapn);
    return win;
}

// dist-win, sol-length, num-jumps, col-diamonds from what a search returned
PCGRL_D void ddf_result(const DdFastLevel& F, uint64_t key, int h, int depth, int jumps, bool win, int* out4) {
    out4[0] = win ? 0 : h;
    out4[1] = win ? depth : 0;
    out4[2] = jumps;
    out4[3] = md_popcount(F.alive0 & ~(key & MDF_ALIVE_MASK));
}
