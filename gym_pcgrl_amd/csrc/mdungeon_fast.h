// Compact MiniDungeons planner search: the same agents as md_search (mdungeon_solver.h) for levels with at most
// MDF_MAXI things on the floor -- every level random play produces; the rest take md_search.
//
// md_search moves 40-byte nodes through global memory and has to look at the pool to recognise a state it has
// seen.  Three out of four pops of this engine are states that are dropped at once (a move into a wall gives the
// parent again, a step back gives the grandparent ...), so here
//   * a state is ONE 64-bit key: which things are left (a bit per thing, numbered row-major) | player cell << 48 |
//     health << 56 -- State.getKey (engine.py:257-269) says exactly that much; treasures, potions and kills follow
//     from the missing bits, the heuristic from the key,
//   * table  visited set in LDS, open addressing on the exact key: a hit needs no look at the pool,
//   * a child that has lost, or whose key is already in the visited set when it is made, is pushed as a *flagged*
//     queue entry without a pool node: the reference pops it, counts the iteration and drops it (a lost node
//     before the visited test, engine.py:67-68 / 116-117; a visited key stays visited), and so does this search --
//     with no memory access beyond the queue.  Its priority is still the real one: the order of the other pops
//     depends on every entry of the heap,
//   * pool   16-byte nodes in global memory for the rest; the next one to be popped is fetched ahead or found in a
//     four-entry LDS cache of the children just made,
//   * heap   CPython heapq on packed (priority << 16 | flag << 15 | node) words in LDS with the two-level sift
//     loops of sokoban_fast.h; BFS keeps its FIFO of (flag << 15 | node) words in the same array.
// Order of exploration, visited-on-pop, iteration counting and best-node rules are those of md_search.
#pragma once
#include "mdungeon_solver.h"
#include "sokoban_fast.h"

#define MDF_MAXI 48
#define MDF_FLAG 0x8000u
#define MDF_ALIVE_MASK ((1ull << MDF_MAXI) - 1)

struct alignas(16) MdFastNode { uint64_t key; uint32_t hd; uint32_t pad; };   // hd = (h + MD_PRIO_BIAS) | depth << 16

struct MdFastLevel {
    uint64_t potion_m, treasure_m, ogre_m;     // over thing numbers (goblins: the rest)
    uint64_t alive0;                           // every thing of the level
    uint8_t item[256];                         // bordered cell -> thing number, 255 = nothing
    int nitems;
};

// Things numbered in row-major order.  Returns their number (the fast search needs <= MDF_MAXI).
PCGRL_D int mdf_level(const MdLevel& L, const MdNode& root, MdFastLevel& F) {
    F.potion_m = 0; F.treasure_m = 0; F.ogre_m = 0;
    int n = 0;
    for (int p = 0; p < L.cells; p++) {
        F.item[p] = 255;
        if (!sok_bit(root.alive, p)) continue;
        if (n < MDF_MAXI) {
            F.item[p] = (uint8_t)n;
            if (sok_bit(L.potion, p)) F.potion_m |= 1ull << n;
            else if (sok_bit(L.treasure, p)) F.treasure_m |= 1ull << n;
            else if (sok_bit(L.ogre, p)) F.ogre_m |= 1ull << n;
        }
        n++;
    }
    F.nitems = n;
    F.alive0 = n >= 64 ? ~0ull : ((1ull << n) - 1);
    return n;
}
PCGRL_D int mdf_heuristic(const MdLevel& L, const MdFastLevel& F, int player, int health, uint64_t alive) {
    return abs((int)L.cx[player] - (int)L.cx[L.door]) + abs((int)L.cy[player] - (int)L.cy[L.door]) + 4 * (5 - health) -
           4 * md_popcount(F.treasure_m & ~alive);
}
template <class TP>
PCGRL_D bool mdf_lookup(TP table, int table_mask, uint64_t key, uint32_t& slot) {
    const uint64_t hs = key * 0x9E3779B97F4A7C15ull;
    slot = (uint32_t)(hs >> 40) & (uint32_t)table_mask;
    for (;;) {
        const uint64_t v = table[slot];
        if (v == 0) return false;
        if (v == key) return true;
        slot = (slot + 1) & (uint32_t)table_mask;
    }
}

// (the same probe where every lane looks up the same key: scalar branches, see SOK_UNI in sokoban_fast.h)
template <class TP>
PCGRL_D bool mdf_lookup_uni(TP table, int table_mask, uint64_t key, uint32_t& slot) {
    const uint64_t hs = key * 0x9E3779B97F4A7C15ull;
    slot = (uint32_t)(hs >> 40) & (uint32_t)table_mask;
    for (;;) {
        const uint64_t v = table[slot];
        if (SOK_UNI(v == 0)) return false;
        if (SOK_UNI(v == key)) return true;
        slot = (slot + 1) & (uint32_t)table_mask;
    }
}

// Child d (0..3 = L, R, U, D) of the state (alive, player, health) with key `key`: its key, heuristic, and whether it
// is dropped when popped (lost, equal to the parent, or already visited).
struct MdChild { uint64_t key; int h; int drop; };
template <class TP>
PCGRL_D MdChild mdf_child(const MdLevel& L, const MdFastLevel& F, TP table, int table_mask, uint64_t key, uint64_t alive, int player,
                          int health, int d) {
    int np = player + L.dirs[d];
    uint64_t al = alive;
    if (sok_bit(L.solid, np)) np = player;                        // checkMovableLocation fails: nothing happens
    else {
        const int it = F.item[np];
        if (it != 255 && ((al >> it) & 1ull)) {
            const uint64_t b = 1ull << it;
            al &= ~b;
            if (F.potion_m & b) { health += 2; if (health > 5) health = 5; }
            else if (!(F.treasure_m & b)) { health -= (F.ogre_m & b) ? 2 : 1; if (health < 0) health = 0; }
        }
    }
    MdChild c;
    c.key = al | ((uint64_t)np << 48) | ((uint64_t)health << 56);
    c.h = mdf_heuristic(L, F, np, health, al);
    uint32_t cslot;
    c.drop = (health == 0 || c.key == key || mdf_lookup(table, table_mask, c.key, cslot)) ? 1 : 0;
    return c;
}
struct MdKidsSerial {     // one lane makes the four children one after the other (host build, tests)
    template <class TP>
    PCGRL_D void operator()(const MdLevel& L, const MdFastLevel& F, TP table, int table_mask, uint64_t key, uint64_t alive, int player,
                            int health, MdChild* out) const {
        for (int d = 0; d < 4; d++) out[d] = mdf_child(L, F, table, table_mask, key, alive, player, health, d);
    }
};

#if defined(PCGRL_SMB_PROF) && defined(__HIP_DEVICE_COMPILE__)
extern __device__ unsigned long long* g_tl_buf;      // worklist.h (developer builds: tools/md_prof.py)
#define MDP_DECL unsigned long long mdp_t = clock64(), mdp_a[5] = {0, 0, 0, 0, 0}
#define MDP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = clock64(); mdp_a[i] += n_ - mdp_t; mdp_t = n_; } while (0)
#define MDP_FLUSH(it) do { if (g_tl_buf && k >= 0) { for (int i_ = 0; i_ < 5; i_++) atomicAdd(&g_tl_buf[16 + i_], mdp_a[i_]); atomicAdd(&g_tl_buf[21], (unsigned long long)(it)); atomicAdd(&g_tl_buf[22], 1ull); } } while (0)
#else
#define MDP_DECL do {} while (0)
#define MDP(i) do {} while (0)
#define MDP_FLUSH(it) do {} while (0)
#endif
// One search.  `table` (64-bit slots) must be all zeros; `cache` is room for four nodes (LDS on the device); `kids` makes
// the four children of a pop (serially, or one per lane on the device: the search then runs on four lanes in lockstep,
// uniform except for that step).
// ret_key / ret_h / ret_depth describe the returned node (winner, or best node).
template <class HP, class TP, class Hook, class Kids, class RSP = SokNoResume>
PCGRL_D bool md_search_fast(const MdLevel& L, const MdFastLevel& F, MdFastNode* pool, HP heap, TP table, int table_mask, MdFastNode* cache,
                            const MdNode& root, int k, int power, uint64_t& ret_key, int& ret_h, int& ret_depth, int& out_iters,
                            bool& out_exhausted, Hook hook, Kids kids, SokDuoBox* duo = nullptr, RSP rsp = RSP()) {
    constexpr bool RS = SokRs<RSP>::on;          // suspend / resume: sokoban_fast.h SokResume
    SokResume* const rst = sok_rs_state(rsp);
    const int rs_limit = sok_rs_limit(rsp);
    const bool resumed = RS && rst->iterations > 0;
    bool suspended = false;
    int npool = 0, head = 0, heapn = 0, iterations = 0, best_h = 0, best_depth = 0;
    bool have_best = false, aborted = false, win = false;
    uint64_t best_key = 0;
    MdFastNode n0;
    n0.key = F.alive0 | ((uint64_t)root.player << 48) | ((uint64_t)root.health << 56);
    n0.hd = (uint32_t)(root.h + MD_PRIO_BIAS); n0.pad = 0;
    MdFastNode ahead = n0;
    int ahead_idx = 0, cache_base = 0, cache_n = 0;   // cache[j] = pool[cache_base + j], j < cache_n
    if (!resumed) {
        pool[0] = n0;
        npool = 1;
        heap[0] = (k >= 0) ? ((uint32_t)(2 * root.h + MD_PRIO_BIAS) << 16) : 0u;
        heapn = 1;                                 // BFS: entries [head, heapn) of the same array are the queue
    } else {
        npool = rst->npool; head = rst->head; heapn = rst->heapn; iterations = rst->iterations;
        best_h = rst->best_h; best_depth = rst->best_depth; have_best = rst->have_best != 0; best_key = rst->best_key;
        ahead_idx = -1;
    }
    ret_key = n0.key; ret_h = root.h; ret_depth = 0;
#if defined(__HIPCC__)
    if (duo && k >= 0) {
        // the search wavefront of a two-wavefront A* search (sokoban_fast.h, SokDuoBox): the heap belongs to the block's heap
        // server, which appends the children of a pop, publishes the new top and removes / repairs for it while this
        // wavefront expands it.  Same operations in the same order as the loop below.
        uint32_t cur_word = (uint32_t)(2 * root.h + MD_PRIO_BIAS) << 16;      // the root's word: pool index 0, not flagged
        int hn = 0;                                      // (RS) the heap's entries after the server's removal for the pending pop
        if (resumed) { cur_word = rst->cur_word; hn = heapn; duo->resume_n = heapn; duo->resume_aw = rst->aw; }
        duo->session = resumed ? 2 : 1;
        sok_duo_sync();                                  // (0)
        SKD_DECL;
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
            if (SOK_UNI(!(ent & MDF_FLAG))) {           // (a flagged entry -- lost, or visited before it was queued -- is only counted)
                const int cur = ent & 0x7FFF;
                MdFastNode nd = ahead;
                if (SOK_UNI(cur != ahead_idx)) {
                    if (SOK_UNI((unsigned)(cur - cache_base) < (unsigned)cache_n)) nd = cache[cur - cache_base];
                    else nd = pool[cur];
                }
                const uint64_t key = nd.key;
                const uint64_t alive = key & MDF_ALIVE_MASK;
                const int node_player = (int)((key >> 48) & 0xFF), node_health = (int)(key >> 56);
                const int node_h = (int)(nd.hd & 0xFFFFu) - MD_PRIO_BIAS, node_depth = (int)(nd.hd >> 16);
                if (SOK_UNI(node_player == L.door)) { win = true; ret_key = key; ret_h = node_h; ret_depth = node_depth; break; }   // checkWin
                uint32_t slot;
                if (!mdf_lookup_uni(table, table_mask, key, slot)) {
                    table[slot] = key;
                    cache_base = npool; cache_n = 0;
                    const bool better = !have_best || node_h < best_h || (node_h == best_h && node_depth < best_depth);
                    have_best = true; best_h = better ? node_h : best_h; best_depth = better ? node_depth : best_depth; best_key = better ? key : best_key;
                    // Node.getChildren: L, R, U, D -- always four.  Lane d makes child d and files it itself: its word in the
                    // server's box and, unless it is dropped, its node in the pool and the cache at its rank among the kept ones
                    const MdChild mine = kids.mine(L, F, table, table_mask, key, alive, node_player, node_health);
                    const uint32_t keepm = (uint32_t)__builtin_amdgcn_ballot_w64(!mine.drop) & 15u;
                    const int rank = __builtin_popcount(keepm & ((1u << (kids.lane & 3)) - 1u));
                    uint32_t ent_c = MDF_FLAG;
                    if (!mine.drop) {
                        MdFastNode c;
                        c.key = mine.key; c.hd = (uint32_t)(mine.h + MD_PRIO_BIAS) | ((uint32_t)(node_depth + 1) << 16); c.pad = 0;
                        pool[npool + rank] = c;
                        cache[rank] = c;
                        ent_c = (uint32_t)(npool + rank);
                    }
                    const uint32_t word = ((uint32_t)(2 * mine.h + k * (node_depth + 1) + MD_PRIO_BIAS) << 16) | ent_c;
                    duo->push[turn & 1][kids.lane & 3] = word;
                    cache_n = __builtin_popcount(keepm); npool += cache_n; npush = 4;
                    cmin = sok_duo_first_smallest(true, word, kids.lane);
                }
            }
            duo->npush[turn & 1] = npush;
            if (RS) { hn += npush; hn -= hn > 0 ? 1 : 0; }
            SKD_MARK(0);
            sok_duo_sync();                              // (A) children one way, the top the repair left the other
            SKD_MARK(1);
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
        SKD_FLUSH(32, iterations);
        if (RS) {
            rst->suspended = suspended ? 1 : 0;
            rst->iterations = iterations; rst->npool = npool; rst->head = 0; rst->heapn = hn;
            rst->cur_word = cur_word; rst->aw = SOK_SCALAR(duo->ahead_word[turn & 1]);
            rst->best_h = best_h; rst->best_depth = best_depth; rst->have_best = have_best ? 1 : 0; rst->best_key = best_key;
        }
        if (!win && have_best) { ret_key = best_key; ret_h = best_h; ret_depth = best_depth; }
        out_iters = iterations;
        out_exhausted = !win && !aborted && empty;
        return win;
    }
#endif
    MDP_DECL;
    while (iterations < power && (k >= 0 ? heapn > 0 : head < heapn)) {
        if (RS && iterations >= rs_limit) { suspended = true; break; }
        iterations++;
        MDP(4);
        if (hook(iterations)) { aborted = true; break; }
        uint32_t ent;
        MdFastNode nd = ahead;
        if (k >= 0) {
            ent = heap[0];
            const uint32_t last = heap[--heapn];
            const int cur = (int)(ent & 0x7FFFu);
            const bool live = !(ent & MDF_FLAG);
            if (live && cur != ahead_idx) {
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
        MDP(0);
        if (ent & MDF_FLAG) continue;                    // lost, or a key that was visited before it was queued
        const uint64_t key = nd.key;
        const uint64_t alive = key & MDF_ALIVE_MASK;
        const int node_player = (int)((key >> 48) & 0xFF), node_health = (int)(key >> 56);
        const int node_h = (int)(nd.hd & 0xFFFFu) - MD_PRIO_BIAS, node_depth = (int)(nd.hd >> 16);
        if (node_player == L.door) { win = true; ret_key = key; ret_h = node_h; ret_depth = node_depth; break; }   // checkWin
        uint32_t slot;
        if (mdf_lookup(table, table_mask, key, slot)) continue;
        table[slot] = key;
        cache_base = npool; cache_n = 0;
        if (!have_best || node_h < best_h || (node_h == best_h && node_depth < best_depth)) {
            have_best = true; best_h = node_h; best_depth = node_depth; best_key = key;
        }
        MDP(1);
        MdChild kid[4];                         // Node.getChildren: L, R, U, D -- always four
        kids(L, F, table, table_mask, key, alive, node_player, node_health, kid);
        MDP(2);
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int d = 0; d < 4; d++) {
            uint32_t ent_c = MDF_FLAG;
            if (!kid[d].drop) {
                MdFastNode c;
                c.key = kid[d].key; c.hd = (uint32_t)(kid[d].h + MD_PRIO_BIAS) | ((uint32_t)(node_depth + 1) << 16); c.pad = 0;
                pool[npool] = c;
                if (k >= 0) cache[cache_n++] = c;
                ent_c = (uint32_t)npool;
                npool++;
            }
            if (k >= 0) {
                heap[heapn] = ((uint32_t)(2 * kid[d].h + k * (node_depth + 1) + MD_PRIO_BIAS) << 16) | ent_c;
                heapn++;
                sokf_siftdown(heap, heapn - 1);
            } else {
                heap[heapn++] = ent_c;
            }
        }
        MDP(3);
    }
    MDP_FLUSH(iterations);
    if (RS) {
        rst->suspended = suspended ? 1 : 0;
        rst->iterations = iterations; rst->npool = npool; rst->head = head; rst->heapn = heapn;
        rst->best_h = best_h; rst->best_depth = best_depth; rst->have_best = have_best ? 1 : 0; rst->best_key = best_key;
    }
    if (!win && have_best) { ret_key = best_key; ret_h = best_h; ret_depth = best_depth; }
    out_iters = iterations;
    out_exhausted = !win && !aborted && !suspended && !(k >= 0 ? heapn > 0 : head < heapn);
    return win;
}

// The five values _run_game hands to get_stats, from the key a search returned.
PCGRL_D void mdf_result(const MdFastLevel& F, uint64_t key, int h, int depth, bool win, int* out5) {
    const uint64_t gone = F.alive0 & ~(key & MDF_ALIVE_MASK);
    out5[0] = win ? 0 : h;
    out5[1] = win ? depth : 0;
    out5[2] = md_popcount(gone & F.potion_m);
    out5[3] = md_popcount(gone & F.treasure_m);
    out5[4] = md_popcount(gone & ~(F.potion_m | F.treasure_m));
}
