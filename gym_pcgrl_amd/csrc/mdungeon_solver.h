// MiniDungeons planner for the reward path (device side).
//
// Restates probs/mdungeon/engine.py as used by MDungeonProblem._run_game (mdungeon_prob.py:91-126):
// AStarAgent with balance 1, 0.5, 0, then BFSAgent (solver_power pops each); the first agent whose returned
// state stands on the exit gives (dist-win 0, sol-length = depth, game status of that state); otherwise the
// BFS agent's best node gives (heuristic, 0, its game status).  Exactness needs the engine's order of
// exploration:
//   * level = map with a solid border (mdungeon_prob.py:92-108); tiles " #@H*$go"
//   * a popped node that has lost (health <= 0) is dropped before anything else, a node on the exit ends the
//     search, a node whose key was seen is dropped -- every pop counts as an iteration (engine.py:62-82, 107-133)
//   * an expanded node always gets FOUR children, L,R,U,D, also when the player could not move (engine.py:14-20)
//   * State.update / updatePlayer (engine.py:215-255): walking onto a potion +2 health (max 5), a treasure +1
//     treasure, a goblin / ogre -1 / -2 health (min 0) and +1 kill; the thing is removed
//   * key (engine.py:257-269) = player cell, health and what is still lying around; here the latter is a bit per
//     cell of the bordered grid (every thing keeps its cell, so the set decides the engine's ordered lists)
//   * heuristic (engine.py:271-275) = manhattan distance to the exit + 4*(5 - health) - 4*treasures: it can be
//     negative, so the packed heap priority 2h + {2,1,0}*depth carries a bias
//   * A* uses queue.PriorityQueue == CPython heapq (sokoban_solver.h has the sift routines, shared)
//   * bestNode = min h, then min depth, first seen.
//
// One wavefront per job, the search driven by lane 0 (a chain of data-dependent pops); heap and visited table
// in LDS (or the global arena for a large solver_power), node pool in the global arena -- the same arena
// layout as the Sokoban solver, a node is 40 bytes here too.
//
// Limits (checked by the host): (W+2)*(H+2) <= 256, solver_power <= 16383.
#pragma once
#include "sokoban_solver.h"

#define MD_PRIO_BIAS 2048      /* 2h >= -8*254 = -2032 */

struct alignas(8) MdNode {     // 40 bytes, moved around as five 64-bit words
    uint64_t alive[4];         // things still on the floor, one bit per bordered cell
    uint8_t player, health;
    int16_t h;
    uint16_t depth;
    uint8_t treasures, pad;
};
struct MdLevel {
    uint64_t solid[4], potion[4], treasure[4], goblin[4], ogre[4];
    uint8_t cx[256], cy[256];
    int w, h, cells, door;
    int dirs[4];
};

PCGRL_D void md_copy(MdNode* dst, const MdNode* src) {
    const uint64_t* s = reinterpret_cast<const uint64_t*>(src);
    uint64_t* d = reinterpret_cast<uint64_t*>(dst);
    const uint64_t a = s[0], b = s[1], c = s[2], e = s[3], f = s[4];
    d[0] = a; d[1] = b; d[2] = c; d[3] = e; d[4] = f;
}
PCGRL_D int md_heuristic(const MdLevel& L, int player, int health, int treasures) {   // engine.py:271-275
    return abs((int)L.cx[player] - (int)L.cx[L.door]) + abs((int)L.cy[player] - (int)L.cy[L.door]) + 4 * (5 - health) - 4 * treasures;
}
PCGRL_D int md_popcount(uint64_t v) {
#if defined(__HIPCC__)
    return __popcll(v);
#else
    return __builtin_popcountll(v);
#endif
}

// mdungeon_prob.py:92-108 + engine.py:143-181.  tiles: 0 empty 1 solid 2 player 3 exit 4 potion 5 treasure 6 goblin 7 ogre
PCGRL_D void md_build_level(const uint8_t* m, int W, int H, MdLevel& L, MdNode& root) {
    L.w = W + 2; L.h = H + 2; L.cells = L.w * L.h; L.door = 0;
    L.dirs[0] = -1; L.dirs[1] = 1; L.dirs[2] = -L.w; L.dirs[3] = L.w;
    for (int k = 0; k < 4; k++) { L.solid[k] = 0; L.potion[k] = 0; L.treasure[k] = 0; L.goblin[k] = 0; L.ogre[k] = 0; root.alive[k] = 0; }
    root.player = 0; root.health = 5; root.h = 0; root.depth = 0; root.treasures = 0; root.pad = 0;
    for (int y = 0; y < L.h; y++)
        for (int x = 0; x < L.w; x++) {
            const int p = y * L.w + x;
            L.cx[p] = (uint8_t)x; L.cy[p] = (uint8_t)y;
            const bool border = x == 0 || y == 0 || x == L.w - 1 || y == L.h - 1;
            const int t = border ? 1 : m[(y - 1) * W + (x - 1)];
            if (t == 1) sok_set(L.solid, p);
            if (t == 2) root.player = (uint8_t)p;
            if (t == 3) L.door = p;
            if (t == 4) sok_set(L.potion, p);
            if (t == 5) sok_set(L.treasure, p);
            if (t == 6) sok_set(L.goblin, p);
            if (t == 7) sok_set(L.ogre, p);
            if (t >= 4) sok_set(root.alive, p);
        }
    root.h = (int16_t)md_heuristic(L, root.player, root.health, 0);
}

PCGRL_D uint32_t md_hash(const MdNode& n) {
    uint64_t x = ((uint64_t)n.player << 8) | n.health;
    for (int i = 0; i < 4; i++) { x = (x ^ n.alive[i]) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; }
    return (uint32_t)(x ^ (x >> 32));
}
PCGRL_D bool md_same(const MdNode& a, const MdNode& b) {
    return a.player == b.player && a.health == b.health && a.alive[0] == b.alive[0] && a.alive[1] == b.alive[1] &&
           a.alive[2] == b.alive[2] && a.alive[3] == b.alive[3];
}

struct MdRaw { uint64_t q[5]; };
PCGRL_D MdRaw md_load(const MdNode* p) {
    const uint64_t* s = reinterpret_cast<const uint64_t*>(p);
    MdRaw r;
    r.q[0] = s[0]; r.q[1] = s[1]; r.q[2] = s[2]; r.q[3] = s[3]; r.q[4] = s[4];
    return r;
}
PCGRL_D void md_store(MdNode* p, const MdRaw& r) {
    uint64_t* d = reinterpret_cast<uint64_t*>(p);
    d[0] = r.q[0]; d[1] = r.q[1]; d[2] = r.q[2]; d[3] = r.q[3]; d[4] = r.q[4];
}

// One search (one lane).  k < 0: BFSAgent, else AStarAgent with integer weight k in {2,1,0}.  `w` is the node
// workspace (LDS on the device).  On return `w` holds the returned node (winner, or best node) and the function
// value is the win flag; out_exhausted: the queue ran empty without a win and without reaching the cap.
// `hook(iterations)` is called at the top of every iteration; returning true abandons the search (k_mdungeon runs
// the agents of one level concurrently and knows when a result cannot be selected any more).
// The node that will be popped next is fetched from the pool one iteration ahead whenever it is already known (BFS:
// the next queue entry; A*: the heap top after the repair), which takes the global-memory latency of the pop off the
// serial chain.
template <class HP, class TP, class Hook>
PCGRL_D bool md_search(const MdLevel& L, MdNode* pool, HP heap, TP table, int table_mask, MdNode& w, const MdNode& root, int k,
                       int power, int& out_iters, bool& out_exhausted, Hook hook) {
    int npool = 0, head = 0, heapn = 0, iterations = 0, best = -1, best_h = 0, best_depth = 0;
    md_copy(pool, &root);
    npool = 1;
    if (k >= 0) { heap[0] = ((uint32_t)(2 * root.h + k * root.depth + MD_PRIO_BIAS) << 16) | 0u; heapn = 1; }
    bool win = false, aborted = false;
    int result = 0;
    MdRaw ahead = md_load(&root);
    int ahead_idx = 0;
    while (iterations < power && (k >= 0 ? heapn > 0 : head < npool)) {
        iterations++;
        if (hook(iterations)) { aborted = true; break; }
        int cur;
        if (k >= 0) {
            const uint32_t last = heap[--heapn];
            cur = (int)((heapn > 0 ? heap[0] : last) & 0xFFFFu);
            MdRaw fetched = ahead;
            if (cur != ahead_idx) fetched = md_load(pool + cur);       // global load in flight while the heap is repaired
            if (heapn > 0) { heap[0] = last; sok_siftup(heap, 0, heapn); }
            md_store(&w, fetched);
            ahead_idx = -1;
            if (heapn > 0) { ahead_idx = (int)(heap[0] & 0xFFFFu); ahead = md_load(pool + ahead_idx); }
        } else {
            cur = head++;
            MdRaw fetched = ahead;
            if (cur != ahead_idx) fetched = md_load(pool + cur);
            md_store(&w, fetched);
            ahead_idx = -1;
            if (head < npool) { ahead_idx = head; ahead = md_load(pool + head); }
        }
        if (w.health == 0) continue;                                   // checkLose
        if (w.player == L.door) { win = true; result = cur; break; }    // checkWin
        const uint32_t hs = md_hash(w);
        uint32_t slot = hs & (uint32_t)table_mask;
        const uint32_t tag = (hs >> 16) << 16;
        bool seen = false;
        for (;;) {
            const uint32_t v = table[slot];
            if (v == 0) break;
            if ((v & 0xFFFF0000u) == tag && md_same(pool[(v & 0xFFFFu) - 1], w)) { seen = true; break; }
            slot = (slot + 1) & (uint32_t)table_mask;
        }
        if (seen) continue;
        table[slot] = tag | (uint32_t)(cur + 1);
        const int node_h = w.h, node_depth = w.depth, node_player = w.player, node_health = w.health, node_tr = w.treasures;
        if (best < 0 || node_h < best_h || (node_h == best_h && node_depth < best_depth)) { best = cur; best_h = node_h; best_depth = node_depth; }
        w.depth = (uint16_t)(node_depth + 1);
        for (int d = 0; d < 4; d++) {          // Node.getChildren: L, R, U, D -- always four
            int np = node_player + L.dirs[d];
            int health = node_health, tr = node_tr;
            int taken = -1;
            if (sok_bit(L.solid, np)) np = node_player;               // checkMovableLocation fails: nothing happens
            else if (sok_bit(w.alive, np)) {
                taken = np;
                if (sok_bit(L.potion, np)) { health += 2; if (health > 5) health = 5; }
                else if (sok_bit(L.treasure, np)) tr += 1;
                else { health -= sok_bit(L.ogre, np) ? 2 : 1; if (health < 0) health = 0; }
            }
            if (taken >= 0) w.alive[taken >> 6] &= ~(1ull << (taken & 63));
            w.player = (uint8_t)np; w.health = (uint8_t)health; w.treasures = (uint8_t)tr;
            w.h = (int16_t)md_heuristic(L, np, health, tr);
            md_copy(pool + npool, &w);
            if (k >= 0) {
                heap[heapn] = ((uint32_t)(2 * w.h + k * w.depth + MD_PRIO_BIAS) << 16) | (uint32_t)npool;
                heapn++;
                sok_siftdown(heap, 0, heapn - 1);
            }
            npool++;
            if (taken >= 0) w.alive[taken >> 6] |= 1ull << (taken & 63);   // undo
        }
    }
    if (!win) result = best < 0 ? 0 : best;
    md_copy(&w, pool + result);
    out_iters = iterations;
    out_exhausted = !win && !aborted && !(k >= 0 ? heapn > 0 : head < npool);
    return win;
}
template <class HP, class TP>
PCGRL_D bool md_search(const MdLevel& L, MdNode* pool, HP heap, TP table, int table_mask, MdNode& w, const MdNode& root, int k,
                       int power, int& out_iters, bool& out_exhausted) {
    return md_search(L, pool, heap, table, table_mask, w, root, k, power, out_iters, out_exhausted, SokNoHook());
}

// The five values _run_game hands to get_stats, from the node a search returned.
PCGRL_D void md_result(const MdLevel& L, const MdNode& root, const MdNode& n, bool win, int* out5) {
    int pot = 0, ene = 0;
    for (int i = 0; i < 4; i++) {
        const uint64_t gone = root.alive[i] & ~n.alive[i];
        pot += md_popcount(gone & L.potion[i]);
        ene += md_popcount(gone & (L.goblin[i] | L.ogre[i]));
    }
    out5[0] = win ? 0 : (int)n.h;
    out5[1] = win ? (int)n.depth : 0;
    out5[2] = pot; out5[3] = n.treasures; out5[4] = ene;
}

// MDungeonProblem._run_game (mdungeon_prob.py:110-126): A*(1), A*(0.5), A*(0), BFS.
//
// Exact shortcut: an A* run that ends because its queue ran empty has expanded every reachable live state and
// none of them stands on the exit.  Every other agent expands exactly the same states (an expanded state always
// gets four children, whatever the order), so it pops exactly as many entries -- it cannot reach the cap either --
// and finds no win.  What _run_game then returns comes from the BFS agent alone, so the remaining A* runs are
// skipped.  `clear_table(size)` zeroes the visited table before each agent.
template <class HP, class TP, class ClearFn>
PCGRL_D void md_run_game(const MdLevel& L, MdNode* pool, HP heap, TP table, int table_size, MdNode& w, const MdNode& root, int power,
                         bool allow_shortcut, ClearFn clear_table, int* out5, int* iters) {
    const int KS[4] = {2, 1, 0, -1};
    bool win = false;
    for (int a = 0; a < 4; a++) iters[a] = 0;
    for (int a = 0; a < 4 && !win; a++) {
        clear_table(table_size);
        bool exhausted = false;
        win = md_search(L, pool, heap, table, table_size - 1, w, root, KS[a], power, iters[a], exhausted);
        if (a < 3 && !win && exhausted && allow_shortcut) a = 2;      // straight to BFS
    }
    md_result(L, root, w, win, out5);
}
