// Dangerous Dave planner for the reward path (device side).
//
// Restates probs/ddave/engine.py as used by DDaveProblem._run_game (ddave_prob.py:92-127): AStarAgent with balance 1,
// 0.5, 0, then BFSAgent (solver_power pops each); the first agent whose returned state has the key and stands on the
// exit gives (dist-win 0, sol-length = depth, game status of that state); otherwise the BFS agent's best node gives
// (heuristic, 0, its game status).  The search loops are those of the mdungeon engine (mdungeon_solver.h); what differs:
//   * children in the order stay, left, right, jump (engine.py:3) -- always four; State.update (engine.py:226-263) is a
//     little platformer: a sideways step if the cell is free, a jump (air time 3, jump counter +1) only from the
//     ground under a free ceiling, then one step up while air time > 1 (a blocked rise leaves air time 1), one hover
//     step at air time 1, else one step down if the cell below is free; updatePlayer (:265-281): a diamond is picked
//     up, else a spike kills, else the key is picked up
//   * State.getKey (engine.py:283-294) = player cell, health, the key if it is still there, the diamonds left -- NOT
//     the air time or the jump counter.  Two states that differ only in those count as the same for the visited set;
//     whichever is popped first is expanded.  The set of states an agent expands therefore depends on its order of
//     exploration, and the "an exhausted agent has seen everything" shortcut of the other two engines does NOT hold
//     here: every agent runs until it wins, exhausts its own queue or reaches the cap
//   * heuristic (engine.py:296-301): distance to the exit, or while the key lies there distance to the key + level
//     width + height; minus 5 per collected diamond (negative values: biased heap priorities).
// One wavefront per search, lane 0 drives it; heap and visited table in LDS (global arena for a large solver_power),
// 40-byte nodes in the global arena shared with the other solvers.
//
// Limits (checked by the host): (W+2)*(H+2) <= 256, solver_power <= 16383.
#pragma once
#include "mdungeon_solver.h"

#define DD_PRIO_BIAS 4096      /* 2h >= -10*254 */
enum { DD_F_HEALTH = 1, DD_F_KEY_THERE = 2, DD_F_AIR_SHIFT = 4 };

struct alignas(8) DdNode {     // 40 bytes, moved around as five 64-bit words
    uint64_t alive[4];         // diamonds still there, one bit per bordered cell
    uint8_t player, flags;     // flags: health | key still on the floor << 1 | air time << 4
    int16_t h;
    uint16_t depth, jumps;
};
struct DdLevel {
    uint64_t solid[4], spike[4], diamond0[4];
    uint8_t cx[256], cy[256];
    int w, h, cells, door, keycell;
};

PCGRL_D int dd_diamonds(const DdLevel& L, const uint64_t* alive) {
    int n = 0;
    for (int i = 0; i < 4; i++) n += md_popcount(L.diamond0[i] & ~alive[i]);
    return n;
}
PCGRL_D int dd_heuristic(const DdLevel& L, int player, bool key_there, int diamonds) {   // engine.py:296-301
    const int t = key_there ? L.keycell : L.door;
    return abs((int)L.cx[player] - (int)L.cx[t]) + abs((int)L.cy[player] - (int)L.cy[t]) + (key_there ? L.w + L.h : 0) - 5 * diamonds;
}
// ddave_prob.py:93-109 + engine.py:141-190.  tiles: 0 empty 1 solid 2 player 3 exit 4 diamond 5 key 6 spike
PCGRL_D void dd_build_level(const uint8_t* m, int W, int H, DdLevel& L, DdNode& root) {
    L.w = W + 2; L.h = H + 2; L.cells = L.w * L.h; L.door = 0; L.keycell = 0;
    for (int k = 0; k < 4; k++) { L.solid[k] = 0; L.spike[k] = 0; L.diamond0[k] = 0; root.alive[k] = 0; }
    root.player = 0; root.flags = DD_F_HEALTH; root.h = 0; root.depth = 0; root.jumps = 0;
    for (int y = 0; y < L.h; y++)
        for (int x = 0; x < L.w; x++) {
            const int p = y * L.w + x;
            L.cx[p] = (uint8_t)x; L.cy[p] = (uint8_t)y;
            const bool border = x == 0 || y == 0 || x == L.w - 1 || y == L.h - 1;
            const int t = border ? 1 : m[(y - 1) * W + (x - 1)];
            if (t == 1) sok_set(L.solid, p);
            if (t == 2) root.player = (uint8_t)p;
            if (t == 3) L.door = p;
            if (t == 4) { sok_set(L.diamond0, p); sok_set(root.alive, p); }
            if (t == 5) { L.keycell = p; root.flags |= DD_F_KEY_THERE; }
            if (t == 6) sok_set(L.spike, p);
        }
    root.h = (int16_t)dd_heuristic(L, root.player, (root.flags & DD_F_KEY_THERE) != 0, 0);
}
PCGRL_D uint32_t dd_hash(const DdNode& n) {
    uint64_t x = ((uint64_t)n.player << 8) | (uint64_t)(n.flags & (DD_F_HEALTH | DD_F_KEY_THERE));
    for (int i = 0; i < 4; i++) { x = (x ^ n.alive[i]) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; }
    return (uint32_t)(x ^ (x >> 32));
}
PCGRL_D bool dd_same(const DdNode& a, const DdNode& b) {      // equality of State.getKey
    return a.player == b.player && ((a.flags ^ b.flags) & (DD_F_HEALTH | DD_F_KEY_THERE)) == 0 && a.alive[0] == b.alive[0] &&
           a.alive[1] == b.alive[1] && a.alive[2] == b.alive[2] && a.alive[3] == b.alive[3];
}

// One search (one lane).  Same contract as md_search.
template <class HP, class TP, class Hook>
PCGRL_D bool dd_search(const DdLevel& L, DdNode* pool, HP heap, TP table, int table_mask, DdNode& w, const DdNode& root, int k,
                       int power, int& out_iters, bool& out_exhausted, Hook hook) {
    int npool = 0, head = 0, heapn = 0, iterations = 0, best = -1, best_h = 0, best_depth = 0;
    MdNode* mpool = reinterpret_cast<MdNode*>(pool);          // same size and alignment: the 5 x 64-bit copy helpers
    md_copy(mpool, reinterpret_cast<const MdNode*>(&root));
    npool = 1;
    if (k >= 0) { heap[0] = ((uint32_t)(2 * root.h + DD_PRIO_BIAS) << 16) | 0u; heapn = 1; }
    bool win = false, aborted = false;
    int result = 0;
    MdRaw ahead = md_load(reinterpret_cast<const MdNode*>(&root));
    int ahead_idx = 0;
    while (iterations < power && (k >= 0 ? heapn > 0 : head < npool)) {
        iterations++;
        if (hook(iterations)) { aborted = true; break; }
        int cur;
        if (k >= 0) {
            const uint32_t last = heap[--heapn];
            cur = (int)((heapn > 0 ? heap[0] : last) & 0xFFFFu);
            MdRaw fetched = ahead;
            if (cur != ahead_idx) fetched = md_load(mpool + cur);
            if (heapn > 0) { heap[0] = last; sok_siftup(heap, 0, heapn); }
            md_store(reinterpret_cast<MdNode*>(&w), fetched);
            ahead_idx = -1;
            if (heapn > 0) { ahead_idx = (int)(heap[0] & 0xFFFFu); ahead = md_load(mpool + ahead_idx); }
        } else {
            cur = head++;
            MdRaw fetched = ahead;
            if (cur != ahead_idx) fetched = md_load(mpool + cur);
            md_store(reinterpret_cast<MdNode*>(&w), fetched);
            ahead_idx = -1;
            if (head < npool) { ahead_idx = head; ahead = md_load(mpool + head); }
        }
        if (!(w.flags & DD_F_HEALTH)) continue;                                              // checkLose
        if (!(w.flags & DD_F_KEY_THERE) && w.player == L.door) { win = true; result = cur; break; }   // checkWin
        const uint32_t hs = dd_hash(w);
        uint32_t slot = hs & (uint32_t)table_mask;
        const uint32_t tag = (hs >> 16) << 16;
        bool seen = false;
        for (;;) {
            const uint32_t v = table[slot];
            if (v == 0) break;
            if ((v & 0xFFFF0000u) == tag && dd_same(pool[(v & 0xFFFFu) - 1], w)) { seen = true; break; }
            slot = (slot + 1) & (uint32_t)table_mask;
        }
        if (seen) continue;
        table[slot] = tag | (uint32_t)(cur + 1);
        const int node_h = w.h, node_depth = w.depth, node_player = w.player, node_flags = w.flags, node_jumps = w.jumps;
        if (best < 0 || node_h < best_h || (node_h == best_h && node_depth < best_depth)) { best = cur; best_h = node_h; best_depth = node_depth; }
        const bool ground = sok_bit(L.solid, node_player + L.w), ceiling = sok_bit(L.solid, node_player - L.w);
        const int node_dia = dd_diamonds(L, w.alive);
        w.depth = (uint16_t)(node_depth + 1);
        for (int d = 0; d < 4; d++) {          // stay, left, right, jump -- always four
            int np = node_player, air = node_flags >> DD_F_AIR_SHIFT, jumps = node_jumps;
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
            int fl = node_flags & (DD_F_HEALTH | DD_F_KEY_THERE), dia = node_dia, taken = -1;
            if (sok_bit(w.alive, np)) { taken = np; dia += 1; }
            else if (sok_bit(L.spike, np)) fl &= ~DD_F_HEALTH;
            else if ((fl & DD_F_KEY_THERE) && np == L.keycell) fl &= ~DD_F_KEY_THERE;
            if (taken >= 0) w.alive[taken >> 6] &= ~(1ull << (taken & 63));
            w.player = (uint8_t)np; w.flags = (uint8_t)(fl | (air << DD_F_AIR_SHIFT)); w.jumps = (uint16_t)jumps;
            w.h = (int16_t)dd_heuristic(L, np, (fl & DD_F_KEY_THERE) != 0, dia);
            md_copy(mpool + npool, reinterpret_cast<const MdNode*>(&w));
            if (k >= 0) {
                heap[heapn] = ((uint32_t)(2 * w.h + k * w.depth + DD_PRIO_BIAS) << 16) | (uint32_t)npool;
                heapn++;
                sok_siftdown(heap, 0, heapn - 1);
            }
            npool++;
            if (taken >= 0) w.alive[taken >> 6] |= 1ull << (taken & 63);   // undo
        }
    }
    if (!win) result = best < 0 ? 0 : best;
    md_copy(reinterpret_cast<MdNode*>(&w), mpool + result);
    out_iters = iterations;
    out_exhausted = !win && !aborted && !(k >= 0 ? heapn > 0 : head < npool);
    return win;
}

// The four values _run_game hands to get_stats, from the node a search returned: dist-win, sol-length, num-jumps,
// col-diamonds.
PCGRL_D void dd_result(const DdLevel& L, const DdNode& n, bool win, int* out4) {
    out4[0] = win ? 0 : (int)n.h;
    out4[1] = win ? (int)n.depth : 0;
    out4[2] = n.jumps;
    out4[3] = dd_diamonds(L, n.alive);
}

// DDaveProblem._run_game (ddave_prob.py:111-127): A*(1), A*(0.5), A*(0), BFS, one after the other (host tests; the
// kernel runs them side by side).
template <class HP, class TP, class ClearFn>
PCGRL_D void dd_run_game(const DdLevel& L, DdNode* pool, HP heap, TP table, int table_size, DdNode& w, const DdNode& root, int power,
                         ClearFn clear_table, int* out4, int* iters) {
    const int KS[4] = {2, 1, 0, -1};
    bool win = false;
    for (int a = 0; a < 4; a++) iters[a] = 0;
    for (int a = 0; a < 4 && !win; a++) {
        clear_table(table_size);
        bool exhausted = false;
        win = dd_search(L, pool, heap, table, table_size - 1, w, root, KS[a], power, iters[a], exhausted, SokNoHook());
    }
    dd_result(L, w, win, out4);
}
