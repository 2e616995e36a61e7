// The planners of the search problems beyond the limits of the compact searches: bordered levels of more than 256 cells (up to
// PCGRL_MAX_LEVEL_CELLS = 16384, e.g. adjust_param(width=20, height=20) -> 22 x 22 = 484), solver_power beyond 16 383, Sokoban levels
// with more than 32 crates (up to SOKB_MAXC).  The reference takes any of these (sokoban_prob.py:60-73, mdungeon_prob.py:68-84,
// ddave_prob.py:67-82).  Part of the single translation unit pcgrl_abi.hip.
//
// Same engines, same exact order of exploration as sokoban_solver.h / mdungeon_solver.h / ddave_solver.h (which cite the engine
// files line by line) -- children in the engine's order, visited on pop with duplicates left in the queue, CPython heapq on
// (h + balance * depth) compared with `<` only, bestNode = min h then min depth then first seen -- with the fixed-width pieces
// widened:
//   * cell indices are 16 bits, level masks and "things still lying there" sets are cells / 64 words (not four);
//   * heap entries are 64 bits (priority << 32 | node index), the visited table holds 32-bit node indices;
//   * node pool, heap and table live in a per-block arena in global memory (DevBufs::big_arena), the level and the node being
//     expanded in LDS.
// One wavefront per job, lane 0 runs the search (a chain of data-dependent pops), all lanes clear the table; the four agents of a
// level run one after the other with the exact shortcuts of the compact versions.  This is the general path, not the tuned one.
#pragma once

#define SOKB_MAXC 256                 /* crates of a Sokoban level (more: reported through the status word) */
#define BIG_MAX_WORDS 256             /* PCGRL_MAX_LEVEL_CELLS / 64: 16 384 bordered cells (126 x 126 maps; round 5: 4 096 before) */

PCGRL_D bool big_hlt(uint64_t a, uint64_t b) { return a < (b & 0xFFFFFFFF00000000ull); }       // priority(a) < priority(b)
PCGRL_D void big_siftdown(uint64_t* heap, int startpos, int pos) {       // heapq._siftdown
    const uint64_t newitem = heap[pos];
    while (pos > startpos) {
        const int parentpos = (pos - 1) >> 1;
        const uint64_t parent = heap[parentpos];
        if (big_hlt(newitem, parent)) { heap[pos] = parent; pos = parentpos; continue; }
        break;
    }
    heap[pos] = newitem;
}
PCGRL_D void big_siftup(uint64_t* heap, int pos, int endpos) {           // heapq._siftup
    const int startpos = pos;
    const uint64_t newitem = heap[pos];
    int childpos = 2 * pos + 1;
    while (childpos < endpos) {
        const int rightpos = childpos + 1;
        uint64_t c = heap[childpos];
        if (rightpos < endpos) {
            const uint64_t r = heap[rightpos];
            if (!big_hlt(c, r)) { childpos = rightpos; c = r; }
        }
        heap[pos] = c;
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    heap[pos] = newitem;
    big_siftdown(heap, startpos, pos);
}
PCGRL_D void big_heappush(uint64_t* heap, int& n, uint64_t item) { heap[n] = item; n++; big_siftdown(heap, 0, n - 1); }
PCGRL_D uint64_t big_heappop(uint64_t* heap, int& n) {
    const uint64_t last = heap[--n];
    if (n == 0) return last;
    const uint64_t top = heap[0];
    heap[0] = last;
    big_siftup(heap, 0, n);
    return top;
}
PCGRL_D bool big_bit(const uint64_t* m, int p) { return (m[p >> 6] >> (p & 63)) & 1ull; }
PCGRL_D void big_set(uint64_t* m, int p) { m[p >> 6] |= 1ull << (p & 63); }
PCGRL_D void big_clr(uint64_t* m, int p) { m[p >> 6] &= ~(1ull << (p & 63)); }

// What a job of k_search_big works with: the block's slice of the arena and the level geometry.
struct BigSearchCtx {
    uint8_t* pool; uint64_t* heap; uint32_t* table;
    int nodes_cap, table_mask, power;
    int w, h, cells, nwb;             // bordered level, words per cell set
    uint16_t* cx; uint16_t* cy;       // LDS: cell -> (x, y)
};
// visited test-and-add on the keys of pool nodes: KEYEQ(a, b) compares the keys of two nodes, `hs` is the key's hash
template <class KeyEq>
PCGRL_D bool big_seen_or_add(const BigSearchCtx& C, uint32_t hs, int cur, KeyEq same) {
    uint32_t slot = hs & (uint32_t)C.table_mask;
    for (;;) {
        const uint32_t v = C.table[slot];
        if (v == 0) break;
        if (same((int)v - 1)) return true;
        slot = (slot + 1) & (uint32_t)C.table_mask;
    }
    C.table[slot] = (uint32_t)cur + 1u;
    return false;
}

// ---------------------------------------------------------------------------------------------------------------- Sokoban
// probs/sokoban/engine.py as used by SokobanProblem._run_game (sokoban_prob.py:85-122); see sokoban_solver.h.
struct SokbLevel {
    uint64_t solid[BIG_MAX_WORDS], dead[BIG_MAX_WORDS], tmask[BIG_MAX_WORDS];
    uint16_t target[SOKB_MAXC];
    int nc;
    int dirs[4];
};
// node: [player u16][h u16][depth u32][crate u16 x ncap]
PCGRL_HD int sokb_stride(int nc) { return 8 + 2 * ((nc + 3) & ~3); }
struct alignas(8) SokbNode { uint16_t player, h; uint32_t depth; uint16_t crate[SOKB_MAXC]; };

PCGRL_D int sokb_crate_at(const SokbLevel& L, const uint16_t* crate, int p) {
    for (int i = 0; i < L.nc; i++) if (crate[i] == p) return i;
    return -1;
}
PCGRL_D bool sokb_win(const SokbLevel& L, const uint16_t* crate) {        // engine.py:272-280
    for (int i = 0; i < L.nc; i++) if (sokb_crate_at(L, crate, L.target[i]) < 0) return false;
    return true;
}
// engine.py:282-296: every crate takes the nearest target still on the list (the first one when none is nearer than w + h).
// `used`: SOKB_MAXC / 64 scratch words.
PCGRL_D int sokb_heuristic(const BigSearchCtx& C, const SokbLevel& L, const uint16_t* crate, uint64_t* used) {
    for (int i = 0; i < SOKB_MAXC / 64; i++) used[i] = 0ull;
    int distance = 0;
    for (int c = 0; c < L.nc; c++) {
        const int cx = C.cx[crate[c]], cy = C.cy[crate[c]];
        int best = C.w + C.h, match = -1, firstfree = -1, matchd = 0, firstd = 0;
        for (int i = 0; i < L.nc; i++) {
            if (big_bit(used, i)) continue;
            const int d = abs(cx - (int)C.cx[L.target[i]]) + abs(cy - (int)C.cy[L.target[i]]);
            if (firstfree < 0) { firstfree = i; firstd = d; }
            if (best > d) { match = i; best = d; matchd = d; }
        }
        if (match < 0) { match = firstfree; matchd = firstd; }
        distance += matchd;
        big_set(used, match);
    }
    return distance;
}
// sokoban_prob.py:85-102 + engine.py:135-184: the bordered level, crates and targets in row-major order.  Returns the number of
// crates found (beyond SOKB_MAXC the lists are truncated and the caller reports it).
PCGRL_D int sokb_build_level(const BigSearchCtx& C, const uint8_t* m, int W, SokbLevel& L, SokbNode& root) {
    L.nc = 0;
    L.dirs[0] = -1; L.dirs[1] = 1; L.dirs[2] = -C.w; L.dirs[3] = C.w;
    for (int k = 0; k < C.nwb; k++) { L.solid[k] = 0; L.tmask[k] = 0; L.dead[k] = 0; }
    int nt = 0, ncr = 0;
    root.player = 0; root.h = 0; root.depth = 0;
    for (int y = 0; y < C.h; y++)
        for (int x = 0; x < C.w; x++) {
            const int p = y * C.w + x;
            C.cx[p] = (uint16_t)x; C.cy[p] = (uint16_t)y;
            const bool border = x == 0 || y == 0 || x == C.w - 1 || y == C.h - 1;
            const int t = border ? 1 : m[(y - 1) * W + (x - 1)];
            if (t == 1) big_set(L.solid, p);
            if (t == 2) root.player = (uint16_t)p;
            if (t == 3) { if (ncr < SOKB_MAXC) root.crate[ncr] = (uint16_t)p; ncr++; }
            if (t == 4) { if (nt < SOKB_MAXC) L.target[nt] = (uint16_t)p; nt++; big_set(L.tmask, p); }
        }
    L.nc = ncr < SOKB_MAXC ? ncr : SOKB_MAXC;
    for (int i = L.nc; i < ((L.nc + 3) & ~3); i++) root.crate[i] = 0;
    return ncr;
}
// engine.py:203-246 intializeDeadlocks.  `corners`: scratch for up to `cells` cell indices.
PCGRL_D void sokb_init_deadlocks(const BigSearchCtx& C, SokbLevel& L, uint16_t* corners) {
    const int w = C.w, h = C.h;
    int nc = 0;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            const int p = y * w + x;
            if (big_bit(L.solid, p)) continue;
            const bool up = big_bit(L.solid, p - w), dn = big_bit(L.solid, p + w), lf = big_bit(L.solid, p - 1), rt = big_bit(L.solid, p + 1);
            if (((up && lf) || (up && rt) || (dn && lf) || (dn && rt)) && !big_bit(L.tmask, p)) { corners[nc++] = (uint16_t)p; big_set(L.dead, p); }
        }
    for (int a = 0; a < nc; a++)
        for (int b = 0; b < nc; b++) {
            const int ax = C.cx[corners[a]], ay = C.cy[corners[a]], bx = C.cx[corners[b]], by = C.cy[corners[b]];
            const int dx = (ax > bx) - (ax < bx), dy = (ay > by) - (ay < by);
            if ((dx == 0 && dy == 0) || (dx != 0 && dy != 0)) continue;
            bool ok = true;
            if (dx != 0) {
                for (int x = bx + dx; x != ax; x += dx) {
                    const int p = by * w + x;
                    if (big_bit(L.tmask, p) || big_bit(L.solid, p) || (!big_bit(L.solid, p - w) && !big_bit(L.solid, p + w))) { ok = false; break; }
                }
                if (ok) for (int x = bx + dx; x != ax; x += dx) big_set(L.dead, by * w + x);
            } else {
                for (int y = by + dy; y != ay; y += dy) {
                    const int p = y * w + bx;
                    if (big_bit(L.tmask, p) || big_bit(L.solid, p) || (!big_bit(L.solid, p - 1) && !big_bit(L.solid, p + 1))) { ok = false; break; }
                }
                if (ok) for (int y = by + dy; y != ay; y += dy) big_set(L.dead, y * w + bx);
            }
        }
}
// (memcpy of eight bytes at a time: a node is written here as words and read back through its fields -- with plain 64-bit
//  lvalues the compiler's type-based alias analysis may move the field reads before the copy)
PCGRL_D void sokb_copy(uint8_t* dst, const uint8_t* src, int stride) {
    for (int i = 0; i < stride; i += 8) { uint64_t v; __builtin_memcpy(&v, src + i, 8); __builtin_memcpy(dst + i, &v, 8); }
}
// One agent.  k < 0: BFSAgent, else AStarAgent with integer weight k in {2, 1, 0} (priority 2h + k * depth).  `w`: the node
// workspace (LDS).  Returns win; out_h / out_depth describe the returned node (winner, or the best node).
PCGRL_D bool sokb_search(const BigSearchCtx& C, const SokbLevel& L, SokbNode& w, const SokbNode& root, int k, uint64_t* used, int& out_h, int& out_depth,
                         int& out_iters, bool& out_exhausted) {
    const int stride = sokb_stride(L.nc);
    int npool = 1, head = 0, heapn = 0, iterations = 0, best = -1, best_h = 0, best_depth = 0;
    sokb_copy(C.pool, reinterpret_cast<const uint8_t*>(&root), stride);
    if (k >= 0) { C.heap[0] = (uint64_t)(2 * root.h + k * (int)root.depth) << 32; heapn = 1; }
    bool win = false;
    int result_h = root.h, result_depth = 0;
    while (iterations < C.power && (k >= 0 ? heapn > 0 : head < npool)) {
        iterations++;
        const int cur = k >= 0 ? (int)(big_heappop(C.heap, heapn) & 0xFFFFFFFFull) : head++;
        sokb_copy(reinterpret_cast<uint8_t*>(&w), C.pool + (size_t)cur * stride, stride);
        const int node_h = w.h, node_depth = (int)w.depth, node_player = w.player;
        if (sokb_win(L, w.crate)) { win = true; result_h = node_h; result_depth = node_depth; break; }
        uint32_t hs = 2166136261u;
        hs = (hs ^ w.player) * 16777619u;
        for (int i = 0; i < L.nc; i++) hs = (hs ^ w.crate[i]) * 16777619u;
        hs ^= hs >> 15;
        const SokbNode* wp = &w;
        const bool seen = big_seen_or_add(C, hs, cur, [&](int other) {
            const SokbNode* o = reinterpret_cast<const SokbNode*>(C.pool + (size_t)other * stride);
            if (o->player != wp->player) return false;
            for (int i = 0; i < L.nc; i++) if (o->crate[i] != wp->crate[i]) return false;
            return true;
        });
        if (seen) continue;
        if (best < 0 || node_h < best_h || (node_h == best_h && node_depth < best_depth)) { best = cur; best_h = node_h; best_depth = node_depth; }
        w.depth = (uint32_t)(node_depth + 1);
        for (int d = 0; d < 4; d++) {          // Node.getChildren: L, R, U, D; State.update engine.py:298-327
            const int np = node_player + L.dirs[d];
            if (big_bit(L.solid, np)) continue;                                  // the player did not move
            const int c = sokb_crate_at(L, w.crate, np);
            int cp = 0;
            if (c >= 0) {
                cp = np + L.dirs[d];
                if (big_bit(L.solid, cp) || sokb_crate_at(L, w.crate, cp) >= 0) continue;          // blocked crate: no move
            }
            w.player = (uint16_t)np;
            bool keep = true;
            if (c >= 0) {
                w.crate[c] = (uint16_t)cp;
                bool deadlock = false;                                            // checkDeadlock looks at every crate
                for (int i = 0; i < L.nc; i++) deadlock = deadlock || big_bit(L.dead, w.crate[i]);
                keep = !deadlock;
                if (keep) w.h = (uint16_t)sokb_heuristic(C, L, w.crate, used);
            }
            if (keep && npool < C.nodes_cap) {
                sokb_copy(C.pool + (size_t)npool * stride, reinterpret_cast<const uint8_t*>(&w), stride);
                if (k >= 0) big_heappush(C.heap, heapn, ((uint64_t)(2 * w.h + k * (int)w.depth) << 32) | (uint32_t)npool);
                npool++;
            }
            w.player = (uint16_t)node_player;                                     // undo
            if (c >= 0) { w.crate[c] = (uint16_t)np; w.h = (uint16_t)node_h; }
        }
    }
    if (!win) { result_h = best_h; result_depth = best_depth; }
    out_h = result_h; out_depth = result_depth; out_iters = iterations;
    out_exhausted = !win && !(k >= 0 ? heapn > 0 : head < npool);
    return win;
}

// ---------------------------------------------------------------------------------------------------------------- MiniDungeons
// probs/mdungeon/engine.py as used by MDungeonProblem._run_game (mdungeon_prob.py:91-126); see mdungeon_solver.h.
struct MdbLevel { uint64_t solid[BIG_MAX_WORDS], potion[BIG_MAX_WORDS], treasure[BIG_MAX_WORDS], goblin[BIG_MAX_WORDS], ogre[BIG_MAX_WORDS]; int door; int dirs[4]; };
// node: [alive u64 x nwb][player u16][treasures u16][h i32][depth u32][health u8, pad x3]
struct MdbNode { uint64_t alive[BIG_MAX_WORDS]; };
struct MdbTail { uint16_t player, treasures; int32_t h; uint32_t depth; uint8_t health, flags, jumps_lo, jumps_hi; };
PCGRL_HD int mdb_stride(int nwb) { return nwb * 8 + 16; }
#define MDB_PRIO_BIAS (1 << 18)       /* 2h >= -8 * 16384 */
PCGRL_D int mdb_heuristic(const BigSearchCtx& C, int door, int player, int health, int treasures) {       // engine.py:271-275
    return abs((int)C.cx[player] - (int)C.cx[door]) + abs((int)C.cy[player] - (int)C.cy[door]) + 4 * (5 - health) - 4 * treasures;
}
// The node being expanded: `alive` words in LDS, the tail in registers of lane 0.
struct MdbWork { uint64_t* alive; MdbTail t; };
PCGRL_D void mdb_store(const BigSearchCtx& C, int idx, const MdbWork& w) {
    uint8_t* p = C.pool + (size_t)idx * mdb_stride(C.nwb);
    for (int i = 0; i < C.nwb; i++) reinterpret_cast<uint64_t*>(p)[i] = w.alive[i];
    *reinterpret_cast<MdbTail*>(p + C.nwb * 8) = w.t;
}
PCGRL_D void mdb_load(const BigSearchCtx& C, int idx, MdbWork& w) {
    const uint8_t* p = C.pool + (size_t)idx * mdb_stride(C.nwb);
    for (int i = 0; i < C.nwb; i++) w.alive[i] = reinterpret_cast<const uint64_t*>(p)[i];
    w.t = *reinterpret_cast<const MdbTail*>(p + C.nwb * 8);
}
PCGRL_D uint32_t mdb_hash(const BigSearchCtx& C, const uint64_t* alive, uint64_t head) {
    uint64_t x = head;
    for (int i = 0; i < C.nwb; i++) { x = (x ^ alive[i]) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; }
    return (uint32_t)(x ^ (x >> 32));
}
// mdungeon_prob.py:92-108 + engine.py:143-181.  tiles: 0 empty 1 solid 2 player 3 exit 4 potion 5 treasure 6 goblin 7 ogre
PCGRL_D void mdb_build_level(const BigSearchCtx& C, const uint8_t* m, int W, MdbLevel& L, MdbWork& root) {
    L.door = 0;
    L.dirs[0] = -1; L.dirs[1] = 1; L.dirs[2] = -C.w; L.dirs[3] = C.w;
    for (int k = 0; k < C.nwb; k++) { L.solid[k] = 0; L.potion[k] = 0; L.treasure[k] = 0; L.goblin[k] = 0; L.ogre[k] = 0; root.alive[k] = 0; }
    root.t.player = 0; root.t.treasures = 0; root.t.h = 0; root.t.depth = 0; root.t.health = 5; root.t.flags = 0; root.t.jumps_lo = 0; root.t.jumps_hi = 0;
    for (int y = 0; y < C.h; y++)
        for (int x = 0; x < C.w; x++) {
            const int p = y * C.w + x;
            C.cx[p] = (uint16_t)x; C.cy[p] = (uint16_t)y;
            const bool border = x == 0 || y == 0 || x == C.w - 1 || y == C.h - 1;
            const int t = border ? 1 : m[(y - 1) * W + (x - 1)];
            if (t == 1) big_set(L.solid, p);
            if (t == 2) root.t.player = (uint16_t)p;
            if (t == 3) L.door = p;
            if (t == 4) big_set(L.potion, p);
            if (t == 5) big_set(L.treasure, p);
            if (t == 6) big_set(L.goblin, p);
            if (t == 7) big_set(L.ogre, p);
            if (t >= 4) big_set(root.alive, p);
        }
    root.t.h = mdb_heuristic(C, L.door, root.t.player, 5, 0);
}
// One agent (contract of md_search): on return `w` holds the returned node.
PCGRL_D bool mdb_search(const BigSearchCtx& C, const MdbLevel& L, MdbWork& w, int k, int& out_iters, bool& out_exhausted) {
    int npool = 1, head = 0, heapn = 0, iterations = 0, best = -1, best_h = 0, best_depth = 0;
    // (the root is node 0 of the pool: the caller stored it there)
    mdb_load(C, 0, w);
    if (k >= 0) { C.heap[0] = (uint64_t)(uint32_t)(2 * w.t.h + k * (int)w.t.depth + MDB_PRIO_BIAS) << 32; heapn = 1; }
    bool win = false;
    int result = 0;
    while (iterations < C.power && (k >= 0 ? heapn > 0 : head < npool)) {
        iterations++;
        const int cur = k >= 0 ? (int)(big_heappop(C.heap, heapn) & 0xFFFFFFFFull) : head++;
        mdb_load(C, cur, w);
        if (w.t.health == 0) continue;                                             // checkLose
        if (w.t.player == L.door) { win = true; result = cur; break; }              // checkWin
        const uint64_t keyhead = ((uint64_t)w.t.player << 8) | w.t.health;
        const uint32_t hs = mdb_hash(C, w.alive, keyhead);
        const MdbWork* wp = &w;
        const bool seen = big_seen_or_add(C, hs, cur, [&](int other) {
            const uint8_t* p = C.pool + (size_t)other * mdb_stride(C.nwb);
            const MdbTail* ot = reinterpret_cast<const MdbTail*>(p + C.nwb * 8);
            if (ot->player != wp->t.player || ot->health != wp->t.health) return false;
            for (int i = 0; i < C.nwb; i++) if (reinterpret_cast<const uint64_t*>(p)[i] != wp->alive[i]) return false;
            return true;
        });
        if (seen) continue;
        const int node_h = w.t.h, node_depth = (int)w.t.depth, node_player = w.t.player, node_health = w.t.health, node_tr = w.t.treasures;
        if (best < 0 || node_h < best_h || (node_h == best_h && node_depth < best_depth)) { best = cur; best_h = node_h; best_depth = node_depth; }
        w.t.depth = (uint32_t)(node_depth + 1);
        for (int d = 0; d < 4; d++) {          // Node.getChildren: L, R, U, D -- always four (engine.py:14-20)
            int np = node_player + L.dirs[d], health = node_health, tr = node_tr, taken = -1;
            if (big_bit(L.solid, np)) np = node_player;                            // checkMovableLocation fails: nothing happens
            else if (big_bit(w.alive, np)) {                                        // updatePlayer engine.py:215-255
                taken = np;
                if (big_bit(L.potion, np)) { health += 2; if (health > 5) health = 5; }
                else if (big_bit(L.treasure, np)) tr += 1;
                else { health -= big_bit(L.ogre, np) ? 2 : 1; if (health < 0) health = 0; }
            }
            if (taken >= 0) big_clr(w.alive, taken);
            w.t.player = (uint16_t)np; w.t.health = (uint8_t)health; w.t.treasures = (uint16_t)tr;
            w.t.h = mdb_heuristic(C, L.door, np, health, tr);
            if (npool < C.nodes_cap) {
                mdb_store(C, npool, w);
                if (k >= 0) big_heappush(C.heap, heapn, ((uint64_t)(uint32_t)(2 * w.t.h + k * (int)w.t.depth + MDB_PRIO_BIAS) << 32) | (uint32_t)npool);
                npool++;
            }
            if (taken >= 0) big_set(w.alive, taken);                                // undo
        }
    }
    if (!win) result = best < 0 ? 0 : best;
    mdb_load(C, result, w);
    out_iters = iterations;
    out_exhausted = !win && !(k >= 0 ? heapn > 0 : head < npool);
    return win;
}

// ---------------------------------------------------------------------------------------------------------------- Dangerous Dave
// probs/ddave/engine.py as used by DDaveProblem._run_game (ddave_prob.py:92-127); see ddave_solver.h.  The node is MiniDungeons'
// with other meanings: alive = diamonds still there, flags = health | key still on the floor << 1 | air time << 4, jumps in two bytes.
struct DdbLevel { uint64_t solid[BIG_MAX_WORDS], spike[BIG_MAX_WORDS], diamond0[BIG_MAX_WORDS]; int door, keycell; };
#define DDB_PRIO_BIAS (1 << 18)       /* 2h >= -10 * 16384 */
PCGRL_D int ddb_diamonds(const BigSearchCtx& C, const DdbLevel& L, const uint64_t* alive) {
    int n = 0;
    for (int i = 0; i < C.nwb; i++) n += md_popcount(L.diamond0[i] & ~alive[i]);
    return n;
}
PCGRL_D int ddb_heuristic(const BigSearchCtx& C, const DdbLevel& L, int player, bool key_there, int diamonds) {     // engine.py:296-301
    const int t = key_there ? L.keycell : L.door;
    return abs((int)C.cx[player] - (int)C.cx[t]) + abs((int)C.cy[player] - (int)C.cy[t]) + (key_there ? C.w + C.h : 0) - 5 * diamonds;
}
// ddave_prob.py:93-109 + engine.py:141-190.  tiles: 0 empty 1 solid 2 player 3 exit 4 diamond 5 key 6 spike
PCGRL_D void ddb_build_level(const BigSearchCtx& C, const uint8_t* m, int W, DdbLevel& L, MdbWork& root) {
    L.door = 0; L.keycell = 0;
    for (int k = 0; k < C.nwb; k++) { L.solid[k] = 0; L.spike[k] = 0; L.diamond0[k] = 0; root.alive[k] = 0; }
    root.t.player = 0; root.t.treasures = 0; root.t.h = 0; root.t.depth = 0; root.t.health = 0; root.t.flags = DD_F_HEALTH; root.t.jumps_lo = 0; root.t.jumps_hi = 0;
    for (int y = 0; y < C.h; y++)
        for (int x = 0; x < C.w; x++) {
            const int p = y * C.w + x;
            C.cx[p] = (uint16_t)x; C.cy[p] = (uint16_t)y;
            const bool border = x == 0 || y == 0 || x == C.w - 1 || y == C.h - 1;
            const int t = border ? 1 : m[(y - 1) * W + (x - 1)];
            if (t == 1) big_set(L.solid, p);
            if (t == 2) root.t.player = (uint16_t)p;
            if (t == 3) L.door = p;
            if (t == 4) { big_set(L.diamond0, p); big_set(root.alive, p); }
            if (t == 5) { L.keycell = p; root.t.flags |= DD_F_KEY_THERE; }
            if (t == 6) big_set(L.spike, p);
        }
    root.t.h = ddb_heuristic(C, L, root.t.player, (root.t.flags & DD_F_KEY_THERE) != 0, 0);
}
PCGRL_D bool ddb_search(const BigSearchCtx& C, const DdbLevel& L, MdbWork& w, int k, int& out_iters, bool& out_exhausted) {
    int npool = 1, head = 0, heapn = 0, iterations = 0, best = -1, best_h = 0, best_depth = 0;
    mdb_load(C, 0, w);
    if (k >= 0) { C.heap[0] = (uint64_t)(uint32_t)(2 * w.t.h + DDB_PRIO_BIAS) << 32; heapn = 1; }
    bool win = false;
    int result = 0;
    while (iterations < C.power && (k >= 0 ? heapn > 0 : head < npool)) {
        iterations++;
        const int cur = k >= 0 ? (int)(big_heappop(C.heap, heapn) & 0xFFFFFFFFull) : head++;
        mdb_load(C, cur, w);
        if (!(w.t.flags & DD_F_HEALTH)) continue;                                                      // checkLose
        if (!(w.t.flags & DD_F_KEY_THERE) && w.t.player == L.door) { win = true; result = cur; break; }    // checkWin
        // State.getKey (engine.py:283-294): cell, health, the key if it is still there, the diamonds left -- NOT air time / jumps
        const uint64_t keyhead = ((uint64_t)w.t.player << 8) | (uint64_t)(w.t.flags & (DD_F_HEALTH | DD_F_KEY_THERE));
        const uint32_t hs = mdb_hash(C, w.alive, keyhead);
        const MdbWork* wp = &w;
        const bool seen = big_seen_or_add(C, hs, cur, [&](int other) {
            const uint8_t* p = C.pool + (size_t)other * mdb_stride(C.nwb);
            const MdbTail* ot = reinterpret_cast<const MdbTail*>(p + C.nwb * 8);
            if (ot->player != wp->t.player || ((ot->flags ^ wp->t.flags) & (DD_F_HEALTH | DD_F_KEY_THERE)) != 0) return false;
            for (int i = 0; i < C.nwb; i++) if (reinterpret_cast<const uint64_t*>(p)[i] != wp->alive[i]) return false;
            return true;
        });
        if (seen) continue;
        const int node_h = w.t.h, node_depth = (int)w.t.depth, node_player = w.t.player, node_flags = w.t.flags;
        const int node_jumps = (int)w.t.jumps_lo | ((int)w.t.jumps_hi << 8);
        if (best < 0 || node_h < best_h || (node_h == best_h && node_depth < best_depth)) { best = cur; best_h = node_h; best_depth = node_depth; }
        const bool ground = big_bit(L.solid, node_player + C.w), ceiling = big_bit(L.solid, node_player - C.w);
        const int node_dia = ddb_diamonds(C, L, w.alive);
        w.t.depth = (uint32_t)(node_depth + 1);
        for (int d = 0; d < 4; d++) {          // stay, left, right, jump -- always four; State.update engine.py:226-263
            int np = node_player, air = node_flags >> DD_F_AIR_SHIFT, jumps = node_jumps;
            if (d == 1) { if (!big_bit(L.solid, np - 1)) np -= 1; }
            else if (d == 2) { if (!big_bit(L.solid, np + 1)) np += 1; }
            else if (d == 3) { if (ground && !ceiling) { air = 3; jumps += 1; } }
            if (air > 1) {
                air -= 1;
                if (!big_bit(L.solid, np - C.w)) np -= C.w; else air = 1;
            } else if (air == 1) {
                air = 0;
            } else {
                if (!big_bit(L.solid, np + C.w)) np += C.w;
            }
            int fl = node_flags & (DD_F_HEALTH | DD_F_KEY_THERE), dia = node_dia, taken = -1;
            if (big_bit(w.alive, np)) { taken = np; dia += 1; }                      // updatePlayer :265-281
            else if (big_bit(L.spike, np)) fl &= ~DD_F_HEALTH;
            else if ((fl & DD_F_KEY_THERE) && np == L.keycell) fl &= ~DD_F_KEY_THERE;
            if (taken >= 0) big_clr(w.alive, taken);
            w.t.player = (uint16_t)np; w.t.flags = (uint8_t)(fl | (air << DD_F_AIR_SHIFT));
            w.t.jumps_lo = (uint8_t)(jumps & 255); w.t.jumps_hi = (uint8_t)((jumps >> 8) & 255);
            w.t.h = ddb_heuristic(C, L, np, (fl & DD_F_KEY_THERE) != 0, dia);
            if (npool < C.nodes_cap) {
                mdb_store(C, npool, w);
                if (k >= 0) big_heappush(C.heap, heapn, ((uint64_t)(uint32_t)(2 * w.t.h + k * (int)w.t.depth + DDB_PRIO_BIAS) << 32) | (uint32_t)npool);
                npool++;
            }
            if (taken >= 0) big_set(w.alive, taken);                                 // undo
        }
    }
    if (!win) result = best < 0 ? 0 : best;
    mdb_load(C, result, w);
    out_iters = iterations;
    out_exhausted = !win && !(k >= 0 ? heapn > 0 : head < npool);
    return win;
}
