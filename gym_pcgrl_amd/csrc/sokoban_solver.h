// Sokoban solvability search for the reward path (device side).
//
// Restates probs/sokoban/engine.py as used by SokobanProblem._run_game (sokoban_prob.py:85-122):
// BFSAgent(power), then AStarAgent with balance 1, 0.5, 0 (power pops each); the first agent whose
// returned state wins gives (dist-win 0, sol-length = depth); otherwise dist-win is the heuristic of
// the last agent's best node.  Exactness needs the engine's precise order of exploration:
//   * level = map with a solid border, crates/targets collected row-major (engine.py:135-184)
//   * children in the order L,R,U,D; dropped if the player did not move or a crate moved and any
//     crate stands on a deadlock cell (engine.py:14-24, 203-252)
//   * visited key = player + *ordered* crate list, tested on pop; duplicates stay queued (engine.py:62-73)
//   * A* uses queue.PriorityQueue == CPython heapq: heappush/_siftdown, heappop/_siftup, comparing
//     only with Node.__lt__ (h + balance*depth; here 2h + {2,1,0}*depth as integers)
//   * bestNode = min h, then min depth, first seen.
//
// Mapping: one wavefront per solver job.  The search itself is a chain of data-dependent pops, so it
// is driven by lane 0; the binary heap and the visited hash table live in LDS (flat pointers: they
// fall back to a global arena when solver_power is too large for LDS), the node pool in a global
// arena.  All lanes cooperate on clearing the tables.
//
// Limits (checked by the host): (W+2)*(H+2) <= 256, solver_power <= 16383; more than SOK_MAXC
// crates raises the sticky status flag instead of returning a wrong answer.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else
#include <stdlib.h>
#endif
#include "pcgrl_common.h"

#define SOK_MAXC 32
#define SOK_LDS_POWER 5000                         /* solver_power up to this keeps heap+table in LDS */
#define SOK_LDS_HEAP (4 * SOK_LDS_POWER + 4)       /* entries (u32) */
#define SOK_LDS_TABLE 8192                         /* slots (u32), power of two */

struct alignas(8) SokNode {  // 40 bytes, moved around as five 64-bit words
    uint8_t crate[SOK_MAXC];
    uint8_t player, pad;
    uint16_t h;
    uint16_t depth, pad2;
};
struct SokRaw { uint64_t q[5]; };
PCGRL_D SokRaw sok_load(const SokNode* p) {
    const uint64_t* s = reinterpret_cast<const uint64_t*>(p);
    SokRaw r;
    r.q[0] = s[0]; r.q[1] = s[1]; r.q[2] = s[2]; r.q[3] = s[3]; r.q[4] = s[4];
    return r;
}
PCGRL_D void sok_store(SokNode* p, const SokRaw& r) {
    uint64_t* d = reinterpret_cast<uint64_t*>(p);
    d[0] = r.q[0]; d[1] = r.q[1]; d[2] = r.q[2]; d[3] = r.q[3]; d[4] = r.q[4];
}

struct SokLevel {
    uint64_t solid[4], dead[4], targetmask[4];
    uint8_t target[SOK_MAXC];
    uint8_t cx[256], cy[256];   // cell -> (x, y): the heuristic runs per child, integer division is slow on the GPU
    int w, h, cells, nc;    // bordered dims, number of crates == targets
    int dirs[4];
};

struct SokArena {           // per wave-slot global scratch
    SokNode* pool;          // [4*power + 4]
    uint32_t* heap;         // LDS or global
    uint32_t* table;        // LDS or global
    int table_mask;
};

PCGRL_D bool sok_bit(const uint64_t* m, int p) { return (m[p >> 6] >> (p & 63)) & 1ull; }
PCGRL_D void sok_set(uint64_t* m, int p) { m[p >> 6] |= 1ull << (p & 63); }

PCGRL_D int sok_crate_at(const SokLevel& L, const uint8_t* crate, int p) {
    for (int i = 0; i < L.nc; i++) if (crate[i] == p) return i;
    return -1;
}
PCGRL_D bool sok_win(const SokLevel& L, const uint8_t* crate) {   // engine.py:272-280
    for (int i = 0; i < L.nc; i++) if (sok_crate_at(L, crate, L.target[i]) < 0) return false;
    return true;
}
PCGRL_D int sok_heuristic(const SokLevel& L, const uint8_t* crate) {   // engine.py:282-296
    uint32_t used = 0;   // targets removed from the shrinking list ("del targets[bestMatch]")
    int distance = 0;
    for (int c = 0; c < L.nc; c++) {
        const int cx = L.cx[crate[c]], cy = L.cy[crate[c]];
        int best = L.w + L.h, match = -1, firstfree = -1, matchd = 0, firstd = 0;
        for (int i = 0; i < L.nc; i++) {
            if ((used >> i) & 1u) continue;
            const int d = abs(cx - (int)L.cx[L.target[i]]) + abs(cy - (int)L.cy[L.target[i]]);
            if (firstfree < 0) { firstfree = i; firstd = d; }
            if (best > d) { match = i; best = d; matchd = d; }
        }
        if (match < 0) { match = firstfree; matchd = firstd; }   // bestMatch stays 0 = first remaining target
        distance += matchd;
        used |= 1u << match;
    }
    return distance;
}

// sokoban_prob.py:85-102 + engine.py:135-184: bordered level, crates/targets collected row-major.
// Returns the number of crates found (may exceed SOK_MAXC; the lists are then truncated).
PCGRL_D int sok_build_level(const uint8_t* m, int W, int H, SokLevel& L, SokNode& root) {
    L.w = W + 2; L.h = H + 2; L.cells = L.w * L.h; L.nc = 0;
    L.dirs[0] = -1; L.dirs[1] = 1; L.dirs[2] = -L.w; L.dirs[3] = L.w;
    for (int k = 0; k < 4; k++) { L.solid[k] = 0; L.targetmask[k] = 0; L.dead[k] = 0; }
    int nt = 0, ncr = 0;
    root.player = 0; root.pad = 0; root.pad2 = 0; root.depth = 0; root.h = 0;
    for (int i = 0; i < SOK_MAXC; i++) { root.crate[i] = 0; L.target[i] = 0; }
    for (int y = 0; y < L.h; y++)
        for (int x = 0; x < L.w; x++) {
            const int p = y * L.w + x;
            L.cx[p] = (uint8_t)x; L.cy[p] = (uint8_t)y;
            const bool border = x == 0 || y == 0 || x == L.w - 1 || y == L.h - 1;
            const int t = border ? 1 : m[(y - 1) * W + (x - 1)];
            if (t == 1) sok_set(L.solid, p);
            if (t == 2) root.player = (uint8_t)p;
            if (t == 3) { if (ncr < SOK_MAXC) root.crate[ncr] = (uint8_t)p; ncr++; }
            if (t == 4) { if (nt < SOK_MAXC) L.target[nt] = (uint8_t)p; nt++; sok_set(L.targetmask, p); }
        }
    L.nc = ncr < SOK_MAXC ? ncr : SOK_MAXC;
    return ncr;
}

// engine.py:203-246 intializeDeadlocks
PCGRL_D void sok_init_deadlocks(SokLevel& L) {
    for (int k = 0; k < 4; k++) L.dead[k] = 0;
    const int w = L.w, h = L.h;
    uint8_t corners[64];
    int nc = 0;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            const int p = y * w + x;
            if (sok_bit(L.solid, p)) continue;
            const bool up = sok_bit(L.solid, p - w), dn = sok_bit(L.solid, p + w), lf = sok_bit(L.solid, p - 1), rt = sok_bit(L.solid, p + 1);
            if ((up && lf) || (up && rt) || (dn && lf) || (dn && rt)) {
                if (!sok_bit(L.targetmask, p)) {
                    if (nc < 64) corners[nc++] = (uint8_t)p;
                    sok_set(L.dead, p);
                }
            }
        }
    for (int a = 0; a < nc; a++)
        for (int b = 0; b < nc; b++) {
            const int ax = L.cx[corners[a]], ay = L.cy[corners[a]], bx = L.cx[corners[b]], by = L.cy[corners[b]];
            const int dx = (ax > bx) - (ax < bx), dy = (ay > by) - (ay < by);
            if ((dx == 0 && dy == 0) || (dx != 0 && dy != 0)) continue;
            bool ok = true;
            if (dx != 0) {
                for (int x = bx + dx; x != ax; x += dx) {
                    const int p = by * w + x;
                    if (sok_bit(L.targetmask, p) || sok_bit(L.solid, p) || (!sok_bit(L.solid, p - w) && !sok_bit(L.solid, p + w))) { ok = false; break; }
                }
                if (ok) for (int x = bx + dx; x != ax; x += dx) sok_set(L.dead, by * w + x);
            } else {
                for (int y = by + dy; y != ay; y += dy) {
                    const int p = y * w + bx;
                    if (sok_bit(L.targetmask, p) || sok_bit(L.solid, p) || (!sok_bit(L.solid, p - 1) && !sok_bit(L.solid, p + 1))) { ok = false; break; }
                }
                if (ok) for (int y = by + dy; y != ay; y += dy) sok_set(L.dead, y * w + bx);
            }
        }
}

// --- CPython heapq on packed entries (priority << 16 | node index); only `<` on priorities ---------
// The heap/table pointer types are template parameters so that, once inlined, the compiler knows the
// address space (LDS vs global) and emits ds_* / global_* instead of flat accesses.
// priority(a) < priority(b) on packed words (priority << 16 | payload): a < (b with its payload cleared) -- one AND, one compare
PCGRL_D bool sok_lt(uint32_t a, uint32_t b) { return a < (b & 0xFFFF0000u); }
template <class HP>
PCGRL_D void sok_siftdown(HP heap, int startpos, int pos) {
    const uint32_t newitem = heap[pos];
    while (pos > startpos) {
        const int parentpos = (pos - 1) >> 1;
        const uint32_t parent = heap[parentpos];
        if (sok_lt(newitem, parent)) { heap[pos] = parent; pos = parentpos; continue; }
        break;
    }
    heap[pos] = newitem;
}
template <class HP>
PCGRL_D void sok_siftup(HP heap, int pos, int endpos) {
    const int startpos = pos;
    const uint32_t newitem = heap[pos];
    int childpos = 2 * pos + 1;
    while (childpos < endpos) {
        const int rightpos = childpos + 1;
        uint32_t c = heap[childpos];
        if (rightpos < endpos) {
            const uint32_t r = heap[rightpos];
            if (!sok_lt(c, r)) { childpos = rightpos; c = r; }
        }
        heap[pos] = c;
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    heap[pos] = newitem;
    sok_siftdown(heap, startpos, pos);
}

PCGRL_D uint32_t sok_hash(const SokLevel& L, const SokNode& n) {
    uint32_t hsh = 2166136261u;
    hsh = (hsh ^ n.player) * 16777619u;
    for (int i = 0; i < L.nc; i++) hsh = (hsh ^ n.crate[i]) * 16777619u;
    return hsh ^ (hsh >> 15);
}
PCGRL_D bool sok_same(const SokLevel& L, const SokNode& a, const SokNode& b) {
    if (a.player != b.player) return false;
    for (int i = 0; i < L.nc; i++) if (a.crate[i] != b.crate[i]) return false;
    return true;
}

// One search (one lane).  k < 0: BFSAgent, else AStarAgent with integer weight k in {2,1,0}.
// `L` and the node workspace `w` should live in LDS on the device (dynamic indexing of per-lane structs
// would otherwise go to scratch memory).  Children are built in place in `w` (a child differs from its
// parent in the player cell and at most one crate) and undone after being written to the pool.
// Returns win; out_h/out_depth describe the returned node (winner, or best node).
// `out_exhausted` reports that the search ended because the queue ran empty (every reachable state was
// expanded), not because of the iteration cap.
// `hook(iterations)` is called at the top of every iteration; returning true abandons the search (the
// caller knows its result is not needed: k_sokoban runs the agents of one level concurrently).
// The node that will be popped next is fetched from the pool one iteration ahead whenever it is already
// known (BFS: the next queue entry; A*: the heap top after the repair), which takes the global-memory
// latency of the pop off the serial chain.
struct SokNoHook { PCGRL_D bool operator()(int) const { return false; } };
template <class HP, class TP, class Hook>
PCGRL_D bool sok_search(const SokLevel& L, SokNode* pool, HP heap, TP table, int table_mask, SokNode& w,
                        const SokNode& root, int k, int power, int& out_h, int& out_depth, int& out_iters,
                        bool& out_exhausted, Hook hook) {
    int npool = 0, head = 0, heapn = 0, iterations = 0, best = -1, best_h = 0, best_depth = 0;
    SokRaw ahead = sok_load(&root);
    int ahead_idx = 0;
    bool aborted = false;
    sok_store(pool, sok_load(&root));
    npool = 1;
    if (k >= 0) { heap[0] = ((uint32_t)(2 * root.h + k * root.depth) << 16) | 0u; heapn = 1; }
    bool win = false;
    int result_h = root.h, result_depth = 0;
    while (iterations < power && (k >= 0 ? heapn > 0 : head < npool)) {
        iterations++;
        if (hook(iterations)) { aborted = true; break; }
        int cur;
        if (k >= 0) {
            const uint32_t last = heap[--heapn];
            cur = (int)((heapn > 0 ? heap[0] : last) & 0xFFFFu);
            SokRaw fetched = ahead;
            if (cur != ahead_idx) fetched = sok_load(pool + cur);   // global load in flight while the heap is repaired
            if (heapn > 0) { heap[0] = last; sok_siftup(heap, 0, heapn); }
            sok_store(&w, fetched);
            ahead_idx = -1;
            if (heapn > 0) { ahead_idx = (int)(heap[0] & 0xFFFFu); ahead = sok_load(pool + ahead_idx); }
        } else {
            cur = head++;
            SokRaw fetched = ahead;
            if (cur != ahead_idx) fetched = sok_load(pool + cur);
            sok_store(&w, fetched);
            ahead_idx = -1;
            if (head < npool) { ahead_idx = head; ahead = sok_load(pool + head); }
        }
        const int node_h = w.h, node_depth = w.depth, node_player = w.player;
        if (sok_win(L, w.crate)) { win = true; result_h = node_h; result_depth = node_depth; break; }
        // visited test-and-add (open addressing; slot = node index + 1, low 16 bits; hash tag in the high bits)
        const uint32_t hs = sok_hash(L, w);
        uint32_t slot = hs & (uint32_t)table_mask;
        const uint32_t tag = (hs >> 16) << 16;
        bool seen = false;
        for (;;) {
            const uint32_t v = table[slot];
            if (v == 0) break;
            if ((v & 0xFFFF0000u) == tag && sok_same(L, pool[(v & 0xFFFFu) - 1], w)) { seen = true; break; }
            slot = (slot + 1) & (uint32_t)table_mask;
        }
        if (seen) continue;
        table[slot] = tag | (uint32_t)(cur + 1);
        if (best < 0 || node_h < best_h || (node_h == best_h && node_depth < best_depth)) { best = cur; best_h = node_h; best_depth = node_depth; }
        w.depth = (uint16_t)(node_depth + 1);
        for (int d = 0; d < 4; d++) {          // Node.getChildren: L, R, U, D
            // State.update engine.py:298-327 (the popped node is never a win)
            const int np = node_player + L.dirs[d];
            if (sok_bit(L.solid, np)) continue;                 // player did not move
            const int c = sok_crate_at(L, w.crate, np);
            int cp = 0;
            if (c >= 0) {
                cp = np + L.dirs[d];
                if (sok_bit(L.solid, cp) || sok_crate_at(L, w.crate, cp) >= 0) continue;   // blocked crate: no move
            }
            w.player = (uint8_t)np;
            bool keep = true;
            if (c >= 0) {
                w.crate[c] = (uint8_t)cp;
                bool deadlock = false;         // checkDeadlock looks at every crate
                for (int i = 0; i < L.nc; i++) deadlock = deadlock || sok_bit(L.dead, w.crate[i]);
                keep = !deadlock;
                if (keep) w.h = (uint16_t)sok_heuristic(L, w.crate);
            }
            if (keep) {
                sok_store(pool + npool, sok_load(&w));
                if (k >= 0) {
                    heap[heapn] = ((uint32_t)(2 * w.h + k * w.depth) << 16) | (uint32_t)npool;
                    heapn++;
                    sok_siftdown(heap, 0, heapn - 1);
                }
                npool++;
            }
            w.player = (uint8_t)node_player;   // undo
            if (c >= 0) { w.crate[c] = (uint8_t)np; w.h = (uint16_t)node_h; }
        }
    }
    if (!win) { result_h = best_h; result_depth = best_depth; }
    out_h = result_h; out_depth = result_depth; out_iters = iterations;
    out_exhausted = !win && !aborted && !(k >= 0 ? heapn > 0 : head < npool);
    return win;
}
template <class HP, class TP>
PCGRL_D bool sok_search(const SokLevel& L, SokNode* pool, HP heap, TP table, int table_mask, SokNode& w,
                        const SokNode& root, int k, int power, int& out_h, int& out_depth, int& out_iters,
                        bool& out_exhausted) {
    return sok_search(L, pool, heap, table, table_mask, w, root, k, power, out_h, out_depth, out_iters, out_exhausted, SokNoHook());
}

// SokobanProblem._run_game (sokoban_prob.py:104-122): BFS, then A* with balance 1, 0.5, 0; first winner gives
// (0, depth), otherwise (heuristic of the last agent's best node, 0).
//
// Exact shortcut: if BFS ends because its queue ran empty without a win, every reachable state was
// expanded and none of them wins.  Each A* run would expand exactly the same states (same children, same
// deadlock pruning), pop exactly as many entries as BFS did (so it cannot hit the cap either), find no win,
// and end with bestNode = a state of minimum heuristic -- the value _run_game returns is that minimum,
// which BFS has already computed.  The three A* runs are skipped in that case.  `clear_table(size)` zeroes
// the visited table before each agent.
template <class HP, class TP, class ClearFn>
PCGRL_D void sok_run_game(const SokLevel& L, SokNode* pool, HP heap, TP table, int table_size, SokNode& w,
                          const SokNode& root, int power, bool allow_shortcut, ClearFn clear_table,
                          int& dist_win, int& sol_len, int* iters) {
    const int KS[4] = {-1, 2, 1, 0};
    bool win = false;
    int hh = 0, dd = 0;
    for (int a = 0; a < 4; a++) iters[a] = 0;
    for (int a = 0; a < 4 && !win; a++) {
        clear_table(table_size);
        bool exhausted = false;
        win = sok_search(L, pool, heap, table, table_size - 1, w, root, KS[a], power, hh, dd, iters[a], exhausted);
        if (a == 0 && !win && exhausted && allow_shortcut) break;
    }
    dist_win = win ? 0 : hh;
    sol_len = win ? dd : 0;
}
