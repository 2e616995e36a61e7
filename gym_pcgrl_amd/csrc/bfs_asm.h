// The BFS level loop of bfs_levels (pcgrl_algos.h, the kHistBfs form) written out for gfx950, for the device lane groups
// (lanegroup_dev.h): pairs of levels until no lane of the WAVEFRONT changes any more or the history word is full (`it`, counted
// up by 2 per pair, reaches a multiple of 32).  Returns whether the last pair still changed something.
//
//   32-bit masks: a level is  t = n | n<<1;  u = n>>1;  t |= up(n);  t |= down(n)  (the row moves folded into v_or_b32_dpp);
//                 n' = (t | u) & pass  (v_bitop3);  compare;  [the copy of the set before the lane's last change;]  the carry of the
//                 compare into the history word (v_addc: hist = 2 hist + changed) -- 8 (7) vector instructions where the compiler's
//                 rendering of the plain C++ has 11.
//   64-bit masks: the same on the two halves (v_alignbit for the bits that cross), 16 (14) instructions of 32 bits where the compiler
//                 has 18 with three 64-bit ones, and two independent chains instead of one.
//   UP / DN: row_shr:1 / row_shl:1 for 16-row groups (four maps per wavefront), wave_shr:1 / wave_shl:1 for a map per wavefront.
//   (The wave form is not used: on the 64-row maps of C5 it measured no faster than the compiler's four-levels-per-test loop --
//   43.5 / 49.7 us against 43.1 / 49.5 -- so DevGroup<64> keeps kHistBfs = 0.)
//
// The DPP reads of a register come at least two instructions after the VALU write of it, as gfx9 requires: the compiler cannot see
// inside the blocks.  (Numeric labels: the blocks are inlined many times per kernel.)
#pragma once
#define PCGRL_DPP_TAIL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define PCGRL_BFS_L32(SRC, DST, UP, DN) \
    "v_lshl_or_b32 %[t], " SRC ", 1, " SRC "\n\t" \
    "v_lshrrev_b32 %[u], 1, " SRC "\n\t" \
    "v_or_b32_dpp %[t], " SRC ", %[t] " UP PCGRL_DPP_TAIL \
    "v_or_b32_dpp %[t], " SRC ", %[t] " DN PCGRL_DPP_TAIL \
    "v_bitop3_b32 " DST ", %[t], %[p], %[u] bitop3:0xc8\n\t" \
    "v_cmp_ne_u32 vcc, " DST ", " SRC "\n\t"
#define PCGRL_BFS_L64(SL, SH, DL, DH, UP, DN) \
    "v_lshl_or_b32 %[tl], " SL ", 1, " SL "\n\t" \
    "v_alignbit_b32 %[th], " SH ", " SL ", 31\n\t" \
    "v_alignbit_b32 %[ul], " SH ", " SL ", 1\n\t" \
    "v_lshrrev_b32 %[uh], 1, " SH "\n\t" \
    "v_or_b32_dpp %[tl], " SL ", %[tl] " UP PCGRL_DPP_TAIL \
    "v_or_b32_dpp %[th], " SH ", %[th] " UP PCGRL_DPP_TAIL \
    "v_or_b32_dpp %[tl], " SL ", %[tl] " DN PCGRL_DPP_TAIL \
    "v_or_b32_dpp %[th], " SH ", %[th] " DN PCGRL_DPP_TAIL \
    "v_or3_b32 %[th], %[th], " SH ", %[uh]\n\t" \
    "v_bitop3_b32 " DL ", %[tl], %[pl], %[ul] bitop3:0xc8\n\t" \
    "v_and_b32 " DH ", %[th], %[ph]\n\t" \
    "v_xor_b32 %[uh], " DH ", " SH "\n\t" \
    "v_bitop3_b32 %[uh], " DL ", " SL ", %[uh] bitop3:0xbe\n\t" \
    "v_cmp_ne_u32 vcc, 0, %[uh]\n\t"
#define PCGRL_BFS_TAIL_N(N) \
    "v_addc_co_u32_e64 %[h], %[c], %[h], %[h], vcc\n\t" \
    "s_add_i32 %[it], %[it], " N "\n\t" \
    "s_cbranch_vccz 2f\n\t" \
    "s_and_b32 %[tmp], %[it], 31\n\t" \
    "s_cbranch_scc1 1b\n" \
    "2:\n\t" \
    "s_mov_b64 %[m], vcc"
#define PCGRL_BFS_TAIL PCGRL_BFS_TAIL_N("2")
#define PCGRL_BFS_RUN32(UP, DN) do { \
    if (WANT_LAST) \
        asm volatile("1:\n\t" \
                     PCGRL_BFS_L32("%[n]", "%[a]", UP, DN) \
                     "v_cndmask_b32 %[prev], %[prev], %[n], vcc\n\t" \
                     "v_addc_co_u32 %[h], vcc, %[h], %[h], vcc\n\t" \
                     PCGRL_BFS_L32("%[a]", "%[n]", UP, DN) \
                     "v_cndmask_b32 %[prev], %[prev], %[a], vcc\n\t" \
                     PCGRL_BFS_TAIL \
                     : [n] "+v"(n), [h] "+v"(hist), [prev] "+v"(prev), [it] "+s"(it), [t] "=&v"(t), [u] "=&v"(u), [a] "=&v"(a), [m] "=s"(m), [c] "=&s"(c), [tmp] "=&s"(tmp) \
                     : [p] "v"(pass) : "vcc", "scc"); \
    else \
        asm volatile("1:\n\t" \
                     PCGRL_BFS_L32("%[n]", "%[a]", UP, DN) \
                     "v_addc_co_u32 %[h], vcc, %[h], %[h], vcc\n\t" \
                     PCGRL_BFS_L32("%[a]", "%[n]", UP, DN) \
                     PCGRL_BFS_TAIL \
                     : [n] "+v"(n), [h] "+v"(hist), [it] "+s"(it), [t] "=&v"(t), [u] "=&v"(u), [a] "=&v"(a), [m] "=s"(m), [c] "=&s"(c), [tmp] "=&s"(tmp) \
                     : [p] "v"(pass) : "vcc", "scc"); \
    } while (0)
#define PCGRL_BFS_RUN64(UP, DN) do { \
    if (WANT_LAST) \
        asm volatile("1:\n\t" \
                     PCGRL_BFS_L64("%[nl]", "%[nh]", "%[al]", "%[ah]", UP, DN) \
                     "v_cndmask_b32 %[prl], %[prl], %[nl], vcc\n\t" \
                     "v_cndmask_b32 %[prh], %[prh], %[nh], vcc\n\t" \
                     "v_addc_co_u32 %[h], vcc, %[h], %[h], vcc\n\t" \
                     PCGRL_BFS_L64("%[al]", "%[ah]", "%[nl]", "%[nh]", UP, DN) \
                     "v_cndmask_b32 %[prl], %[prl], %[al], vcc\n\t" \
                     "v_cndmask_b32 %[prh], %[prh], %[ah], vcc\n\t" \
                     PCGRL_BFS_TAIL \
                     : [nl] "+v"(nl), [nh] "+v"(nh), [h] "+v"(hist), [prl] "+v"(prl), [prh] "+v"(prh), [it] "+s"(it), [tl] "=&v"(tl), [th] "=&v"(th), [ul] "=&v"(ul), \
                       [uh] "=&v"(uh), [al] "=&v"(al), [ah] "=&v"(ah), [m] "=s"(m), [c] "=&s"(c), [tmp] "=&s"(tmp) \
                     : [pl] "v"(pl), [ph] "v"(ph) : "vcc", "scc"); \
    else \
        asm volatile("1:\n\t" \
                     PCGRL_BFS_L64("%[nl]", "%[nh]", "%[al]", "%[ah]", UP, DN) \
                     "v_addc_co_u32 %[h], vcc, %[h], %[h], vcc\n\t" \
                     PCGRL_BFS_L64("%[al]", "%[ah]", "%[nl]", "%[nh]", UP, DN) \
                     PCGRL_BFS_TAIL \
                     : [nl] "+v"(nl), [nh] "+v"(nh), [h] "+v"(hist), [it] "+s"(it), [tl] "=&v"(tl), [th] "=&v"(th), [ul] "=&v"(ul), \
                       [uh] "=&v"(uh), [al] "=&v"(al), [ah] "=&v"(ah), [m] "=s"(m), [c] "=&s"(c), [tmp] "=&s"(tmp) \
                     : [pl] "v"(pl), [ph] "v"(ph) : "vcc", "scc"); \
    } while (0)

// WAVE: a map per wavefront (wave_shr / wave_shl), else four maps per wavefront (row_shr / row_shl)
template <bool WANT_LAST, bool WAVE>
__device__ __forceinline__ bool pcg_bfs_run(uint32_t& n, uint32_t pass, int& hist, uint32_t& prev, int& it) {
    uint32_t t, u, a;
    uint64_t m, c;
    int tmp;
    if (WAVE) PCGRL_BFS_RUN32("wave_shr:1", "wave_shl:1");
    else PCGRL_BFS_RUN32("row_shr:1", "row_shl:1");
    return m != 0;
}
template <bool WANT_LAST, bool WAVE>
__device__ __forceinline__ bool pcg_bfs_run(uint64_t& n, uint64_t pass, int& hist, uint64_t& prev, int& it) {
    uint32_t nl = (uint32_t)n, nh = (uint32_t)(n >> 32), prl = (uint32_t)prev, prh = (uint32_t)(prev >> 32);
    const uint32_t pl = (uint32_t)pass, ph = (uint32_t)(pass >> 32);
    uint32_t tl, th, ul, uh, al, ah;
    uint64_t m, c;
    int tmp;
    if (WAVE) PCGRL_BFS_RUN64("wave_shr:1", "wave_shl:1");
    else PCGRL_BFS_RUN64("row_shr:1", "row_shl:1");
    n = ((uint64_t)nh << 32) | nl;
    if (WANT_LAST) prev = ((uint64_t)prh << 32) | prl;
    return m != 0;
}
