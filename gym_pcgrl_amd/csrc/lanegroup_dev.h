// Device backend for pcgrl_algos.h: a "lane group" is G consecutive lanes of a wavefront,
// lane r holding map row r.  gfx950 only.
//
//   G = 16 : one DPP row per map, 4 maps per wavefront.  up/down are `row_shr:1` / `row_shl:1`
//            (zero fill at the row edge comes for free from bound_ctrl).
//   G = 64 : one wavefront per map.  up/down are `wave_shr:1` / `wave_shl:1`.
//
// Group-wide predicates go through one `v_cmp` + wave ballot; each group reads its own slice of
// the 64-bit ballot, so four maps with different trip counts can share a wavefront (the loop
// conditions are uniform inside a DPP row; the compiler handles the rest with the exec mask).
#pragma once
#include <hip/hip_runtime.h>
#include "pcgrl_common.h"
#include "bfs_asm.h"

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_mov0(uint64_t v) {
    uint32_t lo = dpp_mov0<CTRL>((uint32_t)v), hi = dpp_mov0<CTRL>((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int pcg_popc(uint32_t v) { return __popc(v); }
__device__ __forceinline__ int pcg_popc(uint64_t v) { return __popcll(v); }
__device__ __forceinline__ uint32_t pcg_brev(uint32_t v) { return __brev(v); }
__device__ __forceinline__ uint64_t pcg_brev(uint64_t v) { return __brevll(v); }

// Lane-local pieces shared by both group sizes.
template <class MaskT>
struct DevLaneOps {
    typedef MaskT mask_t;
    typedef int ivec_t;
    __device__ __forceinline__ ivec_t izero() const { return 0; }
    __device__ __forceinline__ bool wave_any(mask_t m) const { return __ballot(m != 0) != 0; }
    __device__ __forceinline__ ivec_t popc_lanes(mask_t m) const { return pcg_popc(m); }
    __device__ __forceinline__ ivec_t isel_ne(mask_t a, mask_t b, int x, ivec_t y) const { return a != b ? x : y; }
    __device__ __forceinline__ mask_t msel_ne(mask_t a, mask_t b, mask_t x, mask_t y) const { return a != b ? x : y; }
    __device__ __forceinline__ mask_t keep_where_eq(ivec_t v, int x, mask_t m) const { return v == x ? m : (mask_t)0; }
    __device__ __forceinline__ mask_t bitrev(mask_t m) const { return pcg_brev(m); }
};

template <int G, class MaskT>
struct DevGroup;

template <class MaskT>
struct DevGroup<16, MaskT> : DevLaneOps<MaskT> {
    typedef MaskT mask_t;
    enum { kGroup = 16, kLog2Group = 4, kHistBfs = 1 };
    // the level loop of bfs_levels (pcgrl_algos.h, the kHistBfs form), written out: bfs_asm.h
    template <bool WANT_LAST>
    __device__ __forceinline__ bool bfs_run(MaskT& n, MaskT pass, int& hist, MaskT& prev, int& it) const { return pcg_bfs_run<WANT_LAST, false>(n, pass, hist, prev, it); }
    __device__ __forceinline__ int hist_fold(int hist, int it, int last_it) const { return hist != 0 ? it - (int)__builtin_ctz((unsigned)hist) : last_it; }
    // row r receives row r - 2^k (rows_down) / r + 2^k (rows_up); k is a compile-time constant after unrolling
    __device__ __forceinline__ mask_t rows_down(mask_t m, int k) const {
        switch (k) { case 0: return dpp_mov0<0x111>(m); case 1: return dpp_mov0<0x112>(m); case 2: return dpp_mov0<0x114>(m); default: return dpp_mov0<0x118>(m); }
    }
    __device__ __forceinline__ mask_t rows_up(mask_t m, int k) const {
        switch (k) { case 0: return dpp_mov0<0x101>(m); case 1: return dpp_mov0<0x102>(m); case 2: return dpp_mov0<0x104>(m); default: return dpp_mov0<0x108>(m); }
    }
    int lane;   // row index inside the group
    int shift;  // bit offset of this group inside the wave ballot
    __device__ __forceinline__ DevGroup() {
        int l = (int)(threadIdx.x & 63);
        lane = l & 15;
        shift = l & 48;
    }
    __device__ __forceinline__ explicit DevGroup(int lane64) { lane = lane64 & 15; shift = lane64 & 48; }
    __device__ __forceinline__ mask_t up(mask_t m) const { return dpp_mov0<0x111>(m); }    // row_shr:1
    __device__ __forceinline__ mask_t down(mask_t m) const { return dpp_mov0<0x101>(m); }  // row_shl:1
    __device__ __forceinline__ uint32_t ballot(bool p) const {
        return (uint32_t)(__ballot(p) >> shift) & 0xFFFFu;
    }
    __device__ __forceinline__ bool any(mask_t m) const { return ballot(m != 0) != 0; }
    __device__ __forceinline__ mask_t rows_between(mask_t m, int lo, int hi) const { return (lane >= lo && lane < hi) ? m : (mask_t)0; }
    __device__ __forceinline__ int first_row(mask_t m) const { return __ffs((int)ballot(m != 0)) - 1; }
    __device__ __forceinline__ bool any_ne(mask_t a, mask_t b) const { return ballot(a != b) != 0; }
    __device__ __forceinline__ mask_t first_bit(mask_t m) const {
        uint32_t b = ballot(m != 0);
        int first = __ffs((int)b) - 1;
        return lane == first ? (m & (mask_t)(0 - m)) : (mask_t)0;
    }
    __device__ __forceinline__ int sum(int v) const {
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
        return v;
    }
    __device__ __forceinline__ int imax(int v) const {   // values are >= 0, so the 0 fill is neutral
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
        return v;
    }
    __device__ __forceinline__ int popcount_sum(mask_t m) const { return sum(pcg_popc(m)); }
    // several counts through ONE reduction: packed into a word (a group of sixteen 32-bit rows has at most 512 cells, of 64-bit rows 1 024)
    __device__ __forceinline__ void popcount_sum2(mask_t a, mask_t b, int& x, int& y) const {
        const int v = sum(pcg_popc(a) | (pcg_popc(b) << 16));
        x = v & 0xFFFF; y = (int)((unsigned)v >> 16);
    }
    __device__ __forceinline__ void popcount_sum3(mask_t a, mask_t b, mask_t c, int& x, int& y, int& z) const {
        if (sizeof(MaskT) == 4) {
            const int v = sum(pcg_popc(a) | (pcg_popc(b) << 10) | (pcg_popc(c) << 20));
            x = v & 1023; y = (v >> 10) & 1023; z = (int)((unsigned)v >> 20);
        } else { popcount_sum2(a, b, x, y); z = popcount_sum(c); }
    }
};

template <class MaskT>
struct DevGroup<64, MaskT> : DevLaneOps<MaskT> {
    typedef MaskT mask_t;
    // One wavefront per map.  Single-row moves cross the whole wave (wave_shr/wave_shl); the doubling
    // steps of the column fill stay inside a 16-lane DPP row (kLog2Group = 4) and the fill adds one
    // whole-wave hop per round to cross row-block boundaries (pcg_fill_cols).
    enum { kGroup = 64, kLog2Group = 4, kHistBfs = 0 };
    template <bool WANT_LAST>
    __device__ __forceinline__ bool bfs_run(MaskT& n, MaskT pass, int& hist, MaskT& prev, int& it) const { return pcg_bfs_run<WANT_LAST, true>(n, pass, hist, prev, it); }
    __device__ __forceinline__ int hist_fold(int hist, int it, int last_it) const { return hist != 0 ? it - (int)__builtin_ctz((unsigned)hist) : last_it; }
    int lane;
    __device__ __forceinline__ DevGroup() { lane = (int)(threadIdx.x & 63); }
    __device__ __forceinline__ mask_t up(mask_t m) const { return dpp_mov0<0x138>(m); }    // wave_shr:1
    __device__ __forceinline__ mask_t down(mask_t m) const { return dpp_mov0<0x130>(m); }  // wave_shl:1
    __device__ __forceinline__ mask_t rows_down(mask_t m, int k) const {
        switch (k) { case 0: return dpp_mov0<0x111>(m); case 1: return dpp_mov0<0x112>(m); case 2: return dpp_mov0<0x114>(m); default: return dpp_mov0<0x118>(m); }
    }
    __device__ __forceinline__ mask_t rows_up(mask_t m, int k) const {
        switch (k) { case 0: return dpp_mov0<0x101>(m); case 1: return dpp_mov0<0x102>(m); case 2: return dpp_mov0<0x104>(m); default: return dpp_mov0<0x108>(m); }
    }
    __device__ __forceinline__ bool any(mask_t m) const { return __ballot(m != 0) != 0; }
    __device__ __forceinline__ mask_t rows_between(mask_t m, int lo, int hi) const { return (lane >= lo && lane < hi) ? m : (mask_t)0; }
    __device__ __forceinline__ int first_row(mask_t m) const { return __ffsll((unsigned long long)__ballot(m != 0)) - 1; }
    __device__ __forceinline__ bool any_ne(mask_t a, mask_t b) const { return __ballot(a != b) != 0; }
    __device__ __forceinline__ mask_t first_bit(mask_t m) const {
        uint64_t b = __ballot(m != 0);
        int first = __ffsll((unsigned long long)b) - 1;
        return lane == first ? (m & (mask_t)(0 - m)) : (mask_t)0;
    }
    // row-wise all-reduce with DPP, then the four row results are combined through SGPRs
    __device__ __forceinline__ int sum(int v) const {
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
        return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
               __builtin_amdgcn_readlane(v, 48);
    }
    __device__ __forceinline__ int imax(int v) const {   // values are >= 0
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
        v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
        return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                   max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
    }
    __device__ __forceinline__ int popcount_sum(mask_t m) const { return sum(pcg_popc(m)); }
    __device__ __forceinline__ void popcount_sum2(mask_t a, mask_t b, int& x, int& y) const {       // (at most 4 096 cells: sixteen bits each)
        const int v = sum(pcg_popc(a) | (pcg_popc(b) << 16));
        x = v & 0xFFFF; y = (int)((unsigned)v >> 16);
    }
    __device__ __forceinline__ void popcount_sum3(mask_t a, mask_t b, mask_t c, int& x, int& y, int& z) const { popcount_sum2(a, b, x, y); z = popcount_sum(c); }
};
