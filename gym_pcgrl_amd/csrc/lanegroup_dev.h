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

template <int G, class MaskT>
struct DevGroup;

template <class MaskT>
struct DevGroup<16, MaskT> {
    typedef MaskT mask_t;
    enum { kGroup = 16 };
    int lane;   // row index inside the group
    int shift;  // bit offset of this group inside the wave ballot
    __device__ __forceinline__ DevGroup() {
        int l = (int)(threadIdx.x & 63);
        lane = l & 15;
        shift = l & 48;
    }
    __device__ __forceinline__ mask_t up(mask_t m) const { return dpp_mov0<0x111>(m); }    // row_shr:1
    __device__ __forceinline__ mask_t down(mask_t m) const { return dpp_mov0<0x101>(m); }  // row_shl:1
    __device__ __forceinline__ uint32_t ballot(bool p) const {
        return (uint32_t)(__ballot(p) >> shift) & 0xFFFFu;
    }
    __device__ __forceinline__ bool any(mask_t m) const { return ballot(m != 0) != 0; }
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
    __device__ __forceinline__ int popcount_sum(mask_t m) const { return sum(pcg_popc(m)); }
};

template <class MaskT>
struct DevGroup<64, MaskT> {
    typedef MaskT mask_t;
    enum { kGroup = 64 };
    int lane;
    __device__ __forceinline__ DevGroup() { lane = (int)(threadIdx.x & 63); }
    __device__ __forceinline__ mask_t up(mask_t m) const { return dpp_mov0<0x138>(m); }    // wave_shr:1
    __device__ __forceinline__ mask_t down(mask_t m) const { return dpp_mov0<0x130>(m); }  // wave_shl:1
    __device__ __forceinline__ bool any(mask_t m) const { return __ballot(m != 0) != 0; }
    __device__ __forceinline__ bool any_ne(mask_t a, mask_t b) const { return __ballot(a != b) != 0; }
    __device__ __forceinline__ mask_t first_bit(mask_t m) const {
        uint64_t b = __ballot(m != 0);
        int first = __ffsll((unsigned long long)b) - 1;
        return lane == first ? (m & (mask_t)(0 - m)) : (mask_t)0;
    }
    __device__ __forceinline__ int sum(int v) const {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
    __device__ __forceinline__ int popcount_sum(mask_t m) const { return sum(pcg_popc(m)); }
};
