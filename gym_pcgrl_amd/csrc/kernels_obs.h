// The wrapped observation: the image the reference's composite wrappers hand to the policy (gym_pcgrl/wrappers.py) --
// Cropped.transform :197-206 (pad with the border tile, window of `size` centred on the cursor), OneHotEncoding.transform
// :101-104, ToImage.transform :53-60 -- as one uint8 tensor [N][oh][ow][depth] (depth 1 = tile ids, depth T = one-hot).
// Part of the single translation unit pcgrl_abi.hip.
//
// This is the one genuinely HBM-bound piece of the trainer-shaped step (zelda, 22 x 22 x 8 window: 254 MB written per step of
// 65 536 environments), so it is written as a store stream: a block owns the images of consecutive environments -- one
// contiguous stretch of the output -- and every thread produces 16-byte pieces of it in registers and stores them as
// dwordx4, a wavefront covering 1 KB of consecutive addresses per store instruction.  Nothing of the image is staged: the
// tile of an output cell comes from the environment's row bit planes (k_step: its LDS copy of the block's state, so the
// fused step writes the observation without reading the byte map at all; binary maps: a whole output row is one shifted
// plane word and 16 cells expand to 16 bytes with four multiplies) or from the byte map (cached narrow loads).
//   obs_write_block   the device routine (k_step calls it at the end of the launch; k_obs is a kernel around it)
//   k_obs             the stand-alone kernel: pcgrl_reset, pcgrl_set_maps, pcgrl_observe and the steps of the configurations
//                     that do not run the fused step kernel
#pragma once

struct ObsView {          // one block's view: environments [0, ne) relative to the block's first one
    uint8_t* out;         // the block's stretch of the output: image of its first environment (16-byte aligned)
    int oh, ow, depth, centered, pad, W, H;
};

// exact floor(i / d) while i * d < 2^32 (cell index / row length, byte / depth: a few thousand at most):  __umulhi(i, obs_magic(d))
__device__ __forceinline__ uint32_t obs_magic(int d) { return 0xFFFFFFFFu / (uint32_t)d + 1u; }
// exact floor(o / d) for 0 <= o < 2^24 (byte offset in a block's stretch / bytes per image): float estimate, corrected
__device__ __forceinline__ int obs_div(int o, int d, float inv) {
    int q = (int)((float)o * inv);
    q -= (q * d > o) ? 1 : 0;
    q += ((q + 1) * d <= o) ? 1 : 0;
    return q;
}

// Tile source: row bit planes [env][G rows][NPL planes] of MaskT (global memory or the LDS copy of k_step)
template <class MaskT, int NPL>
struct ObsPlanes {
    const MaskT* pl; int G;
    static constexpr bool kRows = NPL == 1;
    __device__ __forceinline__ int tile(int e, int y, int x) const {
        const MaskT* p = pl + ((size_t)e * G + y) * NPL;
        int t = (int)((p[0] >> x) & 1);
        if (NPL > 1) t |= ((int)((p[1] >> x) & 1) << 1) | ((int)((p[2] >> x) & 1) << 2);
        return t;
    }
    __device__ __forceinline__ uint64_t row(int e, int y) const { return (uint64_t)pl[((size_t)e * G + y) * NPL]; }
};
// Tile source: the byte map [env][H][W]
struct ObsBytes {
    const uint8_t* map; int W, H;
    static constexpr bool kRows = false;
    __device__ __forceinline__ int tile(int e, int y, int x) const { return (int)map[((size_t)e * H + y) * W + x]; }
    __device__ __forceinline__ uint64_t row(int, int) const { return 0; }
};

// bit c (c < ow <= 64) = tile of output cell (r, c) of a one-plane (binary) map whose window starts at (oy, ox)
template <class Src>
__device__ __forceinline__ uint64_t obs_row_bits(const Src& src, const ObsView& V, int e, int r, int oy, int ox) {
    const uint64_t all = V.ow >= 64 ? ~0ull : ((1ull << V.ow) - 1ull);
    const uint64_t padbits = V.pad ? all : 0ull;
    const int y = r + oy;
    if ((unsigned)y >= (unsigned)V.H) return padbits;
    const uint64_t m = src.row(e, y);
    const uint64_t in = V.W >= 64 ? ~0ull : ((1ull << V.W) - 1ull);
    uint64_t bits, valid;
    if (ox >= 0) { bits = ox < 64 ? (m & in) >> ox : 0ull; valid = ox < 64 ? in >> ox : 0ull; }
    else { bits = ox > -64 ? (m & in) << -ox : 0ull; valid = ox > -64 ? in << -ox : 0ull; }
    valid &= all;
    return (bits & valid) | (padbits & ~valid);
}

template <class Src>
__device__ __forceinline__ int obs_cell(const Src& src, const ObsView& V, int e, int i, uint32_t mg_ow, int oy, int ox) {
    const int r = (int)__umulhi((uint32_t)i, mg_ow), c = i - r * V.ow;
    const int y = r + oy, x = c + ox;
    if ((unsigned)y >= (unsigned)V.H || (unsigned)x >= (unsigned)V.W) return V.pad;
    return src.tile(e, y, x);
}
// one byte of an environment's image (the pieces that straddle two images, and the tail of a partial block)
template <class Src>
__device__ __forceinline__ uint32_t obs_byte(const Src& src, const ObsView& V, const uint8_t* pos, int e, int rem, uint32_t mg_ow, uint32_t mg_d) {
    const int oy = V.centered ? (int)pos[2 * e + 1] - V.oh / 2 : 0, ox = V.centered ? (int)pos[2 * e] - V.ow / 2 : 0;
    const int i = V.depth == 1 ? rem : (int)__umulhi((uint32_t)rem, mg_d), d = rem - i * V.depth;
    const int t = obs_cell(src, V, e, i, mg_ow, oy, ox);
    return V.depth == 1 ? (uint32_t)t : (uint32_t)(d == t);
}

// The images of environments [0, ne) of a block, written by the `nthreads` threads that call this (tid = 0 .. nthreads-1).
// pos: the block's cursors [ne][2] (x, y), read only when the window is centred.
template <class Src>
__device__ __forceinline__ void obs_write_block(const Src& src, const ObsView& V, const uint8_t* pos, int ne, int tid, int nthreads) {
    const int per_env = V.oh * V.ow * V.depth;
    const int total = ne * per_env;
    const uint32_t mg_ow = obs_magic(V.ow), mg_d = obs_magic(V.depth);
    const float inv_env = 1.0f / (float)per_env;
    const bool rows = Src::kRows && V.depth == 1 && V.ow <= 64 && V.pad <= 1;
    uint4* out4 = reinterpret_cast<uint4*>(V.out);
    for (int q = tid; q < (total >> 4); q += nthreads) {
        const int o = q << 4;
        const int e = obs_div(o, per_env, inv_env), rem = o - e * per_env;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (rem + 16 <= per_env) {
            const int oy = V.centered ? (int)pos[2 * e + 1] - V.oh / 2 : 0, ox = V.centered ? (int)pos[2 * e] - V.ow / 2 : 0;
            if (rows) {
                // sixteen cells = sixteen bits taken from one to three consecutive output rows
                int r = (int)__umulhi((uint32_t)rem, mg_ow), c0 = rem - r * V.ow, filled = 0;
                uint32_t acc = 0;
                while (filled < 16) {
                    const uint64_t bits = obs_row_bits(src, V, e, r, oy, ox) >> c0;
                    const int n = (V.ow - c0) < (16 - filled) ? (V.ow - c0) : (16 - filled);
                    acc |= ((uint32_t)bits & ((1u << n) - 1u)) << filled;
                    filled += n; r++; c0 = 0;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) w[k] = (((acc >> (4 * k)) & 15u) * 0x00204081u) & 0x01010101u;   // bit j -> byte j
            } else if (V.depth == 8) {
                const int i = rem >> 3;
                const uint64_t a = 1ull << (8 * obs_cell(src, V, e, i, mg_ow, oy, ox));
                const uint64_t b = 1ull << (8 * obs_cell(src, V, e, i + 1, mg_ow, oy, ox));
                w[0] = (uint32_t)a; w[1] = (uint32_t)(a >> 32); w[2] = (uint32_t)b; w[3] = (uint32_t)(b >> 32);
            } else if (V.depth == 1) {
#pragma unroll
                for (int k = 0; k < 16; k++) w[k >> 2] |= (uint32_t)obs_cell(src, V, e, rem + k, mg_ow, oy, ox) << (8 * (k & 3));
            } else {
                int i = (int)__umulhi((uint32_t)rem, mg_d), d = rem - i * V.depth;
                int t = obs_cell(src, V, e, i, mg_ow, oy, ox);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    w[k >> 2] |= (uint32_t)(d == t) << (8 * (k & 3));
                    if (++d == V.depth) { d = 0; ++i; t = obs_cell(src, V, e, i, mg_ow, oy, ox); }
                }
            }
        } else {      // the piece straddles two images
            int ee = e, rr = rem;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                w[k >> 2] |= obs_byte(src, V, pos, ee, rr, mg_ow, mg_d) << (8 * (k & 3));
                if (++rr == per_env) { rr = 0; ++ee; }
            }
        }
        out4[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    for (int o = (total & ~15) + tid; o < total; o += nthreads) {     // tail of a partial last block
        const int e = obs_div(o, per_env, inv_env), rem = o - e * per_env;
        V.out[o] = (uint8_t)obs_byte(src, V, pos, e, rem, mg_ow, mg_d);
    }
}

// The observation target of a handle (pcgrl_bind_observation): part of DevBufs.
__device__ __forceinline__ ObsView obs_view(const PcgrlParams& P, const ObsSpec& S, int e0) {
    ObsView V;
    V.out = S.out + (size_t)e0 * S.oh * S.ow * S.depth;
    V.oh = S.oh; V.ow = S.ow; V.depth = S.depth; V.centered = S.centered; V.pad = S.pad; V.W = P.width; V.H = P.height;
    return V;
}

#define OBS_EPB 64       /* environments per block of k_obs (a multiple of 16: every block's stretch starts 16-byte aligned) */
// SRC 0: byte map; 1: one u32 plane (binary); 2: one u64 plane (binary, wide or tall maps)
template <int SRC>
__global__ __launch_bounds__(256) void k_obs(PcgrlParams P, DevBufs B, ObsSpec S) {
    const int e0 = blockIdx.x * OBS_EPB;
    const int ne = (P.num_envs - e0) < OBS_EPB ? (P.num_envs - e0) : OBS_EPB;
    const ObsView V = obs_view(P, S, e0);
    const uint8_t* pos = B.pos + (size_t)e0 * 2;
    if (SRC == 0) {
        const ObsBytes src = {B.map + (size_t)e0 * P.width * P.height, P.width, P.height};
        obs_write_block(src, V, pos, ne, (int)threadIdx.x, 256);
    } else if (SRC == 1) {
        const ObsPlanes<uint32_t, 1> src = {reinterpret_cast<const uint32_t*>(B.planes) + (size_t)e0 * P.group, P.group};
        obs_write_block(src, V, pos, ne, (int)threadIdx.x, 256);
    } else {
        const ObsPlanes<uint64_t, 1> src = {reinterpret_cast<const uint64_t*>(B.planes) + (size_t)e0 * P.group, P.group};
        obs_write_block(src, V, pos, ne, (int)threadIdx.x, 256);
    }
}
