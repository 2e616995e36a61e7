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
// fused step writes the observation without reading the byte map at all) or from the byte map (cached narrow loads).
// What a piece costs in instructions decides whether the stream reaches the memory system's rate, so the two shapes the
// trainer uses have their own lean routines, selected once per call (kernel-uniform):
//   rows (binary, tile ids)   an output row is one shifted plane word; sixteen cells = sixteen bits of one or two rows,
//                             expanded to bytes with four 24-bit multiplies
//   hot8 (one-hot, 8 tiles)   zelda, mdungeon: two cells per piece, three plane bits each
//   obs_pieces_any            everything else, byte by byte (other depths, images that are not a multiple of 16 bytes, huge windows)
//   obs_write_block           all images of a block's environments; k_step calls it at the end of the launch, k_obs is a kernel
//                             around it
//   k_obs            the stand-alone kernel: pcgrl_reset, pcgrl_set_maps, pcgrl_observe and the steps of the configurations
//                    that do not run the fused step kernel
#pragma once

// (struct ObsView: worklist.h -- one block's view of the target: environments [0, ne) relative to the block's first one)

// exact floor(i / d) while i * d < 2^32:  __umulhi(i, obs_magic(d))   (the general routine; 32-bit multiplies are slow)
__device__ __forceinline__ uint32_t obs_magic(int d) { return 0xFFFFFFFFu / (uint32_t)d + 1u; }
// exact floor(o / d) for 0 <= o < 2^24: float estimate, corrected
__device__ __forceinline__ int obs_div(int o, int d, float inv) {
    int q = (int)((float)o * inv);
    q -= (q * d > o) ? 1 : 0;
    q += ((q + 1) * d <= o) ? 1 : 0;
    return q;
}
// floor(i / d) as three full-rate instructions; exact where the call sites say why ((i + 1/2) / d is at least 1/(2d) away from
// an integer, and the float product is off by less than that)
__device__ __forceinline__ int obs_fdiv(int i, float inv) { return (int)(((float)i + 0.5f) * inv); }

// Tile source: row bit planes [env][G rows][NPL planes] of MaskT (global memory or the LDS copy of k_step)
template <class MaskT, int NPL>
struct ObsPlanes {
    const MaskT* pl; int G;
    static constexpr int kPlanes = NPL;
    static constexpr bool kWord32 = sizeof(MaskT) == 4;
    __device__ __forceinline__ int tile(int e, int y, int x) const {
        const MaskT* p = pl + ((size_t)e * G + y) * NPL;
        int t = (int)((p[0] >> x) & 1);
        if (NPL > 1) t |= ((int)((p[1] >> x) & 1) << 1) | ((int)((p[2] >> x) & 1) << 2);
        return t;
    }
    __device__ __forceinline__ const MaskT* row(int e, int y) const { return pl + (e * G + y) * NPL; }
};
// Tile source: the byte map [env][H][W]
struct ObsBytes {
    const uint8_t* map; int W, H;
    static constexpr int kPlanes = 0;
    static constexpr bool kWord32 = false;
    __device__ __forceinline__ int tile(int e, int y, int x) const { return (int)map[((size_t)e * H + y) * W + x]; }
    __device__ __forceinline__ const uint32_t* row(int, int) const { return nullptr; }
};

// ---- general routine: any depth, any size; `rem` = byte offset inside the image of environment e
template <class Src>
__device__ __forceinline__ int obs_cell(const Src& src, const ObsView& V, int e, int i, uint32_t mg_ow, int oy, int ox) {
    const int r = (int)__umulhi((uint32_t)i, mg_ow), c = i - r * V.ow;
    const int y = r + oy, x = c + ox;
    if ((unsigned)y >= (unsigned)V.H || (unsigned)x >= (unsigned)V.W) return V.pad;
    return src.tile(e, y, x);
}
template <class Src>
__device__ __forceinline__ uint32_t obs_byte(const Src& src, const ObsView& V, const uint8_t* pos, int e, int rem, uint32_t mg_ow, uint32_t mg_d) {
    const int oy = V.centered ? (int)pos[2 * e + 1] - V.oh / 2 : 0, ox = V.centered ? (int)pos[2 * e] - V.ow / 2 : 0;
    const int i = V.depth == 1 ? rem : (int)__umulhi((uint32_t)rem, mg_d), d = rem - i * V.depth;
    const int t = obs_cell(src, V, e, i, mg_ow, oy, ox);
    return V.depth == 1 ? (uint32_t)t : (uint32_t)(d == t);
}
template <class Src>
__device__ __attribute__((noinline)) void obs_pieces_any(Src src, uint8_t* out, int oh, int ow, int depth, int centered, int pad, int W, int H,
                                                           const uint8_t* pos, int ne, int q0, int q1, int tail, int tid, int nthreads) {
    const ObsView V = {out, oh, ow, depth, centered, pad, W, H};
    const int per_env = oh * ow * depth, total = ne * per_env;
    const uint32_t mg_ow = obs_magic(ow), mg_d = obs_magic(depth);
    const float inv_env = 1.0f / (float)per_env;
    for (int q = q0 + tid; q < q1; q += nthreads) {
        const int o = q << 4;
        int e = obs_div(o, per_env, inv_env), rem = o - e * per_env;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            w[k >> 2] |= obs_byte(src, V, pos, e, rem, mg_ow, mg_d) << (8 * (k & 3));
            if (++rem == per_env) { rem = 0; ++e; }
        }
        reinterpret_cast<uint4*>(out)[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (tail) for (int o = (total & ~15) + tid; o < total; o += nthreads) {     // last bytes of a partial block
        const int e = obs_div(o, per_env, inv_env), rem = o - e * per_env;
        out[o] = (uint8_t)obs_byte(src, V, pos, e, rem, mg_ow, mg_d);
    }
}

// ---- binary, tile ids: W <= 32 (64 with 64-bit plane words), 16 <= ow <= 32, oh * ow <= 4096, images a multiple of 16 bytes
// signed shift: m >> s (negative s = left), low 32 bits; s in [-32, 31] for a 32-bit word, in [-32, 63] for a 64-bit one
__device__ __forceinline__ uint32_t obs_shr(uint32_t m, int s) { return (uint32_t)(((uint64_t)m << 32) >> (32 + s)); }
__device__ __forceinline__ uint32_t obs_shr(uint64_t m, int s) { return s >= 0 ? (uint32_t)(m >> s) : (uint32_t)(m << -s); }
template <class Src>
__device__ __forceinline__ uint32_t obs_row32(const Src& src, const ObsView& V, int e, int r, int oy, int ox, uint32_t valid, uint32_t padbits) {
    // bit c = tile of output cell (r, c): the plane word of map row r + oy moved by ox; cells outside the map = pad
    const int y = r + oy;
    const bool in = (unsigned)y < (unsigned)V.H;
    const auto m = src.row(e, in ? y : 0)[0];            // (uint32_t or uint64_t: the map may be wider than the window)
    const uint32_t v = in ? valid : 0u;
    return (obs_shr(m, ox) & v) | (padbits & ~v);
}
// ---- one-hot over eight tiles, three planes: ow <= 64, oh * ow <= 4096 (an image is then always a multiple of 16 bytes when
// its number of cells is even; odd: the general routine)
template <bool INSIDE, class Src>
__device__ __forceinline__ uint2 obs_hot8(const Src& src, const ObsView& V, int e, int i, int oy, int ox, float inv_ow) {
    // (i < 4096: the quotient is below 4096 / ow, the float product off by < 2^-11 / ow... at most 2^-10 for ow = 1; margin 1/(2 ow) >= 1/128)
    const int r = obs_fdiv(i, inv_ow), c = i - (int)__umul24(r, V.ow);
    const int y = r + oy, x = c + ox;
    const bool in = INSIDE || ((unsigned)y < (unsigned)V.H && (unsigned)x < (unsigned)V.W);
    const auto* p = src.row(e, in ? y : 0);
    const int xs = in ? x : 0;
    int t = (int)((p[0] >> xs) & 1) | ((int)((p[1] >> xs) & 1) << 1) | ((int)((p[2] >> xs) & 1) << 2);
    t = in ? t : V.pad;
    const uint32_t bit = 1u << ((t & 3) << 3);
    return make_uint2(t < 4 ? bit : 0u, t < 4 ? 0u : bit);
}
// Which routine an image takes.  0: general, 1: rows (binary ids), 2: one-hot over eight tiles.  (Host and device: the fused
// step kernel only carries the lean routines -- the host sends the other shapes to k_obs.)
__host__ __device__ inline int obs_lean_mode(int nplanes, bool word32, int W, int oh, int ow, int depth, int pad) {
    const int cells = oh * ow, per_env = cells * depth;
    if ((per_env & 15) != 0 || cells > 4096) return 0;
    if (nplanes == 1 && depth == 1 && ow >= 16 && ow <= 32 && W <= (word32 ? 32 : 64) && pad <= 1) return 1;      // (a window row is one 32-bit word)
    if (nplanes == 3 && word32 && depth == 8 && ow <= 64 && pad <= 7) return 2;
    return 0;
}
template <class Src>
__device__ __forceinline__ int obs_mode(const ObsView& V) { return obs_lean_mode(Src::kPlanes, Src::kWord32, V.W, V.oh, V.ow, V.depth, V.pad); }

// The image of environment e with the lean routine of the source (obs_mode != 0), by one wavefront: environment, window origin
// and validity masks are wave-uniform (scalar registers), a lane's work is one piece.
template <class Src>
__device__ __forceinline__ void obs_write_env_lean(const Src& src, const ObsView& V, const uint8_t* pos, int e, int lane) {
    const int ppe = (V.oh * V.ow * V.depth) >> 4;                 // pieces per image (<= 2048: 32 KB)
    const float inv_ow = 1.0f / (float)V.ow;
    int oy = 0, ox = 0;
    if (V.centered) {
        const uint32_t p = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<const uint16_t*>(pos)[e]);
        ox = (int)(p & 255u) - (V.ow >> 1); oy = (int)(p >> 8) - (V.oh >> 1);
    }
    uint4* oe = reinterpret_cast<uint4*>(V.out) + e * ppe;
    if (Src::kPlanes == 1) {
        const uint32_t all = V.ow >= 32 ? ~0u : ((1u << V.ow) - 1u);
        typedef decltype(src.row(0, 0)[0] + 0) WordT;            // uint32_t / uint64_t
        const WordT in = V.W >= (int)(8 * sizeof(WordT)) ? ~(WordT)0 : (((WordT)1 << V.W) - 1);
        const uint32_t valid = obs_shr(in, ox) & all, padbits = V.pad ? all : 0u;
        for (int j = lane; j < ppe; j += 64) {
            // (rem < 4096, ow >= 16: the quotient is below 256 and the float product off by < 2^-15, against a margin of 1/(2 ow) >= 1/64)
            const int rem = j << 4, r0 = obs_fdiv(rem, inv_ow), c0 = rem - (int)__umul24(r0, V.ow);
            const uint32_t a = obs_row32(src, V, e, r0, oy, ox, valid, padbits);
            const uint32_t b = obs_row32(src, V, e, r0 + 1, oy, ox, valid, padbits);       // (past the last row only when c0 + 16 <= ow: not used then)
            const uint32_t acc = (uint32_t)((((uint64_t)b << V.ow) | a) >> c0);
            uint4 w;
            w.x = __umul24(acc & 15u, 0x204081u) & 0x01010101u;                              // bit j -> byte j
            w.y = __umul24((acc >> 4) & 15u, 0x204081u) & 0x01010101u;
            w.z = __umul24((acc >> 8) & 15u, 0x204081u) & 0x01010101u;
            w.w = __umul24((acc >> 12) & 15u, 0x204081u) & 0x01010101u;
            oe[j] = w;
        }
    } else if (Src::kPlanes == 3) {
        if (!V.centered && V.oh == V.H && V.ow == V.W) {       // the map itself: no cell is outside
            for (int j = lane; j < ppe; j += 64) {
                const uint2 a = obs_hot8<true>(src, V, e, 2 * j, 0, 0, inv_ow), b = obs_hot8<true>(src, V, e, 2 * j + 1, 0, 0, inv_ow);
                oe[j] = make_uint4(a.x, a.y, b.x, b.y);
            }
        } else {
            for (int j = lane; j < ppe; j += 64) {
                const uint2 a = obs_hot8<false>(src, V, e, 2 * j, oy, ox, inv_ow), b = obs_hot8<false>(src, V, e, 2 * j + 1, oy, ox, inv_ow);
                oe[j] = make_uint4(a.x, a.y, b.x, b.y);
            }
        }
    }
}
// The piece of a map image (one-hot over eight tiles, three planes; not centred) that holds map cell (x, y) of environment e -- what one
// change of the wide representation leaves to write when the target still holds the previous image.  One lane.
template <class Src>
__device__ __forceinline__ void obs_write_delta_piece(const Src& src, const ObsView& V, int e, int x, int y) {
    x = x < 0 ? 0 : (x > V.W - 1 ? V.W - 1 : x); y = y < 0 ? 0 : (y > V.H - 1 ? V.H - 1 : y);      // (update_env clamps the same way)
    if (x >= V.ow || y >= V.oh) return;                         // (a window smaller than the map)
    const float inv_ow = 1.0f / (float)V.ow;
    const int ppe = (V.oh * V.ow * V.depth) >> 4;
    const int i = (y * V.ow + x) & ~1;                          // first cell of the piece (8 bytes a cell: a piece is two cells of one image)
    const uint2 a = obs_hot8<false>(src, V, e, i, 0, 0, inv_ow), b = obs_hot8<false>(src, V, e, i + 1, 0, 0, inv_ow);
    reinterpret_cast<uint4*>(V.out)[e * ppe + (i >> 1)] = make_uint4(a.x, a.y, b.x, b.y);
}
// k_step's tasks: the block's view out of its StepLocal (LDS) into scalar registers, where a task that writes images needs it
__device__ __forceinline__ ObsView obs_view_from_lds(const ObsView* v) {
    ObsView V;
    const uint64_t o = (uint64_t)v->out;
    V.out = (uint8_t*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)o));
    V.oh = __builtin_amdgcn_readfirstlane(v->oh); V.ow = __builtin_amdgcn_readfirstlane(v->ow); V.depth = __builtin_amdgcn_readfirstlane(v->depth);
    V.centered = __builtin_amdgcn_readfirstlane(v->centered); V.pad = __builtin_amdgcn_readfirstlane(v->pad);
    V.W = __builtin_amdgcn_readfirstlane(v->W); V.H = __builtin_amdgcn_readfirstlane(v->H);
    return V;
}

// The binary tile-id images (kPlanes == 1) of NB environments e0, e0 + stride, ... (< ne) by one wavefront, side by side: an image is a
// chain of dependent LDS round trips (cursor -> window origin -> two plane rows -> the piece) that one wavefront walks at a fraction of
// the rate the stores could leave at -- 65 536 crops of 28 x 28 cost k_step 8 us of wavefront time, not of store bandwidth (round 6:
// moving the images under the tasks of the step did not shorten it) -- so NB chains are walked together: the loads of all of them
// first, and what does not depend on the environment (row and column of a piece) computed once.
template <int NB, class Src>
__device__ __forceinline__ void obs_write_envs_rows(const Src& src, const ObsView& V, const uint8_t* pos, int e0, int stride, int ne, int lane) {
    const int ppe = (V.oh * V.ow) >> 4;
    const float inv_ow = 1.0f / (float)V.ow;
    const uint32_t all = V.ow >= 32 ? ~0u : ((1u << V.ow) - 1u);
    typedef decltype(src.row(0, 0)[0] + 0) WordT;            // uint32_t / uint64_t
    const WordT in = V.W >= (int)(8 * sizeof(WordT)) ? ~(WordT)0 : (((WordT)1 << V.W) - 1);
    const uint32_t padbits = V.pad ? all : 0u;
    uint32_t p[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int e = e0 + b * stride;
        p[b] = (V.centered && e < ne) ? (uint32_t)reinterpret_cast<const uint16_t*>(pos)[e] : 0u;
    }
    int ox[NB], oy[NB];
    uint32_t valid[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const uint32_t q = __builtin_amdgcn_readfirstlane(p[b]);
        ox[b] = V.centered ? (int)(q & 255u) - (V.ow >> 1) : 0; oy[b] = V.centered ? (int)(q >> 8) - (V.oh >> 1) : 0;
        valid[b] = obs_shr(in, ox[b]) & all;
    }
    for (int j = lane; j < ppe; j += 64) {
        const int rem = j << 4, r0 = obs_fdiv(rem, inv_ow), c0 = rem - (int)__umul24(r0, V.ow);     // (see obs_write_env_lean)
        uint32_t a[NB], c[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int e = e0 + b * stride < ne ? e0 + b * stride : e0;
            a[b] = obs_row32(src, V, e, r0, oy[b], ox[b], valid[b], padbits);
            c[b] = obs_row32(src, V, e, r0 + 1, oy[b], ox[b], valid[b], padbits);
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int e = e0 + b * stride;
            if (e >= ne) continue;                            // wave-uniform
            const uint32_t acc = (uint32_t)((((uint64_t)c[b] << V.ow) | a[b]) >> c0);
            uint4 w;
            w.x = __umul24(acc & 15u, 0x204081u) & 0x01010101u;                              // bit j -> byte j
            w.y = __umul24((acc >> 4) & 15u, 0x204081u) & 0x01010101u;
            w.z = __umul24((acc >> 8) & 15u, 0x204081u) & 0x01010101u;
            w.w = __umul24((acc >> 12) & 15u, 0x204081u) & 0x01010101u;
            (reinterpret_cast<uint4*>(V.out) + e * ppe)[j] = w;
        }
    }
}

// k_step's form (lean shapes only; `nthreads` threads = whole wavefronts call it): every image, a wavefront per environment --
// or, `delta` (the wide representation's map image, one-hot over eight tiles; the target still holds the image of the previous
// state): the piece that holds map cell (x, y) of every environment with `changed[e]` (x, y from its action act[e * 3 ..]) and
// the whole image of every environment with `fresh[e]` (its episode ended: a new map).
template <class Src>
__device__ __forceinline__ void obs_write_block_lean(const Src& src, const ObsView& V, const uint8_t* pos, int ne, bool delta, const int32_t* act,
                                                     const uint8_t* changed, const uint8_t* fresh, int tid, int nthreads) {
    const int lane = tid & 63, nw = nthreads >> 6;
    if (delta && Src::kPlanes == 3) {
        for (int e = tid; e < ne; e += nthreads) {
            if (!changed[e] || fresh[e]) continue;
            obs_write_delta_piece(src, V, e, act[3 * e], act[3 * e + 1]);
        }
    }
    if (Src::kPlanes == 1) {        // binary tile ids: four images a wavefront at a time
        for (int e = __builtin_amdgcn_readfirstlane(tid >> 6); e < ne; e += 4 * nw) obs_write_envs_rows<4>(src, V, pos, e, nw, ne, lane);
        return;
    }
    for (int e = __builtin_amdgcn_readfirstlane(tid >> 6); e < ne; e += nw)
        if (!(delta && Src::kPlanes == 3) || fresh[e]) obs_write_env_lean(src, V, pos, e, lane);
}

// The images of a block's environments [0, ne) (one contiguous stretch of ne * oh * ow * depth bytes), written by the `nthreads`
// threads (whole wavefronts) that call this, tid = 0 .. nthreads-1.  pos: the block's cursors [ne][2] (x, y), read only when
// the window is centred.  The lean routines go environment by environment, a wavefront each: environment, window origin
// and validity masks are then wave-uniform (scalar registers), a lane's work is one piece.
template <class Src>
__device__ __forceinline__ void obs_write_block(const Src& src, const ObsView& V, const uint8_t* pos, int ne, int tid, int nthreads) {
    if (obs_mode<Src>(V) == 0) {
        obs_pieces_any<Src>(src, V.out, V.oh, V.ow, V.depth, V.centered, V.pad, V.W, V.H, pos, ne, 0, (ne * V.oh * V.ow * V.depth) >> 4, 1, tid, nthreads);
        return;
    }
    for (int e = __builtin_amdgcn_readfirstlane(tid >> 6); e < ne; e += nthreads >> 6) obs_write_env_lean(src, V, pos, e, tid & 63);
}

// The observation target of a handle (pcgrl_bind_observation): part of DevBufs.
__device__ __forceinline__ ObsView obs_view(const PcgrlParams& P, const ObsSpec& S, int e0) {
    ObsView V;
    V.out = S.out + (size_t)e0 * S.oh * S.ow * S.depth;
    V.oh = S.oh; V.ow = S.ow; V.depth = S.depth; V.centered = S.centered; V.pad = S.pad; V.W = P.width; V.H = P.height;
    return V;
}

#define OBS_EPB 64       /* environments per block of k_obs (a multiple of 16: every block's stretch starts 16-byte aligned) */
// SRC 0: byte map; 1: one u32 plane (binary); 2: one u64 plane (binary, maps wider than 32 columns); 3: three u32 planes (zelda,
// sokoban, mdungeon, ddave on maps of at most 32 columns).  With planes the block first copies the planes and cursors of its
// environments into LDS (one coalesced burst; dynamic LDS: OBS_EPB * (bytes of an environment's planes + 2)): a piece is a chain of
// two or three dependent reads, which must not be trips to memory.
template <int SRC>
__global__ __launch_bounds__(256) void k_obs(PcgrlParams P, DevBufs B, ObsSpec S) {
    extern __shared__ __attribute__((aligned(16))) uint8_t obs_lds[];
    const int e0 = blockIdx.x * OBS_EPB;
    const int ne = (P.num_envs - e0) < OBS_EPB ? (P.num_envs - e0) : OBS_EPB;
    const ObsView V = obs_view(P, S, e0);
    if (SRC == 0) {
        const ObsBytes src = {B.map + (size_t)e0 * P.width * P.height, P.width, P.height};
        obs_write_block(src, V, B.pos + (size_t)e0 * 2, ne, (int)threadIdx.x, 256);
    } else {
        constexpr int NPL = SRC == 3 ? 3 : 1;
        typedef typename std::conditional<SRC == 2, uint64_t, uint32_t>::type WordT;
        const int env_bytes = P.group * NPL * (int)sizeof(WordT);       // a multiple of 16
        const uint4* g = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(B.planes) + (size_t)e0 * env_bytes);
        for (int i = threadIdx.x; i < (ne * env_bytes) >> 4; i += 256) reinterpret_cast<uint4*>(obs_lds)[i] = g[i];
        uint8_t* lpos = obs_lds + OBS_EPB * env_bytes;
        if ((int)threadIdx.x < ne) reinterpret_cast<uint16_t*>(lpos)[threadIdx.x] = reinterpret_cast<const uint16_t*>(B.pos)[e0 + threadIdx.x];
        __syncthreads();
        const ObsPlanes<WordT, NPL> src = {reinterpret_cast<const WordT*>(obs_lds), P.group};
        obs_write_block(src, V, lpos, ne, (int)threadIdx.x, 256);
    }
}

// Windows beyond the reach of k_obs's 24-bit block offsets (OBS_EPB * oh * ow * depth >= 2^22: a 228 x 228 x 13 crop of an smb level, a
// 128 x 128 one-hot crop of a large map -- the reference's Cropped takes any crop_size, wrappers.py:163-206): the whole output as one
// stream of 16-byte pieces with 64-bit offsets, the position inside an image (row, column, plane) carried along from byte to byte
// instead of divided out.  From the byte map; not a fast path -- the lean routines above are what the trainer's shapes take.
template <int>       // (a template so that the eight parts of the build may all see it: instantiated in the core part only)
__global__ __launch_bounds__(256) void k_obs_huge(PcgrlParams P, DevBufs B, ObsSpec S) {
    const size_t per_env = (size_t)S.oh * S.ow * S.depth, total = (size_t)P.num_envs * per_env;
    const size_t npieces = (total + 15) >> 4;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < npieces; q += (size_t)gridDim.x * 256) {
        const size_t o = q << 4;
        int e = (int)(o / per_env);
        uint32_t rem = (uint32_t)(o - (size_t)e * per_env);               // oh, ow <= 4096, depth <= 8: below 2^27
        uint32_t i = rem / (uint32_t)S.depth;
        int d = (int)(rem - i * (uint32_t)S.depth), r = (int)(i / (uint32_t)S.ow), c = (int)(i - (uint32_t)r * (uint32_t)S.ow);
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        const int nb = (total - o) < 16 ? (int)(total - o) : 16;
        int oy = 0, ox = 0, t = -1;
        for (int k = 0; k < nb; k++) {
            if (t < 0) {        // a new cell: its tile (or the pad value outside the map)
                if (S.centered) { oy = (int)B.pos[2 * (size_t)e + 1] - S.oh / 2; ox = (int)B.pos[2 * (size_t)e] - S.ow / 2; }
                const int y = r + oy, x = c + ox;
                t = ((unsigned)y < (unsigned)P.height && (unsigned)x < (unsigned)P.width) ? (int)B.map[((size_t)e * P.height + y) * P.width + x] : S.pad;
            }
            w[k >> 2] |= (S.depth == 1 ? (uint32_t)t : (uint32_t)(d == t)) << (8 * (k & 3));
            if (++d == S.depth) { d = 0; t = -1; if (++c == S.ow) { c = 0; if (++r == S.oh) { r = 0; ++e; } } }
        }
        if (nb == 16) reinterpret_cast<uint4*>(S.out)[q] = make_uint4(w[0], w[1], w[2], w[3]);
        else for (int k = 0; k < nb; k++) S.out[o + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
    }
}
