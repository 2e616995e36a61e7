// C ABI of the batched PCGRL environment and, through the kernels_*.h headers it includes, its HIP
// kernels (gfx950 / CDNA4).  One source, compiled in eight parts side by side (PCGRL_PART below).
//
// One `pcgrl_step` on the caller's stream is
//   * ONE launch, k_step (kernels_step.h), for the binary and zelda problems on maps of at most 16 rows with the single-cell
//     representations: a block stages the state of its 64 / 128 / 256 environments in LDS and does Representation.update,
//     the statistics, rewards, episode ends and in-kernel resets from there -- and writes the wrapped observation when one is
//     bound (pcgrl_bind_observation, kernels_obs.h);
//   * else the pipeline
//       k_update   thread per environment.  Representation.update (narrow_rep.py:99-114, wide_rep.py:67-70,
//                  turtle_rep.py:101-129), counters + heatmap (pcgrl_env.py:130-137).  Unchanged environments are finished
//                  here (reward 0, done, info); changed ones -- and unchanged ones whose episode ended, flagged "reset
//                  only" -- are compacted into sharded work lists, ordered by what the next kernel will have to do.
//       k_stats    one lane group (16 lanes = one DPP row, or a full wavefront / a block of eight for maps taller than 16) per
//                  work item: Problem.get_stats as row-bitboard programs (pcgrl_algos.h), get_reward, get_episode_over,
//                  get_debug_info (pcgrl_env.py:138-148).  An environment whose episode ended is reset right there
//                  (reset_env.h): PcgrlEnv.reset (pcgrl_env.py:66-76) -- the MT19937 ring staged in LDS, tiles drawn with
//                  numpy's choice() rule, cursor draw, BinaryProblem.reset (binary_prob.py:68-72), start stats
//                  (problem.py:45-46).
//       k_reset    one wavefront per environment on a reset list: pcgrl_reset, and the steps of the search problems.
//       k_sokoban / k_mdungeon / k_ddave / k_smb   the solver jobs parked by k_stats / k_reset: the exact searches of the
//                  reference's engines (two wavefronts per A* search: sokoban_fast.h).
//       k_obs      the wrapped observation, when one is bound.
//   pcgrl_rollout runs a whole tape of actions in one launch (k_step's loop form; k_step_solver for the search problems).
//
// State is structure-of-arrays over the environment axis, all in HBM, owned by the caller
// (include/pcgrl_hip.h).  The uint8 map is the observation; the kernels compute on `planes`
// (row bitboards of the tile-id bits, [N][group][nplanes]) which the update keeps in sync, so the
// statistics never re-read or transpose the byte map.  No MFMA: integer/bit work only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/pcgrl_hip.h"

// Build: this one source is compiled eight times side by side with -DPCGRL_PART=0..7 -- part 0 is the host side of the ABI and the
// small utility kernels, each other part holds the launchers (and with them the instantiations) of one kernel family -- and the
// objects are linked into the library (gym_pcgrl_amd/_lib.py: 2 min 20 s as one unit, ~45 s in parts on eight cores).  Without
// PCGRL_PART it is one translation unit: the developer builds of tools/ (timeline, search profiles) use that form.
#ifdef PCGRL_PART
#define PCGRL_IN_PART(k) (PCGRL_PART == (k))
#else
#define PCGRL_IN_PART(k) 1
#endif
#define PCGRL_LOCAL __attribute__((visibility("hidden")))
#define PART_CORE 0          /* (macros, not an enum: they are compared in #if) */
#define PART_STATS 1
#define PART_UPDATE 2
#define PART_STEP_BINARY 3
#define PART_STEP_ZELDA 4
#define PART_SEARCH 5
#define PART_SMB 6
#define PART_STEP_SOLVER 7

#include "lanegroup_dev.h"
#include "mt19937.h"
#include "pcgrl_algos.h"
#include "sokoban_solver.h"
#include "sokoban_fast.h"
#include "mdungeon_solver.h"
#include "mdungeon_fast.h"
#include "ddave_solver.h"
#include "ddave_fast.h"
#include "level_build_wave.h"

#include "worklist.h"
#include "kernels_update.h"
#include "kernels_obs.h"
#include "reset_env.h"
#include "kernels_stats.h"
#include "kernels_reset.h"
#include "bigmap.h"
#include "bigmap_team.h"
#include "kernels_big.h"
#include "kernels_step.h"
#include "kernels_sokoban.h"
#include "kernels_mdungeon.h"
#include "kernels_ddave.h"
#include "kernels_smb.h"
#include "kernels_step_solver.h"
#include "kernels_search_async.h"
#include "search_big.h"
#include "kernels_search_big.h"
#if PCGRL_IN_PART(PART_CORE)
#include "kernels_misc.h"
#endif

// ------------------------------------------------------------------------------------------
// Host side of the ABI
// the HIP error behind the last PCGRL_EHIP of the calling thread (pcgrl_last_hip_error)
#if PCGRL_IN_PART(PART_CORE)
PCGRL_LOCAL thread_local int g_last_hip = 0;
#else
extern PCGRL_LOCAL thread_local int g_last_hip;
#endif
#if PCGRL_IN_PART(PART_CORE)
#include "step_pool.h"
#endif
struct BigArenaDims { int nblocks, nodes_cap, tsize, stride; size_t heap_off, table_off, block_bytes; };
struct pcgrl_env {
    BigArenaDims big_dims;     // the general searches' arena as it was cut at pcgrl_bind (search_big.h)
    pcgrl_config cfg;
    PcgrlParams P;
    pcgrl_layout L;
    DevBufs B;
    int bound;
    int has_old;      // at least one random reset happened (representation.py:41)
    int was_reset;
    int parity;
    int device;
    // optional per-phase timing with HIP events on the caller's stream (pcgrl_profile)
    int alloc_solver_power;
    // developer switches (pcgrl_set_tuning; A/B measurements and tests), resolved at pcgrl_bind
    pcgrl_tuning tun;
    int no_wide, wide_waves, wide_grid, wide_pairs, fused_zelda, no_fused, step_epb, smb_heap, obs_at_end;
    int profiling;
    int obs_incremental;       // pcgrl_bind_observation(incremental): the bound target is the library's to update in place
    const uint8_t* obs_synced; // the buffer that holds the image of the current state (written by the last step / reset), or NULL
    int obs_hold;     // inside pcgrl_rollout's loop of steps: the bound observation is written once, at the end
    // pcgrl_step_async (kernels_search_async.h): the caller's arena, whether any step may be pending, the tick counter
    AsyncCtl async;
    int async_on, async_dirty, async_split;
    int big_team_waves;       // k_big, binary steps: wavefronts per block (pcgrl_tuning big_team: 1 = the default of 4, else the value)
    std::vector<hipEvent_t> events;
    size_t ev_used;
    int prof_steps;
};
#define PCGRL_NPHASE 6   /* intervals between the 7 event marks of step_one (names: _lib.PHASES) */
static int prof_mark(pcgrl_env* h, hipStream_t st) {
    if (!h->profiling) return PCGRL_OK;
    if (h->ev_used == h->events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return PCGRL_EHIP;
        h->events.push_back(e);
    }
    if (hipEventRecord(h->events[h->ev_used++], st) != hipSuccess) return PCGRL_EHIP;
    return PCGRL_OK;
}

#define HIPCHK(expr) do { hipError_t err_ = (expr); if (err_ != hipSuccess) { g_last_hip = (int)err_; return PCGRL_EHIP; } } while (0)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static int tun_or(int v, int dflt) { return v < 0 ? dflt : v; }       // a pcgrl_tuning field: negative = the library's default

// A handle belongs to the device its buffers live on (found at pcgrl_bind).  Every entry point that launches makes that
// device current for the duration of the call and puts the caller's device back: one host thread may drive several
// GPUs, each through its own handle.
struct DeviceGuard {
    int prev, dev;
    explicit DeviceGuard(int d) : prev(-1), dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

// problems whose statistics need a search kernel after k_stats (the Sokoban solver, the MiniDungeons planner)
static bool solver_prob(int prob) { return prob == PCGRL_SOKOBAN || prob == PCGRL_MDUNGEON || prob == PCGRL_DDAVE || prob == PCGRL_SMB; }
// Sizes (the reference takes any width / height / solver_power: pcgrl_env.py:106-115, probs/problem.py:66-72, sokoban_prob.py:60-73).
//   * maps of up to 64 x 64 cells run on the row-bitboard kernels, larger ones -- up to 255 x 255, the largest map whose cursor
//     fits the observation's uint8 `pos` (narrow_rep.py:60-64) -- on the general path of bigmap.h (big_map below);
//   * a search problem (sokoban, mdungeon, ddave) whose bordered level has at most 256 cells, with solver_power <= 16383, runs the
//     compact searches; larger levels -- up to 4096 bordered cells -- and solver_power up to 1 000 000 run the general searches of
//     search_big.h (big_search below);
//   * smb: its own limits (kernels_smb.h).
#define PCGRL_MAX_DIM 255
#define PCGRL_MAX_DIM_WIDE 4096
#define PCGRL_BIG_LDS_BUDGET ((size_t)150 * 1024)      /* k_big: LDS of one wavefront's masks (launch_big_p) */
#define PCGRL_MAX_LEVEL_CELLS 16384
#define PCGRL_MAX_SOLVER_POWER 1000000
static bool big_map(const pcgrl_config* c) { return c->prob != PCGRL_SMB && (c->width > 64 || c->height > 64); }
static bool big_search(const pcgrl_config* c) {
    return solver_prob(c->prob) && c->prob != PCGRL_SMB && ((c->width + 2) * (c->height + 2) > 256 || c->solver_power > 16383);
}
static int validate_config(const pcgrl_config* c) {
    if (!c) return PCGRL_EINVAL;
    if (c->prob < 0 || c->prob > 5 || c->rep < 0 || c->rep > 5) return PCGRL_EINVAL;
    if (c->num_envs < 1) return PCGRL_EINVAL;
    if (c->prob == PCGRL_SMB) {   // the platformer's grid is (width + 6) x height cells, a column index fits a byte (kernels_smb.h)
        if (c->width < 1 || c->width > 250 || c->height < 3 || c->height > SMB_MAX_H) return PCGRL_EINVAL;
    } else {
        // a representation with a cursor reports it as uint8 `pos` (narrow_rep.py:60-64, turtle_rep.py:73-77): 255 per side.  The wide
        // representation has none (wide_rep.py:42-45, 67-70): any size one wavefront's LDS holds as multi-word row masks (bigmap.h)
        const int max_dim = c->rep == PCGRL_REP_WIDE ? PCGRL_MAX_DIM_WIDE : PCGRL_MAX_DIM;
        if (c->width < 1 || c->width > max_dim || c->height < 1 || c->height > max_dim) return PCGRL_EINVAL;
        if (big_map(c)) {
            const int kw = (c->width + 63) >> 6;
            if (big_wave_lds(c->width, c->height) > PCGRL_BIG_LDS_BUDGET || (long long)big_words(c->width, c->height) * kw >= 65536) return PCGRL_EINVAL;
        }
    }
    if (c->max_changes < 1 || c->max_iterations < 1) return PCGRL_EINVAL;
    if (c->max_changes > 65535) return PCGRL_EINVAL;          // the heat map counts changes per cell in 16 bits (a cell's count <= the episode's changes)
    if (solver_prob(c->prob)) {
        if (c->prob != PCGRL_SMB && (c->width + 2) * (c->height + 2) > PCGRL_MAX_LEVEL_CELLS) return PCGRL_EINVAL;
        if (c->solver_power < 1 || c->solver_power > (c->prob == PCGRL_SMB ? 16383 : PCGRL_MAX_SOLVER_POWER)) return PCGRL_EINVAL;
    }
    return PCGRL_OK;
}
static int ntiles_of(int prob) { return prob == PCGRL_BINARY ? 2 : (prob == PCGRL_SOKOBAN ? 5 : ((prob == PCGRL_DDAVE || prob == PCGRL_SMB) ? 7 : 8)); }

static void fill_params(const pcgrl_config* c, PcgrlParams* P) {
    memset(P, 0, sizeof(*P));
    P->prob = c->prob; P->rep = c->rep; P->num_envs = c->num_envs;
    P->width = c->width; P->height = c->height;
    P->prob_width = c->prob_width > 0 ? c->prob_width : c->width;
    P->prob_height = c->prob_height > 0 ? c->prob_height : c->height;
    P->ntiles = ntiles_of(c->prob);
    P->big = big_map(c) ? 1 : 0;
    P->big_search = big_search(c) ? 1 : 0;
    // smb, and maps beyond 64 x 64 (bigmap.h): no bit planes, everything from the byte map
    P->nplanes = (c->prob == PCGRL_SMB || P->big) ? 0 : (c->prob == PCGRL_BINARY ? 1 : 3);
    P->group = c->height <= 16 ? 16 : 64;
    P->mask_bytes = c->width <= 32 ? 4 : 8;
    P->max_changes = c->max_changes; P->max_iterations = c->max_iterations;
    P->random_start = c->random_start; P->random_tile = c->random_tile; P->warp = c->warp;
    P->random_probs = c->random_probs; P->auto_reset = c->auto_reset;
    P->target_path = c->target_path; P->max_enemies = c->max_enemies; P->target_enemy_dist = c->target_enemy_dist;
    P->max_crates = c->max_crates; P->target_solution = c->target_solution; P->solver_power = c->solver_power;
    P->max_potions = c->max_potions; P->max_treasures = c->max_treasures; P->target_col_enemies = c->target_col_enemies;
    P->max_diamonds = c->max_diamonds; P->min_spikes = c->min_spikes; P->target_jumps = c->target_jumps;
    P->min_empty = c->min_empty; P->min_enemies = c->min_enemies; P->min_jumps = c->min_jumps;
    for (int i = 0; i < PCGRL_MAX_REWARDS; i++) P->rewards[i] = c->rewards[i];
    pcgrl_build_cdf(c->tile_probs, P->ntiles, P->cdf);
}

static const size_t WL_CNT_BYTES = 2 * WL_NLIST * WL_NSHARD * WL_CSTRIDE * sizeof(int32_t);
// shard capacity: the changed list is bucketed by difficulty, so one bucket may receive every environment
static int wl_capacity(int num_envs, int list) {
    return list == WL_CHG ? num_envs : 2 * ((num_envs + WL_NSHARD - 1) / WL_NSHARD + 256);
}
static size_t wl_list_bytes(int num_envs, int list) { return align_up((size_t)WL_NSHARD * wl_capacity(num_envs, list) * 4, 256); }
#define SOK_BLOCKS 256   /* resident solver blocks (one per CU: heap + table fill most of its LDS) */
static size_t wl_bytes(const pcgrl_config* c) {
    size_t b = WL_CNT_BYTES + 256;
    for (int k = 0; k < WL_NLIST; k++) b += wl_list_bytes(c->num_envs, k);
    return b;
}
// per-environment agent results + counters, and the scheduling words of the two solver launches of a step
static size_t sok_sync_bytes() { return align_up(2 * (size_t)(SOK_SY_WORDS + SOK_HARD_CAP) * 4, 256); }
static size_t sok_sched_bytes(int num_envs) { return align_up((size_t)num_envs * 18 * 4, 256) + sok_sync_bytes(); }
static int sok_table_size(int power) { int t = 1024; while (t < 2 * power) t <<= 1; return t; }
// Node pool of one resident solver block: 4 children per pop of the full search -- and never less than the private
// small-tier pools of k_step_solver (SS_SEARCH_WAVES wavefronts x SS_SMALL_NODES at pool + wv * SS_SMALL_NODES), so that a
// small solver_power cannot make them spill into the next block's pool.
static size_t sok_pool_nodes(int power, int prob = -1) {
    if (prob == PCGRL_SMB)     // k_smb: up to SMB_MAX_WAVES searches per block, each with its own arena (kernels_smb.h)
        return (SMB_MAX_WAVES * smb_wave_arena_bytes(power) + sizeof(SokNode) - 1) / sizeof(SokNode);
    const size_t full = 4 * (size_t)power + 4, small = (size_t)SS_SEARCH_WAVES * SS_SMALL_NODES;
    return full > small ? full : small;
}
static_assert(SS_SMALL_NODES >= 4 * SS_SMALL_POPS + 4, "a small-tier search pushes up to four nodes per pop");
// rows of the champion component per environment (binary, maps of at most 16 x 32): the incremental statistics path
// (smb: per map column the rows the last play-through read -- uint32 [N][W], kernels_smb.h -- for the representations that change one tile a step)
static size_t champ_bytes(const pcgrl_config* c) {
    if (c->prob == PCGRL_SMB) return (c->rep <= PCGRL_REP_TURTLE && c->num_envs < SMB_KEEP_PLAY) ? align_up((size_t)c->num_envs * c->width * 4, 256) : 0;
    if (c->prob != PCGRL_BINARY) return 0;
    if (big_map(c))      // maps beyond 64 x 64 (bigmap.h big_incremental): [H][KW] 64-bit words; the work item packs the cell into 8 + 8 bits and the environment into 15
        return (c->rep <= PCGRL_REP_TURTLE && c->width <= 256 && c->height <= 256 && c->num_envs <= WL_INCBIG_ENV_MASK + 1)
                   ? align_up((size_t)c->num_envs * big_words(c->width, c->height) * 8, 256) : 0;
    if (c->height <= 16) return (c->width <= 32 && c->num_envs <= WL_INC_ENV_MASK) ? align_up((size_t)c->num_envs * 64, 256) : 0;
    return c->num_envs <= WL_INC64_ENV_MASK ? align_up((size_t)c->num_envs * 64 * (c->width > 32 ? 8 : 4), 256) : 0;
}
// draw cache of the narrow representation (DevBufs::fifo, fifo_tag)
static size_t fifo_words_bytes(const pcgrl_config* c) { return c->rep == PCGRL_NARROW ? align_up((size_t)c->num_envs * PCGRL_FIFO_N * 4, 256) : 0; }
static size_t fifo_bytes(const pcgrl_config* c) { return c->rep == PCGRL_NARROW ? fifo_words_bytes(c) + align_up((size_t)c->num_envs * 4, 256) : 0; }
static size_t scratch_bytes_base(const pcgrl_config* c);
static size_t wide_sync_bytes(const pcgrl_config* c) { return (c->prob == PCGRL_BINARY && c->height > 16 && !big_map(c)) ? align_up((size_t)c->num_envs * 16, 256) : 0; }
static size_t scratch_bytes(const pcgrl_config* c) { return scratch_bytes_base(c) + champ_bytes(c) + fifo_bytes(c) + wide_sync_bytes(c); }
// The arena of the general searches (search_big.h): per resident block a node pool (4 children per pop), a 64-bit heap and a
// visited table of 32-bit node indices.  As many blocks as fit a 6 GB budget (at most SOK_BLOCKS, at least 4).
static BigArenaDims big_arena_of(const pcgrl_config* c) {
    BigArenaDims A;
    const int cells = (c->width + 2) * (c->height + 2), nwb = (cells + 63) / 64, inner = c->width * c->height;
    A.nodes_cap = 4 * c->solver_power + 4;
    A.tsize = 1024;
    while (A.tsize < 2 * c->solver_power) A.tsize <<= 1;
    // sokb_init_deadlocks keeps its corner list (16-bit cells, fewer than `cells` of them) in the block's heap before the search
    // starts: heap + visited table must hold it even for a tiny solver_power, so that it never reaches the next block's node pool
    while (align_up((size_t)A.nodes_cap * 8, 256) + (size_t)A.tsize * 4 < (size_t)2 * cells) A.tsize <<= 1;
    A.stride = c->prob == PCGRL_SOKOBAN ? sokb_stride(inner < SOKB_MAXC ? inner : SOKB_MAXC) : mdb_stride(nwb);
    A.heap_off = align_up((size_t)A.nodes_cap * A.stride, 256);
    A.table_off = A.heap_off + align_up((size_t)A.nodes_cap * 8, 256);
    A.block_bytes = A.table_off + align_up((size_t)A.tsize * 4, 256);
    const size_t budget = (size_t)6 << 30;
    size_t nb = budget / A.block_bytes;
    nb = nb > SOK_BLOCKS ? SOK_BLOCKS : (nb < 4 ? 4 : nb);
    if (nb > (size_t)c->num_envs) nb = (size_t)c->num_envs;
    A.nblocks = (int)nb;
    return A;
}
static size_t scratch_bytes_base(const pcgrl_config* c) {
    size_t b = wl_bytes(c);
    if (big_search(c)) {
        const BigArenaDims A = big_arena_of(c);
        return b + (size_t)A.nblocks * A.block_bytes + sok_sched_bytes(c->num_envs);
    }
    if (solver_prob(c->prob)) {           // (an MdNode is as large as a SokNode)
        const size_t nodes = sok_pool_nodes(c->solver_power, c->prob);
        b += SOK_BLOCKS * align_up(nodes * sizeof(SokNode), 256);
        b += sok_sched_bytes(c->num_envs);
        const size_t hnodes = 4 * (size_t)c->solver_power + 4;
        if (c->solver_power > SOK_LDS_POWER && c->prob != PCGRL_SMB) b += SOK_BLOCKS * (align_up(hnodes * 4, 256) + (size_t)sok_table_size(c->solver_power) * 4);
    }
    return b;
}

// Per-DEVICE state the kernels need, (re)established at every pcgrl_bind -- never behind a process-wide flag: a process may
// hold handles on several GPUs, and both a function attribute and a __device__ symbol belong to one device.
PCGRL_LOCAL int search_device_setup(pcgrl_env* h);      // PART_SEARCH: dynamic-LDS attribute of k_sokoban / k_mdungeon / k_ddave
PCGRL_LOCAL int smb_device_setup(pcgrl_env* h);         // PART_SMB: of k_smb
#if PCGRL_IN_PART(PART_CORE)
static int device_setup(pcgrl_env* h) {
    {   // init_genrand(19650218): the table every MT19937 init_by_array starts from (k_init_by_array)
        uint32_t tab[PCGRL_MT_N];
        tab[0] = 19650218u;
        for (int i = 1; i < PCGRL_MT_N; i++) tab[i] = 1812433253u * (tab[i - 1] ^ (tab[i - 1] >> 30)) + (uint32_t)i;
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_mt_genrand), tab, sizeof(tab)));
    }
    if (h->cfg.prob == PCGRL_SMB) return smb_device_setup(h);
    if (solver_prob(h->cfg.prob) && !big_search(&h->cfg)) return search_device_setup(h);
    return PCGRL_OK;
}

extern "C" {

int pcgrl_abi_version(void) { return PCGRL_ABI_VERSION; }
int pcgrl_last_hip_error(void) { return g_last_hip; }
const char* pcgrl_error_string(int code) {
    switch (code) {
        case PCGRL_OK: return "ok";
        case PCGRL_EINVAL: return "invalid argument or unsupported configuration";
        case PCGRL_EHIP: return "HIP runtime error";
        case PCGRL_ESTATE: return "call order violated (bind, seed and reset before step)";
        default: return "unknown error";
    }
}

int pcgrl_query_layout(const pcgrl_config* c, pcgrl_layout* L) {
    int rc = validate_config(c);
    if (rc) return rc;
    if (!L) return PCGRL_EINVAL;
    PcgrlParams P;
    fill_params(c, &P);
    const size_t n = (size_t)c->num_envs, cells = (size_t)c->width * c->height;
    memset(L, 0, sizeof(*L));
    L->group = P.group; L->mask_bytes = P.mask_bytes; L->nplanes = P.nplanes; L->nstats = num_stats(c->prob);
    L->map = n * cells; L->old_map = n * cells; L->heatmap = n * cells * 2; L->pos = n * 2;
    L->planes = n * P.nplanes * P.group * P.mask_bytes;
    L->counters = n * 8; L->stats = n * 32; L->start_stats = n * 32; L->info = n * 40;
    L->reward = n * 8; L->done = n; L->tile_p = n * 16;
    L->rng_rep = n * PCGRL_MT_N * 4; L->rng_prob = c->prob == PCGRL_BINARY ? n * PCGRL_MT_N * 4 : 0;
    L->rng_cursor = n * 8;
    L->scratch = scratch_bytes(c);
    return PCGRL_OK;
}

int pcgrl_create(const pcgrl_config* c, pcgrl_env** out) {
    int rc = validate_config(c);
    if (rc) return rc;
    if (!out) return PCGRL_EINVAL;
    pcgrl_env* h = new pcgrl_env();
    h->bound = h->has_old = h->was_reset = h->parity = h->device = 0;
    h->profiling = 0; h->ev_used = 0; h->prof_steps = 0; h->obs_hold = 0; h->obs_incremental = 0; h->obs_synced = nullptr;
    memset(&h->async, 0, sizeof(h->async)); h->async_on = h->async_dirty = 0;
    memset(&h->B, 0, sizeof(h->B));
    pcgrl_tuning_defaults(&h->tun);
    h->cfg = *c;
    fill_params(c, &h->P);
    pcgrl_query_layout(c, &h->L);
    *out = h;
    return PCGRL_OK;
}

void pcgrl_tuning_defaults(pcgrl_tuning* t) {
    if (!t) return;
    int32_t* f = reinterpret_cast<int32_t*>(t);
    for (size_t i = 0; i < sizeof(pcgrl_tuning) / sizeof(int32_t); i++) f[i] = -1;
}
int pcgrl_set_tuning(pcgrl_env* h, const pcgrl_tuning* t) {
    if (!h || !t) return PCGRL_EINVAL;
    if (h->bound) return PCGRL_ESTATE;            // the switches are resolved by pcgrl_bind
    h->tun = *t;
    return PCGRL_OK;
}

int pcgrl_destroy(pcgrl_env* h) {
    if (h) {
        for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    }
    delete h;
    return PCGRL_OK;
}

int pcgrl_bind(pcgrl_env* h, const pcgrl_buffers* b, void* stream) {
    if (!h || !b) return PCGRL_EINVAL;
    if (!b->map || !b->old_map || !b->heatmap || !b->pos || !b->planes || !b->counters || !b->stats ||
        !b->start_stats || !b->info || !b->reward || !b->done || !b->tile_p || !b->rng_rep || !b->rng_cursor || !b->scratch)
        return PCGRL_EINVAL;
    if (h->cfg.prob == PCGRL_BINARY && !b->rng_prob) return PCGRL_EINVAL;
    {   // the handle's device is the one that owns the buffers
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, b->map) == hipSuccess) h->device = at.device;
        else { (void)hipGetLastError(); HIPCHK(hipGetDevice(&h->device)); }
    }
    DeviceGuard guard(h->device);
    int rc0 = device_setup(h);
    if (rc0) return rc0;
    // developer switches (pcgrl_set_tuning): resolved here, once -- nothing on the step path reads them, and the library reads no
    // environment variables
    const pcgrl_tuning& T = h->tun;
    h->no_wide = tun_or(T.no_wide, 0) ? 1 : 0;
    h->wide_waves = tun_or(T.wide_waves, 8);
    h->wide_pairs = tun_or(T.wide_pairs, 1);           // 0 = every full item of a tall map a block of its own
    h->wide_grid = tun_or(T.wide_grid, 2048);           // blocks of k_stats_wide (they loop over the items; C5 steady: 768 .. 4096 -> 59.5 us/step, 8192 -> 63.5: a block costs ~4 us of prefix sums before its first item)
    if (h->wide_grid < 1) h->wide_grid = 2048;
    h->fused_zelda = tun_or(T.fused_zelda, 1) ? 1 : 0;  // 0: zelda steps as k_update + k_stats
    h->no_fused = tun_or(T.no_fused, 0) ? 1 : 0;
    h->obs_at_end = tun_or(T.obs_at_end, 0) ? 1 : 0;
    // pcgrl_step_async: the fresh jobs of a tick in a launch of their own with small search regions.  Pays where a tick has many of them
    // (C4: 1 300 fresh jobs of ~30 pops a tick, 240 -> 275 M env-steps/s); MiniDungeons / Dave have under a hundred and the extra
    // launch -- whose length is one job's chain of pops, like the other's -- costs more than it saves (M1 277 -> 207 M)
    h->async_split = tun_or(T.async_split, h->cfg.prob == PCGRL_SOKOBAN ? 1 : 0) ? 1 : 0;
    {   // k_step: environments per block (see launch_step_pm).  The largest block that still gives (about) every compute unit one and
        // is resident in one round: 256 environments -> one block per CU (LDS), 128 -> two, 64 -> four.
        const int n_ = h->cfg.num_envs;
        h->step_epb = T.step_epb > 0 ? T.step_epb : ((n_ >= 192 * 256 && n_ <= 256 * 256) ? 256 : (n_ >= 192 * 128 ? 128 : 64));
        if (h->step_epb != 128 && h->step_epb != 256) h->step_epb = 64;
        h->smb_heap = T.smb_lds_heap > 0 ? T.smb_lds_heap : SMB_LDS_HEAP;        // heap words a k_smb search keeps in LDS
        if (h->smb_heap < 256 || h->smb_heap > 4096) h->smb_heap = SMB_LDS_HEAP;
    }
    {
        const int f = tun_or(T.full_per_wave, 4), i = tun_or(T.inc_per_wave, 4);
        h->B.step_fpw = (f == 1 || f == 2) ? f : 4;
        // wavefront priorities in k_step: certain resets and full recomputations of the binary problem at level 3, its incremental
        // updates at 0 (C2: 30.4 -> 28.9 us first window, 29.4 -> 28.7 steady).  Zelda's tasks are all of one kind and about one
        // length; every setting measured there was 0.3-0.7 us slower than none.
        // (round 6: with the narrow representation the update wavefronts -- cursor draws, ring refills -- at level 3 as well: C2 steady 29.05 ->
        //  28.6 us, first window unchanged; zelda loses 0.4 us with it, the wrapped steps do not care: profiles/r6_round6/probe/ab_step_prio_update.txt)
        h->B.step_prio = tun_or(T.step_prio, h->cfg.prob == PCGRL_BINARY ? (h->cfg.rep == PCGRL_NARROW ? (15 | (3 << 6)) : 15) : 0) & 0xFFF;
        h->B.step_ipw = (i == 1 || i == 2) ? i : 4;
        h->B.step_touch = tun_or(T.no_touch, 0) ? 0 : 1;
        h->B.step_pair = tun_or(T.step_pair, 6);        // (C3: fifteen certain resets a block and step; 32.2 -> 29.8 us.  C2 has two or three: unaffected)
        h->B.step_tight = h->B.step_touch ? tun_or(T.touch_tight, 1) : 0;
    }
    const bool no_inc = tun_or(T.no_inc, 0) != 0;       // every change takes the full statistics (A/B, tests)
    DevBufs& B = h->B;
    B.map = (uint8_t*)b->map; B.old_map = (uint8_t*)b->old_map; B.heat = (uint16_t*)b->heatmap; B.pos = (uint8_t*)b->pos;
    B.heat_end = B.heat + (size_t)h->cfg.num_envs * h->cfg.width * h->cfg.height;
    B.ep_return = nullptr; B.ep_length = nullptr; B.last_return = nullptr; B.last_length = nullptr;
    B.local = nullptr;
    B.planes = b->planes; B.counters = (int32_t*)b->counters; B.stats = (int32_t*)b->stats;
    B.start_stats = (int32_t*)b->start_stats; B.info = (int32_t*)b->info; B.reward = (double*)b->reward;
    B.done = (uint8_t*)b->done; B.tile_p = (double*)b->tile_p; B.rng_rep = (uint32_t*)b->rng_rep;
    B.rng_prob = (uint32_t*)b->rng_prob; B.rng_cur = (int32_t*)b->rng_cursor;
    uint8_t* s = (uint8_t*)b->scratch;
    B.wl_cnt = (int32_t*)s;
    B.status = (int32_t*)(s + WL_CNT_BYTES);
    {
        uint8_t* q = s + WL_CNT_BYTES + 256;
        for (int k = 0; k < WL_NLIST; k++) {
            B.wl_cap[k] = wl_capacity(h->cfg.num_envs, k);
            B.wl_items[k] = (int32_t*)q;
            q += wl_list_bytes(h->cfg.num_envs, k);
        }
    }
    HIPCHK(hipMemsetAsync(B.wl_cnt, 0, WL_CNT_BYTES + 256, (hipStream_t)stream));
    B.sok_pool = nullptr; B.sok_heap = nullptr; B.sok_table = nullptr;
    B.zelda_inc = (h->cfg.prob == PCGRL_ZELDA && h->cfg.rep <= PCGRL_REP_TURTLE && h->cfg.height <= 16 && h->cfg.width <= 32 &&
                   h->cfg.num_envs <= WL_INC_ENV_MASK && !no_inc) ? 1 : 0;
    B.pair_min = tun_or(T.pair_min, 2048);
    B.big_team = tun_or(T.big_team, 1) ? 1 : 0;
    h->big_team_waves = tun_or(T.big_team, 1) >= 2 ? (tun_or(T.big_team, 1) > BIG_TEAM_MAX_WAVES ? BIG_TEAM_MAX_WAVES : tun_or(T.big_team, 1)) : 4;
    B.champ = nullptr;
    if (champ_bytes(&h->cfg) && !no_inc) {
        B.champ = s + scratch_bytes_base(&h->cfg);
        HIPCHK(hipMemsetAsync(B.champ, h->cfg.prob == PCGRL_SMB ? 0xFF : 0, champ_bytes(&h->cfg), (hipStream_t)stream));   // (smb: "every cell read")
    }
    B.obs = ObsSpec{nullptr, 0, 0, 0, 0, 0, 0, 0};
    B.fifo = nullptr; B.fifo_tag = nullptr;
    if (fifo_bytes(&h->cfg)) {
        uint8_t* f = s + scratch_bytes_base(&h->cfg) + champ_bytes(&h->cfg);     // (wide_sync follows the draw cache: below)
        B.fifo = (uint32_t*)f;
        B.fifo_tag = (int32_t*)(f + fifo_words_bytes(&h->cfg));
        HIPCHK(hipMemsetAsync(B.fifo_tag, 0xFF, (size_t)h->cfg.num_envs * 4, (hipStream_t)stream));     // -1: nothing cached yet
    }
    B.wide_sync = nullptr; B.wide_epoch = 0;
    B.wide_few = tun_or(T.wide_few, WL_WIDE_FEW_REGIONS);
    B.wide_spin = T.wide_spin > 0 ? T.wide_spin : WIDE_SPIN_LIMIT;
    if (wide_sync_bytes(&h->cfg)) {
        B.wide_sync = (int32_t*)(s + scratch_bytes_base(&h->cfg) + champ_bytes(&h->cfg) + fifo_bytes(&h->cfg));
        HIPCHK(hipMemsetAsync(B.wide_sync, 0, wide_sync_bytes(&h->cfg), (hipStream_t)stream));
    }
    // inline_reset = 0 routes resets through the reset list + k_reset instead (A/B measurements)
    B.inline_reset = (!solver_prob(h->cfg.prob) && tun_or(T.inline_reset, 1) != 0) ? 1 : 0;
    B.flat = nullptr;
    B.big_arena = nullptr;
    if (big_search(&h->cfg)) {
        // levels / solver_power beyond the compact searches: the general searches' arena, then the scheduling words
        h->alloc_solver_power = h->cfg.solver_power;
        const BigArenaDims A = h->big_dims = big_arena_of(&h->cfg);        // (the launches keep using the dimensions the arena was cut with)
        uint8_t* a = s + wl_bytes(&h->cfg);
        B.big_arena = a;
        a += (size_t)A.nblocks * A.block_bytes;
        B.sok_res = (int32_t*)a;
        B.sok_cnt = B.sok_res + (size_t)h->cfg.num_envs * 16;
        B.sok_stop = B.sok_cnt + h->cfg.num_envs;
        B.sok_sync = (int32_t*)(a + align_up((size_t)h->cfg.num_envs * 18 * 4, 256));
        HIPCHK(hipMemsetAsync(a, 0, sok_sched_bytes(h->cfg.num_envs), (hipStream_t)stream));
        B.sok_use_lds = 0; B.sok_fast_maxc = -1; B.md_only_agent = -1; B.sok_hard_cap = 0; B.sok_spawn_iters = SOK_SPAWN_ITERS;
        B.sok_table_size = A.tsize; B.sok_pool_stride = 0; B.sok_heap_stride = 0;
    } else if (solver_prob(h->cfg.prob)) {
        // the arena is sized for the solver_power the buffers were allocated with
        const int power = h->alloc_solver_power = h->cfg.solver_power;
        const size_t nodes = sok_pool_nodes(power, h->cfg.prob), hnodes = 4 * (size_t)power + 4;
        uint8_t* a = s + wl_bytes(&h->cfg);
        B.sok_pool = (SokNode*)a;
        B.sok_pool_stride = (int32_t)(align_up(nodes * sizeof(SokNode), 256) / sizeof(SokNode));
        a += SOK_BLOCKS * align_up(nodes * sizeof(SokNode), 256);
        B.sok_res = (int32_t*)a;
        B.sok_cnt = B.sok_res + (size_t)h->cfg.num_envs * 16;
        B.sok_stop = B.sok_cnt + h->cfg.num_envs;
        B.sok_sync = (int32_t*)(a + align_up((size_t)h->cfg.num_envs * 18 * 4, 256));
        HIPCHK(hipMemsetAsync(a, 0, sok_sched_bytes(h->cfg.num_envs), (hipStream_t)stream));
        a += sok_sched_bytes(h->cfg.num_envs);
        B.sok_use_lds = power <= SOK_LDS_POWER || h->cfg.prob == PCGRL_SMB;     // (smb has its own heap split, kernels_smb.h)
        B.sok_fast_maxc = tun_or(T.sok_generic, 0) ? -1 : SOKF_MAXC;       // sok_generic: every level takes the generic search (tests)
        B.md_only_agent = T.md_only_agent;                                  // (-1: all four agents)
        B.sok_hard_cap = tun_or(T.sok_hard_cap, SOK_HARD_CAP);
        B.sok_spawn_iters = T.sok_spawn > 0 ? T.sok_spawn : SOK_SPAWN_ITERS;
        if (B.sok_hard_cap > SOK_HARD_CAP) B.sok_hard_cap = SOK_HARD_CAP;
        B.sok_table_size = sok_table_size(power);
        B.sok_heap_stride = (int32_t)(align_up(hnodes * 4, 256) / 4);
        if (!B.sok_use_lds) {
            B.sok_heap = (uint32_t*)a;
            a += SOK_BLOCKS * align_up(hnodes * 4, 256);
            B.sok_table = (uint32_t*)a;
        }
    }
    h->bound = 1; h->has_old = 0; h->was_reset = 0; h->parity = 0;
    h->async_on = h->async_dirty = 0;      // (pcgrl_bind_async follows a bind)
    return PCGRL_OK;   // tile_p is caller state: call pcgrl_set_tile_probs once after the first bind
}

int pcgrl_configure(pcgrl_env* h, const pcgrl_config* c) {
    if (!h) return PCGRL_EINVAL;
    int rc = validate_config(c);
    if (rc) return rc;
    if (c->prob != h->cfg.prob || c->rep != h->cfg.rep || c->num_envs != h->cfg.num_envs ||
        c->width != h->cfg.width || c->height != h->cfg.height)
        return PCGRL_EINVAL;
    if (h->bound && solver_prob(c->prob) && c->solver_power > h->alloc_solver_power) return PCGRL_EINVAL;   // arena too small: re-create
    // a handle bound for the general searches keeps them when solver_power shrinks below their threshold (its arena is theirs);
    // the other direction needs that arena: re-create
    const bool was_big = h->bound && h->P.big_search;
    if (h->bound && big_search(c) && !was_big) return PCGRL_EINVAL;
    h->cfg = *c;
    fill_params(c, &h->P);
    if (was_big) h->P.big_search = 1;
    return PCGRL_OK;
}

int pcgrl_set_tile_probs(pcgrl_env* h, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    DeviceGuard guard(h->device);
    const int n = h->cfg.num_envs;
    hipLaunchKernelGGL(k_bcast_tile_p, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->B.tile_p, n,
                       h->cfg.tile_probs[0], h->cfg.tile_probs[1]);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}

int pcgrl_seed(pcgrl_env* h, const uint32_t* keys, int32_t first, int32_t count, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    if (!keys || first < 0 || count < 1 || first + count > h->cfg.num_envs) return PCGRL_EINVAL;
    DeviceGuard guard(h->device);
    const size_t bytes = (size_t)count * PCGRL_MT_N * 4, off = (size_t)first * PCGRL_MT_N;
    HIPCHK(hipMemcpyAsync(h->B.rng_rep + off, keys, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    if (h->B.fifo_tag) HIPCHK(hipMemsetAsync(h->B.fifo_tag + first, 0xFF, (size_t)count * 4, (hipStream_t)stream));
    if (h->B.rng_prob) HIPCHK(hipMemcpyAsync(h->B.rng_prob + off, keys, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    hipLaunchKernelGGL(k_zero_cursors, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->B.rng_cur, first, count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));   // `keys` may be pageable host memory
    return PCGRL_OK;
}

// The same seeding with the MT19937 states computed on the device: words HOST u32 [count][3] = (key word 0, key word 1,
// number of key words: 1 or 2) per environment -- what gym's hash_seed leaves of `seed` (gym_pcgrl_amd/seeding.py
// hash_seed_words).  8 bytes of key per environment cross PCIe instead of a 2.5 KB state.
int pcgrl_seed_words(pcgrl_env* h, const uint32_t* words, int32_t first, int32_t count, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    if (!words || first < 0 || count < 1 || first + count > h->cfg.num_envs) return PCGRL_EINVAL;
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemcpy2DAsync(h->B.rng_rep + (size_t)first * PCGRL_MT_N, PCGRL_MT_N * 4, words, 12, 12, (size_t)count, hipMemcpyHostToDevice, st));
    if (h->B.fifo_tag) HIPCHK(hipMemsetAsync(h->B.fifo_tag + first, 0xFF, (size_t)count * 4, st));
    hipLaunchKernelGGL(k_init_by_array, dim3((count + 63) / 64), dim3(64), 0, st, h->B.rng_rep, h->B.rng_prob, h->B.rng_cur, first, count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));   // `words` may be pageable host memory
    return PCGRL_OK;
}

}  // extern "C"
#endif  // PART_CORE

// ---- launch helpers ------------------------------------------------------------------------
struct RolloutArgs { int steps; size_t action_stride; double* reward_out; uint8_t* done_out; int32_t* info_out; };
PCGRL_LOCAL int launch_stats(pcgrl_env* h, int list, int parity, int mode, int clr, int inline_reset, hipStream_t st);
PCGRL_LOCAL int launch_reset(pcgrl_env* h, int list, int park_list, int parity, int clr, hipStream_t st);
PCGRL_LOCAL int launch_planes_from_map(pcgrl_env* h, const uint8_t* maps, hipStream_t st);
PCGRL_LOCAL int launch_update(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st);
PCGRL_LOCAL int launch_step_binary(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st, const RolloutArgs& R);
PCGRL_LOCAL int launch_step_zelda(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st, const RolloutArgs& R);
PCGRL_LOCAL int launch_search(pcgrl_env* h, int32_t* sync, int list_a, int mode_a, int list_b, int mode_b, int parity, int rst_list, int clr, hipStream_t st);
PCGRL_LOCAL int launch_smb(pcgrl_env* h, int32_t* sync, int list_a, int mode_a, int list_b, int mode_b, int parity, int rst_list, int clr, hipStream_t st, int inline_reset);
PCGRL_LOCAL int launch_step_solver(pcgrl_env* h, const int32_t* actions, hipStream_t st, const RolloutArgs& R, int epb);
PCGRL_LOCAL int launch_search_async(pcgrl_env* h, int32_t* tickets, int list_a, int mode_a, int list_b, int mode_b, int parity, int rst_list, int clr, int budget, int resume, int small, hipStream_t st);
// A launch with more than 64 KB of dynamic LDS needs the kernel's cap raised first.  That is a driver call (about a microsecond of
// host time -- a fifth of what issuing a step costs), so it is made only when this process has not yet raised the cap of this kernel
// on this device that far: the cap only ever grows, whatever mix of handles launches the kernel (benign race: two threads may both set it).
template <auto KFN>
static int lds_cap(int device, size_t lds) {
    static size_t cap[64];
    if (lds <= 64 * 1024) return PCGRL_OK;
    const int d = device & 63;
    if (lds > cap[d]) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(KFN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        cap[d] = lds;
    }
    return PCGRL_OK;
}
static int grid_for(int items, int per_block, int cap) {
    int g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return g < cap ? g : cap;
}

#if PCGRL_IN_PART(PART_STATS)
template <int PROB>
static int launch_stats_p(pcgrl_env* h, int list, int parity, int mode, int clr, int inline_reset, hipStream_t st) {
    const PcgrlParams& P = h->P;
    const int gpb = PCGRL_BLOCK / P.group;
    const int grid = grid_for(P.num_envs, gpb, 8192);
    // in-kernel reset: one MT ring + tile-byte staging area per wavefront
    const size_t lds = inline_reset ? 4 * (size_t)(PCGRL_MT_N * 4 + ((P.width * P.height + 15) & ~15)) : 0;
    const int gen = (P.random_start || !h->has_old) ? 1 : 0;
    // where k_update put the environments that are certain to be reset in this launch
    // (1: shard 0 of the bucketed changed list -- single-cell representations of the binary problem on 16-row maps;
    //  2: the list WL_RST -- everything else that resets in k_stats)
    const int lone0 = !(mode == MODE_STEP && inline_reset) ? 0 : ((PROB == PCGRL_PROB_BINARY && P.group == 16 && P.rep <= PCGRL_REP_TURTLE) ? 1 : 2);
    if (PROB == PCGRL_PROB_BINARY && P.group == 64 && !h->no_wide) {   // block per item (k_stats_wide); PCGRL_NO_WIDE=1: A/B switch
        const int nw = h->wide_waves;
        // per wavefront an MT19937 ring + the tile bytes of a map; the block-wide reset keeps its raw words (8 bytes a cell) and the
        // map's bit string in the sets of wavefronts 1.. (kernels_stats.h): at least that much
        const size_t set1 = PCGRL_MT_N * 4 + ((P.width * P.height + 15) & ~15);
        const size_t need1 = set1 + (size_t)8 * P.width * P.height + 8 * ((size_t)(P.width * P.height + 63) / 64 + 2);
        const size_t sets1 = (size_t)(nw == 8 ? 8 : 4) * set1;
        const size_t lds1 = inline_reset ? (sets1 > need1 ? sets1 : need1) : 0;
        int gridw = P.num_envs < h->wide_grid ? P.num_envs : h->wide_grid;
        if (gridw > 1) gridw &= ~1;                                    // even: see the two halves of a certain reset in k_stats_wide
        h->B.wide_epoch = (h->B.wide_epoch % 0x3FFFFFFF) + 1;          // this launch's word in wide_sync (never 0: the cleared state)
        // wavefronts per map: with the incremental route only ~10 % of the changes (and the resets) come here, so the launch
        // is latency-bound and more wavefronts per map pay (PCGRL_WIDE_WAVES overrides for experiments)
        // (a step of a single-cell representation: k_update ranked the list in two classes, the second goes two items to a block)
        const int pair_few = (mode == MODE_STEP && list == WL_CHG && P.rep <= PCGRL_REP_TURTLE && h->wide_pairs) ? 1 : 0;
#define LAUNCH_WIDE(NW) do { if (P.mask_bytes == 4) hipLaunchKernelGGL((k_stats_wide<uint32_t, NW>), dim3(gridw), dim3(NW * 64), lds1, st, P, h->B, list, parity, mode, clr, inline_reset, gen, pair_few); \
                             else hipLaunchKernelGGL((k_stats_wide<uint64_t, NW>), dim3(gridw), dim3(NW * 64), lds1, st, P, h->B, list, parity, mode, clr, inline_reset, gen, pair_few); } while (0)
        if (nw == 8) LAUNCH_WIDE(8); else LAUNCH_WIDE(4);
#undef LAUNCH_WIDE
        HIPCHK(hipGetLastError());
        return PCGRL_OK;
    }
    if (P.group == 16 && P.mask_bytes == 4)
        hipLaunchKernelGGL((k_stats<PROB, 16, uint32_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, list, parity, mode, clr, inline_reset, gen, lone0);
    else if (P.group == 16)
        hipLaunchKernelGGL((k_stats<PROB, 16, uint64_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, list, parity, mode, clr, inline_reset, gen, lone0);
    else if (P.mask_bytes == 4)
        hipLaunchKernelGGL((k_stats<PROB, 64, uint32_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, list, parity, mode, clr, inline_reset, gen, lone0);
    else
        hipLaunchKernelGGL((k_stats<PROB, 64, uint64_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, list, parity, mode, clr, inline_reset, gen, lone0);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
// Maps beyond 64 x 64 (bigmap.h, kernels_big.h): a wavefront per item, as many wavefronts per block as the LDS holds
template <int PROB>
static int launch_big_p(pcgrl_env* h, int list, int parity, int mode, int clr, int inline_reset, int park_list, hipStream_t st) {
    const PcgrlParams& P = h->P;
    const size_t per_wave = big_wave_lds(P.width, P.height);
    int nw = (int)(PCGRL_BIG_LDS_BUDGET / per_wave);
    // (binary, a step: its few full recomputations are made by whole blocks -- four wavefronts a map measured best: 10.2 / 11.2 / 11.5 /
    //  11.2 / 10.6 M env-steps/s on B1 with 2 / 3 / 4 / 6 / 8; more bands = more components that cross one)
    const int nw_max = (PROB == PCGRL_PROB_BINARY && h->B.big_team && mode == MODE_STEP) ? h->big_team_waves : 4;
    nw = nw > nw_max ? nw_max : nw;
    if (nw < 1) return PCGRL_EINVAL;
    const size_t lds = (size_t)nw * per_wave;
    { const int rc = lds_cap<k_big<PROB>>(h->device, lds); if (rc) return rc; }
    const int grid = grid_for(P.num_envs, nw, 2048);
    const int gen = (P.random_start || !h->has_old) ? 1 : 0;
    hipLaunchKernelGGL((k_big<PROB>), dim3(grid), dim3(nw * 64), lds, st, P, h->B, list, parity, mode, clr, inline_reset, gen, park_list);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
static int launch_big(pcgrl_env* h, int list, int parity, int mode, int clr, int inline_reset, int park_list, hipStream_t st) {
    switch (h->P.prob) {
        case PCGRL_PROB_BINARY: return launch_big_p<PCGRL_PROB_BINARY>(h, list, parity, mode, clr, inline_reset, park_list, st);
        case PCGRL_PROB_ZELDA: return launch_big_p<PCGRL_PROB_ZELDA>(h, list, parity, mode, clr, inline_reset, park_list, st);
        case PCGRL_PROB_MDUNGEON: return launch_big_p<PCGRL_PROB_MDUNGEON>(h, list, parity, mode, clr, 0, park_list, st);
        case PCGRL_PROB_DDAVE: return launch_big_p<PCGRL_PROB_DDAVE>(h, list, parity, mode, clr, 0, park_list, st);
        default: return launch_big_p<PCGRL_PROB_SOKOBAN>(h, list, parity, mode, clr, 0, park_list, st);
    }
}
PCGRL_LOCAL int launch_stats(pcgrl_env* h, int list, int parity, int mode, int clr, int inline_reset, hipStream_t st) {
    if (h->P.big) return launch_big(h, list, parity, mode, clr, inline_reset, -1, st);
    switch (h->P.prob) {
        case PCGRL_PROB_BINARY: return launch_stats_p<PCGRL_PROB_BINARY>(h, list, parity, mode, clr, inline_reset, st);
        case PCGRL_PROB_ZELDA: return launch_stats_p<PCGRL_PROB_ZELDA>(h, list, parity, mode, clr, inline_reset, st);
        case PCGRL_PROB_MDUNGEON: return launch_stats_p<PCGRL_PROB_MDUNGEON>(h, list, parity, mode, clr, 0, st);
        case PCGRL_PROB_DDAVE: return launch_stats_p<PCGRL_PROB_DDAVE>(h, list, parity, mode, clr, 0, st);
        default: return launch_stats_p<PCGRL_PROB_SOKOBAN>(h, list, parity, mode, clr, 0, st);
    }
}

#endif  // PART_STATS (continued below: launch_reset, launch_planes_from_map)

#if PCGRL_IN_PART(PART_UPDATE)
template <class MaskT>
static int launch_update_m(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st) {
    const PcgrlParams& P = h->P;
    // (a tick of pcgrl_step_async: the kernel's threads also look through the slots of the suspended searches, one each)
    const int nthreads = P.num_envs > h->B.async_nslots ? P.num_envs : h->B.async_nslots;
    const int grid = (nthreads + PCGRL_BLOCK - 1) / PCGRL_BLOCK;
    switch (P.rep) {
        case PCGRL_REP_NARROW:
            hipLaunchKernelGGL((k_update<PCGRL_REP_NARROW, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
        case PCGRL_REP_WIDE:
            hipLaunchKernelGGL((k_update<PCGRL_REP_WIDE, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
        case PCGRL_REP_TURTLE:
            hipLaunchKernelGGL((k_update<PCGRL_REP_TURTLE, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
        case PCGRL_REP_NARROW_CAST:
            hipLaunchKernelGGL((k_update_block<PCGRL_REP_NARROW_CAST, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
        case PCGRL_REP_NARROW_MULTI:
            hipLaunchKernelGGL((k_update_block<PCGRL_REP_NARROW_MULTI, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
        default:
            hipLaunchKernelGGL((k_update_block<PCGRL_REP_TURTLE_CAST, MaskT>), dim3(grid), dim3(PCGRL_BLOCK), 0, st, P, h->B, actions, parity); break;
    }
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}

PCGRL_LOCAL int launch_update(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st) {
    return h->P.mask_bytes == 4 ? launch_update_m<uint32_t>(h, actions, parity, st) : launch_update_m<uint64_t>(h, actions, parity, st);
}
#endif  // PART_UPDATE

// One fused launch per step (kernels_step.h) where it applies: binary, maps of at most 16 rows, single-cell
// representations, auto-reset with the in-kernel reset.  PCGRL_NO_FUSED=1 keeps the two-launch pipeline (A/B, tests).
static bool fused_step_applies(const pcgrl_env* h, bool rollout = false) {
    const PcgrlParams& P = h->P;
    // (zelda changes 7 of 8 environments per step -- ~18 wavefront tasks per 64 environments; with the block's state in LDS, the
    //  tasks handed out dynamically and 128 environments per block the fused kernel is ahead of the two-launch pipeline:
    //  34.0 vs 38.2 us/step on C3.  PCGRL_FUSED_ZELDA=0 keeps k_update + k_stats (A/B, tests).)
    const bool prob_ok = P.prob == PCGRL_PROB_BINARY || (P.prob == PCGRL_PROB_ZELDA && (rollout || h->fused_zelda));
    return prob_ok && P.group == 16 && !P.big && P.rep <= PCGRL_REP_TURTLE && P.auto_reset &&
           h->B.inline_reset && !h->no_fused;
}
// Environments per block of k_step: 64 (four wavefronts), 128 (eight) or 256 (sixteen) -- chosen at pcgrl_bind from the batch
// size; PCGRL_STEP_EPB=64|128|256 overrides (A/B).
#if PCGRL_IN_PART(PART_STEP_BINARY) || PCGRL_IN_PART(PART_STEP_ZELDA)
template <int PROB, class MaskT, int EPB>
static int launch_step_pme(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st, const RolloutArgs& R) {
    const PcgrlParams& P = h->P;
    // the block's state copy (kernels_step.h) + per wavefront an MT19937 ring and the tile bytes of a map (in-kernel resets)
    const StepLds SL = step_lds_layout(16 * P.nplanes * (int)sizeof(MaskT), PROB == PCGRL_PROB_BINARY ? 16 * (int)sizeof(MaskT) : 0,
                                       P.rep == PCGRL_REP_NARROW, P.rep == PCGRL_REP_WIDE ? 3 : 1, EPB);
    const size_t lds = (size_t)SL.total + (EPB / 16) * (size_t)(PCGRL_MT_N * 4 + ((P.width * P.height + 15) & ~15));
    const int grid = (P.num_envs + EPB - 1) / EPB;
    const int gen = (P.random_start || !h->has_old) ? 1 : 0;
#define PCGRL_LAUNCH_STEP(REPV, MULTI, OBSV) do { \
        { const int rca = lds_cap<k_step<PROB, REPV, MaskT, MULTI, EPB, OBSV>>(h->device, lds); if (rca) return rca; } \
        hipLaunchKernelGGL((k_step<PROB, REPV, MaskT, MULTI, EPB, OBSV>), dim3(grid), dim3(EPB * 4), lds, st, P, h->B, actions, \
                           parity, gen, R.steps, R.action_stride, R.reward_out, R.done_out, R.info_out); } while (0)
    const bool multi = R.steps > 1 || R.reward_out || R.done_out || R.info_out;
    // a single step with a bound observation of a lean shape: the instantiation that writes the images while it runs (32-bit row masks)
    constexpr bool kObsKernel = sizeof(MaskT) == 4;
    const bool obs = kObsKernel && !multi && h->B.obs.out && h->B.obs.fused && !h->obs_at_end;
#define PCGRL_LAUNCH_STEP3(REPV) do { if (multi) PCGRL_LAUNCH_STEP(REPV, true, false); else if (obs) PCGRL_LAUNCH_STEP(REPV, false, kObsKernel); \
                                      else PCGRL_LAUNCH_STEP(REPV, false, false); } while (0)
    switch (P.rep) {
        case PCGRL_REP_NARROW: PCGRL_LAUNCH_STEP3(PCGRL_REP_NARROW); break;
        case PCGRL_REP_WIDE: PCGRL_LAUNCH_STEP3(PCGRL_REP_WIDE); break;
        default: PCGRL_LAUNCH_STEP3(PCGRL_REP_TURTLE); break;
    }
#undef PCGRL_LAUNCH_STEP3
#undef PCGRL_LAUNCH_STEP
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
template <int PROB, class MaskT>
static int launch_step_pm(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st, const RolloutArgs& R) {
    // (the 128-environment form is built for 32-bit row masks only: the common case)
    if (sizeof(MaskT) == 4 && h->step_epb == 128) return launch_step_pme<PROB, uint32_t, 128>(h, actions, parity, st, R);
    if (sizeof(MaskT) == 4 && h->step_epb == 256) return launch_step_pme<PROB, uint32_t, 256>(h, actions, parity, st, R);
    return launch_step_pme<PROB, MaskT, 64>(h, actions, parity, st, R);
}
#endif
#if PCGRL_IN_PART(PART_STEP_BINARY)
PCGRL_LOCAL int launch_step_binary(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st, const RolloutArgs& R) {
    return h->P.mask_bytes == 4 ? launch_step_pm<PCGRL_PROB_BINARY, uint32_t>(h, actions, parity, st, R) : launch_step_pm<PCGRL_PROB_BINARY, uint64_t>(h, actions, parity, st, R);
}
#endif
#if PCGRL_IN_PART(PART_STEP_ZELDA)
PCGRL_LOCAL int launch_step_zelda(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st, const RolloutArgs& R) {
    return h->P.mask_bytes == 4 ? launch_step_pm<PCGRL_PROB_ZELDA, uint32_t>(h, actions, parity, st, R) : launch_step_pm<PCGRL_PROB_ZELDA, uint64_t>(h, actions, parity, st, R);
}
#endif
static int launch_step(pcgrl_env* h, const int32_t* actions, int parity, hipStream_t st, const RolloutArgs& R = RolloutArgs{1, 0, nullptr, nullptr, nullptr}) {
    return h->P.prob == PCGRL_PROB_BINARY ? launch_step_binary(h, actions, parity, st, R) : launch_step_zelda(h, actions, parity, st, R);
}
static int action_width(int rep) {   // int32 values per environment and step
    return rep == PCGRL_REP_WIDE ? 3 : (rep == PCGRL_REP_NARROW_CAST || rep == PCGRL_REP_TURTLE_CAST) ? 2 : rep == PCGRL_REP_NARROW_MULTI ? 9 : 1;
}

// One solver launch: the jobs of list_a (mode_a) and, if list_b >= 0, of list_b (mode_b).  `slot` selects the
// scheduling words (two launches per step), zeroed here.  Episodes the solver ends go to rst_list.
#define SMB_LDS_BUDGET (160 * 1024 - 2048)   /* dynamic LDS of a k_smb block: the attribute set at pcgrl_bind and the launch size both come from here */
#if PCGRL_IN_PART(PART_SMB)
PCGRL_LOCAL int smb_device_setup(pcgrl_env*) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_smb<0>), hipFuncAttributeMaxDynamicSharedMemorySize, SMB_LDS_BUDGET));
    return PCGRL_OK;
}
PCGRL_LOCAL int launch_smb(pcgrl_env* h, int32_t* sync, int list_a, int mode_a, int list_b, int mode_b, int parity, int rst_list, int clr, hipStream_t st, int inline_reset) {
    // per wavefront: the heap (smb_search: its first levels; a deeper heap continues in the arena) + the visited bitmap;
    // as many wavefronts per block (one block per compute unit) as the LDS budget holds, SMB_MAX_WAVES at most
    int heap_n = 4 * h->P.solver_power + 4 < h->smb_heap ? ((4 * h->P.solver_power + 4 + 3) & ~3) : h->smb_heap;
    const int reset_words = PCGRL_MT_N + ((h->P.width * h->P.height + 15) & ~15) / 4;       // the in-kernel reset stages its ring and tiles there
    if (heap_n < reset_words) heap_n = (reset_words + 3) & ~3;
    const size_t vis_words = ((size_t)((h->P.width + 6) * (h->P.height + SMB_YOFF + 1) * 5 + 31) / 32 + 3) & ~(size_t)3;
    const size_t per_wave = ((size_t)heap_n + vis_words) * 4;
    int nw = (int)(SMB_LDS_BUDGET / per_wave);
    nw = nw > SMB_MAX_WAVES ? SMB_MAX_WAVES : (nw < 1 ? 1 : nw);
    if ((size_t)nw * per_wave > SMB_LDS_BUDGET) return PCGRL_EINVAL;      // one search does not fit a compute unit's LDS
    const int gen = (h->P.random_start || !h->has_old) ? 1 : 0;
    // (a step's changed levels come on two lists: WL_INC = the ones expected to take long, first; see k_update)
    hipLaunchKernelGGL(k_smb<0>, dim3(SOK_BLOCKS), dim3(nw * 64), nw * per_wave, st, h->P, h->B, (list_a == WL_CHG && mode_a == MODE_STEP) ? (int)WL_INC : -1,
                       list_a, mode_a, list_b, mode_b, parity, rst_list, sync, clr, heap_n, inline_reset, gen);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
#endif  // PART_SMB
#if PCGRL_IN_PART(PART_SEARCH)
PCGRL_LOCAL int search_device_setup(pcgrl_env* h) {   // the search kernels use most of a compute unit's LDS (heap + 64-bit-key table)
    const int lds = (int)((SOK_LDS_HEAP + 2 * SOK_LDS_TABLE) * 4);
    const void* f = h->cfg.prob == PCGRL_SOKOBAN ? reinterpret_cast<const void*>(k_sokoban<0>)
                  : h->cfg.prob == PCGRL_MDUNGEON ? reinterpret_cast<const void*>(k_mdungeon<0>) : reinterpret_cast<const void*>(k_ddave<0>);
    HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const void* fa = h->cfg.prob == PCGRL_SOKOBAN ? reinterpret_cast<const void*>(k_search_async<PCGRL_PROB_SOKOBAN>)
                   : h->cfg.prob == PCGRL_MDUNGEON ? reinterpret_cast<const void*>(k_search_async<PCGRL_PROB_MDUNGEON>) : reinterpret_cast<const void*>(k_search_async<PCGRL_PROB_DDAVE>);
    HIPCHK(hipFuncSetAttribute(fa, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return PCGRL_OK;
}
// one launch of a tick of pcgrl_step_async (kernels_search_async.h).  small: the launch for the fresh jobs -- a few KB of LDS per
// block, ASYNC_SMALL_BLOCKS blocks (several per compute unit) --, else the full search region and one block per compute unit
PCGRL_LOCAL int launch_search_async(pcgrl_env* h, int32_t* tickets, int list_a, int mode_a, int list_b, int mode_b, int parity, int rst_list, int clr, int budget, int resume, int small, hipStream_t st) {
    const int toff = small ? ASYNC_SMALL_HEAP : SOK_LDS_HEAP, tsize = small ? ASYNC_SMALL_TABLE : SOK_LDS_TABLE;
    const size_t lds = (size_t)(toff + 2 * tsize) * 4;
    const int grid = small ? ASYNC_SMALL_BLOCKS : SOK_BLOCKS;
    if (h->P.prob == PCGRL_PROB_DDAVE)
        hipLaunchKernelGGL(k_search_async<PCGRL_PROB_DDAVE>, dim3(grid), dim3(128), lds, st, h->P, h->B, h->async, list_a, mode_a, list_b, mode_b, parity, rst_list, tickets, clr, budget, resume, toff, tsize, small);
    else if (h->P.prob == PCGRL_PROB_MDUNGEON)
        hipLaunchKernelGGL(k_search_async<PCGRL_PROB_MDUNGEON>, dim3(grid), dim3(128), lds, st, h->P, h->B, h->async, list_a, mode_a, list_b, mode_b, parity, rst_list, tickets, clr, budget, resume, toff, tsize, small);
    else
        hipLaunchKernelGGL(k_search_async<PCGRL_PROB_SOKOBAN>, dim3(grid), dim3(128), lds, st, h->P, h->B, h->async, list_a, mode_a, list_b, mode_b, parity, rst_list, tickets, clr, budget, resume, toff, tsize, small);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
template <int PROB>
static int launch_search_big_p(pcgrl_env* h, int32_t* sync, int list_a, int mode_a, int list_b, int mode_b, int parity, int rst_list, int clr, hipStream_t st) {
    const BigArenaDims& D = h->big_dims;
    const BigSearchArena A = {h->B.big_arena, D.block_bytes, D.heap_off, D.table_off, D.nodes_cap, D.tsize};
    const int cells = (h->P.width + 2) * (h->P.height + 2);
    const size_t lds = (((size_t)cells * 4 + 15) & ~(size_t)15) + (size_t)(BIG_MAX_WORDS + SOKB_MAXC / 64) * 8;
    { const int rca = lds_cap<k_search_big<PROB>>(h->device, lds); if (rca) return rca; }       // (levels beyond ~15 000 cells: more than 64 KB of cell coordinates)
    hipLaunchKernelGGL((k_search_big<PROB>), dim3(D.nblocks), dim3(64), lds, st, h->P, h->B, A, list_a, mode_a, list_b, mode_b, parity, rst_list, sync, clr);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
PCGRL_LOCAL int launch_search(pcgrl_env* h, int32_t* sync, int list_a, int mode_a, int list_b, int mode_b, int parity, int rst_list, int clr, hipStream_t st) {
    if (h->P.big_search) {
        if (h->P.prob == PCGRL_PROB_DDAVE) return launch_search_big_p<PCGRL_PROB_DDAVE>(h, sync, list_a, mode_a, list_b, mode_b, parity, rst_list, clr, st);
        if (h->P.prob == PCGRL_PROB_MDUNGEON) return launch_search_big_p<PCGRL_PROB_MDUNGEON>(h, sync, list_a, mode_a, list_b, mode_b, parity, rst_list, clr, st);
        return launch_search_big_p<PCGRL_PROB_SOKOBAN>(h, sync, list_a, mode_a, list_b, mode_b, parity, rst_list, clr, st);
    }
    const size_t lds = h->B.sok_use_lds ? (size_t)(SOK_LDS_HEAP + 2 * SOK_LDS_TABLE) * 4 : 0;   // heap + 64-bit-key table
    if (h->P.prob == PCGRL_PROB_DDAVE)
        hipLaunchKernelGGL(k_ddave<0>, dim3(SOK_BLOCKS), dim3(128), lds, st, h->P, h->B, list_a, mode_a, list_b, mode_b, parity, rst_list, sync, clr);
    else if (h->P.prob == PCGRL_PROB_MDUNGEON)
        hipLaunchKernelGGL(k_mdungeon<0>, dim3(SOK_BLOCKS), dim3(128), lds, st, h->P, h->B, list_a, mode_a, list_b, mode_b, parity, rst_list, sync, clr);
    else
        hipLaunchKernelGGL(k_sokoban<0>, dim3(SOK_BLOCKS), dim3(128), lds, st, h->P, h->B, list_a, mode_a, list_b, mode_b, parity, rst_list,
                           sync, sync + SOK_SY_WORDS, clr);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
int pcgrl_selftest_heap(const uint32_t* ops, int32_t n_ops, uint32_t* pops, uint32_t* heap_out, int32_t* n_out, void* stream) {
    if (!ops || !pops || !heap_out || !n_out || n_ops < 0) return PCGRL_EINVAL;
    const int cap = 16384;                               // 64 KB of LDS: fifteen levels, deeper than any search's heap
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_selftest_heap<0>), hipFuncAttributeMaxDynamicSharedMemorySize, cap * 4));
    hipLaunchKernelGGL(k_selftest_heap<0>, dim3(1), dim3(64), (size_t)cap * 4, (hipStream_t)stream, ops, n_ops, cap, pops, heap_out, n_out);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
#endif  // PART_SEARCH
static int launch_solver(pcgrl_env* h, int slot, int list_a, int mode_a, int list_b, int mode_b, int parity, int rst_list, int clr,
                         hipStream_t st, int inline_reset = 0) {
    int32_t* sync = h->B.sok_sync + (size_t)slot * (SOK_SY_WORDS + SOK_HARD_CAP);
    HIPCHK(hipMemsetAsync(sync, 0, (size_t)(SOK_SY_WORDS + SOK_HARD_CAP) * 4, st));
    if (h->P.prob == PCGRL_PROB_SMB) return launch_smb(h, sync, list_a, mode_a, list_b, mode_b, parity, rst_list, clr, st, inline_reset);
    return launch_search(h, sync, list_a, mode_a, list_b, mode_b, parity, rst_list, clr, st);
}

#if PCGRL_IN_PART(PART_STATS)
template <int PROB>
static int launch_reset_p(pcgrl_env* h, int list, int park_list, int parity, int clr, hipStream_t st) {
    const PcgrlParams& P = h->P;
    const int cells = P.width * P.height;
    const size_t lds = 4 * (size_t)(PCGRL_MT_N * 4 + ((cells + 15) & ~15));
    const int grid = grid_for(P.num_envs, 4, h->was_reset ? 512 : 4096);
    const int gen = (P.random_start || !h->has_old) ? 1 : 0;
    if (P.group == 16 && P.mask_bytes == 4)
        hipLaunchKernelGGL((k_reset<PROB, 16, uint32_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, list, park_list, parity, gen, clr);
    else if (P.group == 16)
        hipLaunchKernelGGL((k_reset<PROB, 16, uint64_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, list, park_list, parity, gen, clr);
    else if (P.mask_bytes == 4)
        hipLaunchKernelGGL((k_reset<PROB, 64, uint32_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, list, park_list, parity, gen, clr);
    else
        hipLaunchKernelGGL((k_reset<PROB, 64, uint64_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, list, park_list, parity, gen, clr);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
// map generation + start stats of every environment on the reset list
PCGRL_LOCAL int launch_reset(pcgrl_env* h, int list, int park_list, int parity, int clr, hipStream_t st) {
    if (h->P.big) return launch_big(h, list, parity, MODE_START, clr, 0, park_list, st);
    switch (h->P.prob) {
        case PCGRL_PROB_BINARY: return launch_reset_p<PCGRL_PROB_BINARY>(h, list, park_list, parity, clr, st);
        case PCGRL_PROB_ZELDA: return launch_reset_p<PCGRL_PROB_ZELDA>(h, list, park_list, parity, clr, st);
        case PCGRL_PROB_MDUNGEON: return launch_reset_p<PCGRL_PROB_MDUNGEON>(h, list, park_list, parity, clr, st);
        case PCGRL_PROB_DDAVE: return launch_reset_p<PCGRL_PROB_DDAVE>(h, list, park_list, parity, clr, st);
        case PCGRL_PROB_SMB: return launch_reset_p<PCGRL_PROB_SMB>(h, list, park_list, parity, clr, st);
        default: return launch_reset_p<PCGRL_PROB_SOKOBAN>(h, list, park_list, parity, clr, st);
    }
}

PCGRL_LOCAL int launch_planes_from_map(pcgrl_env* h, const uint8_t* maps, hipStream_t st) {     // pcgrl_set_maps
    const PcgrlParams& P = h->P;
    if (P.big) {            // no planes to rebuild: the byte maps are the state
        const size_t total = (size_t)P.num_envs * P.width * P.height;
        const size_t g = (total + 255) / 256;
        hipLaunchKernelGGL(k_copy_map<0>, dim3((unsigned)(g < 16384 ? g : 16384)), dim3(256), 0, st, P, h->B, maps);
        HIPCHK(hipGetLastError());
        return PCGRL_OK;
    }
    const size_t lds = 4 * (size_t)((P.width * P.height + 15) & ~15);
    const int grid = grid_for(P.num_envs, 4, 4096);
    if (P.mask_bytes == 4)
        hipLaunchKernelGGL((k_planes_from_map<uint32_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, maps);
    else
        hipLaunchKernelGGL((k_planes_from_map<uint64_t>), dim3(grid), dim3(PCGRL_BLOCK), lds, st, P, h->B, maps);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
#endif  // PART_STATS

// pcgrl_rollout for the search problems: persistent blocks that own their environments for the whole tape (kernels_step_solver.h)
static bool solver_rollout_applies(const pcgrl_env* h, int* envs_per_block) {
    const PcgrlParams& P = h->P;
    // (P.big: maps wider than 64 cells take the general map path and keep no bit planes; P.big_search: the general searches)
    if (!solver_prob(P.prob) || P.prob == PCGRL_PROB_SMB || P.rep > PCGRL_REP_TURTLE || !P.auto_reset || !h->B.sok_use_lds || P.group != 16 || h->no_fused || P.big ||
        P.big_search) return false;
    int epb = 64 * ((P.num_envs + SOK_BLOCKS * 64 - 1) / (SOK_BLOCKS * 64));      // one block per compute unit when the batch is large enough
    epb = epb < 64 ? 64 : epb;
    if (epb > WL_LOCAL_CAP) return false;                                            // more than 256 x 512 environments: the sequence of steps
    *envs_per_block = epb;
    return true;
}
#if PCGRL_IN_PART(PART_STEP_SOLVER)
template <int PROB, int REP, class MaskT>
static int launch_step_solver_t(pcgrl_env* h, const int32_t* actions, hipStream_t st, const RolloutArgs& R, int epb) {
    const size_t lds = (size_t)(SOK_LDS_HEAP + 2 * SOK_LDS_TABLE) * 4;
    { const int rca = lds_cap<k_step_solver<PROB, REP, MaskT>>(h->device, lds); if (rca) return rca; }
    const int grid = (h->P.num_envs + epb - 1) / epb;
    const int gen = (h->P.random_start || !h->has_old) ? 1 : 0;
    hipLaunchKernelGGL((k_step_solver<PROB, REP, MaskT>), dim3(grid), dim3(SS_THREADS), lds, st, h->P, h->B, actions, gen, R.steps, R.action_stride, epb,
                       R.reward_out, R.done_out, R.info_out);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}
template <int PROB, class MaskT>
static int launch_step_solver_p(pcgrl_env* h, const int32_t* actions, hipStream_t st, const RolloutArgs& R, int epb) {
    switch (h->P.rep) {
        case PCGRL_REP_NARROW: return launch_step_solver_t<PROB, PCGRL_REP_NARROW, MaskT>(h, actions, st, R, epb);
        case PCGRL_REP_WIDE: return launch_step_solver_t<PROB, PCGRL_REP_WIDE, MaskT>(h, actions, st, R, epb);
        default: return launch_step_solver_t<PROB, PCGRL_REP_TURTLE, MaskT>(h, actions, st, R, epb);
    }
}
PCGRL_LOCAL int launch_step_solver(pcgrl_env* h, const int32_t* actions, hipStream_t st, const RolloutArgs& R, int epb) {
    const bool m4 = h->P.mask_bytes == 4;
    switch (h->P.prob) {
        case PCGRL_PROB_SOKOBAN: return m4 ? launch_step_solver_p<PCGRL_PROB_SOKOBAN, uint32_t>(h, actions, st, R, epb) : launch_step_solver_p<PCGRL_PROB_SOKOBAN, uint64_t>(h, actions, st, R, epb);
        case PCGRL_PROB_MDUNGEON: return m4 ? launch_step_solver_p<PCGRL_PROB_MDUNGEON, uint32_t>(h, actions, st, R, epb) : launch_step_solver_p<PCGRL_PROB_MDUNGEON, uint64_t>(h, actions, st, R, epb);
        default: return m4 ? launch_step_solver_p<PCGRL_PROB_DDAVE, uint32_t>(h, actions, st, R, epb) : launch_step_solver_p<PCGRL_PROB_DDAVE, uint64_t>(h, actions, st, R, epb);
    }
}

#endif  // PART_STEP_SOLVER

#if PCGRL_IN_PART(PART_CORE)
// the wrapped observation of every environment with the stand-alone kernel (kernels_obs.h)
static bool obs_is_huge(const ObsSpec& S) { return (size_t)OBS_EPB * S.oh * S.ow * S.depth >= ((size_t)1 << 24) / 4; }   // offsets inside a block's stretch stay small (obs_div)
static bool obs_same_spec(const ObsSpec& a, const ObsSpec& b) {
    return a.out == b.out && a.oh == b.oh && a.ow == b.ow && a.depth == b.depth && a.centered == b.centered && a.pad == b.pad;
}
static int launch_obs(pcgrl_env* h, const ObsSpec& S, hipStream_t st) {
    const PcgrlParams& P = h->P;
    const int grid = (P.num_envs + OBS_EPB - 1) / OBS_EPB;
    // from the row bit planes (staged in LDS) where the lean routines of kernels_obs.h apply -- binary tile ids, one-hot over
    // eight tiles --, else from the byte map
    const size_t lds = (size_t)OBS_EPB * (P.group * P.nplanes * P.mask_bytes + 2);
    if (obs_is_huge(S)) {                  // beyond k_obs's 24-bit block offsets: the 64-bit stream (kernels_obs.h)
        const size_t pieces = ((size_t)P.num_envs * S.oh * S.ow * S.depth + 15) >> 4;
        const size_t g = (pieces + 255) / 256;
        hipLaunchKernelGGL(k_obs_huge<0>, dim3((unsigned)(g < 65536 ? g : 65536)), dim3(256), 0, st, P, h->B, S);
    } else if (P.nplanes == 1 && P.mask_bytes == 4 && S.depth == 1) hipLaunchKernelGGL(k_obs<1>, dim3(grid), dim3(256), lds, st, P, h->B, S);
    else if (P.nplanes == 1 && P.mask_bytes == 8 && S.depth == 1) hipLaunchKernelGGL(k_obs<2>, dim3(grid), dim3(256), lds, st, P, h->B, S);
    else if (P.nplanes == 3 && P.mask_bytes == 4 && S.depth == 8) hipLaunchKernelGGL(k_obs<3>, dim3(grid), dim3(256), lds, st, P, h->B, S);
    else hipLaunchKernelGGL(k_obs<0>, dim3(grid), dim3(256), 0, st, P, h->B, S);
    HIPCHK(hipGetLastError());
    // a full image of the current state in the bound target -- only when it was written with the bound geometry (pcgrl_observe
    // may be handed the same tensor with another window: the target then does not hold what an in-place update builds on)
    if (obs_same_spec(S, h->B.obs)) h->obs_synced = S.out;
    else if (S.out == h->B.obs.out) h->obs_synced = nullptr;
    return PCGRL_OK;
}
static int obs_spec(const pcgrl_env* h, uint8_t* out, int32_t out_h, int32_t out_w, int32_t centered, int32_t pad_value, int32_t onehot, ObsSpec* S) {
    if (out_h < 1 || out_w < 1 || out_h > 4096 || out_w > 4096 || pad_value < 0 || pad_value > 255) return PCGRL_EINVAL;
    if (centered && h->P.rep == PCGRL_REP_WIDE) return PCGRL_EINVAL;   // Cropped needs a cursor (wrappers.py:170)
    if (((uintptr_t)out & 15) != 0) return PCGRL_EINVAL;
    const int depth = onehot ? h->P.ntiles : 1;
    *S = ObsSpec{out, out_h, out_w, depth, centered ? 1 : 0, pad_value, 0, 0};
    S->fused = obs_lean_mode(h->P.nplanes, h->P.mask_bytes == 4, h->P.width, out_h, out_w, depth, pad_value) != 0 ? 1 : 0;
    return PCGRL_OK;
}

extern "C" {

static int reset_one(pcgrl_env* h, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int n = h->P.num_envs, par = h->parity;
    hipLaunchKernelGGL(k_fill_all, dim3((n + 255) / 256), dim3(256), 0, st, h->B, n, par, (int)WL_RST);
    HIPCHK(hipGetLastError());
    const bool sok = solver_prob(h->P.prob);
    int rc = launch_reset(h, WL_RST, WL_SOL2, par, sok ? -1 : (par ^ 1), st);
    if (rc) return rc;
    if (sok && (rc = launch_solver(h, 0, WL_SOL2, MODE_START, -1, 0, par, WL_RST2, par ^ 1, st))) return rc;
    if (h->B.obs.out && (rc = launch_obs(h, h->B.obs, st))) return rc;
    return PCGRL_OK;
}

// *used_lists: the step went through the global work lists (every path but the fused kernel).  Invariant of the handle:
// at the start of every call the counters of h->parity are zero.  A pass through the lists leaves them dirty and has its last
// kernel zero the other parity's, so the caller flips h->parity after it; the fused kernels never touch the lists and
// must NOT flip it (an odd number of fused steps followed by a list step would otherwise land on uncleared counters).
static int step_one(pcgrl_env* h, const int32_t* actions, void* stream, bool* used_lists) {
    hipStream_t st = (hipStream_t)stream;
    const int par = h->parity;
    int rc;
    *used_lists = true;
    if ((rc = prof_mark(h, st))) return rc;
    if (fused_step_applies(h)) {      // the whole step in one launch
        *used_lists = false;
        if ((rc = launch_step(h, actions, par, st))) return rc;
        for (int k = 0; k < 6; k++) if ((rc = prof_mark(h, st))) return rc;
        return PCGRL_OK;
    }
    if ((rc = launch_update(h, actions, par, st))) return rc;
    if ((rc = prof_mark(h, st))) return rc;
    // The last kernel of the step clears the other parity's work-list counters.  Every problem but Sokoban is
    // two launches: k_stats also resets the environments whose episode ended (auto_reset).
    const bool sok = solver_prob(h->P.prob), ar = h->P.auto_reset != 0;
    if (!sok) {
        const bool inl = ar && h->B.inline_reset;
        rc = launch_stats(h, WL_CHG, par, MODE_STEP, (ar && !inl) ? -1 : (par ^ 1), inl ? 1 : 0, st);
        if (rc) return rc;
        if ((rc = prof_mark(h, st))) return rc;
        if ((rc = prof_mark(h, st))) return rc;
        if (ar && !inl && (rc = launch_reset(h, WL_RST, WL_SOL2, par, par ^ 1, st))) return rc;
        for (int k = 0; k < 3; k++) if ((rc = prof_mark(h, st))) return rc;
        return PCGRL_OK;
    }
    // Sokoban: k_stats parks the maps that need the solver (SOL) and sends finished episodes to RST; k_reset
    // regenerates those and parks their solver jobs (SOL2); ONE solver launch then works on SOL and SOL2
    // together, so that the step waits for its slowest search once, not twice.  The few episodes that only the
    // solver could end (RST2) get a second, almost empty reset + solver pass.
    // (smb has no separate statistics pass: k_smb computes everything from the byte map, so its changed list is the job list)
    const bool smb = h->P.prob == PCGRL_PROB_SMB;
    if (!smb && (rc = launch_stats(h, WL_CHG, par, MODE_STEP, -1, 0, st))) return rc;
    if ((rc = prof_mark(h, st))) return rc;
    if (ar && (rc = launch_reset(h, WL_RST, WL_SOL2, par, -1, st))) return rc;
    if ((rc = prof_mark(h, st))) return rc;
    if (smb && ar) {
        // k_smb resets the environments whose episode its play-through ends itself (most levels can be won, and winning ends
        // the episode): one launch, the last of the step
        if ((rc = launch_solver(h, 0, WL_CHG, MODE_STEP, WL_SOL2, MODE_START, par, WL_RST2, par ^ 1, st, 1))) return rc;
        for (int k = 0; k < 3; k++) if ((rc = prof_mark(h, st))) return rc;
        return PCGRL_OK;
    }
    if ((rc = launch_solver(h, 0, smb ? WL_CHG : WL_SOL, MODE_STEP, ar ? WL_SOL2 : -1, MODE_START, par, WL_RST2, ar ? -1 : (par ^ 1), st))) return rc;
    if ((rc = prof_mark(h, st))) return rc;
    if (ar && (rc = launch_reset(h, WL_RST2, WL_SOL3, par, -1, st))) return rc;
    if ((rc = prof_mark(h, st))) return rc;
    if (ar && (rc = launch_solver(h, 1, WL_SOL3, MODE_START, -1, 0, par, WL_RST2, par ^ 1, st))) return rc;
    if ((rc = prof_mark(h, st))) return rc;
    return PCGRL_OK;
}

// ---- pcgrl_step_async (kernels_search_async.h) ------------------------------------------------
static bool async_applies(const pcgrl_config* c) {
    return solver_prob(c->prob) && c->prob != PCGRL_SMB && !big_search(c) && !big_map(c) && c->solver_power <= SOK_LDS_POWER;
}
// head of the arena: pending [N] | counters + the shards of the first one | the tick's list of runnable slots
static size_t async_stats_bytes() { return 64 + ASYNC_NSHARD * 64; }
static size_t async_small_pool_bytes() { return align_up((size_t)ASYNC_SMALL_BLOCKS * ASYNC_SMALL_NODES * 16, 256); }
static size_t async_head_bytes(const pcgrl_config* c, int nslots) {       // ... | node pools of the small launch | its hand-over list
    return align_up((size_t)c->num_envs, 256) + async_stats_bytes() + align_up((size_t)nslots * 4, 256) + async_small_pool_bytes() + align_up((size_t)c->num_envs * 4, 256);
}
static size_t async_slot_bytes(const pcgrl_config* c) {
    return align_up((size_t)ASYNC_SLOT_HDR + (size_t)(4 * c->solver_power + 4) * 16 + (size_t)(SOK_LDS_HEAP + 2 * SOK_LDS_TABLE) * 4, 256);
}
size_t pcgrl_async_bytes(const pcgrl_config* c, int32_t nslots) {
    if (validate_config(c) != PCGRL_OK || nslots < 1 || !async_applies(c)) return 0;
    return async_head_bytes(c, nslots) + (size_t)nslots * async_slot_bytes(c);
}
int pcgrl_bind_async(pcgrl_env* h, void* arena, size_t bytes, int32_t nslots, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    if (!arena) { h->async_on = h->async_dirty = 0; return PCGRL_OK; }
    if (!async_applies(&h->cfg) || nslots < 1 || ((uintptr_t)arena & 255) != 0 || bytes < pcgrl_async_bytes(&h->cfg, nslots)) return PCGRL_EINVAL;
    if (h->alloc_solver_power != h->cfg.solver_power) return PCGRL_EINVAL;         // (the slots are cut for the bound solver_power)
    DeviceGuard guard(h->device);
    uint8_t* a = (uint8_t*)arena;
    AsyncCtl& A = h->async;
    A.pending = a;
    A.stats = (unsigned long long*)(a + align_up((size_t)h->cfg.num_envs, 256));
    A.runlist = (int32_t*)(a + align_up((size_t)h->cfg.num_envs, 256) + async_stats_bytes());
    A.small_pool = (uint8_t*)A.runlist + align_up((size_t)nslots * 4, 256);
    A.overflow = (int32_t*)(A.small_pool + async_small_pool_bytes());
    A.slots = a + async_head_bytes(&h->cfg, nslots);
    A.slot_bytes = async_slot_bytes(&h->cfg);
    A.nslots = nslots; A.nodes_cap = 4 * h->cfg.solver_power + 4; A.tick = 0; A.pad = 0;
    HIPCHK(hipMemsetAsync(a, 0, async_head_bytes(&h->cfg, nslots), (hipStream_t)stream));
    HIPCHK(hipMemset2DAsync(A.slots, A.slot_bytes, 0, ASYNC_SLOT_HDR, (size_t)nslots, (hipStream_t)stream));
    h->async_on = 1; h->async_dirty = 0;
    return PCGRL_OK;
}
// the pending steps are dropped (pcgrl_reset: every environment starts over)
static int async_drop(pcgrl_env* h, hipStream_t st) {
    if (!h->async_on || !h->async_dirty) return PCGRL_OK;
    HIPCHK(hipMemsetAsync(h->async.pending, 0, (size_t)h->cfg.num_envs, st));
    HIPCHK(hipMemset2DAsync(h->async.slots, h->async.slot_bytes, 0, ASYNC_SLOT_HDR, (size_t)h->async.nslots, st));
    h->async_dirty = 0;
    return PCGRL_OK;
}
// what the update kernel of a tick (and the collect kernel of a flush) needs of the arena: set for that one launch
static void async_devbufs(pcgrl_env* h, bool on) {
    DevBufs& B = h->B;
    const AsyncCtl& A = h->async;
    B.pending = on ? A.pending : nullptr; B.async_stats = on ? A.stats : nullptr;
    B.async_slots = on ? A.slots : nullptr; B.async_slot_bytes = A.slot_bytes; B.async_nslots = on ? A.nslots : 0;
    B.async_run = on ? A.runlist : nullptr; B.async_run_n = on ? h->B.sok_sync + 2 : nullptr;
}
// the searches of a tick: the fresh jobs of the step's lists in the small launch, then the suspended ones (and what the small launch
// could not take) in the full one (an episode a search ends is reset by the next tick).  async_split = 0: one full launch for all.
static int async_tick_searches(pcgrl_env* h, int budget, hipStream_t st) {
    const int par = h->parity;
    const bool ar = h->P.auto_reset != 0;
    int rc;
    h->async.tick = (h->async.tick + 1) & 0x3FFFFFFF;
    if (!h->async_split)
        return launch_search_async(h, h->B.sok_sync, WL_SOL, MODE_STEP, ar ? (int)WL_SOL2 : -1, MODE_START, par, -1, par ^ 1, budget, 1, 0, st);
    if ((rc = launch_search_async(h, h->B.sok_sync, WL_SOL, MODE_STEP, ar ? (int)WL_SOL2 : -1, MODE_START, par, -1, -1, budget, 0, 1, st))) return rc;
    return launch_search_async(h, h->B.sok_sync, -1, MODE_STEP, -1, MODE_START, par, -1, par ^ 1, budget, 1, 0, st);
}
// everything that is pending, to its end: the suspended searches, then the resets and searches of the episodes they ended
static int async_finish_all(pcgrl_env* h, hipStream_t st) {
    const int par = h->parity, n = h->P.num_envs, budget = 0x3FFFFFFF;
    const bool ar = h->P.auto_reset != 0;
    int rc;
    h->async.tick = (h->async.tick + 1) & 0x3FFFFFFF;
    int32_t* sync0 = h->B.sok_sync, *sync1 = h->B.sok_sync + (SOK_SY_WORDS + SOK_HARD_CAP);
    HIPCHK(hipMemsetAsync(sync0, 0, 8 * sizeof(int32_t), st));
    async_devbufs(h, true);
    {
        const int nn = n > h->async.nslots ? n : h->async.nslots;
        hipLaunchKernelGGL(k_async_collect, dim3((nn + 255) / 256), dim3(256), 0, st, h->B, h->async.pending, n, par, (int)WL_RST2);
    }
    async_devbufs(h, false);
    HIPCHK(hipGetLastError());
    if ((rc = launch_search_async(h, sync0, -1, MODE_STEP, -1, MODE_START, par, WL_RST2, ar ? -1 : (par ^ 1), budget, 1, 0, st))) return rc;
    if (ar) {
        if ((rc = launch_reset(h, WL_RST2, WL_SOL3, par, -1, st))) return rc;
        HIPCHK(hipMemsetAsync(sync1, 0, 8 * sizeof(int32_t), st));
        if ((rc = launch_search_async(h, sync1, WL_SOL3, MODE_START, -1, 0, par, WL_RST2, par ^ 1, budget, 0, 0, st))) return rc;
    }
    return PCGRL_OK;
}
int pcgrl_async_flush(pcgrl_env* h, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    if (!h->async_on || !h->async_dirty) return PCGRL_OK;
    DeviceGuard guard(h->device);
    int rc = async_finish_all(h, (hipStream_t)stream);
    if (rc) return rc;
    h->parity ^= 1;
    h->async_dirty = 0;
    if (h->B.obs.out && (rc = launch_obs(h, h->B.obs, (hipStream_t)stream))) return rc;
    return PCGRL_OK;
}
int pcgrl_step_async(pcgrl_env* h, const int32_t* actions, int32_t pop_budget, void* stream) {
    if (!h || !h->bound || !h->was_reset) return PCGRL_ESTATE;
    if (!actions || pop_budget < 1) return PCGRL_EINVAL;
    if (!h->async_on) return PCGRL_ESTATE;          // pcgrl_bind_async first
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    const int par = h->parity;
    const bool ar = h->P.auto_reset != 0;
    int rc;
    HIPCHK(hipMemsetAsync(h->B.sok_sync, 0, 8 * sizeof(int32_t), st));       // the search launches' tickets, the counts of runnable slots and of handed-over jobs
    async_devbufs(h, true);
    rc = launch_update(h, actions, par, st);        // (also lists the runnable slots)
    async_devbufs(h, false);
    if (rc) return rc;
    if ((rc = launch_stats(h, WL_CHG, par, MODE_STEP, -1, 0, st))) return rc;
    if (ar && (rc = launch_reset(h, WL_RST, WL_SOL2, par, -1, st))) return rc;
    if ((rc = async_tick_searches(h, pop_budget, st))) return rc;
    h->parity ^= 1;
    h->async_dirty = 1;
    if (h->B.obs.out) { h->B.obs.delta = 0; if ((rc = launch_obs(h, h->B.obs, st))) return rc; }
    return PCGRL_OK;
}

int pcgrl_reset(pcgrl_env* h, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    DeviceGuard guard(h->device);
    int rc = async_drop(h, (hipStream_t)stream);
    if (rc) return rc;
    rc = reset_one(h, stream);
    if (rc) return rc;
    h->parity ^= 1;
    h->has_old = 1;
    h->was_reset = 1;
    return PCGRL_OK;
}

int pcgrl_step(pcgrl_env* h, const int32_t* actions, void* stream) {
    if (!h || !h->bound || !h->was_reset) return PCGRL_ESTATE;
    if (!actions) return PCGRL_EINVAL;
    if (h->async_dirty) { const int rf = pcgrl_async_flush(h, stream); if (rf) return rf; }
    DeviceGuard guard(h->device);
    bool used_lists = true;
    h->B.obs.delta = (h->B.obs.out && h->B.obs.fused && h->obs_incremental && !h->obs_hold && h->obs_synced == h->B.obs.out) ? 1 : 0;
    int rc = step_one(h, actions, stream, &used_lists);
    if (rc) return rc;
    if (h->B.obs.out && !h->obs_hold) h->obs_synced = h->B.obs.out;      // (every path below leaves the image of the new state there)
    if (used_lists) h->parity ^= 1;
    if (h->profiling) h->prof_steps++;
    // the wrapped observation: the fused step kernel wrote it; every other pipeline gets one more launch
    if ((used_lists || !h->B.obs.fused) && h->B.obs.out && !h->obs_hold && (rc = launch_obs(h, h->B.obs, (hipStream_t)stream))) return rc;
    return PCGRL_OK;
}

// pcgrl_step on `count` handles (the shards of one batch: one per GPU of a node, or several on one GPU) in ONE call: step k is issued
// on every handle -- its device made current, on its own stream -- before the call returns; nothing is waited for.  A host thread
// that drives eight GPUs pays one foreign-function call per step instead of eight (node.MultiGpuPcgrlEnv; SURVEY 8e: the scaling
// risk of a 28 us step is host launch latency).  The handles are issued side by side by a pool of threads (step_pool.h).  Returns the
// first error seen (every handle has been tried).
int pcgrl_step_multi(pcgrl_env* const* envs, const int32_t* const* actions, void* const* streams, int32_t count) {
    if (!envs || !actions || !streams || count < 1) return PCGRL_EINVAL;
    if (count > 1) {       // the handles side by side on the issuing threads (step_pool.h); pcgrl_step_threads(0): all of them here
        StepPool* pool = StepPool::get(count);
        if (pool) return pool->step(&pcgrl_step, envs, actions, streams, count);
    }
    for (int i = 0; i < count; i++) {
        const int rc = pcgrl_step(envs[i], actions[i], streams[i]);
        if (rc) return rc;
    }
    return PCGRL_OK;
}

// pcgrl_selftest_range_reward: get_range_reward (helper.py:366-376) as the DEVICE evaluates it -- range_reward_i on `n` rows of
// (new value, old value, low, high), INT_MAX / INT_MIN standing for +-inf -- so that the reference's exhaustive table is held against
// the kernels' own integer form on the GPU and not only through trajectories.
int pcgrl_selftest_range_reward(const int32_t* rows, int32_t n, int32_t* out, void* stream) {
    if (!rows || !out || n < 0) return PCGRL_EINVAL;
    if (n == 0) return PCGRL_OK;
    hipLaunchKernelGGL(k_selftest_range_reward, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rows, n, out);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}

// How many issuing threads pcgrl_step_multi uses besides the caller's: n >= 0 sets it (0: none; at most 7; only before the first
// multi-handle call has made them), n < 0 only asks.  Returns the number in effect.
int32_t pcgrl_step_threads(int32_t n) {
    int eff = 0;
    StepPool::get(0, n, &eff);
    return eff;
}

// pcgrl_selftest_step_pool: the issuing threads of pcgrl_step_multi without a GPU -- `calls` calls of `count` stand-in handles; hits[i]
// counts how often stand-in i was stepped (the stand-in whose index is fail_at reports PCGRL_EINVAL).  Returns how many calls
// returned an error, -1 when the pool is switched off.
static int selftest_pool_fn(pcgrl_env* e, const int32_t* a, void*) {
    int32_t* hit = reinterpret_cast<int32_t*>(e);
    __atomic_fetch_add(hit, 1, __ATOMIC_RELAXED);
    if (a) g_last_hip = 9000 + *a;      // a stand-in "HIP error" of the failing handle: the caller's pcgrl_last_hip_error() must show it
    return a ? PCGRL_EINVAL : PCGRL_OK;
}
int pcgrl_selftest_step_pool(int32_t count, int32_t calls, int32_t fail_at, int32_t* hits) {
    if (count < 2 || count > 64 || calls < 1 || !hits) return PCGRL_EINVAL;
    StepPool* pool = StepPool::get(count);
    if (!pool) return -1;
    pcgrl_env* envs[64]; const int32_t* acts[64]; void* streams[64];
    const int32_t marker = fail_at;
    for (int i = 0; i < count; i++) { envs[i] = reinterpret_cast<pcgrl_env*>(hits + i); acts[i] = i == fail_at ? &marker : nullptr; streams[i] = nullptr; }
    int failed = 0;
    for (int c = 0; c < calls; c++) failed += pool->step(&selftest_pool_fn, envs, acts, streams, count) != PCGRL_OK;
    return failed;
}

// ActionMap.step + PcgrlEnv.step: where the fused step kernel applies the flat indices are decoded by its update wavefronts (one launch
// less: the decode kernel was 4-5 us of a 35 us step with the image); elsewhere k_action_map fills xyv and the step takes that.
int pcgrl_step_flat(pcgrl_env* h, const int32_t* flat, int32_t* xyv, void* stream) {
    if (!h || !h->bound || !h->was_reset) return PCGRL_ESTATE;
    if (!flat || !xyv || h->cfg.rep != PCGRL_REP_WIDE) return PCGRL_EINVAL;
    if (fused_step_applies(h, false)) {
        h->B.flat = flat;
        const int rc = pcgrl_step(h, xyv, stream);        // (xyv is not read: the action segment of the kernel's prefetch is off)
        h->B.flat = nullptr;
        return rc;
    }
    const int rc = pcgrl_action_map(h, flat, xyv, stream);
    return rc ? rc : pcgrl_step(h, xyv, stream);
}

// `steps` consecutive pcgrl_step calls on a tape of actions.  Where the fused step kernel applies this is ONE launch: a block
// of k_step owns 64 environments that depend on nothing outside the block, so it walks down the tape on its own.
int pcgrl_rollout(pcgrl_env* h, const int32_t* actions, int32_t steps, double* reward_out, uint8_t* done_out, int32_t* info_out, void* stream) {
    if (!h || !h->bound || !h->was_reset) return PCGRL_ESTATE;
    if (!actions || steps < 1) return PCGRL_EINVAL;
    if (h->async_dirty) { const int rf = pcgrl_async_flush(h, stream); if (rf) return rf; }
    DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)h->P.num_envs, stride = n * action_width(h->P.rep);
    h->obs_synced = nullptr;          // a tape ends with a full image of the state it ends in
    h->B.obs.delta = 0;
    if (fused_step_applies(h, true) && !h->profiling) {
        const RolloutArgs R = {steps, stride, reward_out, done_out, info_out};
        int rc = launch_step(h, actions, h->parity, st, R);     // no work lists, no parity flip (see step_one)
        if (rc == PCGRL_OK && h->B.obs.out && !h->B.obs.fused) rc = launch_obs(h, h->B.obs, st);
        if (rc == PCGRL_OK) h->obs_synced = h->B.obs.out;
        return rc;
    }
    int epb = 0;
    if (solver_rollout_applies(h, &epb) && !h->profiling) {
        const RolloutArgs R = {steps, stride, reward_out, done_out, info_out};
        int rc = launch_step_solver(h, actions, st, R, epb);       // block-local work lists: the global lists and their parity are not touched
        if (rc == PCGRL_OK && h->B.obs.out) rc = launch_obs(h, h->B.obs, st);
        return rc;
    }
    h->obs_hold = 1;            // one image at the end of the tape, not one per step
    // The step kernels write reward / done / info through the handle's pointers and never read an earlier step's: step t of the tape
    // writes its rows of the caller's [steps, N] outputs directly (no copy launch per step); the bound buffers get the last step's
    // values at the end, so that the state is what `steps` calls of pcgrl_step leave.
    double* const reward0 = h->B.reward; uint8_t* const done0 = h->B.done; int32_t* const info0 = h->B.info;
    int rc = PCGRL_OK;
    for (int t = 0; t < steps && rc == PCGRL_OK; t++) {
        if (reward_out) h->B.reward = reward_out + (size_t)t * n;
        if (done_out) h->B.done = done_out + (size_t)t * n;
        if (info_out) h->B.info = info_out + (size_t)t * n * 10;
        rc = pcgrl_step(h, actions + (size_t)t * stride, stream);
    }
    h->B.reward = reward0; h->B.done = done0; h->B.info = info0;
    h->obs_hold = 0;
    if (rc) return rc;
    {
        const size_t last = (size_t)(steps - 1);
        if (reward_out) HIPCHK(hipMemcpyAsync(reward0, reward_out + last * n, n * sizeof(double), hipMemcpyDeviceToDevice, st));
        if (done_out) HIPCHK(hipMemcpyAsync(done0, done_out + last * n, n, hipMemcpyDeviceToDevice, st));
        if (info_out) HIPCHK(hipMemcpyAsync(info0, info_out + last * n * 10, n * 10 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    }
    if (h->B.obs.out) return launch_obs(h, h->B.obs, st);
    return PCGRL_OK;
}

// Optional per-environment episode statistics, kept by the step kernels.  All four DEVICE pointers or all NULL
// (off, the default).  Call after pcgrl_bind; the buffers are zeroed here.
int pcgrl_bind_episode_stats(pcgrl_env* h, double* ep_return, int32_t* ep_length, double* last_return, int32_t* last_length,
                             void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    const int any = (ep_return != nullptr) + (ep_length != nullptr) + (last_return != nullptr) + (last_length != nullptr);
    if (any != 0 && any != 4) return PCGRL_EINVAL;
    DeviceGuard guard(h->device);
    const size_t n = (size_t)h->cfg.num_envs;
    if (any) {
        HIPCHK(hipMemsetAsync(ep_return, 0, n * 8, (hipStream_t)stream));
        HIPCHK(hipMemsetAsync(ep_length, 0, n * 4, (hipStream_t)stream));
        HIPCHK(hipMemsetAsync(last_return, 0, n * 8, (hipStream_t)stream));
        HIPCHK(hipMemsetAsync(last_length, 0, n * 4, (hipStream_t)stream));
    }
    h->B.ep_return = ep_return; h->B.ep_length = ep_length; h->B.last_return = last_return; h->B.last_length = last_length;
    return PCGRL_OK;
}

int pcgrl_profile(pcgrl_env* h, int enable) {
    if (!h) return PCGRL_EINVAL;
    h->profiling = enable ? 1 : 0;
    h->ev_used = 0;
    h->prof_steps = 0;
    return PCGRL_OK;
}

// Sums the per-phase GPU time (ms) of every step issued since pcgrl_profile(h, 1); synchronises.
int pcgrl_profile_read(pcgrl_env* h, double* phase_ms, int32_t* steps) {
    if (!h || !phase_ms || !steps) return PCGRL_EINVAL;
    for (int k = 0; k < PCGRL_NPHASE; k++) phase_ms[k] = 0.0;
    *steps = h->prof_steps;
    if (h->ev_used == 0) return PCGRL_OK;
    DeviceGuard guard(h->device);
    HIPCHK(hipEventSynchronize(h->events[h->ev_used - 1]));
    const size_t per = PCGRL_NPHASE + 1;
    for (size_t s0 = 0; s0 + per <= h->ev_used; s0 += per)
        for (int k = 0; k < PCGRL_NPHASE; k++) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, h->events[s0 + k], h->events[s0 + k + 1]));
            phase_ms[k] += ms;
        }
    return PCGRL_OK;
}

int pcgrl_observe(pcgrl_env* h, uint8_t* out, int32_t out_h, int32_t out_w, int32_t centered, int32_t pad_value, int32_t onehot, void* stream) {
    if (!h || !h->bound || !h->was_reset) return PCGRL_ESTATE;
    if (!out) return PCGRL_EINVAL;
    DeviceGuard guard(h->device);
    ObsSpec S;
    int rc = obs_spec(h, out, out_h, out_w, centered, pad_value, onehot, &S);
    if (rc) return rc;
    return launch_obs(h, S, (hipStream_t)stream);
}

int pcgrl_bind_observation(pcgrl_env* h, uint8_t* out, int32_t out_h, int32_t out_w, int32_t centered, int32_t pad_value, int32_t onehot,
                           int32_t incremental) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    if (!out) { h->B.obs = ObsSpec{nullptr, 0, 0, 0, 0, 0, 0, 0}; h->obs_synced = nullptr; return PCGRL_OK; }
    ObsSpec S;
    int rc = obs_spec(h, out, out_h, out_w, centered, pad_value, onehot, &S);
    if (rc) return rc;
    const ObsSpec& O = h->B.obs;
    if (!obs_same_spec(O, S)) h->obs_synced = nullptr;
    h->B.obs = S;
    h->obs_incremental = incremental ? 1 : 0;
    return PCGRL_OK;
}

int pcgrl_action_map(pcgrl_env* h, const int32_t* flat, int32_t* xyv, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    if (!flat || !xyv) return PCGRL_EINVAL;
    DeviceGuard guard(h->device);
    const int n = h->P.num_envs;
    hipLaunchKernelGGL(k_action_map, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, flat, xyv, n, h->P.width, h->P.height, h->P.ntiles, h->B.status);
    HIPCHK(hipGetLastError());
    return PCGRL_OK;
}

int pcgrl_status(pcgrl_env* h, void* stream, int32_t* status) {
    if (!h || !h->bound || !status) return PCGRL_ESTATE;
    DeviceGuard guard(h->device);
    HIPCHK(hipMemcpyAsync(status, h->B.status, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return PCGRL_OK;
}

int pcgrl_clear_status(pcgrl_env* h, void* stream) {
    if (!h || !h->bound) return PCGRL_ESTATE;
    DeviceGuard guard(h->device);
    HIPCHK(hipMemsetAsync(h->B.status, 0, sizeof(int32_t), (hipStream_t)stream));
    return PCGRL_OK;
}

static int set_maps_one(pcgrl_env* h, const uint8_t* maps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const PcgrlParams& P = h->P;
    const int n = P.num_envs, par = h->parity;
    // (a caller that restores a checkpoint rewrites the rings and cursors and then comes here: drop the draw cache)
    if (h->B.fifo_tag) HIPCHK(hipMemsetAsync(h->B.fifo_tag, 0xFF, (size_t)n * 4, st));
    int rc0 = launch_planes_from_map(h, maps, st);
    if (rc0) return rc0;
    hipLaunchKernelGGL(k_fill_all, dim3((n + 255) / 256), dim3(256), 0, st, h->B, n, par, (int)WL_CHG);
    HIPCHK(hipGetLastError());
    const bool sok = solver_prob(P.prob), smb = P.prob == PCGRL_PROB_SMB;
    int rc = smb ? PCGRL_OK : launch_stats(h, WL_CHG, par, MODE_SETMAP, sok ? -1 : (par ^ 1), 0, st);
    if (rc) return rc;
    if (sok && (rc = launch_solver(h, 0, smb ? WL_CHG : WL_SOL2, MODE_SETMAP, -1, 0, par, WL_RST2, par ^ 1, st))) return rc;
    if (h->B.obs.out && (rc = launch_obs(h, h->B.obs, st))) return rc;
    return PCGRL_OK;
}

#if defined(PCGRL_TIMELINE) || defined(PCGRL_SMB_PROF) || defined(PCGRL_BIG_PROF) || defined(PCGRL_WIDE_TL)
// debug build only (tools/timeline.py, tools/smb_prof.py): where the kernels write their timeline marks (NULL: off)
int pcgrl_debug_timeline(void* buf) {
    unsigned long long* p = (unsigned long long*)buf;
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tl_buf), &p, sizeof(p)));
    return PCGRL_OK;
}
#endif

int pcgrl_set_maps(pcgrl_env* h, const uint8_t* maps, void* stream) {
    if (!h || !h->bound || !h->was_reset) return PCGRL_ESTATE;
    if (!maps) return PCGRL_EINVAL;
    DeviceGuard guard(h->device);
    int rc = async_drop(h, (hipStream_t)stream);          // (every map is replaced: what was in flight is void)
    if (rc) return rc;
    rc = set_maps_one(h, maps, stream);
    if (rc) return rc;
    h->parity ^= 1;
    return PCGRL_OK;
}

}  // extern "C"
#endif  // PART_CORE
