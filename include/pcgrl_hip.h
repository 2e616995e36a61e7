/*
 * pcgrl_hip.h -- C ABI of the MI355X batched PCGRL environment (libpcgrl_hip.so).
 *
 * The reference (amidos2006/gym-pcgrl) has no FFI of its own: its hot path sits behind a Python
 * class.  This ABI is what a replacement for that path binds.  Each entry point names the
 * reference interface it replaces (paths relative to gym_pcgrl/envs/):
 *
 *   pcgrl_create / pcgrl_configure   PcgrlEnv.__init__ pcgrl_env.py:27-42, adjust_param :106-115,
 *                                    Problem/Representation.adjust_param (probs/problem.py:66-72,
 *                                    binary_prob.py:49-59, zelda_prob.py:59-71, sokoban_prob.py:60-73, mdungeon_prob.py:68-84,
 *                                    ddave_prob.py:67-82, smb_prob.py:40-53,
 *                                    reps/representation.py:53-54, narrow_rep.py:86-88, turtle_rep.py:42-44)
 *   pcgrl_seed                       PcgrlEnv.seed pcgrl_env.py:54-57 (host passes MT19937 keys)
 *   pcgrl_reset                      PcgrlEnv.reset pcgrl_env.py:66-76 for every environment
 *   pcgrl_step                       PcgrlEnv.step pcgrl_env.py:129-150 for every environment, plus the
 *                                    vector-env auto-reset the reference delegates to SubprocVecEnv (utils.py:60-71)
 *   pcgrl_set_tile_probs             Problem.adjust_param(probs=...) probs/problem.py:68-72
 *   pcgrl_set_maps                   direct assignment of Representation._map (tests, curriculum loaders)
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * PCGRL_E* code (no exceptions cross the ABI); all device buffers are OWNED BY THE CALLER (hipMalloc,
 * or a torch tensor's data_ptr()) and described by pcgrl_layout; launches are asynchronous on the
 * given hipStream_t (passed as void*); one handle per GPU, one host thread per handle.
 * There is no CPU fallback: without a HIP device every launch returns PCGRL_EHIP.
 */
#ifndef PCGRL_HIP_H
#define PCGRL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: + pcgrl_bind_episode_stats.  3: planes buffer laid out [N,group,nplanes] (was [N,nplanes,group]); + pcgrl_seed_words.
 * 4: + the mdungeon problem: pcgrl_config grew (max_potions, max_treasures, target_col_enemies, rewards[12]).
 * 5: + the ddave problem: pcgrl_config grew (max_diamonds, min_spikes, target_jumps).  6: + pcgrl_rollout.
 * 7: + the smb problem: pcgrl_config grew (min_empty, min_enemies, min_jumps; `reserved_` is gone); pcgrl_status reports
 *    clamped actions.  10: + pcgrl_tuning / pcgrl_set_tuning (the library reads no environment variables any more); pcgrl_config
 *    grew (prob_width, prob_height); maps up to 255 x 255, search levels up to 16 384 bordered cells, solver_power up to 1 000 000. */
#define PCGRL_ABI_VERSION 14
#define PCGRL_OK 0
#define PCGRL_EINVAL (-1)   /* bad argument / unsupported configuration */
#define PCGRL_EHIP (-2)     /* a HIP runtime call failed (see pcgrl_last_hip_error) */
#define PCGRL_ESTATE (-3)   /* call order violated (e.g. step before bind/reset) */

enum { PCGRL_BINARY = 0, PCGRL_ZELDA = 1, PCGRL_SOKOBAN = 2, PCGRL_MDUNGEON = 3, PCGRL_DDAVE = 4, PCGRL_SMB = 5 };
enum { PCGRL_NARROW = 0, PCGRL_WIDE = 1, PCGRL_TURTLE = 2, PCGRL_NARROW_CAST = 3, PCGRL_NARROW_MULTI = 4, PCGRL_TURTLE_CAST = 5 };

/* Batch-wide parameters (everything the reference keeps as attributes of PcgrlEnv/Problem/Representation). */
typedef struct pcgrl_config {
    int32_t prob, rep;
    int32_t num_envs;
    int32_t width, height;                 /* the maps' width / height: 1..255 each (search problems: (width + 2) * (height + 2) <= 16384;
                                              smb: width up to 250, height 3..32).  Up to 64 x 64 the row-bitboard kernels, beyond
                                              them the general path (csrc/bigmap.h) */
    int32_t prob_width, prob_height;       /* Problem._width/_height when they differ from the maps' -- adjust_param(width, height)
                                              without a reset(): the reference goes on stepping the old maps with the problem's new
                                              size in its formulas (pcgrl_env.py:106-115, zelda_prob.py:99); 0 = same as width / height */
    int32_t max_changes, max_iterations;   /* pcgrl_env.py:33-34 / :108-110, computed by the host */
    int32_t random_start, random_tile, warp, random_probs;
    int32_t auto_reset;                    /* 1: a done env is reset inside step (vector-env semantics) */
    int32_t target_path;                   /* binary 20, zelda 16 */
    int32_t max_enemies, target_enemy_dist;            /* zelda (max_enemies: mdungeon and smb too) */
    int32_t max_crates, target_solution, solver_power; /* sokoban (target_solution, solver_power: mdungeon too) */
    int32_t max_potions, max_treasures;                /* mdungeon */
    int32_t max_diamonds, min_spikes, target_jumps;    /* ddave (+ target_solution, solver_power) */
    int32_t min_empty, min_enemies, min_jumps;         /* smb (+ max_enemies, solver_power) */
    double target_col_enemies;                         /* mdungeon */
    double tile_probs[8];                  /* Problem._prob in tile order (un-normalised) */
    double rewards[12];                    /* Problem._rewards in the problem's own key order */
} pcgrl_config;

/* Byte sizes of the caller-provided device buffers for a configuration (N = num_envs). */
typedef struct pcgrl_layout {
    int32_t group;        /* lanes per map: 16 (height <= 16) or 64 */
    int32_t mask_bytes;   /* bytes per row mask: 4 (width <= 32) or 8 */
    int32_t nplanes;      /* bit planes of the tile id: 1 binary, 3 zelda/sokoban/mdungeon/ddave, 0 smb (statistics from the byte map) */
    int32_t nstats;       /* used slots of a stats row: 2 binary, 7 zelda, 6 sokoban, 8 mdungeon / ddave (packed, see below), 8 smb */
    size_t map;           /* u8  [N,H,W]   observation "map" */
    size_t old_map;       /* u8  [N,H,W]   Representation._old_map */
    size_t heatmap;       /* i16 [N,H,W]   observation "heatmap" (counts) */
    size_t pos;           /* u8  [N,2]     observation "pos" (x,y); unused for wide */
    size_t planes;        /* mask[N,group,nplanes] row bitboards of the tile-id bits (the planes of a row are adjacent) */
    size_t counters;      /* i32 [N,2]     iteration, changes */
    size_t stats;         /* i32 [N,8]     current _rep_stats.  mdungeon keeps its eleven values in eight slots:
                                            player, exit, potions, treasures, enemies, regions, then slot 6 = sol-length if
                                            the planner won else dist-win, slot 7 = col-potions | col-treasures << 8 |
                                            col-enemies << 16 | won << 24; ddave likewise: player | exit << 8 | key << 16,
                                            dist-floor, diamonds, spikes, regions, num-jumps, then slot 6 = sol-length if the
                                            planner won else dist-win, slot 7 = col-diamonds | won << 24 */
    size_t start_stats;   /* i32 [N,8]     Problem._start_stats */
    size_t info;          /* i32 [N,10]    per-step info: stats[8], iterations, changes */
    size_t reward;        /* f64 [N] */
    size_t done;          /* u8  [N] */
    size_t tile_p;        /* f64 [N,2]     binary: per-env (p_empty, p_solid) */
    size_t rng_rep;       /* u32 [N,624]   representation MT19937 ring */
    size_t rng_prob;      /* u32 [N,624]   problem MT19937 ring (binary only; may be NULL otherwise) */
    size_t rng_cursor;    /* i32 [N,2]     ring cursors (rep, prob) */
    size_t scratch;       /* work lists, counters, sokoban solver arena */
} pcgrl_layout;

typedef struct pcgrl_buffers {
    void *map, *old_map, *heatmap, *pos, *planes, *counters, *stats, *start_stats, *info, *reward,
         *done, *tile_p, *rng_rep, *rng_prob, *rng_cursor, *scratch;
} pcgrl_buffers;

typedef struct pcgrl_env pcgrl_env;

/* Developer switches: which of several equivalent kernels / schedules a handle uses.  Every setting gives the same results (the
 * tests run the alternatives against the same fixtures); they exist for A/B measurements and to force the rarely taken paths in
 * tests.  A negative field = the library's default (pcgrl_tuning_defaults sets them all).  Set with pcgrl_set_tuning between
 * pcgrl_create and pcgrl_bind; the library reads no environment variables and keeps no process-wide state. */
typedef struct pcgrl_tuning {
    int32_t no_fused;        /* 1: binary / zelda steps as k_update + k_stats instead of the one-launch k_step */
    int32_t fused_zelda;     /* 0: zelda steps as two launches (binary unaffected) */
    int32_t step_epb;        /* environments per block of k_step: 64, 128 or 256 (default: from the batch size) */
    int32_t no_inc;          /* 1: no incremental statistics -- every change recomputes */
    int32_t inline_reset;    /* 0: resets through the reset list + k_reset instead of inside the statistics kernel */
    int32_t pair_min;        /* certain resets per launch from which a wavefront of k_stats takes two (default 2048) */
    int32_t no_wide;         /* 1: tall binary maps (17..64 rows) on one wavefront per map instead of k_stats_wide */
    int32_t wide_waves;      /* wavefronts per tall map: 4 or 8 (default 8) */
    int32_t wide_grid;       /* blocks of k_stats_wide (default 2048) */
    int32_t wide_pairs;      /* 0: every full item of a tall map a block of its own */
    int32_t wide_few;        /* region count up to which a tall map's full item takes half a block (default 32) */
    int32_t sok_generic;     /* 1: every Sokoban / MiniDungeons / Dave level takes the generic (not the register-resident) search */
    int32_t sok_hard_cap;    /* levels k_sokoban may publish to idle blocks per launch (default 4096) */
    int32_t sok_spawn;       /* pops after which a Sokoban BFS publishes its level (default 128) */
    int32_t md_only_agent;   /* >= 0: the MiniDungeons planner runs only this agent (timing experiments: results are then wrong) */
    int32_t smb_lds_heap;    /* heap words a k_smb search keeps in LDS (default 2048; tests: the overflow path) */
    int32_t full_per_wave;   /* k_step: maps of full recomputations per wavefront task: 1, 2 or 4 */
    int32_t inc_per_wave;    /* k_step: incremental items per wavefront task: 1, 2 or 4 */
    int32_t wide_spin;       /* k_stats_wide: sleeps a reset's block waits for its partner block before it computes the old map's statistics
                                itself (default 400, ~50 us; tests: 1 forces the take-over path) */
    int32_t step_prio;       /* k_step: s_setprio levels of the wavefronts by what they do, two bits each: bits 0-1 a certain reset, 2-3 a full
                                recomputation, 4-5 an incremental update, 6-7 the update wavefronts; bits 8-11: how many of the leading
                                (dearest) full tasks of a block get the level of bits 2-3 (0 = all of them) */
    int32_t no_touch;        /* 1: k_step recomputes the binary statistics in full when a change is in or next to the champion component
                                (default 0: binary_touch first -- the champion's pieces / its union with the cell and one double sweep) */
    int32_t touch_tight;     /* where a full computation also sweeps the second largest component, for a tight bound on "the others": bit 0 the
                                recomputations of a step, bit 1 the resets (default 1) */
    int32_t step_pair;       /* k_step, zelda: from this many certain resets in a block's step on, a wavefront takes two of them (default 6; 0: never) */
    int32_t async_split;     /* pcgrl_step_async: 1 = the fresh jobs of a tick in a launch of their own with small search regions, several blocks per
                                compute unit (the default for sokoban); 0 = one search launch per tick, the full region for every job */
    int32_t big_team;        /* k_big, binary maps beyond 64 x 64: 1 = a step's few full recomputations are made by all wavefronts of a block
                                together (csrc/bigmap_team.h; the default: four wavefronts a block), 2..8 = that many, 0 = a wavefront
                                per map throughout */
    int32_t obs_at_end;      /* k_step with a bound observation (pcgrl_bind_observation): 1 = every image at the end of the launch (the form up to
                                round 5); default 0 = the images leave while the step runs -- a reset's wavefront writes its new map's, a wide
                                change's lane its piece, "observation tasks" between the statistics tasks the rest (csrc/kernels_step.h) */
} pcgrl_tuning;

int pcgrl_abi_version(void);
const char* pcgrl_error_string(int code);
int pcgrl_last_hip_error(void);

int pcgrl_query_layout(const pcgrl_config* cfg, pcgrl_layout* out);
int pcgrl_create(const pcgrl_config* cfg, pcgrl_env** out);
int pcgrl_destroy(pcgrl_env* env);
void pcgrl_tuning_defaults(pcgrl_tuning* t);
int pcgrl_set_tuning(pcgrl_env* env, const pcgrl_tuning* t);   /* between pcgrl_create and pcgrl_bind (PCGRL_ESTATE afterwards) */
int pcgrl_bind(pcgrl_env* env, const pcgrl_buffers* bufs, void* stream);
/* Change parameters that do not alter buffer sizes (anything but prob/rep/num_envs/width/height). */
int pcgrl_configure(pcgrl_env* env, const pcgrl_config* cfg);
/* keys: HOST pointer, [count,624] u32 = MT19937 init_by_array state of environment first..first+count-1;
 * seeds both streams identically (pcgrl_env.py:54-57). */
int pcgrl_seed(pcgrl_env* env, const uint32_t* keys, int32_t first, int32_t count, void* stream);
/* The same, with the MT19937 states computed on the device (init_by_array): words HOST u32 [count][3] = (key word 0,
 * key word 1, number of key words 1|2) -- the key gym's seeding.hash_seed derives from the integer seed. */
int pcgrl_seed_words(pcgrl_env* env, const uint32_t* words, int32_t first, int32_t count, void* stream);
/* Broadcast cfg.tile_probs[0..1] into tile_p (binary); call once after the first bind and whenever
 * adjust_param(probs=...) touched them.  tile_p otherwise persists (it carries BinaryProblem._prob). */
int pcgrl_set_tile_probs(pcgrl_env* env, void* stream);
int pcgrl_reset(pcgrl_env* env, void* stream);
/* actions: DEVICE pointer, i32 [N] (narrow, turtle), [N,3] = (x, y, tile) (wide), [N,2] = (type, tile)
 * (narrowcast, turtlecast) or [N,9] (narrowmulti: tile+1 per cell of the 3x3 block, 0 = keep). */
int pcgrl_step(pcgrl_env* env, const int32_t* actions, void* stream);
/* pcgrl_step on `count` handles in one call -- the shards of one batch of environments, one handle per GPU of a node (or several
 * on one GPU), each with its own stream: envs / actions / streams are HOST arrays of `count` entries.  Step k is issued on every
 * handle before the call returns and nothing is waited for; what SubprocVecEnv.step_async does for the reference's worker
 * processes (utils.py:60-71), as one foreign-function call per step of the whole node.  The handles are issued side by side by a
 * small pool of host threads inside the library (one handle each; pcgrl_step_threads(0) beforehand: all on the calling thread);
 * every handle is tried, the first error seen is returned.  One call at a time per process (calls take turns). */
int pcgrl_step_multi(pcgrl_env* const* envs, const int32_t* const* actions, void* const* streams, int32_t count);
/* The issuing threads of pcgrl_step_multi besides the caller's (process-wide; default 7 = a thread per GPU of an eight-GPU node):
 * n >= 0 sets the number -- only until the first multi-handle call has started them --, n < 0 only asks.  Returns the number in
 * effect.  No reference counterpart (its SubprocVecEnv has a process per environment, utils.py:60-71). */
int32_t pcgrl_step_threads(int32_t n);
/* `steps` consecutive pcgrl_step calls on a tape of actions (a random-action rollout as in the reference's README
 * loop `env.step(env.action_space.sample())`, a recorded episode, an evaluation run): actions DEVICE i32
 * [steps, N(, k)] laid out like `steps` action arrays of pcgrl_step one after the other.  Optional DEVICE outputs, one
 * row per step: reward_out f64 [steps, N], done_out u8 [steps, N], info_out i32 [steps, N, 10]; NULL = not wanted.
 * The state buffers end up exactly as after the equivalent sequence of pcgrl_step calls.  Where one kernel does the
 * whole step (binary and zelda maps of at most 16 rows) the rollout is a single launch -- blocks of environments run ahead
 * of each other, there is nothing to wait for between steps; the search problems (sokoban, mdungeon, ddave; up to 131 072
 * environments) run on persistent blocks that own their environments for the whole tape, so that only the block that meets
 * a long search waits for it; everything else is the sequence of steps. */
int pcgrl_rollout(pcgrl_env* env, const int32_t* actions, int32_t steps, double* reward_out, uint8_t* done_out,
                  int32_t* info_out, void* stream);
/* maps: DEVICE pointer u8 [N,H,W]; replaces every map, recomputes stats (start stats unchanged). */
int pcgrl_set_maps(pcgrl_env* env, const uint8_t* maps, void* stream);
/* Observation formatting of the reference's composite wrappers (gym_pcgrl/wrappers.py): Cropped.transform
 * :197-206 + OneHotEncoding.transform :101-104 + ToImage.transform :53-60.  out: DEVICE u8 [N,out_h,out_w,D],
 * D = 1 (tile ids) or number of tiles (one-hot).  centered = 1: window centred on the cursor, cells outside
 * the map = pad_value (CroppedImagePCGRLWrapper :215-231, out_h = out_w = crop_size); centered = 0: the
 * map itself from its origin (ActionMapImagePCGRLWrapper :234-248, out_h = H, out_w = W). */
int pcgrl_observe(pcgrl_env* env, uint8_t* out, int32_t out_h, int32_t out_w, int32_t centered, int32_t pad_value,
                  int32_t onehot, void* stream);
/* The same image kept up to date by the step itself: after this call every pcgrl_reset / pcgrl_step / pcgrl_rollout /
 * pcgrl_set_maps leaves the image of the state it returns in `out` (DEVICE uint8 [N][out_h][out_w][D], 16-byte aligned,
 * caller-owned).  Where the step is one fused kernel (binary, zelda; maps of at most 16 rows) that kernel writes the image from
 * its on-chip copy of the state -- no extra launch, no read of the byte map; elsewhere one extra kernel follows the step.  This
 * is what wrappers.py:215-248 + utils.make_vec_envs :60-71 hand the policy per step.  May be called again at any time with
 * another `out` (a rollout buffer row); out = NULL switches it off.  Only records the target: nothing is written by the call.
 * incremental != 0: the caller will not write into `out`; a step whose target is the buffer the previous step (or reset) left
 * its image in may then update it in place -- for a window that does not follow a cursor (wide representation) that is one
 * 16-byte piece per changed environment and the full image only where an episode ended, instead of every byte.  Binding
 * another buffer, or incremental = 0, always gives full images. */
int pcgrl_bind_observation(pcgrl_env* env, uint8_t* out, int32_t out_h, int32_t out_w, int32_t centered,
                           int32_t pad_value, int32_t onehot, int32_t incremental);
/* ActionMap.step for the wide representation (wrappers.py:139-154): flat DEVICE i32 [N] index into
 * (H, W, tiles) -> xyv DEVICE i32 [N,3] = (x, y, tile), the action pcgrl_step takes. */
int pcgrl_action_map(pcgrl_env* env, const int32_t* flat, int32_t* xyv, void* stream);
/* ActionMap.step followed by PcgrlEnv.step (wrappers.py:139-154 + pcgrl_env.py:129-150) for the wide representation in one call:
 * flat DEVICE i32 [N]; xyv DEVICE i32 [N,3] scratch of the caller (filled where the decode is a kernel of its own; the fused step
 * kernel decodes inside and leaves it alone).  PCGRL_EINVAL for another representation. */
int pcgrl_step_flat(pcgrl_env* env, const int32_t* flat, int32_t* xyv, void* stream);
/* ---- Asynchronous stepping of the search problems (sokoban, mdungeon, ddave; csrc/kernels_search_async.h).  No reference
 * counterpart as such: it is what the reference gets from running every environment in a process of its own behind
 * SubprocVecEnv (utils.py:60-71) -- no environment waits for another environment's solver (sokoban_prob.py:85-122,
 * mdungeon_prob.py:110-126, ddave_prob.py) -- on one GPU and one stream.
 * pcgrl_step_async is one *tick*: every environment whose previous step is complete takes its action from `actions` (laid out
 * as for pcgrl_step) and steps; every search gets at most `pop_budget` pops; an environment whose search is not finished by then
 * is marked pending (its search is suspended and goes on in the following ticks) and takes no further action -- the entries of
 * `actions` for pending environments are ignored -- until a tick completes its step.  After a tick, pending[e] == 0 means: the
 * observation / reward / done / info of environment e are those of its last taken action, and it takes the next one.  Per
 * environment, the sequence of (taken action -> outputs) is bitwise what pcgrl_step gives for the same actions.
 *   pcgrl_async_bytes      DEVICE bytes the caller provides for `nslots` suspended searches (0: the configuration has no
 *                          asynchronous form -- another problem, levels or solver_power beyond the compact searches).
 *   pcgrl_bind_async       after pcgrl_bind; arena DEVICE, 256-byte aligned, zeroed by the call.  Its head is the caller's to
 *                          read: pending u8 [N] at offset 0 (0: complete; 1: a search is suspended; 2: the search ended the episode,
 *                          the next tick resets the environment), then (256-byte aligned) eight u64 counters: -, searches suspended,
 *                          jobs finished from a slot, slot overflows (no free slot: that search ran to its end inside its tick), pops
 *                          of the resumable searches; then sixteen u64 words 64 bytes apart whose sum is the number of actions taken.
 *   pcgrl_async_flush      finishes every pending step (unbounded budget) -- pcgrl_step, pcgrl_rollout, pcgrl_set_maps and
 *                          pcgrl_reset do it themselves (pcgrl_reset drops the pending steps instead).
 * pop_budget < 1: PCGRL_EINVAL. */
size_t pcgrl_async_bytes(const pcgrl_config* cfg, int32_t nslots);
int pcgrl_bind_async(pcgrl_env* env, void* arena, size_t bytes, int32_t nslots, void* stream);
int pcgrl_step_async(pcgrl_env* env, const int32_t* actions, int32_t pop_budget, void* stream);
int pcgrl_async_flush(pcgrl_env* env, void* stream);
/* Episode statistics kept by the step kernels -- what stable-baselines' Monitor keeps around the reference env in
 * utils.make_env / make_vec_envs (utils.py:13-29, 60-71).  ep_return f64 [N], ep_length i32 [N]: reward sum (in step
 * order) and step count of the running episode; last_return / last_length: the same, latched when an episode ends
 * (read them where done == 1).  DEVICE pointers, zeroed by the call; pass four NULLs to switch the feature off (the
 * default).  Call after pcgrl_bind. */
int pcgrl_bind_episode_stats(pcgrl_env* env, double* ep_return, int32_t* ep_length, double* last_return,
                             int32_t* last_length, void* stream);
/* Test hook, no reference counterpart: runs the heap primitives of the two-wavefront searches (the CPython heapq of
 * sokoban/engine.py:96-119, mdungeon/engine.py, ddave/engine.py: heappush / heappop on entries compared by priority only)
 * over a tape of operations on the current device.  ops DEVICE u32 [n_ops]: a packed word (priority << 16 | payload) to push, or
 * 0xFFFFFFFF = pop.  pops DEVICE u32 [number of pops]: the popped words in order (0xFFFFFFFF: the heap was empty); heap_out DEVICE
 * u32 [16384] and n_out DEVICE i32 [1]: the heap array afterwards.  At most 16 384 entries are kept (further pushes are dropped).
 * tests/test_gpu_parity.py holds the result against Python's heapq slot for slot. */
int pcgrl_selftest_heap(const uint32_t* ops, int32_t n_ops, uint32_t* pops, uint32_t* heap_out, int32_t* n_out, void* stream);
/* Test hook, no reference counterpart, no GPU needed: the issuing threads of pcgrl_step_multi driven with `count` (2..64) stand-in
 * handles for `calls` calls; hits HOST i32 [count] is incremented once per stand-in and call, stand-in `fail_at` (-1: none) reports an
 * error.  Returns the number of calls that returned an error, -1 when the pool is switched off (pcgrl_step_threads(0)). */
int pcgrl_selftest_step_pool(int32_t count, int32_t calls, int32_t fail_at, int32_t* hits);
/* Test hook for get_range_reward (helper.py:366-376), which every reward term of every problem goes through: the integer form the
 * step kernels evaluate (bounds are integers or +-inf, statistics are small integers; INT32_MAX / INT32_MIN stand for +-inf), run
 * on the current device over a table.  rows DEVICE i32 [n][4] = (new value, old value, low, high); out DEVICE i32 [n].
 * tests/test_gpu_parity.py holds it against the reference's exhaustive table (tests/golden/range_reward.npz). */
int pcgrl_selftest_range_reward(const int32_t* rows, int32_t n, int32_t* out, void* stream);
/* Sticky device status word, 0 = fine.  Bit 0 (1): a level was outside a search kernel's limits -- a Sokoban level with more crates
 * than the search takes (32 in the compact searches, 256 in the general ones of csrc/search_big.h), or more than 255 tiles / collected
 * things of one kind in a packed statistics row (Dave, MiniDungeons on large maps): the statistics of that level are then not exact.
 * Bit 1 (2): an action outside the action space was clamped into it (the reference raises IndexError or writes the
 * bad value: narrow_rep.py:101-103, wide_rep.py:68-69, turtle_rep.py:101-129, wrappers.py:139-154).  Bit 2 (4):
 * pcgrl_set_maps was handed a tile id >= the number of tiles (clamped).  Synchronises the stream. */
int pcgrl_status(pcgrl_env* env, void* stream, int32_t* status);
/* Zero the status word (asynchronous on the stream): a caller that turns bit 1 into the reference's IndexError at the offending
 * call clears it afterwards, so that the next call reports only its own actions. */
int pcgrl_clear_status(pcgrl_env* env, void* stream);


/* Per-phase GPU timing of pcgrl_step with HIP events recorded on the caller's stream (bench.py's
 * roofline leg).  Phases: 0 update, 1 stats(step), 2 solver(step), 3 mapgen, 4 stats(start), 5 solver(start). */
#define PCGRL_NPHASE 6
int pcgrl_profile(pcgrl_env* env, int enable);
int pcgrl_profile_read(pcgrl_env* env, double* phase_ms /*[PCGRL_NPHASE]*/, int32_t* steps);

#ifdef __cplusplus
}
#endif
#endif
