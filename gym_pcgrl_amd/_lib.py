"""ctypes binding of the C ABI in include/pcgrl_hip.h (gym_pcgrl_amd/lib/libpcgrl_hip.so).

There is no CPU fallback: if the HIP library is missing the import of the product path fails
loudly with instructions to build it.
"""
import ctypes as C
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
SO = os.path.join(LIBDIR, "libpcgrl_hip.so")
SOURCES = [os.path.join(CSRC, "pcgrl_abi.hip")]
# (-amdgpu-sched-strategy=max-ilp, round 6: the step kernels are chains of dependent instructions at a fraction of the issue peak -- the
#  compiler's default strategy schedules for occupancy first; same box, alternating libraries: C2 28.6 -> 28.2 / 28.6 -> 28.05 us, C3 28.9 ->
#  28.3 / 27.8 -> 27.1, C3w 31.5 -> 30.2: profiles/r6_round6/probe/ab_compiler_sched*.txt)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]

PCGRL_OK, PCGRL_EINVAL, PCGRL_EHIP, PCGRL_ESTATE = 0, -1, -2, -3


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "prob", "rep", "num_envs", "width", "height", "prob_width", "prob_height", "max_changes", "max_iterations",
        "random_start", "random_tile", "warp", "random_probs", "auto_reset", "target_path",
        "max_enemies", "target_enemy_dist", "max_crates", "target_solution", "solver_power", "max_potions",
        "max_treasures", "max_diamonds", "min_spikes", "target_jumps", "min_empty", "min_enemies", "min_jumps")] + [
        ("target_col_enemies", C.c_double), ("tile_probs", C.c_double * 8), ("rewards", C.c_double * 12)]


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("group", "mask_bytes", "nplanes", "nstats")] + [
        (n, C.c_size_t) for n in ("map", "old_map", "heatmap", "pos", "planes", "counters", "stats", "start_stats",
                                  "info", "reward", "done", "tile_p", "rng_rep", "rng_prob", "rng_cursor", "scratch")]


TUNING_FIELDS = ("no_fused", "fused_zelda", "step_epb", "no_inc", "inline_reset", "pair_min", "no_wide", "wide_waves", "wide_grid",
                 "wide_pairs", "wide_few", "sok_generic", "sok_hard_cap", "sok_spawn", "md_only_agent", "smb_lds_heap", "full_per_wave", "inc_per_wave", "wide_spin", "step_prio", "no_touch", "touch_tight", "step_pair", "async_split", "big_team", "obs_at_end")


class Tuning(C.Structure):
    """include/pcgrl_hip.h pcgrl_tuning: developer switches, -1 = the library's default."""
    _fields_ = [(n, C.c_int32) for n in TUNING_FIELDS]


# Process-wide overrides of the developer switches for tools/ and tests (the library itself reads no environment variables and
# keeps no global state: this dict lives in the Python binding).  {field name: value}; BatchedPcgrlEnv(tuning={...}) takes
# precedence.  tools/ fill it from PCGRL_* environment variables (tools/_tuning_env.py).
TUNING_OVERRIDES = {}


def make_tuning(overrides=None):
    t = Tuning(*([-1] * len(TUNING_FIELDS)))
    for k, v in list(TUNING_OVERRIDES.items()) + list((overrides or {}).items()):
        if k not in TUNING_FIELDS:
            raise KeyError("unknown tuning switch %r (known: %s)" % (k, ", ".join(TUNING_FIELDS)))
        setattr(t, k, int(v))
    return t


BUFFER_NAMES = ("map", "old_map", "heatmap", "pos", "planes", "counters", "stats", "start_stats", "info", "reward",
                "done", "tile_p", "rng_rep", "rng_prob", "rng_cursor", "scratch")


class Buffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in BUFFER_NAMES]


ABI_VERSION = 14         # include/pcgrl_hip.h PCGRL_ABI_VERSION
EXPORTS = ("pcgrl_abi_version", "pcgrl_error_string", "pcgrl_last_hip_error", "pcgrl_query_layout", "pcgrl_create",
           "pcgrl_destroy", "pcgrl_bind", "pcgrl_configure", "pcgrl_seed", "pcgrl_set_tile_probs", "pcgrl_reset",
           "pcgrl_step", "pcgrl_set_maps", "pcgrl_observe", "pcgrl_action_map", "pcgrl_status", "pcgrl_profile",
           "pcgrl_profile_read", "pcgrl_bind_episode_stats", "pcgrl_seed_words", "pcgrl_rollout", "pcgrl_bind_observation", "pcgrl_selftest_heap",
           "pcgrl_tuning_defaults", "pcgrl_set_tuning", "pcgrl_clear_status", "pcgrl_step_flat",
           "pcgrl_async_bytes", "pcgrl_bind_async", "pcgrl_step_async", "pcgrl_async_flush", "pcgrl_step_multi", "pcgrl_selftest_step_pool", "pcgrl_step_threads", "pcgrl_selftest_range_reward")
NPHASE = 6
# the six intervals between the seven event marks of a step; sokoban: update, stats, reset, solver, reset2, solver2;
# other problems: update, stats(+resets), -, reset (only with PCGRL_INLINE_RESET=0), -, -
PHASES = ("update", "stats", "reset_or_idle", "solver_or_reset", "reset2", "solver2")


def sources_newer_than_lib():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "pcgrl_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash():
    """Content hash of the kernel sources (csrc/* and include/pcgrl_hip.h), 16 hex digits: the profile tools stamp it into
    profiles/*/<W>_traffic.json / _pmc.json and bench.py compares it with the tree it runs from, so that a bench line cannot
    silently carry counter figures measured on other code (no git on the GPU box: the hash is over file contents)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(ROOT, "include", "pcgrl_hip.h")]
    h.update(" ".join(HIPCC_FLAGS).encode() + b"\0")          # (the compiler flags are part of what the library was built from)
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


NPARTS = 8               # csrc/pcgrl_abi.hip: PCGRL_PART=0..7 (host ABI; stats; update; step binary; step zelda; search; smb; step_solver)


def build(force=False, verbose=False, jobs=None):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU): the one source is compiled NPARTS times
    side by side (-DPCGRL_PART=k, see the head of csrc/pcgrl_abi.hip) and the objects are linked."""
    if not force and not sources_newer_than_lib():
        return SO
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs = jobs or int(os.environ.get("PCGRL_BUILD_JOBS", "0")) or min(NPARTS, os.cpu_count() or 1)
    objs = [os.path.join(objdir, "part%d.o" % k) for k in range(NPARTS)]
    cmds = [[hipcc] + flags + ["-DPCGRL_PART=%d" % k, "-c", SOURCES[0], "-o", objs[k]] for k in range(NPARTS)]
    running, todo, failed = [], list(range(NPARTS)), []
    while todo or running:
        while todo and len(running) < jobs:
            k = todo.pop(0)
            if verbose:
                print(" ".join(cmds[k]), flush=True)
            running.append((k, subprocess.Popen(cmds[k])))
        k, p = running.pop(0)
        if p.wait() != 0:
            failed.append(k)
    if failed:
        raise subprocess.CalledProcessError(1, cmds[failed[0]])
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", SO]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return SO


_lib = None


def load():
    global _lib, SO
    if _lib is not None:
        return _lib
    # developer switch (A/B builds of the kernels, tools/): another build of the SAME library; it must exist
    SO = os.environ.get("PCGRL_HIP_SO", SO)
    if not os.path.exists(SO):
        raise RuntimeError(
            "gym_pcgrl_amd: HIP library %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the batched environment." % SO)
    # torch must be imported first: it ships its own libamdhip64, and the kernels have to run in the same
    # HIP runtime instance that owns the tensors whose pointers we are handed (two runtimes in one
    # process do not share devices or allocations: hipErrorNoDevice on the first call).
    import torch  # noqa: F401
    L = C.CDLL(SO)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError("gym_pcgrl_amd: %s does not export %s (stale build?)" % (SO, name))
    L.pcgrl_abi_version.restype = C.c_int
    if L.pcgrl_abi_version() != ABI_VERSION:
        raise RuntimeError("gym_pcgrl_amd: %s has ABI version %d, this package needs %d (stale build?)" % (SO, L.pcgrl_abi_version(), ABI_VERSION))
    L.pcgrl_error_string.restype = C.c_char_p
    L.pcgrl_error_string.argtypes = [C.c_int]
    L.pcgrl_last_hip_error.restype = C.c_int
    L.pcgrl_query_layout.argtypes = [C.POINTER(Config), C.POINTER(Layout)]
    L.pcgrl_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.pcgrl_destroy.argtypes = [C.c_void_p]
    L.pcgrl_tuning_defaults.argtypes = [C.POINTER(Tuning)]
    L.pcgrl_tuning_defaults.restype = None
    L.pcgrl_set_tuning.argtypes = [C.c_void_p, C.POINTER(Tuning)]
    L.pcgrl_bind.argtypes = [C.c_void_p, C.POINTER(Buffers), C.c_void_p]
    L.pcgrl_configure.argtypes = [C.c_void_p, C.POINTER(Config)]
    L.pcgrl_seed.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.pcgrl_seed_words.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.pcgrl_set_tile_probs.argtypes = [C.c_void_p, C.c_void_p]
    L.pcgrl_reset.argtypes = [C.c_void_p, C.c_void_p]
    L.pcgrl_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcgrl_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcgrl_set_maps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcgrl_observe.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.pcgrl_bind_observation.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.pcgrl_action_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcgrl_step_flat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcgrl_selftest_heap.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcgrl_selftest_step_pool.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.pcgrl_selftest_range_reward.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.pcgrl_step_threads.argtypes = [C.c_int32]
    L.pcgrl_step_threads.restype = C.c_int32
    L.pcgrl_status.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.pcgrl_clear_status.argtypes = [C.c_void_p, C.c_void_p]
    L.pcgrl_profile.argtypes = [C.c_void_p, C.c_int]
    L.pcgrl_bind_episode_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pcgrl_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.pcgrl_async_bytes.argtypes = [C.POINTER(Config), C.c_int32]
    L.pcgrl_async_bytes.restype = C.c_size_t
    L.pcgrl_bind_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]
    L.pcgrl_step_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.pcgrl_async_flush.argtypes = [C.c_void_p, C.c_void_p]
    L.pcgrl_step_multi.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32]
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        L = load()
        msg = L.pcgrl_error_string(rc).decode()
        extra = " (hipError %d)" % L.pcgrl_last_hip_error() if rc == PCGRL_EHIP else ""
        raise RuntimeError("gym_pcgrl_amd: %s failed: %s%s" % (what, msg, extra))
