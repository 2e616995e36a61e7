// Host side of pcgrl_step_multi: the handles of a node are stepped by a small pool of issuing threads, one handle each, so that the
// host time of a step of eight GPUs is one launch's (~3.5 us) plus the hand-over, not eight launches back to back (a
// 28 us step leaves the one driving thread less than that to issue all eight).  utils.py:60-71 is what this replaces: the
// reference's SubprocVecEnv has a worker PROCESS per environment; here the workers only issue launches.
//
// One pool per process, made at the first call that has more than one handle (and grown when a later call has more); worker w takes
// the handles w + 1, w + 1 + W, ...
// (the calling thread takes handle 0 and every handle beyond the pool).  A call publishes its arrays, bumps the generation and runs
// its own share; workers spin on the generation for a short while after a call (steps of a training loop follow each other within
// tens of microseconds) and then sleep on a condition variable.  The call returns when every handle's step has been ISSUED -- nothing
// waits for the GPU.  pcgrl_tuning is per handle and this is per process, so the switch is an entry point of its own:
// pcgrl_step_threads(0) before the first call keeps everything on the calling thread (default: min(handles - 1, 7) workers).
// Part of the single translation unit pcgrl_abi.hip (host code, part 0 only).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <unistd.h>

#if defined(__x86_64__) || defined(__i386__)
#define PCGRL_CPU_RELAX() __builtin_ia32_pause()
#else
#define PCGRL_CPU_RELAX() std::this_thread::yield()
#endif

struct StepPool {
    typedef int (*StepFn)(pcgrl_env*, const int32_t*, void*);
    enum { MAX_WORKERS = 7, SPIN_US = 200 };
    int nworkers = 0;
    pid_t pid = 0;                      // the process the threads belong to (a forked child has none: it steps on its own thread)
    // the call in flight (written by the caller before the generation is published)
    StepFn fn = nullptr;
    pcgrl_env* const* envs = nullptr;
    const int32_t* const* actions = nullptr;
    void* const* streams = nullptr;
    int count = 0;
    std::atomic<unsigned> gen{0};
    std::atomic<int> left{0};           // workers that have not finished the call in flight
    std::atomic<int> rc{0};             // first error of the call in flight
    std::atomic<int> hip{0};            // ... and the HIP error code behind it: pcgrl_last_hip_error is per thread, and the step that
                                        // failed may have run on a worker -- step() hands the code to the calling thread
    std::atomic<int> sleepers{0};
    std::mutex m;                       // callers take turns; also the condition variable's mutex
    std::mutex call_m;
    std::condition_variable cv;

    void run_share(int w, int stride) {
        for (int i = w; i < count; i += stride) {
            const int r = fn(envs[i], actions[i], streams[i]);
            if (r) { int zero = 0; if (rc.compare_exchange_strong(zero, r)) hip.store(g_last_hip, std::memory_order_release); }
        }
    }
    void worker(int w, unsigned seen) {        // seen: the generation at the time the worker was made (no call was in flight)
        for (;;) {
            // wait for the next generation: spin first, then sleep
            const auto t0 = std::chrono::steady_clock::now();
            unsigned g;
            int spins = 0;
            while ((g = gen.load(std::memory_order_seq_cst)) == seen) {
                if ((++spins & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(SPIN_US)) {
                    std::unique_lock<std::mutex> lk(m);
                    sleepers.fetch_add(1, std::memory_order_seq_cst);
                    // (seq_cst on both sides of the sleepers / gen handshake: the caller bumps gen and then reads sleepers, the worker
                    //  announces itself and then re-reads gen -- Dekker-style, which acquire / release alone does not order)
                    cv.wait(lk, [&] { return gen.load(std::memory_order_seq_cst) != seen; });
                    sleepers.fetch_sub(1, std::memory_order_seq_cst);
                } else {
                    PCGRL_CPU_RELAX();
                }
            }
            seen = g;
            run_share(w + 1, nworkers + 1);
            left.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    int step(StepFn f, pcgrl_env* const* e, const int32_t* const* a, void* const* s, int n) {
        std::lock_guard<std::mutex> turn(call_m);
        fn = f; envs = e; actions = a; streams = s; count = n;
        rc.store(0, std::memory_order_relaxed);
        hip.store(0, std::memory_order_relaxed);
        left.store(nworkers, std::memory_order_relaxed);
        gen.fetch_add(1, std::memory_order_seq_cst);
        if (sleepers.load(std::memory_order_seq_cst) > 0) { std::lock_guard<std::mutex> lk(m); cv.notify_all(); }
        run_share(0, nworkers + 1);
        while (left.load(std::memory_order_acquire) > 0) PCGRL_CPU_RELAX();
        const int r = rc.load(std::memory_order_acquire);
        if (r) g_last_hip = hip.load(std::memory_order_acquire);      // what pcgrl_last_hip_error() reports on the caller's thread
        return r;
    }
    // handles > 0: the pool for a call of that many handles (made now if there is none), or null when the calling thread is to do
    // everything (threads switched off, or a forked child of the process that owns them).  handles == 0: pcgrl_step_threads -- sets
    // the number of workers to make (`set` >= 0; only before the pool exists) and returns the number in effect.
    static StepPool* get(int handles, int set = -1, int* in_effect = nullptr) {
        static std::mutex make_m;
        static StepPool* pool = nullptr;
        static int want = MAX_WORKERS;
        std::lock_guard<std::mutex> lk(make_m);
        if (handles == 0) {
            if (set >= 0 && !pool) want = set > MAX_WORKERS ? MAX_WORKERS : set;
            if (in_effect) *in_effect = pool ? pool->nworkers : want;
            return nullptr;
        }
        if (pool && pool->pid != getpid()) return nullptr;
        if (!pool) {
            if (want <= 0) return nullptr;
            pool = new StepPool();          // (never destroyed: its threads sleep until the process ends)
            pool->pid = getpid();
        }
        const int need = want < handles - 1 ? want : handles - 1;
        if (need > pool->nworkers) {        // (a larger node than the calls so far: more workers, made while no call is in flight)
            std::lock_guard<std::mutex> turn(pool->call_m);
            const unsigned g = pool->gen.load(std::memory_order_acquire);
            StepPool* const p = pool;
            for (int w = pool->nworkers; w < need; w++) std::thread([p, w, g] { p->worker(w, g); }).detach();
            pool->nworkers = need;
        }
        return pool;
    }
};
