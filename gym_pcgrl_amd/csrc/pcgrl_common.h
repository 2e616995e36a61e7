// Common definitions for the PCGRL HIP kernels (gfx950).
//
// The algorithm headers (mt19937.h, pcgrl_algos.h) are written against a small "lane group"
// interface so that the same source can also be instantiated by the CPU-side lane-group simulator
// used in tests/ (tests/hostsim).  The product only ever compiles them with hipcc for gfx950;
// the macros below exist so a plain C++ compiler can parse the headers for that test build.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PCGRL_HD __host__ __device__ __forceinline__
#define PCGRL_D __device__ __forceinline__
#else
#define PCGRL_HD inline
#define PCGRL_D inline
#endif

#define PCGRL_MAX_TILES 8
#define PCGRL_MAX_STATS 8
#define PCGRL_MAX_REWARDS 10
#define PCGRL_MT_N 624
#define PCGRL_MT_M 397

enum { PCGRL_PROB_BINARY = 0, PCGRL_PROB_ZELDA = 1, PCGRL_PROB_SOKOBAN = 2, PCGRL_PROB_MDUNGEON = 3, PCGRL_PROB_DDAVE = 4, PCGRL_PROB_SMB = 5 };
enum { PCGRL_REP_NARROW = 0, PCGRL_REP_WIDE = 1, PCGRL_REP_TURTLE = 2, PCGRL_REP_NARROW_CAST = 3, PCGRL_REP_NARROW_MULTI = 4,
       PCGRL_REP_TURTLE_CAST = 5 };
