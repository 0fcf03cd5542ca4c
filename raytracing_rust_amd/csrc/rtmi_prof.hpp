// rtmi_prof.hpp — lane-activity and phase-time profiling hooks (PROF instantiations only).
// Part of the single translation unit rtmi_device.hip (device code is header-only so that every
// kernel instantiation inlines the whole path); arithmetic contract as stated there.
#pragma once
#include "rtmi_types.hpp"

// ---- lane-activity profiling (PROF instantiation only; diagnostics, never on the timed path) -----
// slot s: prof[2s] += active lanes, prof[2s+1] += 64 (one wave-iteration).  Accumulated in LDS,
// flushed to global once per block.
#define RTMI_PROF_SLOTS 32
template <bool PROF>
__device__ __forceinline__ void prof_tick(unsigned long long *prof_lds, int slot, bool active) {
    if (PROF) {
        const unsigned long long m = __ballot(active);
        if (m != 0ull && (int)(__ffsll((long long)m) - 1) == (int)(threadIdx.x & 63)) {
            atomicAdd(&prof_lds[2 * slot], (unsigned long long)__popcll(m));
            atomicAdd(&prof_lds[2 * slot + 1], 64ull);
        }
    }
}

// section timing (PROF only): elapsed shader cycles of this wave since the previous stamp are added
// to slot `slot` (word 0 = cycles, word 1 = number of stamps)
template <bool PROF>
__device__ __forceinline__ void prof_time(unsigned long long *prof_lds, int slot, unsigned long long &t_prev) {
    if (PROF) {
        const unsigned long long t = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&prof_lds[2 * slot], t - t_prev);
            atomicAdd(&prof_lds[2 * slot + 1], 1ull);
        }
        t_prev = __builtin_readcyclecounter();
    }
}
