// Philox4x32-10 counter-based generator (Salmon et al., SC'11), the device twin of
// oracle/rng.py: word-for-word the same streams so seeded runs agree bit for bit.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define DIE_STREAM_TURN 0u
#define DIE_STREAM_BROWNIAN 1u
#define DIE_STREAM_NOISE 2u
#define DIE_STREAM_INIT_AGENTS 3u
#define DIE_STREAM_INIT_FOOD 4u
#define DIE_STREAM_INIT_HEADING 5u
#define DIE_STREAM_INIT_AGENT_FOOD 6u

struct die_u32x4 { uint32_t v[4]; };

__host__ __device__ inline die_u32x4 die_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return die_u32x4{{c0, c1, c2, c3}};
}

__host__ __device__ inline die_u32x4 die_draw(uint64_t seed, uint32_t step, uint64_t slot, uint32_t stream) {
    return die_philox((uint32_t)slot, (uint32_t)(slot >> 32), step, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// numerator r in [0, 1000] of `random_sample().round(3)` (core/data_init.py:168-169)
__host__ __device__ inline int die_round3_units(uint32_t bits) {
    return (int)(((uint64_t)bits * 1000ull + 0x80000000ull) >> 32);
}
