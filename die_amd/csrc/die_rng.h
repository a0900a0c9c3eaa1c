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
#define DIE_STREAM_BUILDER 7u

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


// 2-D gradient ("Perlin") noise — Perlin 1985 / 2002: a unit gradient on every integer lattice point, the four corner
// dot products blended with the quintic fade 6t^5 − 15t^4 + 10t^3.  Stands in for the un-vendored `perlin_noise`
// package of core/data_init.py:190-196, whose `octaves` is the number of lattice cells per unit length: the caller
// passes (x·octaves, y·octaves).  The lattice gradients come from Philox(seed, lattice point); oracle/cpu_ref.py
// perlin2 is the same function in numpy.  Values lie in [−√½, √½].
__device__ inline double die_perlin_dot(uint64_t seed, int64_t i, int64_t j, double dx, double dy) {
    const die_u32x4 r = die_draw(seed, 0u, (uint64_t)(i & 0xFFFFF) | ((uint64_t)(j & 0xFFFFF) << 20), DIE_STREAM_INIT_FOOD);
    const double th = 6.283185307179586476925 * ((double)r.v[0] * (1.0 / 4294967296.0));
    return cos(th) * dx + sin(th) * dy;
}
__device__ inline double die_perlin2(uint64_t seed, double x, double y) {
    const double fx0 = floor(x), fy0 = floor(y);
    const int64_t i = (int64_t)fx0, j = (int64_t)fy0;
    const double tx = x - fx0, ty = y - fy0;
    const double d00 = die_perlin_dot(seed, i, j, tx, ty), d10 = die_perlin_dot(seed, i + 1, j, tx - 1.0, ty);
    const double d01 = die_perlin_dot(seed, i, j + 1, tx, ty - 1.0), d11 = die_perlin_dot(seed, i + 1, j + 1, tx - 1.0, ty - 1.0);
    const double u = tx * tx * tx * (tx * (tx * 6.0 - 15.0) + 10.0), v = ty * ty * ty * (ty * (ty * 6.0 - 15.0) + 10.0);
    const double a = d00 + u * (d10 - d00), b = d01 + u * (d11 - d01);
    return a + v * (b - a);
}
