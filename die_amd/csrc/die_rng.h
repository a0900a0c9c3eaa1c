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

// Random turn sign of PhysarumAgent._choose_turn (core/agent/gradient.py:183: np.random.randint(0, 2) per slot, from an
// unseeded global generator).  One Philox block serves 128 slots: word w = slot >> 5 of the step's bit table is word
// (w & 3) of Philox(counter = (w >> 2, 0, step, TURN), key = seed), slot's bit is bit (slot & 31) of it.  A kernel either
// evaluates the block itself (die_turn_bit) or reads the table a generator kernel has filled (die_turn_bits_fill,
// die_pic.hip: 1/128 of the Philox work per step) — same bits; oracle/rng.py turn_signs is the numpy twin.
__host__ __device__ inline uint32_t die_turn_word(uint64_t seed, uint32_t step, uint32_t w) {
    const die_u32x4 r = die_philox(w >> 2, 0u, step, DIE_STREAM_TURN, (uint32_t)seed, (uint32_t)(seed >> 32));
    // (selects, not r.v[w & 3]: a dynamically indexed register array is moved to LDS by the compiler — in the classic agent kernel
    // that took 5 KB of LDS per workgroup and, at 1024², 34 µs instead of 13)
    const uint32_t lo = (w & 1u) ? r.v[1] : r.v[0], hi = (w & 1u) ? r.v[3] : r.v[2];
    return (w & 2u) ? hi : lo;
}
__host__ __device__ inline uint32_t die_turn_bit(uint64_t seed, uint32_t step, uint32_t slot) {
    return (die_turn_word(seed, step, slot >> 5) >> (slot & 31u)) & 1u;
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

// 3-D gradient noise for PerlinNoiseSequence (core/data_init.py:55-69: noise((x, y, t))): a unit gradient on every lattice
// point of the integer grid — uniform on the sphere: z = 2·u1 − 1, azimuth 2π·u2 from one Philox draw keyed by the point
// (step word 1 keeps it apart from the 2-D lattice) — the eight corner dot products blended with the quintic fade.
// oracle/cpu_ref.py perlin3 is the same function in numpy.
__device__ inline double die_perlin_dot3(uint64_t seed, int64_t i, int64_t j, int64_t k, double dx, double dy, double dz) {
    const uint64_t key = (uint64_t)(i & 0xFFFFF) | ((uint64_t)(j & 0xFFFFF) << 20) | ((uint64_t)(k & 0xFFFFF) << 40);
    const die_u32x4 r = die_draw(seed, 1u, key, DIE_STREAM_INIT_FOOD);
    const double gz = 2.0 * ((double)r.v[0] * (1.0 / 4294967296.0)) - 1.0;
    const double gr = sqrt(fmax(1.0 - gz * gz, 0.0));
    const double th = 6.283185307179586476925 * ((double)r.v[1] * (1.0 / 4294967296.0));
    return gr * cos(th) * dx + gr * sin(th) * dy + gz * dz;
}
__device__ inline double die_fade5(double t) { return t * t * t * (t * (t * 6.0 - 15.0) + 10.0); }
__device__ inline double die_perlin3(uint64_t seed, double x, double y, double z) {
    const double fx0 = floor(x), fy0 = floor(y), fz0 = floor(z);
    const int64_t i = (int64_t)fx0, j = (int64_t)fy0, k = (int64_t)fz0;
    const double tx = x - fx0, ty = y - fy0, tz = z - fz0;
    const double u = die_fade5(tx), v = die_fade5(ty), w = die_fade5(tz);
    double plane[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const double dz = tz - (double)c;
        const double d00 = die_perlin_dot3(seed, i, j, k + c, tx, ty, dz), d10 = die_perlin_dot3(seed, i + 1, j, k + c, tx - 1.0, ty, dz);
        const double d01 = die_perlin_dot3(seed, i, j + 1, k + c, tx, ty - 1.0, dz), d11 = die_perlin_dot3(seed, i + 1, j + 1, k + c, tx - 1.0, ty - 1.0, dz);
        const double a = d00 + u * (d10 - d00), b = d01 + u * (d11 - d01);
        plane[c] = a + v * (b - a);
    }
    return plane[0] + w * (plane[1] - plane[0]);
}

