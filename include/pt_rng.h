/*
 * pt_rng.h -- the counter-based random number generator that defines the
 * sample sequences of the MI355X render loop.
 *
 * It replaces class Prng (reference src/15/prng.hpp:7-22: one mt19937_64
 * stream per pixel, 2.5 KB of state) with a stateless function of
 * (seed, pixel, sample, draw number), so that any (pixel, sample) can be
 * evaluated on any GPU in any order and gives the same numbers.  The draw
 * number advances by one exactly where the reference calls Prng::in01().
 *
 *   mix(z)                 = the SplitMix64 output function
 *   key(seed, pixel, samp) = mix(mix(seed + G * (pixel + 1)) + G * (samp + 1))
 *   bits(key, n)           = mix(key + G * (n + 1))          n = 0, 1, 2, ...
 *   in01(key, n)           = (bits >> 40) * 2^-24            in [0, 1)
 *
 * with G = 0x9E3779B97F4A7C15 and pixel = y * width + x of the whole image.
 * Usable from C, C++ and HIP device code.
 */
#ifndef PT_RNG_H
#define PT_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PT_RNG_FN __host__ __device__ static inline
#else
#define PT_RNG_FN static inline
#endif

#define PT_RNG_GOLDEN 0x9E3779B97F4A7C15ull

PT_RNG_FN uint64_t pt_rng_mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

PT_RNG_FN uint64_t pt_rng_key(uint64_t seed, uint64_t pixel, uint64_t sample)
{
    return pt_rng_mix(pt_rng_mix(seed + PT_RNG_GOLDEN * (pixel + 1)) + PT_RNG_GOLDEN * (sample + 1));
}

PT_RNG_FN float pt_rng_in01(uint64_t key, uint32_t n)
{
    uint64_t bits = pt_rng_mix(key + PT_RNG_GOLDEN * ((uint64_t)n + 1));
    return (float)(uint32_t)(bits >> 40) * 5.9604644775390625e-08f; /* 2^-24, exact */
}

#endif /* PT_RNG_H */
