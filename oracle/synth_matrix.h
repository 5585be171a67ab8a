/* =============================================================================
 *  TEST INFRASTRUCTURE — CPU statement of the synthetic benchmark matrices
 *  (SURVEY.md §8(d) "M-band" / "M-rand" / C4 non-symmetric variant).
 *
 *  The product generates the same matrices on the GPU
 *  (spectra_amd/csrc/synth.hip); this file is the independent CPU statement the
 *  tests compare against.  Everything is pure 64-bit integer hashing followed
 *  by one exact int->double conversion, so CPU and GPU agree bit for bit.
 *
 *  value(seed, a, b) = U(-0.5, 0.5) from a splitmix64-style counter hash.
 *  Symmetric matrices use (a, b) = (min(i,j), max(i,j)); the non-symmetric
 *  variant uses (i, j).
 * ============================================================================= */
#ifndef ORACLE_SYNTH_MATRIX_H
#define ORACLE_SYNTH_MATRIX_H

#include <stdint.h>

static inline uint64_t synth_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline double synth_value(uint64_t seed, uint64_t a, uint64_t b)
{
    const uint64_t k = synth_mix64(synth_mix64(seed ^ a) ^ (b * 0xD6E8FEB86659FD93ULL));
    /* top 53 bits -> [0,1) exactly, then centre */
    return (double) (k >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}

#endif
