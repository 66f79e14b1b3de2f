/*
 * spl_synth.h — specification of the synthetic workloads.
 *
 * Counter-based generators: every entry of every synthetic matrix / vector is a
 * pure function of (seed, row, draw-index), so the CPU oracle, the HIP
 * generator kernels and every rank of a multi-GPU run regenerate identical
 * data without any transfer.  Integer arithmetic only (plus one exact
 * int->double conversion), so host and device agree bit for bit.
 *
 * This header is valid C99, C++ and HIP.  It specifies DATA, not the
 * reference's algorithms; both the product (csrc/generate.hip) and the test
 * oracle (oracle/sparse_oracle.c) include it.
 */
#ifndef SPL_SYNTH_H
#define SPL_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define SPL_HD __host__ __device__ static inline
#else
#define SPL_HD static inline
#endif

#define SPL_SEED_A 0x5EEDull   /* matrix seed   */
#define SPL_SEED_X 0xBEEFull   /* vector seed   */
#define SPL_VAL_SALT 0xA5A5A5A5A5A5A5A5ull
#define SPL_GOLDEN 0x9E3779B97F4A7C15ull
#define SPL_MAX_DRAWS 64       /* draws per row must be <= 64 */
#define SPL_BAND_DIAGS 20      /* banded variant: 20 diagonals ... */
#define SPL_BAND_HALF 1000     /* ... at fixed offsets within +-1000 */

/* splitmix64 finaliser */
SPL_HD uint64_t spl_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* hash of (seed, i, k), k < SPL_MAX_DRAWS */
SPL_HD uint64_t spl_hash(uint64_t seed, uint64_t i, uint64_t k) {
  return spl_mix64(seed + SPL_GOLDEN * (i * SPL_MAX_DRAWS + k + 1));
}

/* high 64 bits of a*b */
SPL_HD uint64_t spl_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

/* uniform integer in [0, n) */
SPL_HD uint64_t spl_uniform_index(uint64_t h, uint64_t n) { return spl_mulhi64(h, n); }

/* uniform double in [0.5, 1.5): 53 random mantissa bits, exact */
SPL_HD double spl_uniform_value(uint64_t h) {
  return 0.5 + (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

/* k-th column draw / value draw of row r of the `random(n, K)` matrix */
SPL_HD uint64_t spl_random_col(uint64_t seed, uint64_t r, uint64_t k, uint64_t n) {
  return spl_uniform_index(spl_hash(seed, r, k), n);
}
SPL_HD double spl_random_val(uint64_t seed, uint64_t r, uint64_t k) {
  return spl_uniform_value(spl_hash(seed ^ SPL_VAL_SALT, r, k));
}

/* dense vector entry j */
SPL_HD double spl_vector_entry(uint64_t seed, uint64_t j) {
  return spl_uniform_value(spl_hash(seed, j, 0));
}

/* d-th diagonal offset of the banded variant, d in [0, SPL_BAND_DIAGS):
 * strictly increasing, spans [-1000, +1000], never 0 (d*2000/19 is never 1000) */
SPL_HD int64_t spl_band_offset(int d) {
  return (int64_t)(-SPL_BAND_HALF) + ((int64_t)d * (2 * SPL_BAND_HALF)) / (SPL_BAND_DIAGS - 1);
}

/* R-MAT: endpoint pair of edge e for a 2^scale graph with quadrant
 * probabilities (a,b,c,d) given as 32-bit fixed point thresholds
 * ta = a*2^32, tb = (a+b)*2^32, tc = (a+b+c)*2^32.  One hash per level. */
SPL_HD void spl_rmat_edge(uint64_t seed, uint64_t e, int scale, uint32_t ta, uint32_t tb,
                          uint32_t tc, uint64_t *row, uint64_t *col) {
  uint64_t r = 0, c = 0;
  for (int l = 0; l < scale; ++l) {
    uint32_t u = (uint32_t)(spl_hash(seed, e, (uint64_t)l) >> 32);
    unsigned rb, cb;
    if (u < ta) { rb = 0; cb = 0; }
    else if (u < tb) { rb = 0; cb = 1; }
    else if (u < tc) { rb = 1; cb = 0; }
    else { rb = 1; cb = 1; }
    r = (r << 1) | rb;
    c = (c << 1) | cb;
  }
  *row = r;
  *col = c;
}

#endif /* SPL_SYNTH_H */
