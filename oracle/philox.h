/* philox.h -- Philox4x32-10 counter-based RNG (Salmon, Moraes, Dror, Shaw,
 * "Parallel random numbers: as easy as 1, 2, 3", SC'11), restated from the
 * published algorithm for the ORACLE.  TEST INFRASTRUCTURE: the engine has
 * its own copy (radiative3d_amd/csrc/r3d_rng.h); tests check the two agree
 * with the paper's known-answer vectors.
 *
 * Draw convention shared by oracle and engine (the reference itself uses
 * libc rand(), model.cpp:235, which is neither reproducible nor parallel):
 *   draw k (k = 0,1,2,...) of history `id` under key `seed` is the 53-bit
 *   uniform in (0,1] built from words [2*(k&1), 2*(k&1)+1] of
 *   Philox4x32-10(counter = {id_lo, id_hi, k>>1, 0}, key = {seed_lo, seed_hi}).
 *   An event that takes two uniforms (reflection / transmission: S polarisation kind and outcome;
 *   scattering: conversion and deflection) takes the two halves of ONE block: it first skips to
 *   the next even k (oracle_rng_draw_pair).
 */
#ifndef R3D_ORACLE_PHILOX_H_
#define R3D_ORACLE_PHILOX_H_

#include <stdint.h>

static inline void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                                        uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0, c1 = n1, c2 = n2, c3 = n3;
    k0 += W0, k1 += W1;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

/* 53-bit uniform in (0,1] from two 32-bit words. */
static inline double oracle_u01(uint32_t hi, uint32_t lo) {
  uint64_t m = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6); /* 27 + 26 bits */
  return (double)(m + 1) * (1.0 / 9007199254740992.0);
}

typedef struct oracle_rng {
  uint32_t id_lo, id_hi, key[2];
  uint32_t k;         /* next draw index */
  uint32_t cache[4];
} oracle_rng;

static inline void oracle_rng_init(oracle_rng* g, uint64_t seed, uint64_t id) {
  g->id_lo = (uint32_t)id, g->id_hi = (uint32_t)(id >> 32);
  g->key[0] = (uint32_t)seed, g->key[1] = (uint32_t)(seed >> 32);
  g->k = 0;
}

static inline double oracle_rng_draw(oracle_rng* g) {
  if ((g->k & 1u) == 0) {
    uint32_t ctr[4] = {g->id_lo, g->id_hi, g->k >> 1, 0};
    oracle_philox4x32_10(ctr, g->key, g->cache);
  }
  uint32_t w = 2 * (g->k & 1u);
  g->k++;
  return oracle_u01(g->cache[w], g->cache[w + 1]);
}

/* The two uniforms of a two-draw event: the next whole block. */
static inline void oracle_rng_draw_pair(oracle_rng* g, double* u0, double* u1) {
  g->k = (g->k + 1u) & ~1u;
  *u0 = oracle_rng_draw(g);
  *u1 = oracle_rng_draw(g);
}

#endif
