// r3d_rng.h -- counter-based Philox4x32-10 stream of the HIP engine.
//
// The reference draws from libc rand() seeded with time(NULL)
// (model.cpp:235; call sites probability.cpp:111, scatterers.cpp:299,
// rtcoef.cpp:414,447), which is neither reproducible nor parallel.  The
// engine keys every history by its id instead:
//
//   draw k (k = 0,1,2,...) of history `id` under key `seed` is the 53-bit
//   uniform in (0,1] made of words [2(k&1), 2(k&1)+1] of
//   Philox4x32-10(counter = {id_lo, id_hi, k>>1, 0}, key = {seed_lo, seed_hi});
//   an event that takes TWO uniforms -- a reflection / transmission (S polarisation kind, outcome),
//   a scattering (conversion, deflection) -- takes the two halves of ONE block: it first skips to
//   the next even k.  (Ten rounds then serve both; drawn one by one they would straddle two
//   blocks on half the lanes of a wave, which costs the wave two evaluations every time.)
//
// so results do not depend on which lane, wave or GPU runs a history.  All of
// the reference's uniform conventions ([0,1], (0,1], 1-[0,1)) are mapped to
// (0,1]; they differ from it by at most one part in 2^31.
#ifndef R3D_RNG_H_
#define R3D_RNG_H_

#include <stdint.h>

#include "r3d_math.h"

namespace r3d {

struct Rng {
  uint32_t id_lo, id_hi;   // history id = Philox counter words 0,1
  uint32_t k;              // next draw index
};
// (The second half of a block is not cached between draws: lanes of a wave sit at
//  different parities of k, so a draw site runs the ten rounds for the wave whether or not
//  this lane would have had its words at hand -- and the cache cost two registers.)
struct RngKey {            // the run's seed: the same for every history, so not kept per lane
  uint32_t k0, k1;
};
R3D_HD RngKey rng_key(uint64_t seed) { return RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)}; }

R3D_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

R3D_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                 uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    // (one 32 x 32 -> 64 multiply per product: v_mad_u64_u32 gives both halves)
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

// (m + 1) 2^-53 with m = (hi >> 5) 2^26 + (lo >> 6), formed from its two halves: each converts exactly,
// and their scaled sum -- a multiple of 2^-53 in (0, 1] -- is exact too, so this IS (double)(m + 1) * 2^-53
// (oracle/philox.h) without the 64-bit integer arithmetic and the 64-bit conversion.
R3D_HD double u01_from_words(uint32_t hi, uint32_t lo) {
  const double a = (double)(hi >> 5), b = (double)((lo >> 6) + 1u);
  return __builtin_fma(a, 1.0 / 134217728.0, b * (1.0 / 9007199254740992.0));
}

R3D_HD void rng_init(Rng& g, uint64_t id) {
  g.id_lo = (uint32_t)id, g.id_hi = (uint32_t)(id >> 32);
  g.k = 0;
}

// the two uniforms of a two-draw event: the next whole block (see above)
R3D_HD void rng_draw_pair(Rng& g, RngKey key, double& u0, double& u1) {
  const uint32_t k = (g.k + 1u) & ~1u;
  uint32_t w[4];
  philox4x32_10(g.id_lo, g.id_hi, k >> 1, 0u, key.k0, key.k1, w);
  g.k = k + 2u;
  u0 = u01_from_words(w[0], w[1]), u1 = u01_from_words(w[2], w[3]);
}

R3D_HD double rng_draw(Rng& g, RngKey key) {
  uint32_t w[4];
  philox4x32_10(g.id_lo, g.id_hi, g.k >> 1, 0u, key.k0, key.k1, w);
  const bool second = (g.k & 1u) != 0;
  g.k++;
  return u01_from_words(second ? w[2] : w[0], second ? w[3] : w[1]);
}

}  // namespace r3d
#endif
