// r3d_wave.h -- wave-level helpers and the per-workgroup bin accumulators of the traversal kernel
// (device code; included by r3d_kernels_kind.hip ahead of r3d_pool.h).
#ifndef R3D_WAVE_H_
#define R3D_WAVE_H_

#include <hip/hip_runtime.h>

#include "r3d_kernels.h"
#include "r3d_step.h"

namespace r3d {

// ---------------------------------------------------- wave-level helpers ----
// the lanes for which c holds, as the compare instruction leaves them (HIP's __ballot(int) goes
// through a 0 / 1 value and a second compare)
__device__ __forceinline__ unsigned long long ballot(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ bool any_lane(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
// Ordering of this wave's LDS accesses as other waves of the workgroup see them (queue entries
// against slot state).  Named for the LDS alone: the generic fence also waits for every outstanding
// access to HBM -- a collection's bin updates, microseconds -- which no queue protocol depends on.
#define R3D_LDS_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define R3D_LDS_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")
// number of set bits of m below this lane
__device__ __forceinline__ unsigned rank_in(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// Inclusive scans over the wave's 64 lanes on the vector unit's data-parallel primitives (row shifts within rows of 16,
// then the row broadcasts of this architecture): six instructions, no trip through the LDS crossbar -- a scan by
// __shfl_up is six DEPENDENT ds_bpermute round trips, 300-400 cycles each behind the other waves' slot traffic.
// (All 64 lanes must be active.)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {   // (of non-negative values: the identity is 0)
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return v;
}
__device__ __forceinline__ double bcast(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ V3 bcast(V3 v, int src) { return v3(bcast(v.x, src), bcast(v.y, src), bcast(v.z, src)); }

// Per-workgroup accumulators for seismometer bins, in LDS.  First arrivals pile
// onto a handful of (seismometer, time-bin) records -- in the LopNor runs one bin
// takes a quarter of all catches and sixteen take 58 % -- and atomics on one
// address are served one after the other by a single L2 channel: measured, that
// contention alone was half of the LopNor kernel time.  So a catch first tries a
// small open-addressed table here (first come, first admitted; hot bins show up
// early and often); the block adds each entry to HBM once, at the end.  A catch
// that finds no entry takes the queue above.
struct BinCache {
  double* e;        // [n][5] energies X, Y, Z, P, S
  uint32_t* key;    // [n]    seismometer * n_bins + bin, or kEmpty
  uint32_t* cnt;    // [n][2] catches by type
  uint32_t mask, shift;   // n - 1, 32 - log2 n
  bool on;
};
constexpr uint32_t kEmpty = 0xFFFFFFFFu;

__device__ __forceinline__ bool bin_cache_add(const BinCache& bc, uint32_t bin, uint32_t type, double ex,
                                              double ey, double ez, double et) {
  uint32_t idx = (bin * 2654435761u) >> bc.shift;
  for (int probe = 0; probe < 4; probe++) {
    const uint32_t old = atomicCAS(&bc.key[idx], kEmpty, bin);
    if (old == kEmpty || old == bin) {
      double* e = bc.e + idx * 5u;
      unsafeAtomicAdd(e + 0, ex);
      unsafeAtomicAdd(e + 1, ey);
      unsafeAtomicAdd(e + 2, ez);
      unsafeAtomicAdd(e + 3 + type, et);
      atomicAdd(&bc.cnt[idx * 2u + type], 1u);
      return true;
    }
    idx = (idx + 1u) & bc.mask;
  }
  return false;
}

// RES: which of the small tables are staged in LDS (the enum itself: r3d_kernels.h).  RES_ALL: the cell records and the
// scatterer heads (layered and spherical models: a few dozen cells); RES_TABLES: the scatterer
// heads only (tetra models: the cell records come through L1 / L2); RES_NONE: neither (models
// with thousands of scatterers, whose heads alone would crowd out the phonon pool).

}  // namespace r3d
#endif
