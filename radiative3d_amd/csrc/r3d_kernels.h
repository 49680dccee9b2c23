// r3d_kernels.h -- what the engine (r3d_engine.hip) and the three traversal-kernel translation units
// (r3d_kernels_kind.hip compiled with -DR3D_KIND=0 / 1 / 2) agree on: the pool's geometry in LDS and
// the launch entry points of each cell kind.
#ifndef R3D_KERNELS_H_
#define R3D_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "r3d_tables.h"

namespace r3d {

// 12 waves = 3 per SIMD, i.e. a budget of 168 registers per lane.  At two waves per SIMD a wave that
// waits (memory round trips, dependent fp64 chains) is covered by one other only, so a third is
// worth 15-25 % -- once every phase fits the budget: the R/T solve in two halves with nothing but
// the choice carried across, event counters that live for one batch, the launch arguments fetched
// per batch (0-4 vector registers spilled).  Measured at 512 / 768 threads, same code otherwise:
// NSCP 15.8 / 12.8 ms, LopNor 13.9 / 11.0, SphereEarth 40.8 / 33.0 per 3e6 histories.
constexpr int kPoolBlock = 768;

enum { Q_MOVE = 0, Q_COLLECT = 1, Q_RT = 2, Q_SCATTER = 3, Q_FREE = 4, Q_NUM = 5 };

// One history in flight = one slot number; its state is eight 16-byte records (twelve doubles, eight
// words: r3d_pool.h), each kind of record an array over the slots in LDS, so that a batch of slot
// numbers reads and writes a record across the banks (slot-major 128-byte blocks measured 77 % of
// the LDS cycles as conflicts).  meta: bit 0 ray type | bits 1-3 pending face + 1 (0: none) |
// bits 8-15 that face's flags | bits 16-18 the queue the slot is in (for carry-over).
constexpr size_t kSlotBytes = 128;
constexpr uint32_t kSlotStride = 1024;   // entries per record array of the pool (>= the slots in circulation)

// bytes of one bin accumulator of the per-workgroup table (BinCache, r3d_wave.h)
constexpr size_t kAccEntryBytes = 5 * sizeof(double) + 3 * sizeof(uint32_t);

// RES: which of the small tables are staged in LDS.  RES_ALL: the cell records and the
// scatterer heads (layered and spherical models: a few dozen cells); RES_TABLES: the scatterer
// heads only (tetra models: the cell records come through L1 / L2); RES_NONE: neither (models
// with thousands of scatterers, whose heads alone would crowd out the phonon pool).
enum { RES_ALL = 0, RES_TABLES = 1, RES_NONE = 2 };

// One launch of the traversal kernel of a cell kind: res = RES_*, trace = the diagnostic variant
// (final records, report stream), drain_only = the flush launch of a carry chain (pool_drain_kernel).
// grid workgroups of kPoolBlock threads, lds_bytes of dynamic LDS, on stream s.
hipError_t launch_pool_cyl(int res, bool trace, bool drain_only, unsigned grid, size_t lds_bytes, hipStream_t s, const KArgs& a);
hipError_t launch_pool_tet(int res, bool trace, bool drain_only, unsigned grid, size_t lds_bytes, hipStream_t s, const KArgs& a);
hipError_t launch_pool_sph(int res, bool trace, bool drain_only, unsigned grid, size_t lds_bytes, hipStream_t s, const KArgs& a);
// hipFuncAttributeMaxDynamicSharedMemorySize for the three kernels of (kind, res)
hipError_t pool_lds_attr_cyl(int res, int lds_bytes);
hipError_t pool_lds_attr_tet(int res, int lds_bytes);
hipError_t pool_lds_attr_sph(int res, int lds_bytes);
#ifdef R3D_PHASE_TIMING
int pool_stats_cyl(unsigned long long out[40]);   // (diagnostic builds: read and reset the unit's counters)
int pool_stats_tet(unsigned long long out[40]);
int pool_stats_sph(unsigned long long out[40]);
#endif

}  // namespace r3d
#endif
