// r3d_tables.h -- the engine's own HBM/LDS data layout.
//
// The C-ABI hands over AoS structs that mirror the reference's objects
// (include/r3d.h).  The engine repacks them once at r3d_engine_create into
// what the kernels want:
//   * one fixed-size, 16-B aligned record per cell and per cell kind, holding
//     only what the traversal reads, with derived constants folded in
//     (plane offsets n.p, 1/|grad v|, the attenuation exponent -pi f / Q);
//   * densities (needed only by the R/T solve) in a side array;
//   * take-off directions as unit vectors instead of (theta, phi);
//   * seismometers split into a small "scan" record (position + gather radii,
//     staged in LDS) and a "hit" record (axes, 1/(bin width * area)) fetched
//     from HBM only on a hit, plus a uniform spatial hash over the scan
//     records so a surface arrival tests a handful of candidates, not all.
#ifndef R3D_TABLES_H_
#define R3D_TABLES_H_

#include <stdint.h>

// Timing-only developer switches: R3D_ABLATE_* cut a part of the work out of the kernel (one of them,
// R3D_ABLATE_RT, changes the physics) and R3D_PHASE_TIMING adds per-phase cycle counters.  They exist
// in `make variant` builds, which say -DR3D_DEV_BUILD; the shipped libraries (`make all`) never define
// it, and with this guard no build without it can contain any of them (tests/test_abi.py holds the
// Makefile and this list against the sources).
#if !defined(R3D_DEV_BUILD) && (defined(R3D_ABLATE_CATCH) || defined(R3D_ABLATE_COLLECT) || defined(R3D_ABLATE_RT) || \
                                defined(R3D_ABLATE_SPRAY_SEARCH) || defined(R3D_ABLATE_SCATTER) || defined(R3D_PHASE_TIMING) || \
                                defined(R3D_STEP_FINALS))
#error "R3D_ABLATE_* / R3D_PHASE_TIMING / R3D_STEP_FINALS are developer-build switches: build with -DR3D_DEV_BUILD (make variant)"
#endif

namespace r3d {

// face flag byte (same bit values as include/r3d.h)
enum : uint32_t { F_COLLECT = 1u, F_REFLECT = 2u, F_ADJOIN = 4u, F_DISCON = 8u,
                  // engine-only, derived at pack time for ADJOIN faces without DISCON (see
                  // classify_velocity_step in r3d_pack.h): the reference's run-time test
                  // "fractional velocity step > 1e-5" (phonons.cpp:243-252) has the same
                  // outcome everywhere on the face
                  F_SMOOTH = 16u,   // never exceeds it: plain hand-over
                  F_STEP = 32u };   // always exceeds it: Snell bend

// ---- cells -----------------------------------------------------------------
// EVERY KIND HAS ONE RECORD PER CELL AND RAY TYPE (index 2 * cell + type), holding what a move of
// that type reads and nothing of the other type: a move copies its record into registers in one
// go of 16-byte loads (r3d_step.h step_move), where fields indexed by the ray type would be a table
// in scratch memory -- or loads left at their uses, each a round trip of its own.
// Layered cylinder cell (reference RCUCylinder, media.hpp:312-331): uniform
// velocity, top and bottom planes; the lateral wall radius is a model constant.
struct alignas(16) CellCyl {
  double v, att;      // velocity of this ray type; -pi f / Q
  double n[2][3];     // outward unit normals of top, bottom
  double d[2];        // plane offsets n . point
  int32_t nbr[2];
  uint32_t flags;     // byte f = flags of face f
  int32_t scat;
  double rho, pad_;
};
static_assert(sizeof(CellCyl) == 112, "seven 16-byte loads");

// Tetrahedral cell with linear velocity (reference Tetra, media.hpp:400-408): ONE RECORD PER CELL
// AND RAY TYPE (index 2 * cell + type), three 64-byte lines holding everything a move of that type
// reads -- twelve 16-byte loads per lane.  A gather from thousands of records costs the L1 path
// ~55-70 cycles per 64-lane load instruction whatever its width (tools/microbench/gather.hip: 445 ns
// per wave for a 256-byte record fetched as 16 x 16 bytes, 275 ns for 192 bytes as 12 x 16, 730 ns for
// 192 bytes as 24 x 8), and the tetra move's fetch kept the texture path busy two thirds of the
// launch (profiles/r02: TA_BUSY 66 %): so no field of the other ray type, no 8- or 4-byte loads.
// link[f]: the neighbour behind face f + 1 in bits 0-21 (0: none), the face's flag byte's low six bits
// in bits 22-27, and in bits 28-31 nibble f of the cell's scatterer index.
struct alignas(64) CellTet {
  double g[3];        // grad v (this ray type)
  double v0;          // velocity at the origin
  double inv_gmag;    // 1 / |grad v|
  double att;         // -pi f / Q
  double n[4][3];
  double d[4];
  uint32_t link[4];
};
static_assert(sizeof(CellTet) == 192, "a tetra record is three 64-byte lines");
constexpr uint32_t kTetNbrBits = 22, kTetNbrMask = (1u << kTetNbrBits) - 1u;
#if defined(__HIPCC__)
#define R3D_TBL_HD __host__ __device__
#else
#define R3D_TBL_HD
#endif
R3D_TBL_HD inline int tet_link_neighbor(uint32_t link) { return (int)(link & kTetNbrMask) - 1; }
R3D_TBL_HD inline uint32_t tet_link_flags(uint32_t link) { return (link >> kTetNbrBits) & 0x3Fu; }
R3D_TBL_HD inline int tet_link_scat(const uint32_t link[4]) {
  return (int)((link[0] >> 28) | ((link[1] >> 28) << 4) | ((link[2] >> 28) << 8) | ((link[3] >> 28) << 12));
}

// Spherical shell, v(r) = a r^2 + c (reference SphereShell, media.hpp:467-478); per ray type, as above.
struct alignas(16) CellSph {
  double a, c, zero_rad2, att;
  double radius[2];   // signed: +top (outward normal), -bottom (inward)
  int32_t nbr[2];
  uint32_t flags;
  int32_t scat;
  double rho_a, rho_c;
};
static_assert(sizeof(CellSph) == 80, "five 16-byte loads");

struct RhoLin {       // density side table for tetra cells
  double g[3], c;
};

// One cell of a cumulative table's search guide: a draw u in [j / G, (j + 1) / G), G = 2^guide_bits,
// has its answer inside [k1, k2] (k1 = smallest k with total j / G <= cdf[k], k2 the same for
// j + 1).  c[] holds, for a bracket of up to seven entries -- 97 % of the draws --, the table's
// entries k1 .. k2 - 1: the draw is decided by this ONE 64-byte fetch.  For a longer bracket it
// holds seven pivots, the entries at guide_pivot(k1, k2, i): they cut the bracket into eighths, and
// one further fetch of eight neighbouring entries finishes all but the longest (a lane in a long
// bracket is what the whole wave waits for: bisecting it took three to six dependent round trips).
struct alignas(64) GuideCell {
  uint32_t k1, k2;
  double c[7];
};
static_assert(sizeof(GuideCell) == 64, "a guide cell is one 64-byte sector");
constexpr int kGuideVals = 7;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t guide_pivot(uint64_t k1, uint64_t k2, int i) { return k1 + (((uint64_t)(i + 1) * (k2 - k1)) >> 3); }

// ---- scatterers ------------------------------------------------------------
struct ScatHead {     // small per-scatterer record, staged in LDS
  double mfp[2];
  double whole[2][4]; // cumulative conversion weights for incoming P / S
  double total[4];    // cdf[k][n_toa-1]
};
struct ScatPtrs {     // HBM-resident tables of one scatterer
  const double* cdf[4];
  // S -> S polarisation of each deflection as (cosine, sine) pairs: what the transform takes, one 16-byte
  // fetch in place of an angle and its sine / cosine polynomials at every S -> S scattering
  const double* spol_cs;
  const GuideCell* guide[4]; // search guides of the four CDFs (see sample_cdf_guided)
};

// ---- seismometers ----------------------------------------------------------
struct SeisScan {     // 56 B, staged in LDS
  double loc[3];
  double r_in[2], r_out[2];
};
struct SeisHit {      // fetched on a hit
  double axes[3][3];
  double inv_norm[2]; // 1 / (time_per_bin * area[type])
};
struct SeisGrid {     // uniform hash over seismometer gather spheres
  double origin[3];
  double inv_h;
  int32_t dim[3];
  int32_t n_cells;
  double dim_f[3];         // dim as doubles (the bounds test runs in fp64)
  const uint32_t* start;   // n_cells + 1 offsets into items
  const uint32_t* items;   // seismometer indices
};

// ---- everything a launch needs (passed by value) ---------------------------
struct KArgs {
  // model
  const void* cells;         // CellCyl / CellTet / CellSph array
  const RhoLin* rho;         // tetra only
  int32_t n_cells;
  int32_t n_scat;
  int32_t n_seis;
  uint32_t n_bins;
  const ScatHead* scat_head;
  const ScatPtrs* scat_ptrs;
  const SeisScan* seis_scan;
  const SeisHit* seis_hit;
  SeisGrid grid;
  uint64_t n_toa;
  // take-off directions as {cos theta, cos phi, sin phi, sin theta} (theta already nudged): two 16-byte loads give
  // the unit vector (st cp, st sp, ct) AND its theta^ / phi^ axes (ct cp, ct sp, -st), (-sp, cp, 0) -- what a
  // scattering would otherwise rebuild from the vector with a reciprocal square root
  const double* toa_dir;     // n_toa x 4
  const double* src_cdf[3];
  const GuideCell* src_guide[3];
  double src_total[3];       // src_cdf[k][n_toa-1]
  uint32_t guide_bits;       // guides have 2^guide_bits cells
  uint32_t pad0_;
  double src_whole[3];
  double src_loc[3];
  int32_t src_cell;
  uint32_t no_deflect;
  // scalars
  double ttl, time_per_bin, inv_time_per_bin, slow_concern;
  uint64_t loop_concern;
  double nodeflect_dir[4];   // the same four numbers at theta = min_theta, phi = 0
  double cyl_radius2;        // cylinder models: wall radius squared
  double earth_center[3];
  // work
  uint64_t n, first_id, seed;
  unsigned long long* next;  // device work counter
  // results
  double* energy;
  unsigned long long* counts;
  unsigned long long* scalars;
  void* finals;              // r3d_final[n] or null (the diagnostic kernel's: indexed by id - first_id)
  // final records out of the PRODUCTION kernels (r3d_engine_set_production_finals): the record of history `id` is
  // pfinals[id] -- the buffer's address less its first id, so that the kernel holds ONE word for it (the engine
  // refuses launches whose ids the buffer does not cover); null = off
  void* pfinals;
  // optional volumetric scatter-event grid (null = off): count[type][frame][z][y][x]
  unsigned int* vol;
  double vol_origin[3], vol_inv_cell[3], vol_inv_dt;
  uint32_t vol_dim[3], vol_frames;
  double vol_dim_f[3], vol_frames_f;   // the same as doubles, for the bounds test
  double n_bins_f;           // (double)n_bins
  // optional per-event report buffer (null = off): r3d_event records, see include/r3d.h
  void* evlog;
  unsigned long long* evlog_count;
  uint64_t evlog_cap;
  uint32_t evlog_mask;
  uint32_t pad2_;
  // optional carry-over of unfinished histories between launches (null = off): one CarrySlot
  // per work-item of the grid, see r3d_engine.hip
  void* carry_in;
  void* carry_out;
  // LDS carve-up (bytes from the start of dynamic shared memory)
  uint32_t lds_cells_off;    // 0xFFFFFFFF: cells stay in HBM
  uint32_t lds_scat_off;
  uint32_t lds_scatptr_off;  // ScatPtrs[n_scat] beside the heads
  uint32_t lds_seis_off;
  uint32_t lds_hit_off;
  uint32_t lds_grid_off;     // seismometer hash (start u32[n_cells+1], items u16[n_items]); 0xFFFFFFFF: in HBM
  uint32_t grid_n_items;
  uint32_t lds_acc_off;      // bin accumulators (see BinCache in r3d_engine.hip)
  uint32_t acc_bits;         // 2^acc_bits accumulator entries; 0: none
  // phonon pool (r3d_pool.h): slots of histories in flight and the rings of the phase queues
  uint32_t pool_slots;       // S, a multiple of 64
  uint32_t pool_ring_mask;   // ring capacity - 1 (capacity: the power of two >= S)
  uint32_t lds_pool_off;
  uint32_t lds_ring_off;
};

}  // namespace r3d
#endif
