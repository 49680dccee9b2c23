// r3d_step.h -- one history: the source spray and one iteration of the
// propagation loop, for one work-item.
//
//   spray()  = ShearDislocation::GenerateEventPhonon (reference events.cpp:111-124,
//              sources.cpp:156-170, phonons.hpp:193-207)
//   step()   = one pass through the body of Phonon::Propagate
//              (reference phonons.cpp:540-682)
//
// Both are templates over the cell kind (a model is homogeneous in kind, so
// the choice is made once per launch, not per cell) and over where the small
// tables live.  They are __host__ __device__: r3d_pool.h calls the halves of
// the iteration from the phases of the pool kernel; tests/emul wraps them in
// a plain CPU loop.
#ifndef R3D_STEP_H_
#define R3D_STEP_H_

#include <type_traits>

#include "r3d_physics.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define R3D_ADD_F64(ptr, val) unsafeAtomicAdd((ptr), (val))
#define R3D_ADD_U64(ptr, val) atomicAdd((ptr), (unsigned long long)(val))
#define R3D_ADD_U32(ptr, val) atomicAdd((ptr), (unsigned int)(val))
#else
#define R3D_ADD_F64(ptr, val) (*(ptr) += (val))
#define R3D_ADD_U64(ptr, val) (*(ptr) += (unsigned long long)(val))
#define R3D_ADD_U32(ptr, val) (*(ptr) += (unsigned int)(val))
#endif

// (tests/emul counts the moves that take the reference's construction; nothing in a device build)
#ifndef R3D_COUNT_SLOW_MOVE
#define R3D_COUNT_SLOW_MOVE() ((void)0)
#endif

namespace r3d {

template <int KIND> struct CellOf;
template <> struct CellOf<CELL_CYL> { using type = CellCyl; };
template <> struct CellOf<CELL_TET> { using type = CellTet; };
template <> struct CellOf<CELL_SPH> { using type = CellSph; };

// Per-lane event tallies (reported through r3d_result.events).
struct LaneStats {
  uint32_t iterations, scatter, collect, n_catch, reflect, transfer, rtsolve;
  uint32_t vol_out;   // SCT / REF events outside an attached event grid (r3d_result.events[R3D_EV_VOLUME_OUT])
  uint32_t quiet;     // this lane only helps another lane's history along (r3d_pool.h kQuad): it leaves no mark in the grid
};

// Where the step finds its tables (pointers may be LDS or HBM).
template <int KIND>
struct Tables {
  const typename CellOf<KIND>::type* cells;
  const ScatHead* scat_head;
  const ScatPtrs* scat_ptrs;   // where each scatterer's tables are (the pointers themselves: LDS or HBM)
  const SeisScan* seis_scan;
  const SeisHit* seis_hit;
};

// ---- per-kind property lookups --------------------------------------------
// The record a phonon of ray type `type` in cell `cell` works from (one per cell and ray type:
// r3d_tables.h).  Where a record has been copied into registers, its small arrays are picked from
// by selects: an index would make the copy a table in scratch memory.
template <int KIND>
R3D_HD const typename CellOf<KIND>::type& cell_rec(const Tables<KIND>& T, int cell, int type) {
  return T.cells[2 * cell + type];
}
// velocity of its ray type at p, from the record that holds it
R3D_HD double cell_velocity(const CellCyl& c, V3, int) { return c.v; }
R3D_HD double cell_velocity(const CellTet& c, V3 p, int) { return dot(p, v3(c.g)) + c.v0; }
R3D_HD double cell_velocity(const CellSph& c, V3 p, int) { return c.c + c.a * mag2(p); }
template <int KIND>
R3D_HD double velocity_in(const Tables<KIND>& T, int cell, V3 p, int t) {
  return cell_velocity(cell_rec<KIND>(T, cell, t), p, t);
}

R3D_HD double cell_density(const KArgs&, const CellCyl& c, int, V3) { return c.rho; }
R3D_HD double cell_density(const KArgs& a, const CellTet&, int idx, V3 p) {
  const RhoLin r = a.rho[idx];
  return dot(p, v3(r.g)) + r.c;
}
R3D_HD double cell_density(const KArgs&, const CellSph& c, int, V3 p) { return c.rho_c + c.rho_a * mag2(p); }

R3D_HD V3 cell_face_normal(const CellCyl& c, int f, V3) {
  return f == 0 ? v3(c.n[0]) : v3(c.n[1]);   // (the wall, face 2, is never asked for: a phonon is lost there)
}
R3D_HD V3 cell_face_normal(const CellTet& c, int f, V3) { return v3(c.n[f]); }
R3D_HD V3 cell_face_normal(const CellSph& c, int f, V3 loc) {  // media_cellface.cpp:624-627
  V3 u = unit_else(loc, v3(0, 0, 1));
  return (f == 0 ? c.radius[0] : c.radius[1]) > 0 ? u : -u;
}
R3D_HD int cell_neighbor(const CellCyl& c, int f) { return f == 0 ? c.nbr[0] : f == 1 ? c.nbr[1] : -1; }
R3D_HD int cell_neighbor(const CellTet& c, int f) { return tet_link_neighbor(c.link[f]); }
R3D_HD int cell_neighbor(const CellSph& c, int f) { return f == 0 ? c.nbr[0] : c.nbr[1]; }
R3D_HD uint32_t cell_face_flags(const CellCyl& c, int f) { return face_flags(c.flags, f); }
R3D_HD uint32_t cell_face_flags(const CellTet& c, int f) { return tet_link_flags(c.link[f]); }
R3D_HD uint32_t cell_face_flags(const CellSph& c, int f) { return face_flags(c.flags, f); }
R3D_HD int cell_scat(const CellCyl& c) { return c.scat; }
R3D_HD int cell_scat(const CellTet& c) { return tet_link_scat(c.link); }
R3D_HD int cell_scat(const CellSph& c) { return c.scat; }

// ---- source spray ----------------------------------------------------------
R3D_HD void spray(const KArgs& a, Phonon& p, Rng& rng) {
  // 0 P, 1 SH, 2 SV: smallest k with r <= whole[k] (probability.cpp:104-128 on 3 entries)
  double u_type, u_dir;   // (a history's first two uniforms: one block of the generator)
  rng_draw_pair(rng, rng_key(a.seed), u_type, u_dir);
  const double r3 = a.src_whole[2] * u_type;
  const int rt3 = (r3 <= a.src_whole[0]) ? 0 : (r3 <= a.src_whole[1]) ? 1 : 2;
#ifdef R3D_ABLATE_SPRAY_SEARCH
  uint64_t k = (uint64_t)(u_dir * (double)(a.n_toa - 1));
#else
  // (selects, not a[rt3]: a dynamic index into the by-value argument block would
  //  make the compiler copy the arrays to scratch memory)
  const double* cdf = rt3 == 0 ? a.src_cdf[0] : rt3 == 1 ? a.src_cdf[1] : a.src_cdf[2];
  const GuideCell* guide = rt3 == 0 ? a.src_guide[0] : rt3 == 1 ? a.src_guide[1] : a.src_guide[2];
  const double total = rt3 == 0 ? a.src_total[0] : rt3 == 1 ? a.src_total[1] : a.src_total[2];
  uint64_t k = sample_cdf_guided(cdf, guide, a.guide_bits, total, u_dir);
#endif
  p.t = p.path = p.recent = 0.0;
  p.lamp = 0.0;
  p.moves = 0;
  {
    const double* d = a.toa_dir + 4 * k;   // {cos theta, cos phi, sin phi, sin theta}
    p.dir = v3(d[3] * d[1], d[3] * d[2], d[0]);
  }
  // mPol = pi/2 for SH, else 0 (phonons.hpp:200); cos(pi/2) in fp64 is 6.1e-17, not 0
  p.pc = (rt3 == 1) ? 6.123233995736766e-17 : 1.0;
  p.ps = (rt3 == 1) ? 1.0 : 0.0;
  p.type = (rt3 == 0) ? RAY_P : RAY_S;
  p.loc = v3(a.src_loc);
  p.cell = a.src_cell;
}

// The same draws, stopped early: spray_touch(level 0) requests one word of the guide cell the take-off draw
// will read, spray_touch(level 1) goes on to the take-off index and requests one word of its direction
// record.  A caller that starts several batches of histories runs these ahead of the sprays proper, so that
// the two dependent fetches of a spray -- from tables of hundreds of MB -- are on their way for all of
// them at once; the words themselves mean nothing (the caller only keeps them alive until the sprays).
// The generator is evaluated again by each: instructions are what a refill has to spare.
template <int LEVEL>
R3D_HD uint32_t spray_touch(const KArgs& a, uint64_t hid) {
  Rng rng;
  rng_init(rng, hid);
  double u_type, u_dir;
  rng_draw_pair(rng, rng_key(a.seed), u_type, u_dir);
  const double r3 = a.src_whole[2] * u_type;
  const int rt3 = (r3 <= a.src_whole[0]) ? 0 : (r3 <= a.src_whole[1]) ? 1 : 2;
  const GuideCell* guide = rt3 == 0 ? a.src_guide[0] : rt3 == 1 ? a.src_guide[1] : a.src_guide[2];
  if (LEVEL == 0) return guide_touch(guide, a.guide_bits, u_dir);
  const double* cdf = rt3 == 0 ? a.src_cdf[0] : rt3 == 1 ? a.src_cdf[1] : a.src_cdf[2];
  const double total = rt3 == 0 ? a.src_total[0] : rt3 == 1 ? a.src_total[1] : a.src_total[2];
  const uint64_t k = sample_cdf_guided(cdf, guide, a.guide_bits, total, u_dir);
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(1))) const uint32_t gword;
  return *(gword*)(a.toa_dir + 4 * k);
#else
  return (uint32_t)k;
#endif
}

// ---- seismometers ----------------------------------------------------------
// reference DataReporter::ReportPhononCollected (dataout.cpp:545-568) +
// Seismometer::CatchPhonon (dataout.cpp:103-216).  The reference tests every
// seismometer on every collection event (mPassthrough is hard-wired true);
// the hash grid narrows that to the seismometers whose gather sphere can
// contain the arrival point, which gives the identical set of catches.
template <int KIND>
R3D_HD void collect(const KArgs& a, const Tables<KIND>& T, const Phonon& p, double vel,
                    LaneStats& st) {
  st.collect++;
  const SeisGrid& g = a.grid;
  double fx = (p.loc.x - g.origin[0]) * g.inv_h;
  double fy = (p.loc.y - g.origin[1]) * g.inv_h;
  double fz = (p.loc.z - g.origin[2]) * g.inv_h;
  if (!(fx >= 0 && fy >= 0 && fz >= 0 && fx < g.dim_f[0] && fy < g.dim_f[1] && fz < g.dim_f[2])) return;
  int cellid = ((int)fz * g.dim[1] + (int)fy) * g.dim[0] + (int)fx;
  uint32_t k0 = g.start[cellid], k1 = g.start[cellid + 1];
  if (k0 == k1) return;
  const int t = p.type;
  V3 dopm = v3(0, 0, 0);
  double amp2 = 0.0;
  bool have_dopm = false;
  for (uint32_t k = k0; k < k1; k++) {
    const uint32_t s = g.items[k];
    const SeisScan& S = T.seis_scan[s];
    V3 to = v3(S.loc) - p.loc;
    double dist = mag(to);
    if (dist > S.r_out[t] || dist < S.r_in[t]) continue;
    double arv = p.t;
    if (S.r_in[t] <= 0) arv += dot(to, p.dir) / vel;   // plane-wave arrival-time correction
    double scaled = arv / a.time_per_bin;
    if (!(scaled >= 0.0)) continue;
    double fl = floor(scaled);
    if (!(fl < a.n_bins_f)) continue;
    uint32_t bin = (uint32_t)fl;
    if (!have_dopm) dopm = direction_of_motion(p), amp2 = amplitude2(p), have_dopm = true;
    const SeisHit& H = T.seis_hit[s];
    double xf = dot(dopm, v3(H.axes[0])), yf = dot(dopm, v3(H.axes[1])), zf = dot(dopm, v3(H.axes[2]));
    double energy = amp2 * H.inv_norm[t];
    size_t slot = (size_t)s * a.n_bins + bin;
    double* e = a.energy + slot * 5;
    R3D_ADD_F64(e + 0, energy * (xf * xf));
    R3D_ADD_F64(e + 1, energy * (yf * yf));
    R3D_ADD_F64(e + 2, energy * (zf * zf));
    R3D_ADD_F64(e + 3 + t, energy);
    R3D_ADD_U64(a.counts + slot * 2 + t, 1);
    st.n_catch++;
  }
}

// ---- volumetric scatter-event grid (include/r3d.h r3d_volume_desc) ------------
// One count per SCT / REF event, binned by resulting wave type, frame
// floor(t / dt) and model-space cell: the histogram the reference's video
// scripts build from its per-event text stream.
R3D_HD void volume_count(const KArgs& a, const Phonon& p, LaneStats& st) {
  if (!a.vol) return;
  const double f = p.t * a.vol_inv_dt;
  const double x = (p.loc.x - a.vol_origin[0]) * a.vol_inv_cell[0];
  const double y = (p.loc.y - a.vol_origin[1]) * a.vol_inv_cell[1];
  const double z = (p.loc.z - a.vol_origin[2]) * a.vol_inv_cell[2];
  if (!(f >= 0 && x >= 0 && y >= 0 && z >= 0 && f < a.vol_frames_f && x < a.vol_dim_f[0] &&
        y < a.vol_dim_f[1] && z < a.vol_dim_f[2])) {
    st.vol_out++;
    return;
  }
  const size_t idx = ((((size_t)p.type * a.vol_frames + (size_t)f) * a.vol_dim[2] + (size_t)z) *
                          a.vol_dim[1] + (size_t)y) * a.vol_dim[0] + (size_t)x;
  if (!st.quiet) R3D_ADD_U32(a.vol + idx, 1u);
}

// ---- one loop iteration, in two halves --------------------------------------
// What the first half leaves for the second.
struct Pending {
  double vel;        // phonon's velocity at the arrival point (for the seismometers)
  int32_t face;      // face reached, or -1 if the phonon scattered inside the cell
  uint32_t flags;    // that face's flags
  int32_t nbr;       // the cell behind that face (-1: none), read while the cell's record is at hand
};

// First half (phonons.cpp:549-623): termination checks, boundary search,
// free-path draw, advance + Move.  Returns FATE_ALIVE to continue with
// step_event(), else the fate; *reason gets the invalid-reason slot
// (include/r3d.h R3D_INV_*) when FATE_INVALID.
// QUAD (device, tetra, the kernels with a tail): when `quad` holds -- wave-uniform -- the four lanes of a quad carry the
// same history and share its boundary search, lane `fl` of the quad taking face `fl` (r3d_physics.h tet_fast_exit_quad).
template <int KIND, bool QUAD = false>
R3D_HD int step_move(const KArgs& a, const Tables<KIND>& T, Phonon& p, Rng& rng, LaneStats& st,
                     int* reason, Pending& ev, bool quad = false, unsigned fl = 0u) {
  using Cell = typename CellOf<KIND>::type;
  // phonons.cpp:549-552
  if (p.t > a.ttl) return FATE_TIMEOUT;
  // phonons.cpp:554-584: sanity checks every 128th move
  if ((p.moves & 127u) == 127u) {
    if (isnan(p.path)) return *reason = 0, FATE_INVALID;
    if (isnan(p.t)) return *reason = 1, FATE_INVALID;
    if (p.path < 0) return *reason = 2, FATE_INVALID;
    if (p.t < 0 || p.recent < 0) return *reason = 3, FATE_INVALID;
    if (p.recent == 0) return *reason = 4, FATE_INVALID;
    if (p.recent < a.slow_concern) return *reason = 5, FATE_INVALID;
    if (p.moves > a.loop_concern) return *reason = 6, FATE_INVALID;
    p.recent = 0;
  }
  st.iterations++;
  // The WHOLE cell record is read into registers here, ahead of the move's random number: the fence
  // below keeps later code from moving up, so loads written at their uses -- inside the boundary
  // search, and the links and the attenuation constant at the very end of the move -- would be
  // issued after the draw and each cost the wave a round trip of its own: a microsecond under load
  // for a tetra record (L2), a few hundred cycles for a layered or spherical one (LDS, behind the
  // other waves' slot traffic), eight times a move.
  using CellHere = const Cell;
  CellHere c = cell_rec<KIND>(T, p.cell, p.type);
#if defined(__HIP_DEVICE_COMPILE__)
  V3 quad_n = v3(0, 0, 0);
  double quad_d = 0;
  if constexpr (QUAD && KIND == CELL_TET) {
    if (quad) quad_n = v3(cell_rec<KIND>(T, p.cell, p.type).n[fl]), quad_d = cell_rec<KIND>(T, p.cell, p.type).d[fl];
  }
#endif
  R3D_SCHED_FENCE();   // (the loads above, THEN the draw: left to itself the scheduler puts the draw first)
  // The move's one uniform (for the free path, below) is drawn here: it depends on nothing, and
  // its hundred integer instructions fill the wait for the cell record, which everything else
  // in the move needs.
  double u_free = rng_draw(rng, rng_key(a.seed));
#if defined(__HIP_DEVICE_COMPILE__)
  // (pinned: without this the optimiser sinks the whole generator down to the free-path test, its
  //  first use, and the wave sits out the record's round trip with nothing to do)
  asm volatile("" : "+v"(u_free));
#endif
  R3D_SCHED_FENCE();
  // (the mean free path hangs on the record's scatterer index: for records in LDS it is asked for as
  //  soon as that is in, not where the free path is formed, a boundary search later; the tetra
  //  kernel has no register to spare for it across the search: +2 % with two spilled)
  double mfp = 0;
  if constexpr (KIND != CELL_TET) {
    mfp = T.scat_head[cell_scat(c)].mfp[p.type];
    R3D_SCHED_FENCE();
  }

  // --- where does the ray leave the cell? (phonons.cpp:590)
  Exit e;
  bool scatters;
  if constexpr (KIND == CELL_TET) {
    // The move in local form (r3d_physics.h tet_fast_exit) wherever it certifies its answer; a lane that
    // does not -- a start outside the cell, a tie between faces, a tangent plane, an exit at the phonon's
    // feet: one move in 1e4 in a tetrahedral grid -- takes the reference's own construction, as before.
    // (the certified lanes' whole move first, then the others': nothing of the local form is alive across the
    //  reference's construction, which needs every register the kernel has)
    bool slow;
    e.len = 0.0, scatters = false;
    {
      TetLocal L;
      TetFast F;
#if defined(__HIP_DEVICE_COMPILE__)
      if (QUAD && quad) F = tet_fast_exit_quad(c, quad_n, quad_d, p, L);
      else
#endif
        F = tet_fast_exit(c, p, L);
      slow = !F.ok;
      e.face = F.face;
      if (!slow) {
        e.len = L.R * two_atan(F.t, F.sn, F.cs);
        // free path (scatterers.cpp:297-307, phonons.cpp:601), screened as below
        mfp = T.scat_head[cell_scat(c)].mfp[p.type];
        double scatlen = pos_inf();
        if (!((1.0 - u_free) * mfp >= e.len)) scatlen = -log_lean(u_free) * mfp;
        scatters = scatlen < e.len;
        double len = e.len, sn = F.sn, cs = F.cs, omc = F.omc;
        if (any_lanes(scatters)) {
          if (scatters) {                       // scatter leg: the arc angle is len / R
            len = scatlen;
            rotation(len * frcp(L.R), &sn, &cs);
            omc = (sn * sn) * frcp(1.0 + cs);   // 1 - cos without the cancellation (|th| < pi: a leg inside one cell)
          }
        }
        tet_advance_local(c, L, p, len, sn, cs, omc);
      }
    }
    if (any_lanes(slow)) {
      if (slow) {
        R3D_COUNT_SLOW_MOVE();
        const TetSlowOut o = tet_move_reference_inline(&cell_rec<KIND>(T, p.cell, p.type), p, u_free,
                                                       T.scat_head[cell_scat(c)].mfp[p.type]);
        if (o.fate != FATE_ALIVE) return o.fate;
        p = o.p, e.face = o.face, scatters = o.scatters != 0;
      }
    }
  } else if constexpr (KIND == CELL_SPH) {
    // The shell move in local form (r3d_physics.h sph_fast_exit), certified as the tetra's; the other lanes -- straight
    // and vertical rays, starts outside the shell, tangent arcs, exits at the phonon's feet -- take the reference's.
    bool slow;
    e.len = 0.0, scatters = false;
    const V3 ec = v3(a.earth_center[0], a.earth_center[1], a.earth_center[2]);
    {
      TetLocal L;
      const SphFast F = sph_fast_exit(c, ec, p, L);
      slow = !F.ok;
      e.face = F.face;
      if (!slow) {
        e.len = L.R * two_atan(F.t, F.sn, F.cs);
        double scatlen = pos_inf();
        if (!((1.0 - u_free) * mfp >= e.len)) scatlen = -log_lean(u_free) * mfp;
        scatters = scatlen < e.len;
        double len = e.len, sn = F.sn, cs = F.cs, omc = F.omc, t = F.t;
        if (any_lanes(scatters)) {
          if (scatters) {                       // scatter leg: the arc angle is len / R (up to 178 degrees: inside the exit's)
            len = scatlen;
            rotation(len * frcp(L.R), &sn, &cs);
            // 1 - cos th and tan(th / 2) without cancellation, either side of 90 degrees
            const bool near = cs > 0.0;
            const double ih = frcp(near ? 1.0 + cs : sn);
            omc = near ? (sn * sn) * ih : 1.0 - cs;
            t = near ? sn * ih : omc * ih;
          }
        }
        sph_advance_local(c, L, F, p, len, sn, cs, omc, t);
      }
    }
    if (any_lanes(slow)) {
      if (slow) {
        R3D_COUNT_SLOW_MOVE();
        const TetSlowOut o = sph_move_reference_inline(&cell_rec<KIND>(T, p.cell, p.type), ec, p, u_free, mfp);
        if (o.fate != FATE_ALIVE) return o.fate;
        p = o.p, e.face = o.face, scatters = o.scatters != 0;
      }
    }
  } else {   // layered models: straight rays in a cylinder (media.cpp:236-330)
    e = cyl_exit(c, a.cyl_radius2, p);
    if (e.len == pos_inf()) return FATE_TIMEOUT;  // phonons.cpp:595-598

    // --- free path to the next scattering event, drawn afresh every iteration
    //     (scatterers.cpp:297-307, phonons.cpp:601)
    // scatlen = -ln(u) mfp.  Since -ln(u) >= 1 - u, (1-u) mfp >= len already rules a scatter
    // out, and the logarithm is only taken for the lanes that pass this screen.
    double scatlen = pos_inf();
    if (!((1.0 - u_free) * mfp >= e.len)) scatlen = -log_lean(u_free) * mfp;
    scatters = scatlen < e.len;
    // --- advance (both branches) and Move (phonons.cpp:608-609, :623)
    cyl_advance(c, p, scatters ? scatlen : e.len);
  }

  ev.face = scatters ? -1 : e.face;
  if constexpr (KIND == CELL_TET) {   // (selects, not an index into the four words: that would be a table in scratch)
    const uint32_t lk = e.face == 0 ? c.link[0] : e.face == 1 ? c.link[1] : e.face == 2 ? c.link[2] : c.link[3];
    ev.flags = scatters ? 0u : tet_link_flags(lk);
    ev.nbr = scatters ? -1 : tet_link_neighbor(lk);
  } else {
    ev.flags = scatters ? 0u : cell_face_flags(c, e.face);
    ev.nbr = scatters ? -1 : cell_neighbor(c, e.face);
  }
  ev.vel = cell_velocity(c, p.loc, p.type);
  return FATE_ALIVE;
}

// Elastic properties either side of the face the phonon sits on (CellFace::GetRTBasis,
// media_cellface.cpp:122-149).
template <int KIND>
R3D_HD Iface rt_interface(const KArgs& a, const Tables<KIND>& T, const Phonon& p, const Pending& ev, int nbr) {
  using Cell = typename CellOf<KIND>::type;
  const Cell& c = cell_rec<KIND>(T, p.cell, p.type);
  const bool adjoin = (ev.flags & F_ADJOIN) != 0;
  Iface f;
  f.normal = cell_face_normal(c, ev.face, p.loc);
  f.vR[0] = velocity_in<KIND>(T, p.cell, p.loc, 0), f.vR[1] = velocity_in<KIND>(T, p.cell, p.loc, 1);
  f.rhoR = cell_density(a, c, p.cell, p.loc);
  f.has_neighbor = adjoin;
  f.vT[0] = f.vT[1] = f.rhoT = 0;
  if (adjoin) {
    f.vT[0] = velocity_in<KIND>(T, nbr, p.loc, 0), f.vT[1] = velocity_in<KIND>(T, nbr, p.loc, 1);
    f.rhoT = cell_density(a, cell_rec<KIND>(T, nbr, p.type), nbr, p.loc);
  }
  return f;
}

// Second half (phonons.cpp:611-618, :640-676): scatter, or act on the face
// reached.  Seismometer collection (phonons.cpp:629-631) happens BETWEEN the
// halves, with the incident state: per lane in step(), wave-cooperatively in
// the kernel.
// PART selects what is compiled in: EV_ALL everything; EV_LIGHT everything but the
// reflection/transmission solve (the caller guarantees the event is not one); EV_RT only
// that solve (the caller guarantees it is one); EV_SCATTER only the scattering branch (the
// caller guarantees ev.face < 0); EV_BEND only the face branch without the solve (Snell bend
// or hand-over; the caller guarantees a face with a neighbour and no solve).  The kernel calls
// the parts from separate, wave-wide phases, which keeps each call site's code small.
enum { EV_ALL = 0, EV_LIGHT = 1, EV_RT = 2, EV_SCATTER = 3, EV_BEND = 4 };
template <int KIND, int PART = EV_ALL>
R3D_HD int step_event(const KArgs& a, const Tables<KIND>& T, Phonon& p, Rng& rng, LaneStats& st,
                      const Pending& ev, int nbr_known = -2) {
  using Cell = typename CellOf<KIND>::type;
  const Cell& c = cell_rec<KIND>(T, p.cell, p.type);
  if (PART == EV_SCATTER || (PART != EV_RT && PART != EV_BEND && ev.face < 0)) {
    // Scatterer::GetRandomScatteredRelativePhonon, scatterers.cpp:318-363
    st.scatter++;
    if (a.no_deflect) {
      scatter_transform(p, a.nodeflect_dir, 1.0, 0.0, p.type);
    } else {
      const int scat = cell_scat(c);
      const ScatHead& sh = T.scat_head[scat];
      const ScatPtrs* sp = T.scat_ptrs + scat;   // (indexed in place: a local copy would go to scratch)
      double u_conv, u_dir;   // the event's two uniforms: one block of the generator
      rng_draw_pair(rng, rng_key(a.seed), u_conv, u_dir);
      int conv = sample_small(sh.whole[p.type], 4, u_conv);  // GPP GPS GSP GSS
#ifdef R3D_ABLATE_SCATTER   // timing-only developer build: no table search
      uint64_t k = (uint64_t)(u_dir * (double)(a.n_toa - 1));
#else
      uint64_t k = sample_cdf_guided(sp->cdf[conv], sp->guide[conv], a.guide_bits, sh.total[conv], u_dir);
#endif
      double rc = 1.0, rs = 0.0;      // relative polarisation 0 except S->S (scatterers.cpp:341-356)
      if (conv == 3) {   // (the angle's cosine and sine as the table holds them: r3d_tables.h ScatPtrs)
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(1))) const double gdouble;   // (HBM: a global, not a FLAT, load)
        const gdouble* cs = (gdouble*)sp->spol_cs + 2 * k;
#else
        const double* cs = sp->spol_cs + 2 * k;
#endif
        rc = cs[0], rs = cs[1];
      }
      scatter_transform(p, a.toa_dir + 4 * k, rc, rs, (conv & 1) ? RAY_S : RAY_P);
    }
    volume_count(a, p, st);   // SCT
    return FATE_ALIVE;
  }
  if (PART == EV_SCATTER) return FATE_ALIVE;   // (not reached)
  const uint32_t fl = ev.flags;
  if (!(fl & (F_REFLECT | F_ADJOIN))) return FATE_LOST;  // phonons.cpp:675
  // (a caller that already knows the neighbour passes it, so that both cells' records can be
  //  fetched at once instead of the neighbour's waiting for this cell's)
  const int nbr = (nbr_known != -2) ? nbr_known : cell_neighbor(c, ev.face);
  const bool adjoin = (fl & F_ADJOIN) != 0;
  (void)adjoin;   // (only the timing-only R3D_ABLATE_RT build reads it here)
  bool crossed;
  if (PART != EV_LIGHT && PART != EV_BEND && (PART == EV_RT || (fl & F_REFLECT) || (fl & F_DISCON))) {
    st.rtsolve++;
#ifdef R3D_ABLATE_RT   // timing-only developer build: specular bounce / coin-flip transmission
    {
      const Iface f = rt_interface<KIND>(a, T, p, ev, nbr);
      double dn = dot(f.normal, p.dir);
      crossed = adjoin && (rng_draw(rng, rng_key(a.seed)) < 0.5);
      if (!crossed) p.dir = p.dir - (2.0 * dn) * f.normal;
    }
#else
    // (the event's uniforms first: they wait for nothing, the interface's records have to be fetched)
    double u_pol, u_out;
    rt_draws(p, rng, rng_key(a.seed), u_pol, u_out);
    crossed = rt_event<KIND == CELL_CYL>(p, rt_interface<KIND>(a, T, p, ev, nbr), u_pol, u_out);
#endif
  } else if (PART == EV_RT) {
    crossed = true;   // (not reached)
  } else if (fl & F_SMOOTH) {
    crossed = true;   // velocity step below 1e-5 everywhere on this face: plain hand-over
  } else {
    // Phonon::Refract without a grid discontinuity (phonons.cpp:243-252):
    // bend on a fractional velocity step > 1e-5, else plain hand-over.
    // (tetra cells hold one record per ray type: the phonon's own type first -- the bend needs nothing else)
    const int ty = p.type;
    const double vi = cell_velocity(c, p.loc, ty), vo = velocity_in<KIND>(T, nbr, p.loc, ty);
    bool step = (fl & F_STEP) != 0;
    if (!step) {      // straddling face: evaluate the reference's test here, on both ray types
      const double ui = velocity_in<KIND>(T, p.cell, p.loc, 1 - ty), uo = velocity_in<KIND>(T, nbr, p.loc, 1 - ty);
      const double v1 = ty == RAY_P ? vi : ui, v2 = ty == RAY_P ? vo : uo;
      const double w1 = ty == RAY_P ? ui : vi, w2 = ty == RAY_P ? uo : vo;
      double dvp = fabs(2 * (v2 - v1) / (v2 + v1));
      double dvs = fabs(2 * (w2 - w1) / (w2 + w1));
      step = (dvp > dvs ? dvp : dvs) > 0.00001;
    }
    if (step) {
      crossed = bend<KIND == CELL_CYL>(p, cell_face_normal(c, ev.face, p.loc), vi, vo);
    } else {
      crossed = true;
    }
  }
  if (crossed) {
    p.cell = nbr, st.transfer++;
  } else {
    st.reflect++;
    volume_count(a, p, st);   // REF
  }
  return FATE_ALIVE;
}

// The reflection / transmission event in two calls (see rt_choose / rt_apply in r3d_physics.h):
// a caller short of registers -- the pool kernel at three waves per SIMD -- draws the outcome, lets
// go of everything but the four words of the choice, reloads the phonon and applies it.  Together
// they are step_event<KIND, EV_RT>.
template <int KIND>
R3D_HD RtChoice rt_event_choose(const KArgs& a, const Tables<KIND>& T, const Phonon& p, Rng& rng, LaneStats& st,
                                const Pending& ev, int nbr) {
  st.rtsolve++;
  // (the event's uniforms first: they wait for nothing, the interface's records have to be fetched)
  double u_pol, u_out;
  rt_draws(p, rng, rng_key(a.seed), u_pol, u_out);
  return rt_choose<KIND == CELL_CYL>(p, rt_interface<KIND>(a, T, p, ev, nbr), u_pol, u_out);
}
template <int KIND>
R3D_HD void rt_event_apply(const KArgs& a, const Tables<KIND>& T, Phonon& p, LaneStats& st, const Pending& ev,
                           int nbr, RtChoice ch) {
  // what is read again of the tables: the face normal
  const bool crossed = rt_apply<KIND == CELL_CYL>(p, cell_face_normal(cell_rec<KIND>(T, p.cell, p.type), ev.face, p.loc), ch);
  if (crossed) {
    p.cell = nbr, st.transfer++;
  } else {
    st.reflect++;
    volume_count(a, p, st);   // REF
  }
}

// The whole iteration for one work-item on its own (host emulation).
template <int KIND>
R3D_HD int step(const KArgs& a, const Tables<KIND>& T, Phonon& p, Rng& rng, LaneStats& st,
                int* reason) {
  Pending ev;
  int fate = step_move<KIND>(a, T, p, rng, st, reason, ev);
  if (fate != FATE_ALIVE) return fate;
  if (ev.flags & F_COLLECT) collect<KIND>(a, T, p, ev.vel, st);
  return step_event<KIND>(a, T, p, rng, st, ev, ev.face < 0 ? -2 : ev.nbr);
}

}  // namespace r3d
#endif
