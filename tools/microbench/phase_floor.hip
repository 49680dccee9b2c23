// phase_floor.hip -- the vector instructions of each phase's COMMON path, counted statically
// (tools/phase_floor.py compiles this for gfx950 with -S and counts v_* per kernel): what a history's
// events cost when nothing but the physics is executed -- no queues, no slot traffic, no tallies, no
// rare branch -- with the short tier of every wave-voted series (-DR3D_FLOOR_TIERS makes every vote
// pass).  bench.py multiplies these by the run's event counts: roofline.floor_lane_insts_per_history.
// Never linked into the engine; never run.
#define R3D_DEV_BUILD 1
#include "../../radiative3d_amd/csrc/r3d_step.h"
using namespace r3d;

// ---- one move of each cell kind: the draw, the boundary search, the free-path screen, the advance
extern "C" __global__ void floor_move_tet(const CellTet* cells, Phonon* ps, double mfp, uint64_t seed) {
  const int i = threadIdx.x;
  Phonon p = ps[i];
  Rng rng;
  rng_init(rng, (uint64_t)i);
  const CellTet c = cells[2 * p.cell + p.type];
  const double u = rng_draw(rng, rng_key(seed));
  TetLocal L;
  const TetFast F = tet_fast_exit(c, p, L);
  const double len = L.R * two_atan(F.t, F.sn, F.cs);
  if ((1.0 - u) * mfp >= len && F.ok) {   // (no scattering, certified: the common case)
    tet_advance_local(c, L, p, len, F.sn, F.cs, F.omc);
    p.t += cell_velocity(c, p.loc, p.type);   // (the velocity at the arrival point, for the receivers)
    p.cell = tet_link_neighbor(F.face == 0 ? c.link[0] : F.face == 1 ? c.link[1] : F.face == 2 ? c.link[2] : c.link[3]);
  }
  ps[i] = p;
}
extern "C" __global__ void floor_move_cyl(const CellCyl* cells, Phonon* ps, double mfp, double rad2, uint64_t seed) {
  const int i = threadIdx.x;
  Phonon p = ps[i];
  Rng rng;
  rng_init(rng, (uint64_t)i);
  const CellCyl c = cells[2 * p.cell + p.type];
  const double u = rng_draw(rng, rng_key(seed));
  const Exit e = cyl_exit(c, rad2, p);
  if ((1.0 - u) * mfp >= e.len) {
    cyl_advance(c, p, e.len);
    p.cell = e.face == 0 ? c.nbr[0] : c.nbr[1];
  }
  ps[i] = p;
}
extern "C" __global__ void floor_move_sph(const CellSph* cells, Phonon* ps, double mfp, uint64_t seed) {
  const int i = threadIdx.x;
  Phonon p = ps[i];
  Rng rng;
  rng_init(rng, (uint64_t)i);
  const CellSph c = cells[2 * p.cell + p.type];
  const double u = rng_draw(rng, rng_key(seed));
  TetLocal L;
  const SphFast F = sph_fast_exit(c, v3(0, 0, 0), p, L);
  const double len = L.R * two_atan(F.t, F.sn, F.cs);
  if ((1.0 - u) * mfp >= len && F.ok) {
    sph_advance_local(c, L, F, p, len, F.sn, F.cs, F.omc, F.t);
    p.cell = F.face == 0 ? c.nbr[0] : c.nbr[1];
  }
  ps[i] = p;
}
// ---- a reflection / transmission: the event's draws, the interface (four velocities, two densities of linear
//      cells), the outcome weights and the choice, the outgoing ray and its polarisation
extern "C" __global__ void floor_rt(const CellTet* cells, const RhoLin* rho, Phonon* ps, uint64_t seed) {
  const int i = threadIdx.x;
  Phonon p = ps[i];
  Rng rng;
  rng_init(rng, (uint64_t)i);
  double u_pol, u_out;
  rt_draws(p, rng, rng_key(seed), u_pol, u_out);
  Iface f;
  const int nbr = p.cell + 1;
  f.normal = v3(cells[2 * p.cell].n[1]);
  f.vR[0] = cell_velocity(cells[2 * p.cell], p.loc, 0), f.vR[1] = cell_velocity(cells[2 * p.cell + 1], p.loc, 1);
  f.vT[0] = cell_velocity(cells[2 * nbr], p.loc, 0), f.vT[1] = cell_velocity(cells[2 * nbr + 1], p.loc, 1);
  f.rhoR = dot(p.loc, v3(rho[p.cell].g)) + rho[p.cell].c, f.rhoT = dot(p.loc, v3(rho[nbr].g)) + rho[nbr].c;
  f.has_neighbor = true;
  const bool crossed = rt_event(p, f, u_pol, u_out);
  p.cell = crossed ? nbr : p.cell;
  ps[i] = p;
}
// ---- a scattering: the event's draws, the conversion, the deflection from the guided table, the rotation
extern "C" __global__ void floor_scatter(const ScatHead* sh, const ScatPtrs* sp, const double* toa_dir, Phonon* ps,
                                         uint32_t bits, uint64_t seed) {
  const int i = threadIdx.x;
  Phonon p = ps[i];
  Rng rng;
  rng_init(rng, (uint64_t)i);
  double u_conv, u_dir;
  rng_draw_pair(rng, rng_key(seed), u_conv, u_dir);
  const int conv = sample_small(sh->whole[p.type], 4, u_conv);
  // (the guide cell decides 97 % of the draws: its eight words, seven compares)
  const double r = sh->total[conv] * u_dir;
  uint32_t j = (uint32_t)(u_dir * (double)(1u << bits));
  const GuideCell g = sp->guide[conv][j];
  uint32_t below = 0;
#pragma unroll
  for (int k = 0; k < kGuideVals; k++) below += !(r <= g.c[k]) ? 1u : 0u;
  const uint64_t k = g.k1 + below;
  double rc = 1.0, rs = 0.0;
  if (conv == 3) rc = sp->spol_cs[2 * k], rs = sp->spol_cs[2 * k + 1];
  scatter_transform(p, toa_dir + 4 * k, rc, rs, (conv & 1) ? RAY_S : RAY_P);
  ps[i] = p;
}
// ---- a fresh history: two draws, the wave type, the take-off direction from the guided table
extern "C" __global__ void floor_spray(const double* whole, const GuideCell* guide, const double* toa_dir, double total,
                                       Phonon* ps, uint32_t bits, uint64_t seed) {
  const int i = threadIdx.x;
  Phonon p;
  Rng rng;
  rng_init(rng, (uint64_t)i);
  double u_type, u_dir;
  rng_draw_pair(rng, rng_key(seed), u_type, u_dir);
  const double r3 = whole[2] * u_type;
  const int rt3 = (r3 <= whole[0]) ? 0 : (r3 <= whole[1]) ? 1 : 2;
  const double r = total * u_dir;
  uint32_t j = (uint32_t)(u_dir * (double)(1u << bits));
  const GuideCell g = guide[j];
  uint32_t below = 0;
#pragma unroll
  for (int k = 0; k < kGuideVals; k++) below += !(r <= g.c[k]) ? 1u : 0u;
  const double* d = toa_dir + 4 * (g.k1 + below);
  p.dir = v3(d[3] * d[1], d[3] * d[2], d[0]);
  p.t = p.path = p.recent = p.lamp = 0.0, p.moves = 0, p.cell = 0;
  p.pc = (rt3 == 1) ? 6.123233995736766e-17 : 1.0, p.ps = (rt3 == 1) ? 1.0 : 0.0, p.type = rt3 == 0 ? RAY_P : RAY_S;
  p.loc = v3(0, 0, -10);
  ps[i] = p;
}
// ---- an arrival at a collection face: the receiver hash's cell; ONE candidate receiver tested; ONE catch binned
extern "C" __global__ void floor_collect_arrival(const SeisGrid* gp, const Phonon* ps, uint32_t* out) {
  const int i = threadIdx.x;
  const Phonon p = ps[i];
  const SeisGrid& g = *gp;
  const double fx = (p.loc.x - g.origin[0]) * g.inv_h, fy = (p.loc.y - g.origin[1]) * g.inv_h, fz = (p.loc.z - g.origin[2]) * g.inv_h;
  uint32_t k0 = 0, k1 = 0;
  if (fx >= 0 && fy >= 0 && fz >= 0 && fx < g.dim_f[0] && fy < g.dim_f[1] && fz < g.dim_f[2]) {
    const int cellid = ((int)fz * g.dim[1] + (int)fy) * g.dim[0] + (int)fx;
    k0 = g.start[cellid], k1 = g.start[cellid + 1];
  }
  out[2 * i] = k0, out[2 * i + 1] = k1;
}
extern "C" __global__ void floor_collect_candidate(const SeisScan* scan, const Phonon* ps, double inv_vel, double tpb, double* out) {
  const int i = threadIdx.x;
  const Phonon p = ps[i];
  const SeisScan& S = scan[p.cell];
  const V3 to = v3(S.loc) - p.loc;
  const double dist = mag(to);
  double fl = -1.0;
  if (!(dist > S.r_out[p.type] || dist < S.r_in[p.type])) {
    double arv = p.t;
    if (S.r_in[p.type] <= 0) arv += dot(to, p.dir) * inv_vel;
    fl = floor(arv / tpb);
  }
  out[i] = fl;
}
extern "C" __global__ void floor_collect_catch(const SeisHit* hit, const Phonon* ps, double* e) {
  const int i = threadIdx.x;
  const Phonon p = ps[i];
  const SeisHit& H = hit[p.cell];
  const V3 dm = direction_of_motion(p);
  const double xf = dot(dm, v3(H.axes[0])), yf = dot(dm, v3(H.axes[1])), zf = dot(dm, v3(H.axes[2]));
  const double et = amplitude2(p) * H.inv_norm[p.type];
  e[4 * i] = et * (xf * xf), e[4 * i + 1] = et * (yf * yf), e[4 * i + 2] = et * (zf * zf), e[4 * i + 3] = et;
}
