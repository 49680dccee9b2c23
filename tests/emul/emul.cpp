// emul.cpp -- TEST-ONLY host build of the engine's per-work-item code.
//
// Compiles radiative3d_amd/csrc/r3d_step.h (the exact functions the HIP kernel
// runs per lane) with g++ and drives them one history at a time, so kernel
// logic can be single-stepped and compared with the oracle in a container
// that has no GPU.  It is NOT a product path: it is not part of the C-ABI,
// not shipped in radiative3d_amd/lib, and nothing outside tests/ loads it.
#include <cstring>

#include "../../include/r3d.h"
// tetra moves that did not certify the local form and took the reference's construction (r3d_step.h)
static unsigned long long g_slow_moves = 0;
#define R3D_COUNT_SLOW_MOVE() (++g_slow_moves)
static unsigned long long g_loc_reason[8] = {0};
#define R3D_LOC_REASON(k, mask) (g_loc_reason[k] += (mask) ? 1 : 0)
#include <cstdlib>
#include "../../radiative3d_amd/csrc/r3d_pack.h"
#include "../../radiative3d_amd/csrc/r3d_step.h"

using namespace r3d;

template <int KIND>
static void run_kind(const KArgs& a, uint64_t n, uint64_t first_id, uint64_t seed, r3d_result* out,
                     r3d_final* finals) {
  Tables<KIND> T;
  T.cells = reinterpret_cast<const typename CellOf<KIND>::type*>(a.cells);
  T.scat_head = a.scat_head;
  T.scat_ptrs = a.scat_ptrs;
  T.seis_scan = a.seis_scan;
  T.seis_hit = a.seis_hit;
  for (uint64_t i = 0; i < n; i++) {
    Phonon p;
    Rng rng;
    LaneStats st = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    rng_init(rng, first_id + i);
    spray(a, p, rng);
    out->events[R3D_EV_GENERATED]++;
    int fate, reason = 0;
    while ((fate = step<KIND>(a, T, p, rng, st, &reason)) == FATE_ALIVE) {
    }
    if (fate == FATE_LOST) out->n_lost++;
    else if (fate == FATE_TIMEOUT) out->n_timeout++;
    else out->n_invalid++, out->invalid_reasons[reason]++;
    out->events[R3D_EV_ITERATIONS] += st.iterations;
    out->events[R3D_EV_SCATTER] += st.scatter;
    out->events[R3D_EV_COLLECT] += st.collect;
    out->events[R3D_EV_CATCH] += st.n_catch;
    out->events[R3D_EV_REFLECT] += st.reflect;
    out->events[R3D_EV_TRANSFER] += st.transfer;
    out->events[R3D_EV_RTSOLVE] += st.rtsolve;
    out->events[R3D_EV_VOLUME_OUT] += st.vol_out;
    if (finals) {
      r3d_final& f = finals[i];
      std::memset(&f, 0, sizeof f);
      f.time = p.t, f.path = p.path, f.amp = amplitude(p);
      f.loc[0] = p.loc.x, f.loc[1] = p.loc.y, f.loc[2] = p.loc.z;
      f.dir[0] = p.dir.x, f.dir[1] = p.dir.y, f.dir[2] = p.dir.z;
      f.moves = p.moves;
      f.fate = (uint8_t)fate;
      f.type = (uint8_t)p.type;
      f.n_catch = (uint16_t)(st.n_catch > 65535u ? 65535u : st.n_catch);
    }
  }
}

extern "C" unsigned long long r3d_emul_slow_moves(int reset) {
  const unsigned long long v = g_slow_moves;
  if (reset) g_slow_moves = 0;
  return v;
}

extern "C" void r3d_emul_loc_reasons(unsigned long long* out, int reset) {
  for (int i = 0; i < 8; i++) {
    out[i] = g_loc_reason[i];
    if (reset) g_loc_reason[i] = 0;
  }
}

static r3d_volume_desc g_vol_desc;
static uint32_t* g_vol = nullptr;
extern "C" void r3d_emul_set_volume(const r3d_volume_desc* v, uint32_t* counters) {
  g_vol = v ? counters : nullptr;
  if (v) g_vol_desc = *v;
}

extern "C" int r3d_emul_run(const r3d_model_desc* m, uint64_t n, uint64_t first_id, uint64_t seed,
                            r3d_result* out, r3d_final* finals) {
  if (!m || !out) return 1;
  PackedModel pm;
  pack_model(*m, pm);
  KArgs a = pm.args;
  a.seed = seed;
  a.energy = out->energy;
  a.counts = reinterpret_cast<unsigned long long*>(out->counts);
  if (g_vol) {
    for (int k = 0; k < 3; k++) {
      a.vol_origin[k] = g_vol_desc.origin[k], a.vol_inv_cell[k] = 1.0 / g_vol_desc.cell_size[k];
      a.vol_dim[k] = g_vol_desc.dims[k], a.vol_dim_f[k] = (double)g_vol_desc.dims[k];
    }
    a.vol_frames = g_vol_desc.n_frames, a.vol_frames_f = (double)g_vol_desc.n_frames, a.vol_inv_dt = 1.0 / g_vol_desc.frame_dt, a.vol = g_vol;
  }
  switch (m->cell_kind) {
    case R3D_CELL_CYLINDER: run_kind<CELL_CYL>(a, n, first_id, seed, out, finals); break;
    case R3D_CELL_TETRA: run_kind<CELL_TET>(a, n, first_id, seed, out, finals); break;
    default: run_kind<CELL_SPH>(a, n, first_id, seed, out, finals);
  }
  return 0;
}

// Pack-time class (F_SMOOTH / F_STEP / 0) of face f of cell ci: r3d_pack.h classify_velocity_step.
extern "C" uint32_t r3d_emul_face_class(const r3d_model_desc* m, int ci, int f) {
  return classify_velocity_step(*m, ci, f);
}
// ... and of bare corner values steps[corner][type] of the signed fractional velocity step
extern "C" uint32_t r3d_emul_class_from_corners(const double* steps, int n) {
  return classify_from_corner_steps(reinterpret_cast<const double (*)[2]>(steps), n);
}

// The lean elementary functions of r3d_math.h, one value at a time (host build: a "wave" is one
// lane, so every tier of the wave-voted routines is reached by its own arguments).
// which: 0 exp_lean(x)  1 log_lean(x)  2 atanh_lean(x)  3 asin_small(x)  4 angle_from_sincos(x, y)
//        5 / 6 sine / cosine of rotation(x)
extern "C" double r3d_emul_math(int which, double x, double y) {
  double s = 0, c = 0;
  switch (which) {
    case 0: return exp_lean(x);
    case 1: return log_lean(x);
    case 2: return atanh_lean(x);
    case 3: return asin_small(x);
    case 4: return angle_from_sincos(x, y);
    case 5: rotation(x, &s, &c); return s;
    case 6: rotation(x, &s, &c); return c;
  }
  return 0.0 / 0.0;
}

// Inverse-CDF draws two ways (r3d_physics.h): out_guided[i] from the search guide's cells
// (sample_cdf_guided), out_plain[i] by bisecting the whole table (sample_cdf) -- the reference's
// ProbDist::GetRandomIndex.  bits = 0: the width the engine would choose for a table of n entries.
extern "C" void r3d_emul_sample_cdf(const double* cdf, uint64_t n, uint32_t bits, const double* u, uint64_t m,
                                    uint64_t* out_guided, uint64_t* out_plain, uint64_t* longest_bracket) {
  if (!bits) bits = guide_bits_for(n);
  std::vector<GuideCell> cells;
  build_guide_cells(cdf, n, bits, cells);
  uint64_t longest = 0;
  for (const GuideCell& c : cells) longest = std::max<uint64_t>(longest, c.k2 - c.k1);
  if (longest_bracket) *longest_bracket = longest;
  for (uint64_t i = 0; i < m; i++) {
    out_guided[i] = sample_cdf_guided(cdf, cells.data(), bits, cdf[n - 1], u[i]);
    out_plain[i] = sample_cdf(cdf, n, u[i]);
  }
}

// ---- the tetra move's two searches side by side (tests/test_face_filter.py) ------------------------------
// Random tetrahedra with a linear velocity, a start and a direction per case; the local form
// (tet_fast_exit) and the reference's construction (tet_arc / tet_exit / tet_exit_length) both run.  Where
// the local form CERTIFIES its answer the two must agree: same face, same arc length.  mode:
//   0 interior starts, any direction          1 starts on a face (to rounding, either side), moving in
//   2 starts on / near an edge or a vertex    3 directions aimed at an edge or a vertex (ties)
//   4 starts outside a face by 1e-17 .. 1e-7 R, moving in or out (retrograde micro-steps)
//   5 sliver cells (faces meeting at shallow angles)   6 arcs tangent to a face (strong gradients)
//   7 strongly graded cells 1e3 .. 1e5 from the origin, starts on a face moving in
// out[0] cases, out[1] certified, out[2] certified and face differs, out[3] certified, same face, arc
// lengths differ by more than tol * R, out[4] the reference gave no exit (len = inf) among the certified, out[5]
// (with an oracle) certified cases on which the engine's sine-space search and the oracle's differ;
// dev[0] largest |len_local - len_ref| / R among the certified, dev[1] the same / max(len, 1e-300).
namespace {
struct SplitMix {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double u() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
  double sym() { return 2.0 * u() - 1.0; }
  double logu(double lo, double hi) { return lo * std::pow(hi / lo, u()); }
};
static V3 rnd_unit(SplitMix& g) {
  for (;;) {
    V3 v = v3(g.sym(), g.sym(), g.sym());
    const double m = mag2(v);
    if (m > 1e-4 && m <= 1.0) return (1.0 / std::sqrt(m)) * v;
  }
}
}  // namespace
typedef int (*tet_search_fn)(const double normals[12], const double points[12], const double g[3], double v0,
                             const double loc[3], const double dir[3], double* len);
extern "C" void r3d_emul_face_filter(int mode, uint64_t n, uint64_t seed, double tol, uint64_t* out, double* dev,
                                     double* first_bad /* 16 doubles or null */,
                                     tet_search_fn oracle /* oracle/r3d_oracle.cpp r3d_oracle_tet_search, or null */) {
  SplitMix g{seed * 0x2545F4914F6CDD1Dull + (uint64_t)mode};
  for (int i = 0; i < 6; i++) out[i] = 0;
  dev[0] = dev[1] = 0.0;
  for (uint64_t it = 0; it < n; it++) {
    // the cell: four corners in a box of edge ~10; slivers: the fourth corner close to the others' plane
    V3 x[4];
    for (;;) {
      for (int k = 0; k < 4; k++) x[k] = v3(10 * g.u(), 10 * g.u(), 10 * g.u());
      if (mode == 7) {   // a cell far from the origin (strongly graded: the arc's radius is much less than |loc|)
        const V3 off = g.logu(1e3, 1e5) * rnd_unit(g);
        for (int k = 0; k < 4; k++) x[k] = x[k] + off;
      }
      if (mode == 5) {
        const V3 nn = unit(cross(x[1] - x[0], x[2] - x[0]));
        const V3 c3 = (1.0 / 3.0) * (x[0] + x[1] + x[2]);
        x[3] = c3 + g.logu(1e-4, 1e-1) * nn + 3.0 * (g.sym() * unit(x[1] - x[0]) + g.sym() * unit(x[2] - x[0]));
      }
      const double vol6 = std::fabs(dot(cross(x[1] - x[0], x[2] - x[0]), x[3] - x[0]));
      if (vol6 > (mode == 5 ? 1e-6 : 1.0)) break;
    }
    CellTet c;
    std::memset(&c, 0, sizeof c);
    double points[12];
    for (int f = 0; f < 4; f++) {   // face f: opposite corner f, normal pointing away from it
      const V3 a = x[(f + 1) & 3], b = x[(f + 2) & 3], d = x[(f + 3) & 3];
      V3 nn = unit(cross(b - a, d - a));
      if (dot(nn, x[f] - a) > 0) nn = -nn;
      c.n[f][0] = nn.x, c.n[f][1] = nn.y, c.n[f][2] = nn.z;
      c.d[f] = dot(nn, a);
      points[3 * f] = a.x, points[3 * f + 1] = a.y, points[3 * f + 2] = a.z;
    }
    const V3 centre = 0.25 * (x[0] + x[1] + x[2] + x[3]);
    // velocity: 5 at the centre, gradient of any direction; arc radius from ~the cell's size to 1e5 of it
    const V3 gd = rnd_unit(g);
    const double gm = (mode == 6 || mode == 7) ? 5.0 / g.logu(12.0, 60.0) : 5.0 / g.logu(20.0, 1e6);   // (velocity > 0 throughout the cell)
    c.g[0] = gm * gd.x, c.g[1] = gm * gd.y, c.g[2] = gm * gd.z;
    c.v0 = 5.0 - dot(v3(c.g), centre);
    c.inv_gmag = 1.0 / gm;
    c.att = -0.01;
    Phonon p;
    std::memset(&p, 0, sizeof p);
    p.pc = 1.0, p.type = 0;
    // start
    double w[4];
    double sum = 0;
    for (int k = 0; k < 4; k++) sum += (w[k] = -std::log(1.0 - g.u()));
    for (int k = 0; k < 4; k++) w[k] /= sum;
    const int f0 = (int)(g.next() & 3), f1 = (f0 + 1 + (int)(g.next() % 3)) & 3;
    if (mode == 1 || mode == 4 || mode == 7) w[f0] = 0;        // on face f0 (corner f0 has no weight)
    if (mode == 2) {
      w[f0] = 0, w[f1] = (g.u() < 0.5) ? 0.0 : g.logu(1e-16, 1e-6);   // on / near the edge shared by f0 and f1
      if (g.u() < 0.3) w[(f1 + 1) & 3 == f0 ? (f1 + 2) & 3 : (f1 + 1) & 3] = g.logu(1e-16, 1e-6);   // near a vertex
    }
    sum = w[0] + w[1] + w[2] + w[3];
    p.loc = v3(0, 0, 0);
    for (int k = 0; k < 4; k++) p.loc = p.loc + (w[k] / sum) * x[k];
    p.dir = rnd_unit(g);
    const V3 nf0 = v3(c.n[f0]);
    if (mode == 1 || mode == 2 || mode == 7) {   // moving in through f0 (any angle, grazing included)
      if (dot(nf0, p.dir) > 0) p.dir = p.dir - (2.0 * dot(nf0, p.dir)) * nf0;
      if (g.u() < 0.2) {             // grazing: mostly along the face
        V3 tang = unit(cross(nf0, rnd_unit(g)));
        p.dir = unit(tang + (-g.logu(1e-9, 1e-1)) * nf0);
      }
      if (mode == 1 && g.u() < 0.5) p.loc = p.loc + (g.sym() * g.logu(1e-17, 1e-12) * 10.0) * nf0;   // rounding, either side
    }
    if (mode == 3) {   // aimed at a point of an edge or at a vertex, from inside
      V3 target = x[f0];
      if (g.u() < 0.7) {
        const double s = g.u();
        target = s * x[f0] + (1 - s) * x[f1];
      }
      if (g.u() < 0.5) target = target + (g.logu(1e-14, 1e-4)) * rnd_unit(g);
      p.dir = unit(target - p.loc);   // (a straight aim: the arc misses by its sagitta, which varies with the gradient)
    }
    if (mode == 4) {   // outside f0 by a hair (or more), moving in or out
      const double R_guess = 5.0 / gm;
      p.loc = p.loc + (g.logu(1e-17, 1e-7) * R_guess) * nf0;
      if (g.u() < 0.5 && dot(nf0, p.dir) > 0) p.dir = p.dir - (2.0 * dot(nf0, p.dir)) * nf0;
    }
    if (mode == 6) {   // a direction nearly parallel to face f1, so that the arc bends to or away from it
      V3 tang = unit(cross(v3(c.n[f1]), rnd_unit(g)));
      p.dir = unit(tang + (g.sym() * g.logu(1e-6, 3e-1)) * v3(c.n[f1]));
    }
    out[0]++;
    TetLocal L;
    const TetFast F = tet_fast_exit(c, p, L);
    if (!F.ok) continue;
    out[1]++;
    const double len_loc = L.R * two_atan(F.t, F.sn, F.cs);
    const TetArc A = tet_arc(c, p);
    const TetExit E = tet_exit(c, A);
    double len_ref = tet_exit_length(A, E);
    int face_ref = E.face;
    bool bad = false;
    if (oracle) {   // the reference's construction as the ORACLE has it (angles, acos, atan2): it must say the same
      const double loc[3] = {p.loc.x, p.loc.y, p.loc.z}, dir[3] = {p.dir.x, p.dir.y, p.dir.z};
      double len_o = 0;
      const int face_o = oracle(&c.n[0][0], points, c.g, c.v0, loc, dir, &len_o);
      if (face_o != face_ref || !(std::fabs(len_o - len_ref) <= tol * L.R)) out[5]++;   // (engine's sine-space search vs oracle)
      face_ref = face_o, len_ref = len_o;
    }
    if (!(len_ref < pos_inf())) out[4]++, bad = true;
    else if (face_ref != F.face) out[2]++, bad = true;
    else {
      const double d = std::fabs(len_loc - len_ref);
      if (!(d <= tol * L.R)) out[3]++, bad = true;
      if (d / L.R > dev[0]) dev[0] = d / L.R;
      if (d / std::fmax(std::fabs(len_ref), 1e-300) > dev[1]) dev[1] = d / std::fmax(std::fabs(len_ref), 1e-300);
    }
    static int want_kind = getenv("R3D_FF_KIND") ? atoi(getenv("R3D_FF_KIND")) : 0;   // debugging: which kind of mismatch to record
    const bool rec = want_kind == 0 ? (out[2] + out[3] + out[4] == 1) : (want_kind == 4 ? (!(len_ref < pos_inf()) && out[4] == 1) : (want_kind == 2 ? (face_ref != F.face && out[2] == 1) : false));
    if (bad && first_bad && rec) {
      first_bad[0] = (double)it, first_bad[1] = F.face, first_bad[2] = face_ref, first_bad[3] = len_loc, first_bad[4] = len_ref;
      first_bad[5] = L.R, first_bad[6] = F.t;
      for (int f = 0; f < 4; f++) first_bad[7 + f] = c.d[f] - dot(v3(c.n[f]), p.loc);
      for (int f = 0; f < 4; f++) first_bad[11 + f] = dot(v3(c.n[f]), p.dir);
    }
  }
}

// ---- the shell move's two searches side by side (tests/test_face_filter.py) -------------------------------
// Random shells v = a r^2 + c (a < 0, velocity 3 .. 9 growing 0.1 .. 30 % from top to bottom, top radius 1000 .. 6371,
// thickness 5 .. 2000) about the origin; the local form (sph_fast_exit) and the reference's construction
// (sph_arc / sph_exit) both run; where the local form certifies they must agree.  mode:
//   0 interior starts, any direction        1 on the top face (to rounding, either side), moving in
//   2 on the bottom face, moving in (up)    3 near the bottom, nearly horizontal (arcs tangent to the bottom face)
//   4 outside either face by 1e-17 .. 1e-7 of the radius, moving in or out
//   5 nearly vertical rays                  6 thin shells with strong gradients (legs of many degrees)
// out[] / dev[] as r3d_emul_face_filter.
typedef int (*shell_search_fn)(double a, double c0, double r_top, double r_bottom, const double loc[3], const double dir[3], double* len);
extern "C" void r3d_emul_shell_filter(int mode, uint64_t n, uint64_t seed, double tol, uint64_t* out, double* dev,
                                      double* first_bad, shell_search_fn oracle) {
  SplitMix g{seed * 0x9E3779B97F4A7C15ull + 77u + (uint64_t)mode};
  for (int i = 0; i < 6; i++) out[i] = 0;
  dev[0] = dev[1] = 0.0;
  for (uint64_t it = 0; it < n; it++) {
    const double rt = 1000.0 + 5371.0 * g.u();
    const double thick = (mode == 6) ? g.logu(2.0, 60.0) : std::fmin(g.logu(5.0, 2000.0), 0.8 * rt);
    const double rb = rt - thick;
    const double vt = 3.0 + 6.0 * g.u();
    const double vb = vt * (1.0 + ((mode == 6) ? g.logu(0.05, 0.6) : g.logu(1e-3, 0.3)));
    CellSph c;
    std::memset(&c, 0, sizeof c);
    c.a = (vt - vb) / (rt * rt - rb * rb);
    c.c = vt - c.a * rt * rt;
    c.zero_rad2 = -c.c / c.a;
    c.att = -0.01;
    c.radius[0] = rt, c.radius[1] = -rb;
    Phonon p;
    std::memset(&p, 0, sizeof p);
    p.pc = 1.0;
    const V3 up = rnd_unit(g);
    double r = rb + thick * g.u();
    if (mode == 1) r = rt;
    if (mode == 2) r = rb;
    if (mode == 3) r = rb + thick * g.logu(1e-12, 1e-2);
    if (mode == 4) r = (g.u() < 0.5) ? rt * (1.0 + g.logu(1e-17, 1e-7)) : rb * (1.0 - g.logu(1e-17, 1e-7));
    p.loc = r * up;
    if ((mode == 1 || mode == 2) && g.u() < 0.5) p.loc = (1.0 + g.sym() * g.logu(1e-17, 1e-13)) * p.loc;
    p.dir = rnd_unit(g);
    const double vert = dot(p.dir, up);
    if (mode == 1 && vert > 0) p.dir = p.dir - (2.0 * vert) * up;      // down into the shell
    if (mode == 2 && vert < 0) p.dir = p.dir - (2.0 * vert) * up;      // up into the shell
    if ((mode == 1 || mode == 2) && g.u() < 0.2) {                     // grazing entries
      const V3 tang = unit(cross(up, rnd_unit(g)));
      p.dir = unit(tang + ((mode == 1 ? -1.0 : 1.0) * g.logu(1e-9, 1e-1)) * up);
    }
    if (mode == 3) {
      const V3 tang = unit(cross(up, rnd_unit(g)));
      p.dir = unit(tang + (g.sym() * g.logu(1e-8, 2e-1)) * up);
    }
    if (mode == 4 && g.u() < 0.5) {   // moving in
      const double vv = dot(p.dir, up);
      if ((r > rt) == (vv > 0)) p.dir = p.dir - (2.0 * vv) * up;
    }
    if (mode == 5) {
      const V3 tang = unit(cross(up, rnd_unit(g)));
      p.dir = unit((g.u() < 0.5 ? 1.0 : -1.0) * up + g.logu(1e-12, 1e-2) * tang);
    }
    out[0]++;
    TetLocal L;
    const SphFast F = sph_fast_exit(c, v3(0, 0, 0), p, L);
    if (!F.ok) continue;
    out[1]++;
    const double len_loc = L.R * two_atan(F.t, F.sn, F.cs);
    const SphArc A = sph_arc(c, v3(0, 0, 0), p);
    const SphExit E = sph_exit(c, A, p);
    double len_ref = E.len;
    int face_ref = E.face;
    bool bad = false;
    if (oracle) {
      const double loc[3] = {p.loc.x, p.loc.y, p.loc.z}, dir[3] = {p.dir.x, p.dir.y, p.dir.z};
      double len_o = 0;
      const int face_o = oracle(c.a, c.c, rt, rb, loc, dir, &len_o);
      if (face_o != face_ref || !(std::fabs(len_o - len_ref) <= tol * L.R)) out[5]++;
      face_ref = face_o, len_ref = len_o;
    }
    if (!(len_ref < pos_inf())) out[4]++, bad = true;
    else if (face_ref != F.face) out[2]++, bad = true;
    else {
      const double d = std::fabs(len_loc - len_ref);
      if (!(d <= tol * L.R)) out[3]++, bad = true;
      if (d / L.R > dev[0]) dev[0] = d / L.R;
      if (d / std::fmax(std::fabs(len_ref), 1e-300) > dev[1]) dev[1] = d / std::fmax(std::fabs(len_ref), 1e-300);
    }
    if (bad && first_bad && out[2] + out[3] + out[4] == 1) {
      first_bad[0] = (double)it, first_bad[1] = F.face, first_bad[2] = face_ref, first_bad[3] = len_loc, first_bad[4] = len_ref;
      first_bad[5] = L.R, first_bad[6] = F.t, first_bad[7] = rt, first_bad[8] = rb, first_bad[9] = r, first_bad[10] = dot(p.dir, up);
      first_bad[11] = c.a, first_bad[12] = c.c;
    }
  }
}

// ---- the reflection / transmission event on random bare interfaces: the engine's slowness-form solve (r3d_physics.h
//      rt_choose / rt_apply) beside the oracle's restatement of RTCoef (oracle/r3d_oracle.cpp r3d_oracle_rt_event),
//      the same interface, phonon and uniforms for both.
// mode: 0 solid on solid, moderate contrasts; 1 free surface; 2 incidence within 1e-12 .. 1e-2 of a critical angle of one
//       of the outgoing rays; 3 grazing incidence, cos(i) = 1e-6 .. 1e-2; 4 nearly identical media; 5 a fluid (beta = 0)
//       on either side: the reference's default outcome; 6 contrasts up to 1e3; 7 near-normal incidence, sin(i) = 1e-9 ..
//       1e-3; 8 along the normal exactly (the reference's substitute axis, geom_r3.cpp:146-171); 9 HORIZONTAL faces, normal
//       (0, 0, +-1), through the layered models' flat-face form (rt_choose<true> / rt_apply<true>), vertical rays included
// Two places where the REFERENCE's formulation is the ill-conditioned side, and the comparison allows what it loses:
//   * it takes cos(i) as sqrt(1 - sin(i)^2), good to 1.1e-16 / cos(i) (absolutely), where the engine has n.d itself: the
//     weights (which carry cos(i)) then differ by 1e-16 / cos(i)^2 of themselves, the reflected direction by 1e-16 / cos(i);
//   * its axis normal to the plane of incidence is unit(n x d), good to 1e-16 / sin(i): so is the SH fraction an S ray's
//     polarisation draw is compared with, and the outgoing polarisation.
// out[0] cases, [1] outcomes that differ (type or side), [2] of those: with both draws further than the margin from their
//       thresholds, [3] directions off by more than the tolerance, [4] polarisations off by more than the tolerance,
//       [5] transmitted, [6] outgoing S rays, [7] incident S rays counted as SH
// dev[0] largest direction error, dev[1] largest polarisation error (radians) x sin(theta) of the outgoing ray, over the
//       cases with equal outcome
typedef void (*rt_event_fn)(const double media[6], int has_neighbor, const double normal[3], double theta, double phi,
                            double pol, int type, double u_pol, double u_out, double out[8]);
extern "C" void r3d_emul_rt_events(int mode, uint64_t n, uint64_t seed, double tol, double margin, uint64_t* out,
                                   double* dev, double* first_bad /* 16 doubles or null */, rt_event_fn oracle) {
  SplitMix g{seed * 0x2545F4914F6CDD1Dull + 977u * (uint64_t)mode};
  for (int i = 0; i < 8; i++) out[i] = 0;
  dev[0] = dev[1] = 0.0;
  for (uint64_t it = 0; it < n; it++) {
    double media[6];
    int has_nbr = 1;
    auto medium = [&](double* m) {
      m[1] = 1.5 + 11.5 * g.u();              // alpha
      m[2] = m[1] / (1.45 + 0.7 * g.u());     // beta
      m[0] = 1.0 + 5.0 * g.u();               // rho
    };
    medium(media), medium(media + 3);
    if (mode == 1) has_nbr = 0;
    if (mode == 4) {
      const double e = g.logu(1e-10, 1e-2);
      for (int k = 0; k < 3; k++) media[3 + k] = media[k] * (1.0 + e * g.sym());
    }
    if (mode == 5) {
      if (g.u() < 0.5) media[5] = 0.0;
      else media[2] = 0.0;
    }
    if (mode == 6) {
      const double f = g.logu(1.0, 1e3);
      const bool up = g.u() < 0.5;
      for (int k = 0; k < 3; k++) media[3 + k] = up ? media[k] * f * (0.5 + g.u()) : media[k] / f * (0.5 + g.u());
    }
    int type = g.u() < 0.5 ? RAY_P : RAY_S;
    if (mode == 5 && media[2] == 0.0) type = RAY_P;   // (no S ray travels in a fluid)
    // (mode 8: +z is the one normal whose (theta, phi) give the unit vector back exactly)
    const V3 nrm = mode == 8 ? v3(0, 0, 1) : mode == 9 ? v3(0, 0, g.u() < 0.5 ? 1.0 : -1.0) : rnd_unit(g);
    // incidence: the sine of the angle to the normal
    double sini = std::sqrt(g.u());
    if (mode == 2) {
      // a critical angle of one of the other rays: v_k sin(i) / v_in = 1
      const double v_in = media[type == RAY_P ? 1 : 2];
      double vk[4] = {media[1], media[2], media[4], media[5]};
      double s_crit = 2.0;
      for (int tries = 0; tries < 8 && !(s_crit < 1.0); tries++) s_crit = v_in / vk[(int)(4 * g.u()) & 3];
      if (s_crit < 1.0) sini = s_crit * (1.0 + g.logu(1e-12, 1e-2) * (g.u() < 0.5 ? 1.0 : -1.0));
      if (!(sini < 1.0)) sini = s_crit;
    }
    if (mode == 3) sini = std::sqrt(1.0 - std::pow(g.logu(1e-6, 1e-2), 2));   // grazing
    if (mode == 7) sini = g.logu(1e-9, 1e-3);                                  // near normal
    if (mode == 8) sini = 0.0;                                                 // along the normal exactly
    if (mode == 9 && g.u() < 0.05) sini = 0.0;
    // a direction with that incidence: normal cos(i) + tangent sin(i)
    V3 tng = cross(nrm, rnd_unit(g));
    while (mag2(tng) < 1e-6) tng = cross(nrm, rnd_unit(g));
    tng = (1.0 / std::sqrt(mag2(tng))) * tng;
    V3 dir = std::sqrt(std::fmax(0.0, 1.0 - sini * sini)) * nrm + sini * tng;
    if (mode == 8 || (mode == 9 && sini == 0.0)) dir = nrm;
    // the reference's phonon carries (theta, phi): both sides start from the direction those angles give
    const double theta = std::acos(std::fmax(-1.0, std::fmin(1.0, dir.z))), phi = std::atan2(dir.y, dir.x);
    dir = v3(std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta));
    if (!(dot(nrm, dir) > 0.0)) {   // (rounding turned a grazing ray away from the face: not an arrival)
      it--;
      continue;
    }
    const double pol = kPi * g.sym();
    const double u_pol = 1.0 - g.u(), u_out = 1.0 - g.u();   // (0, 1]
    Phonon p;
    p.t = p.path = p.recent = p.lamp = 0.0, p.loc = v3(0, 0, 0), p.cell = 0, p.moves = 0;
    p.dir = dir, p.pc = std::cos(pol), p.ps = std::sin(pol), p.type = type;
    Iface f;
    f.normal = nrm, f.has_neighbor = has_nbr != 0;
    f.rhoR = media[0], f.vR[0] = media[1], f.vR[1] = media[2];
    f.rhoT = media[3], f.vT[0] = media[4], f.vT[1] = media[5];
    const RtChoice ch = mode == 9 ? rt_choose<true>(p, f, u_pol, u_out) : rt_choose<false>(p, f, u_pol, u_out);
    const bool crossed = mode == 9 ? rt_apply<true>(p, f.normal, ch) : rt_apply<false>(p, f.normal, ch);
    double o[8];
    const double nn[3] = {nrm.x, nrm.y, nrm.z};
    oracle(media, has_nbr, nn, theta, phi, pol, type, u_pol, u_out, o);
    out[0]++;
    out[5] += crossed ? 1 : 0, out[6] += p.type == RAY_S ? 1 : 0, out[7] += (ch.code & 4) ? 1 : 0;
    bool bad = false;
    // what the reference's own formulation is good to at this incidence (above)
    const double ci = dot(nrm, dir), si = std::sqrt(mag2(cross(nrm, dir)));
    const double lost_w = 4e-16 / (ci * ci), lost_dir = 4e-16 / ci, lost_axis = si > 0 ? 4e-16 / si : 0.0;
    if ((int)o[0] != p.type || (o[4] != 0.0) != crossed) {
      out[1]++;
      if (o[6] > margin + lost_w && o[7] > margin + lost_axis) out[2]++, bad = true;
    } else if (((ch.code & 4) != 0) != ((int)o[5] == 2 || (int)o[5] == 5) && p.type == RAY_S && type == RAY_S &&
               !(o[7] > margin + lost_axis)) {
      // (the polarisation draw sat on its threshold and fell the other way: SH one side, SV the other -- same ray, the
      //  particle motion a quarter turn apart; counted with the differing outcomes)
      out[1]++;
    } else {
      const V3 od = v3(std::sin(o[1]) * std::cos(o[2]), std::sin(o[1]) * std::sin(o[2]), std::cos(o[1]));
      const double dd = std::sqrt(mag2(p.dir - od));
      if (!(dd <= tol + lost_dir)) out[3]++, bad = true;
      if (dd > dev[0]) dev[0] = dd;
      if (p.type == RAY_S) {
        // (the polarisation angle is defined about the direction: both as unit vectors in the (theta^, phi^) plane)
        const double dp = std::sqrt((p.pc - std::cos(o[3])) * (p.pc - std::cos(o[3])) + (p.ps - std::sin(o[3])) * (p.ps - std::sin(o[3])));
        // a ray along the pole has no azimuth of its own: phi = atan2(0, 0) on one side, the limit on the other; near
        // the pole the (theta^, phi^) frame turns by (error of the direction) / sin(theta)
        const double st = std::sqrt(od.x * od.x + od.y * od.y);
        if (st > 1e-6) {
          if (!(dp <= tol + (lost_dir + lost_axis) / st)) out[4]++, bad = true;
          if (dp * st > dev[1]) dev[1] = dp * st;
        }
      }
    }
    if (bad && first_bad && first_bad[4] == 0.0) {
      first_bad[0] = (double)it, first_bad[1] = type, first_bad[2] = sini, first_bad[3] = has_nbr;
      for (int k = 0; k < 6; k++) first_bad[4 + k] = media[k];
      first_bad[10] = o[5], first_bad[11] = ch.code, first_bad[12] = o[6], first_bad[13] = u_out, first_bad[14] = u_pol, first_bad[15] = pol;
    }
  }
}
// rt_weights for one interface and incidence sine: w[0..5] the six weights, w[6] the squared determinant they carry
extern "C" void r3d_emul_rt_weights(const double media[6], double sini, int intype, double* w) {
  Iface f;
  f.normal = v3(0, 0, 1), f.has_neighbor = true;
  f.rhoR = media[0], f.vR[0] = media[1], f.vR[1] = media[2];
  f.rhoT = media[3], f.vT[0] = media[4], f.vT[1] = media[5];
  double det2;
  rt_weights(f, sini, intype, w, det2);
  w[6] = det2;
}

// ---- the Snell bend without conversion on random bare faces: bend<FLAT_FACES> (r3d_physics.h) beside the oracle's
//      restatement of Phonon::Refraction_Bend (oracle/r3d_oracle.cpp r3d_oracle_bend_event).
// mode: 0 random faces, moderate velocity steps; 1 steps of 1e-5 .. 1e-3 (what the STEP class of a smooth model sees);
//       2 within 1e-12 .. 1e-2 of total reflection (either side); 3 grazing and near-normal incidence; 4 HORIZONTAL faces
//       through the layered models' flat-face form, vertical rays included
// out[0] cases, [1] crossed / reflected differs, [2] of those: outgoing sine further than `margin` from 1, [3] directions
//       off, [4] polarisations off, [5] crossed, [6] S rays;  dev[0] / dev[1]: largest direction / polarisation x sin(theta) error
typedef void (*bend_event_fn)(const double normal[3], double theta, double phi, double pol, int type, double veli, double velo,
                              double out[4]);
extern "C" void r3d_emul_bend_events(int mode, uint64_t n, uint64_t seed, double tol, double margin, uint64_t* out, double* dev,
                                     bend_event_fn oracle) {
  SplitMix g{seed * 0x2545F4914F6CDD1Dull + 7919u * (uint64_t)mode};
  for (int i = 0; i < 8; i++) out[i] = 0;
  dev[0] = dev[1] = 0.0;
  for (uint64_t it = 0; it < n; it++) {
    const V3 nrm = mode == 4 ? v3(0, 0, g.u() < 0.5 ? 1.0 : -1.0) : rnd_unit(g);
    const double veli = 1.5 + 7.0 * g.u();
    double velo = veli * (0.6 + 0.9 * g.u());
    if (mode == 1) velo = veli * (1.0 + g.logu(1e-5, 1e-3) * (g.u() < 0.5 ? 1.0 : -1.0));
    double sini = std::sqrt(g.u());
    if (mode == 2) {
      velo = veli * (1.05 + g.u());
      sini = (veli / velo) * (1.0 + g.logu(1e-12, 1e-2) * (g.u() < 0.5 ? 1.0 : -1.0));
      if (!(sini < 1.0)) sini = veli / velo;
    }
    if (mode == 3) sini = g.u() < 0.5 ? std::sqrt(1.0 - std::pow(g.logu(1e-6, 1e-2), 2)) : g.logu(1e-9, 1e-3);
    if (mode == 4 && g.u() < 0.05) sini = 0.0;
    V3 tng = cross(nrm, rnd_unit(g));
    while (mag2(tng) < 1e-6) tng = cross(nrm, rnd_unit(g));
    tng = (1.0 / std::sqrt(mag2(tng))) * tng;
    V3 dir = std::sqrt(std::fmax(0.0, 1.0 - sini * sini)) * nrm + sini * tng;
    if (mode == 4 && sini == 0.0 && nrm.z > 0) dir = nrm;
    const double theta = std::acos(std::fmax(-1.0, std::fmin(1.0, dir.z))), phi = std::atan2(dir.y, dir.x);
    dir = v3(std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta));
    if (!(dot(nrm, dir) > 0.0)) {
      it--;
      continue;
    }
    const int type = g.u() < 0.4 ? RAY_P : RAY_S;
    const double pol = type == RAY_P ? 0.0 : kPi * g.sym();
    Phonon p;
    p.t = p.path = p.recent = p.lamp = 0.0, p.loc = v3(0, 0, 0), p.cell = 0, p.moves = 0;
    p.dir = dir, p.pc = std::cos(pol), p.ps = std::sin(pol), p.type = type;
    const bool crossed = mode == 4 ? bend<true>(p, nrm, veli, velo) : bend<false>(p, nrm, veli, velo);
    double o[4];
    const double nn[3] = {nrm.x, nrm.y, nrm.z};
    oracle(nn, theta, phi, pol, type, veli, velo, o);
    out[0]++, out[5] += crossed ? 1 : 0, out[6] += type == RAY_S ? 1 : 0;
    const double ci = dot(nrm, dir), si = std::sqrt(mag2(cross(nrm, dir)));
    const double sino = (velo / veli) * si;
    // (the reference's cos(i) = sqrt(1 - sin^2) and its unit axis unit(n x d): good to 1e-16 / cos(i), 1e-16 / sin(i))
    const double lost_dir = 4e-16 / std::fmax(ci, 1e-300) + (sino < 1.0 ? 4e-16 / std::sqrt(std::fmax(1.0 - sino * sino, 1e-300)) : 0.0);
    const double lost_axis = si > 0 ? 4e-16 / si : 0.0;
    if ((o[3] != 0.0) != crossed) {
      out[1]++;
      if (std::fabs(sino - 1.0) > margin) out[2]++;
      continue;
    }
    const V3 od = v3(std::sin(o[0]) * std::cos(o[1]), std::sin(o[0]) * std::sin(o[1]), std::cos(o[0]));
    const double dd = std::sqrt(mag2(p.dir - od));
    if (!(dd <= tol + lost_dir)) out[3]++;
    if (dd > dev[0]) dev[0] = dd;
    if (type == RAY_S) {
      const double dp = std::sqrt((p.pc - std::cos(o[2])) * (p.pc - std::cos(o[2])) + (p.ps - std::sin(o[2])) * (p.ps - std::sin(o[2])));
      const double st = std::sqrt(od.x * od.x + od.y * od.y);
      if (st > 1e-6) {
        if (!(dp <= tol + (lost_dir + lost_axis) / st)) out[4]++;
        if (dp * st > dev[1]) dev[1] = dp * st;
      }
    }
  }
}

// ---- the scattering's rotation: scatter_transform (r3d_physics.h) beside the oracle's Phonon::Transform
//      (oracle/r3d_oracle.cpp r3d_oracle_transform), random phonons and deflections; mode 1: deflections within 1e-9 .. 1e-3
//      of forward and backward, mode 2: phonons within 1e-9 .. 1e-3 of the poles (where theta^, phi^ turn fastest)
// out[0] cases, [1] directions off, [2] polarisations off;  dev[0] / dev[1] largest direction / polarisation x sin(theta) error
typedef void (*transform_fn)(double theta, double phi, double pol, double rth, double rph, double rpol, double out[3]);
extern "C" void r3d_emul_transforms(int mode, uint64_t n, uint64_t seed, double tol, uint64_t* out, double* dev, transform_fn oracle) {
  SplitMix g{seed * 0x2545F4914F6CDD1Dull + 104729u * (uint64_t)mode};
  out[0] = out[1] = out[2] = 0;
  dev[0] = dev[1] = 0.0;
  for (uint64_t it = 0; it < n; it++) {
    double theta = std::acos(g.sym()), phi = kPi * g.sym(), pol = kPi * g.sym();
    double rth = std::acos(g.sym()), rph = kPi * g.sym(), rpol = g.u() < 0.5 ? 0.0 : kPi * g.sym();
    if (mode == 1) rth = g.u() < 0.5 ? g.logu(1e-9, 1e-3) : kPi - g.logu(1e-9, 1e-3);
    if (mode == 2) theta = g.u() < 0.5 ? g.logu(1e-9, 1e-3) : kPi - g.logu(1e-9, 1e-3);
    Phonon p;
    p.t = p.path = p.recent = p.lamp = 0.0, p.loc = v3(0, 0, 0), p.cell = 0, p.moves = 0, p.type = RAY_S;
    p.dir = v3(std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta));
    p.pc = std::cos(pol), p.ps = std::sin(pol);
    const double dirn[4] = {std::cos(rth), std::cos(rph), std::sin(rph), std::sin(rth)};
    scatter_transform(p, dirn, std::cos(rpol), std::sin(rpol), RAY_S);
    double o[3];
    oracle(theta, phi, pol, rth, rph, rpol, o);
    out[0]++;
    const V3 od = v3(std::sin(o[0]) * std::cos(o[1]), std::sin(o[0]) * std::sin(o[1]), std::cos(o[0]));
    const double dd = std::sqrt(mag2(p.dir - od));
    if (!(dd <= tol)) out[1]++;
    if (dd > dev[0]) dev[0] = dd;
    const double dp = std::sqrt((p.pc - std::cos(o[2])) * (p.pc - std::cos(o[2])) + (p.ps - std::sin(o[2])) * (p.ps - std::sin(o[2])));
    const double st = std::sqrt(od.x * od.x + od.y * od.y);
    if (st > 1e-6) {
      // (both sides carry the polarisation as an angle about axes that turn by (direction error) / sin(theta) near a pole)
      if (!(dp * st <= tol)) out[2]++;
      if (dp * st > dev[1]) dev[1] = dp * st;
    }
  }
}
