// emul.cpp -- TEST-ONLY host build of the engine's per-work-item code.
//
// Compiles radiative3d_amd/csrc/r3d_step.h (the exact functions the HIP kernel
// runs per lane) with g++ and drives them one history at a time, so kernel
// logic can be single-stepped and compared with the oracle in a container
// that has no GPU.  It is NOT a product path: it is not part of the C-ABI,
// not shipped in radiative3d_amd/lib, and nothing outside tests/ loads it.
#include <cstring>

#include "../../include/r3d.h"
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
    LaneStats st = {0, 0, 0, 0, 0, 0, 0};
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
