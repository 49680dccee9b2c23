// r3d_engine.hip -- the MI355X (gfx950) phonon-transport engine: persistent
// traversal kernel + the C-ABI of include/r3d.h.
//
// Kernel design (one phonon per work-item, 64-wide wavefronts):
//   * persistent grid, one 768-thread workgroup per CU; each WAVE pulls chunks of
//     history ids from one global counter (one atomic per 256 histories) and
//     deals them to its lanes with a ballot + prefix-popcount, so a lane whose
//     history ended is refilled within a few iterations and the wave stays full
//     until the id range is exhausted; what is still in flight then can be carried
//     into the engine's next launch instead of draining the GPU (CarrySlot);
//   * every iteration all running lanes do the same thing: boundary search ->
//     free-path draw -> advance (up to three times when a move ends in a plain
//     hand-over); then the receivers, the light events, and -- for the lanes that
//     parked for it -- the reflection / transmission solve;
//   * small read-only tables (cells of layered and spherical models, scatterer
//     heads, receiver scan / hit records, the receiver hash) are staged in LDS once
//     per workgroup; tetra cells (0.6 MB for the crust-pinch model), CDFs
//     (GBs at TOA degree 9) and bins stay in HBM / L2;
//   * bins are accumulated through per-workgroup LDS accumulators and per-wave
//     catch queues into native fp64 / u64 global atomics;
//   * RNG is counter-based Philox keyed by history id (r3d_rng.h): results are
//     independent of lane, wave, launch geometry, launch boundaries and GPU count.
//
// The product has no CPU path: without a HIP device every entry point fails
// with an error message.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/r3d.h"
#include "r3d_pack.h"
#include "r3d_step.h"
#include "r3d_tables_build.h"

namespace r3d {

#ifndef R3D_BLOCK
#define R3D_BLOCK 768
#endif
// One workgroup per CU shares one copy of the LDS tables.  12 waves = 3 per SIMD, i.e. a
// budget of 168 registers per lane (__launch_bounds__).  Measured on the four models
// (TOA degree 9): 512 threads (2 per SIMD, 183 registers, no spills) 31.6 / 35.5 / 78.4 /
// 10.9 ms, 768 threads (a few spills in the R/T solve) 30.2 / 33.6 / 67.5 / 8.8 ms,
// 1024 threads (128 registers, heavy spills) slower than either.
constexpr int kBlock = R3D_BLOCK;
constexpr int kWaves = kBlock / 64;
constexpr unsigned kQueueCap = 128;  // per-wave catch queue entries (flushed 64 at a time)
constexpr unsigned kChunk = 256;     // history ids a wave claims per global atomic
#ifndef R3D_REFILL_MIN
#define R3D_REFILL_MIN 8
#endif
constexpr unsigned kRefillMin = R3D_REFILL_MIN;   // idle lanes that trigger a refill
#ifndef R3D_RT_BATCH
#define R3D_RT_BATCH 24
#endif
constexpr unsigned kRtBatch = R3D_RT_BATCH;      // parked R/T lanes that trigger the solve (<= 1: no parking)
#ifndef R3D_MOVES_PER_ITER
#define R3D_MOVES_PER_ITER 3
#endif
// Moves a lane may make per loop iteration (see the loop).  Measured at TOA degree 9 on chained
// launches, 1 / 2 / 3 / 4 moves: NSCP 19.5 / 16.8 / 16.4 / 16.5 ms per 1e7; LopNor flat; on
// self-contained launches SphereEarth 60.8 -> 66 ms with 2 (few plain hand-overs there, and the
// loop costs registers): the spherical kernel keeps one move per iteration.
constexpr int kMovesPerIterLayeredTetra = R3D_MOVES_PER_ITER;
#ifndef R3D_MOVE_AGAIN_MIN
#define R3D_MOVE_AGAIN_MIN 16
#endif
constexpr unsigned kMoveAgainMin = R3D_MOVE_AGAIN_MIN;   // lanes that make an extra move worth its while

// ---- optional in-kernel phase timing (diagnostic build only: -DR3D_PHASE_TIMING) ----
#ifdef R3D_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[8];
#define R3D_STAMP(slot)                                                        \
  do {                                                                         \
    unsigned long long now__ = __builtin_readcyclecounter();                   \
    if (lane == 0) atomicAdd(&s_phase[slot], now__ - t_phase);                 \
    t_phase = now__;                                                           \
  } while (0)
#else
#define R3D_STAMP(slot) do { } while (0)
#endif

// ---------------------------------------------------- wave-level helpers ----
// number of set bits of m below this lane
__device__ __forceinline__ unsigned rank_in(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ double bcast(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ V3 bcast(V3 v, int src) { return v3(bcast(v.x, src), bcast(v.y, src), bcast(v.z, src)); }

// Per-wave queue of seismometer catches, in LDS.  Bin updates are global atomics,
// and on CDNA a wave's vector-memory operations retire in issue order: a load
// issued after an atomic waits for it (~1-3 us under load).  Issuing the five
// atomics of every catch where it happens would stall the next cell fetch a couple
// of times per iteration.  Instead catches are parked here and flushed 64 at a
// time, one catch per lane: one such stall per 64 catches, and full-width atomics.
struct CatchQueue {
  uint32_t slot[kQueueCap];      // (seismometer * n_bins + bin) * 2 + type
  double e[4][kQueueCap];        // energy on X, Y, Z and total
};

__device__ __forceinline__ void flush_catches(const KArgs& a, CatchQueue& q, unsigned n, unsigned lane) {
#ifdef R3D_ABLATE_CATCH   // timing-only developer build: drop the bin updates
  return;
#endif
  if (lane < n) {
    const uint32_t sl = q.slot[lane];
    const size_t bin = sl >> 1;
    const uint32_t type = sl & 1u;
    double* e = a.energy + bin * 5;
    unsafeAtomicAdd(e + 0, q.e[0][lane]);
    unsafeAtomicAdd(e + 1, q.e[1][lane]);
    unsafeAtomicAdd(e + 2, q.e[2][lane]);
    unsafeAtomicAdd(e + 3 + type, q.e[3][lane]);
    atomicAdd(a.counts + bin * 2 + type, 1ull);
  }
}

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
constexpr size_t kAccEntryBytes = 5 * sizeof(double) + 3 * sizeof(uint32_t);

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

// Seismometer collection for all of the wave's arrivals at once: same tests and same bin
// updates as collect() in r3d_step.h (reference dataout.cpp:103-216, :545-568).  Lane l
// arrives with the candidate receivers [k0, k1) of its hash cell (k0 == k1: none).  The
// (arrival, candidate) pairs of the whole wave are numbered through a prefix sum of the
// candidate counts and dealt to the 64 lanes, 64 pairs per pass: a pair's lane finds its
// arrival by bisection over the prefix sums and fetches the arrival's state from that lane
// (ds_bpermute).  A typical iteration has 3-4 arrivals with a few candidates each, i.e.
// one pass, where serving the arrivals one after the other took one pass each.
// q_count is the wave-uniform fill of the wave's catch queue.
template <int KIND, bool TRACE>
__device__ __forceinline__ void collect_pairs(const KArgs& a, const Tables<KIND>& T, const Phonon& p,
                                              double vel_lane, uint32_t k0, uint32_t k1,
                                              const uint16_t* lds_items /* or null: a.grid.items */,
                                              unsigned lane, uint32_t& lane_catches,
                                              const BinCache& bc, CatchQueue& q, unsigned& q_count) {
  const uint32_t cnt = k1 - k0;
  uint32_t incl = cnt;   // inclusive prefix sum over the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(incl, off);
    if (lane >= (unsigned)off) incl += y;
  }
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  const uint32_t excl = incl - cnt;
  // this lane's arrival record (meaningful where cnt > 0): Phonon::DirectionOfMotion,
  // squared amplitude, and direction / velocity for the plane-wave arrival correction
  V3 dopm = p.dir;
  if (cnt && p.type != RAY_P) {
    V3 th, ph;
    sph_basis(p.dir, th, ph);
    dopm = p.pc * th + p.ps * ph;
  }
  const double amp2 = p.amp * p.amp;
  const double inv_vel = 1.0 / vel_lane;
  for (uint32_t base = 0; base < total; base += 64u) {
    const uint32_t j = base + lane;
    uint32_t src = 0;   // smallest lane whose inclusive sum exceeds j
#pragma unroll
    for (uint32_t step = 32u; step; step >>= 1) {
      const uint32_t v = (uint32_t)__shfl((int)incl, (int)(src + step - 1u));
      if (v <= j) src += step;
    }
    const bool valid = j < total;
    src = valid ? src : lane;
    const uint32_t k = (uint32_t)__shfl((int)k0, (int)src) + (j - (uint32_t)__shfl((int)excl, (int)src));
    const V3 loc = v3(__shfl(p.loc.x, (int)src), __shfl(p.loc.y, (int)src), __shfl(p.loc.z, (int)src));
    const V3 dir = v3(__shfl(p.dir.x, (int)src), __shfl(p.dir.y, (int)src), __shfl(p.dir.z, (int)src));
    const V3 dm = v3(__shfl(dopm.x, (int)src), __shfl(dopm.y, (int)src), __shfl(dopm.z, (int)src));
    const double t = __shfl(p.t, (int)src), a2 = __shfl(amp2, (int)src), iv = __shfl(inv_vel, (int)src);
    const int type = __shfl(p.type, (int)src);
    bool hit = false;
    uint32_t hit_slot = 0;
    double ex = 0, ey = 0, ez = 0, et = 0;
    if (valid) {
      const uint32_t s = lds_items ? (uint32_t)lds_items[k] : a.grid.items[k];
      const SeisScan& S = T.seis_scan[s];
      const V3 to = v3(S.loc) - loc;
      const double dist = mag(to);
      if (!(dist > S.r_out[type] || dist < S.r_in[type])) {
        double arv = t;
        if (S.r_in[type] <= 0) arv += dot(to, dir) * iv;
        const double scaled = arv / a.time_per_bin;
        const double fl = floor(scaled);
        if (scaled >= 0.0 && fl < a.n_bins_f) {
          const uint32_t bin = (uint32_t)fl;
          const SeisHit& H = T.seis_hit[s];
          const double xf = dot(dm, v3(H.axes[0])), yf = dot(dm, v3(H.axes[1])), zf = dot(dm, v3(H.axes[2]));
          et = a2 * H.inv_norm[type];
          ex = et * (xf * xf), ey = et * (yf * yf), ez = et * (zf * zf);
          hit_slot = ((s * a.n_bins + bin) << 1) | (uint32_t)type;
          hit = true;
        }
      }
    }
    unsigned long long hm = __ballot(hit);
    if (!hm) continue;
    if (TRACE) {   // per-history catch counts for the final records
      for (unsigned long long r = hm; r; r &= r - 1ull) {
        const int b = __ffsll((long long)r) - 1;
        if ((int)lane == __builtin_amdgcn_readlane((int)src, b)) lane_catches++;
      }
    } else if (lane == 0) {
      lane_catches += (uint32_t)__popcll(hm);   // (only the wave's total is tallied)
    }
    if (bc.on && hit && bin_cache_add(bc, hit_slot >> 1, (uint32_t)type, ex, ey, ez, et)) hit = false;
    const unsigned long long m = __ballot(hit);   // catches the accumulators did not take
    if (m) {
      if (hit) {
        const unsigned at = q_count + rank_in(m);
        q.slot[at] = hit_slot;
        q.e[0][at] = ex, q.e[1][at] = ey, q.e[2][at] = ez, q.e[3][at] = et;
      }
      q_count += (unsigned)__popcll(m);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (q_count >= 64u) {   // flush the oldest 64, slide the rest down
        flush_catches(a, q, 64u, lane);
        const unsigned rest = q_count - 64u;
        uint32_t ms = 0;
        double m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        if (lane < rest) {
          ms = q.slot[64u + lane];
          m0 = q.e[0][64u + lane], m1 = q.e[1][64u + lane], m2 = q.e[2][64u + lane], m3 = q.e[3][64u + lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < rest) {
          q.slot[lane] = ms;
          q.e[0][lane] = m0, q.e[1][lane] = m1, q.e[2][lane] = m2, q.e[3][lane] = m3;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        q_count = rest;
      }
    }
  }
}

// --------------------------------------------------------------- the kernel --
// A batch ends in a drain phase: the work counter is exhausted, ever fewer lanes still
// carry a history, and the longest histories are ~40 times the mean -- about 8 of a lone
// 1e7-history NSCP launch's 25 ms.  With carry-over a wave that finds the counter
// exhausted parks each unfinished history in its work-item's own slot in HBM and exits;
// the engine's next launch resumes it in the same work-item before handing out new ids.
// Histories are keyed by id and draw from per-history counters, so which launch runs
// which part of a history changes no result.
struct CarrySlot {
  Phonon p;
  Rng rng;
  Pending ev;
  uint32_t state;   // 0 empty, 1 in flight, 3 in flight and parked on a reflection/transmission
};

// RES: which of the small tables are staged in LDS.  RES_ALL: the cell records and the
// scatterer / receiver tables (layered and spherical models: a few dozen cells);
// RES_TABLES: the tables only (tetra models: the cell records come through L1 / L2);
// RES_NONE: neither (models with thousands of scatterers or receivers, whose tables
// alone would not fit the CU's 160 KB).
enum { RES_ALL = 0, RES_TABLES = 1, RES_NONE = 2 };
template <int KIND, int RES, bool TRACE>
__device__ __forceinline__ void propagate_body(const KArgs& a) {
  using Cell = typename CellOf<KIND>::type;
  constexpr bool LDS_CELLS = (RES == RES_ALL), LDS_TABLES = (RES != RES_NONE);
  extern __shared__ __align__(16) unsigned char smem[];

  // ---- stage the small tables in LDS ----
  {
    auto copy_words = [&](void* dst, const void* src, size_t bytes) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
      const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
      for (size_t i = threadIdx.x; i < bytes / 8; i += kBlock) d[i] = s[i];
    };
    if (LDS_CELLS) copy_words(smem + a.lds_cells_off, a.cells, (size_t)a.n_cells * sizeof(Cell));
    if (LDS_TABLES) {
      copy_words(smem + a.lds_scat_off, a.scat_head, (size_t)a.n_scat * sizeof(ScatHead));
      copy_words(smem + a.lds_seis_off, a.seis_scan, (size_t)a.n_seis * sizeof(SeisScan));
    }
    if (a.lds_hit_off != 0xFFFFFFFFu)
      copy_words(smem + a.lds_hit_off, a.seis_hit, (size_t)a.n_seis * sizeof(SeisHit));
    if (a.lds_grid_off != 0xFFFFFFFFu) {   // seismometer hash: offsets as they are, items as u16
      uint32_t* gs = reinterpret_cast<uint32_t*>(smem + a.lds_grid_off);
      for (uint32_t i = threadIdx.x; i <= (uint32_t)a.grid.n_cells; i += kBlock) gs[i] = a.grid.start[i];
      uint16_t* gi = reinterpret_cast<uint16_t*>(gs + a.grid.n_cells + 1);
      for (uint32_t i = threadIdx.x; i < a.grid_n_items; i += kBlock) gi[i] = (uint16_t)a.grid.items[i];
    }
    if (a.acc_bits) {   // energies and counts zero, keys empty
      const size_t n = (size_t)1 << a.acc_bits;
      unsigned long long* z = reinterpret_cast<unsigned long long*>(smem + a.lds_acc_off);
      for (size_t i = threadIdx.x; i < n * 5; i += kBlock) z[i] = 0ull;
      uint32_t* k = reinterpret_cast<uint32_t*>(smem + a.lds_acc_off + n * 5 * sizeof(double));
      for (size_t i = threadIdx.x; i < n * 3; i += kBlock) k[i] = (i < n) ? kEmpty : 0u;
    }
    __syncthreads();
  }
  const bool grid_in_lds = a.lds_grid_off != 0xFFFFFFFFu;   // wave-uniform
  const uint32_t* lds_gstart = reinterpret_cast<const uint32_t*>(smem + (grid_in_lds ? a.lds_grid_off : 0u));
  const uint16_t* lds_gitems = reinterpret_cast<const uint16_t*>(lds_gstart + a.grid.n_cells + 1);
  BinCache bc;
  bc.on = a.acc_bits != 0;
  bc.e = reinterpret_cast<double*>(smem + a.lds_acc_off);
  bc.key = reinterpret_cast<uint32_t*>(smem + a.lds_acc_off + ((size_t)5 * sizeof(double) << a.acc_bits));
  bc.cnt = bc.key + ((size_t)1 << a.acc_bits);
  bc.mask = (1u << a.acc_bits) - 1u, bc.shift = 32u - a.acc_bits;
  const unsigned lane = threadIdx.x & 63u;
  Tables<KIND> T;
  T.cells = LDS_CELLS ? reinterpret_cast<const Cell*>(smem + a.lds_cells_off)
                      : reinterpret_cast<const Cell*>(a.cells);
  T.scat_head = LDS_TABLES ? reinterpret_cast<const ScatHead*>(smem + a.lds_scat_off) : a.scat_head;
  T.seis_scan = LDS_TABLES ? reinterpret_cast<const SeisScan*>(smem + a.lds_seis_off) : a.seis_scan;
  T.seis_hit = (a.lds_hit_off != 0xFFFFFFFFu) ? reinterpret_cast<const SeisHit*>(smem + a.lds_hit_off)
                                              : a.seis_hit;

  // Tallies live in LDS, not in registers: each iteration the wave adds its lanes'
  // 0/1 events with one ballot + one LDS atomic per counter; the block flushes them
  // to HBM once at the end.  Slot order = r3d_run_device's d_scalars.
  __shared__ CatchQueue s_queue[kWaves];
  CatchQueue& queue = s_queue[threadIdx.x >> 6];
  unsigned q_count = 0;   // wave-uniform
  __shared__ unsigned long long s_tally[R3D_N_SCALARS];
  if (threadIdx.x < R3D_N_SCALARS) s_tally[threadIdx.x] = 0ull;
  __syncthreads();
  auto tally = [&](bool cond, int slot) {
    const unsigned long long m = __ballot(cond);
    if (lane == 0 && m) atomicAdd(&s_tally[slot], (unsigned long long)__popcll(m));
  };
  constexpr int kEv = 3 + R3D_INV_NUM;
  // Report stream (diagnostic kernel only; include/r3d.h r3d_event): the lanes for which
  // `cond` holds append one record each; the wave claims the slots with one atomic.
  uint64_t my_id = 0;        // (only read when TRACE)
  auto report = [&](bool cond, int tag, const Phonon& q) {
    if (!TRACE || !a.evlog || !((a.evlog_mask >> tag) & 1u)) return;
    const unsigned long long m = __ballot(cond);
    if (!m) return;
    const int first = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if ((int)lane == first) base = atomicAdd(a.evlog_count, (unsigned long long)__popcll(m));
    base = __shfl(base, first);
    const unsigned long long at = base + (unsigned long long)rank_in(m);
    if (cond && at < a.evlog_cap) {
      r3d_event* r = reinterpret_cast<r3d_event*>(a.evlog) + at;
      r->id = my_id;
      r->time = q.t, r->path = q.path, r->amp = q.amp;
      r->loc[0] = q.loc.x, r->loc[1] = q.loc.y, r->loc[2] = q.loc.z;
      r->dir[0] = q.dir.x, r->dir[1] = q.dir.y, r->dir[2] = q.dir.z;
      r->cell = (uint32_t)q.cell, r->moves = q.moves;
      r->tag = (uint8_t)tag, r->type = (uint8_t)q.type;
    }
  };
#ifdef R3D_PHASE_TIMING
  __shared__ unsigned long long s_phase[8];
  if (threadIdx.x < 8) s_phase[threadIdx.x] = 0ull;
  __syncthreads();
  unsigned long long t_phase = __builtin_readcyclecounter();
#endif

  Phonon p;
  Rng rng;
  uint32_t lane_catches = 0; // catches of the current history (only read when TRACE)
  bool alive = false;
  bool parked = false;   // holds a reflection/transmission event in `ev`, waiting for company
  Pending ev;
  ev.vel = 0.0, ev.face = -1, ev.flags = 0u;
  unsigned long long w_next = 0, w_end = 0;  // wave-uniform: ids this wave still owns
  bool drained = false;                      // wave-uniform: the global counter ran out
  const size_t gtid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (a.carry_in) {   // resume the history this work-item parked at the end of the previous launch
    CarrySlot* c = reinterpret_cast<CarrySlot*>(a.carry_in) + gtid;
    const uint32_t state = c->state;
    if (state) {
      p = c->p, rng = c->rng, ev = c->ev;
      alive = true, parked = (state & 2u) != 0;
      my_id = ((uint64_t)rng.id_hi << 32) | rng.id_lo;
      c->state = 0u;
    }
  }

  for (;;) {
    // ---- refill idle lanes from the wave's id range.  A refill costs a dependent
    //      table search, so wait until kRefillMin lanes are idle -- or none is left
    //      running -- and serve them together ----
    unsigned long long need = __ballot(!alive);
    const bool refill_now = (unsigned)__popcll(need) >= a.refill_min || need == ~0ull;
    while (refill_now && need != 0ull && !drained) {
      if (w_next == w_end) {
        // claim ids: 256 at a time while the batch is far from its end, tapering to 64 so
        // that no wave sits on unstarted histories while others have run dry (judged from
        // where this wave's previous claim ended: no extra look at the counter)
        unsigned long long base = 0, chunk = kChunk;
        if (lane == 0) {
          const unsigned long long seen = w_end;   // (where this wave's previous claim ended: a lower bound)
          const unsigned long long left = (seen < a.n) ? a.n - seen : 0ull;
          const unsigned long long share = left / (2ull * gridDim.x * kWaves);   // per wave, halved
          chunk = (share >= kChunk) ? kChunk : (share >= 128ull ? 128ull : 64ull);
          base = atomicAdd(a.next, chunk);
        }
        base = __shfl(base, 0), chunk = __shfl(chunk, 0);
        if (base >= a.n) {
          drained = true;
          break;
        }
        w_next = base;
        w_end = (base + chunk < a.n) ? base + chunk : a.n;
      }
      const unsigned want = (unsigned)__popcll(need);
      const unsigned long long avail = w_end - w_next;
      const unsigned take = (avail < want) ? (unsigned)avail : want;
      const unsigned rank = rank_in(need);
      const bool fresh = !alive && rank < take;
      if (fresh) {
        my_id = a.first_id + w_next + rank;
        rng_init(rng, my_id);
        spray(a, p, rng);
        alive = true;
        lane_catches = 0;
      }
      report(fresh, 0, p);   // GEN
      if (lane == 0 && take) atomicAdd(&s_tally[kEv + R3D_EV_GENERATED], (unsigned long long)take);
      w_next += take;
      need = __ballot(!alive);
    }
    if (drained && a.carry_out) {   // no ids left: park what is in flight for the next launch
      if (alive) {
        CarrySlot* c = reinterpret_cast<CarrySlot*>(a.carry_out) + gtid;
        c->p = p, c->rng = rng, c->ev = ev;
        c->state = parked ? 3u : 1u;
      }
      break;
    }
    if (!__any(alive)) break;  // every lane idle and nothing left to hand out
    R3D_STAMP(0);  // refill

    // ---- first half of the iteration for every running lane: search, draw, advance ----
    int fate = FATE_ALIVE, reason = 0;
    LaneStats st = {0, 0, 0, 0, 0, 0, 0};   // this iteration's events of this lane
    const bool run = alive && !parked;
#if defined(R3D_PHASE_TIMING) && defined(R3D_PROBE_FETCH)
    if (run) {   // exposed latency of the cell fetch: touch both cache lines of the record, wait
      const volatile double* rec = reinterpret_cast<const volatile double*>(&T.cells[p.cell]);
      double x0 = rec[0], x1 = rec[sizeof(Cell) / 8 - 1];
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(x0), "v"(x1));
    }
    R3D_STAMP(6);
#endif
    // Most moves end on a face with nothing to do but hand the phonon to the neighbour cell
    // (velocity step below 1e-5 everywhere on the face, no receiver surface, no discontinuity:
    // 86 % of the NSCP iterations).  Those lanes take the hand-over here and move again right
    // away, up to kMovesPerIter moves per iteration, so that the per-iteration phases below
    // (receivers, events, refill, book-keeping) are paid once for more than one move.  The
    // order of operations within a history does not change.  (One call site in a loop that
    // is kept rolled: a second copy of the move code would not fit the instruction cache.)
    {
      constexpr int kMovesPerIter = (KIND == CELL_SPH) ? 1 : kMovesPerIterLayeredTetra;
      bool go = run;
      int rep = 0;
#pragma nounroll
      for (;;) {
        if (go) fate = step_move<KIND>(a, T, p, rng, st, &reason, ev);
        if (++rep >= kMovesPerIter) break;
        const bool again = go && fate == FATE_ALIVE && ev.face >= 0 && (ev.flags & F_SMOOTH) != 0 &&
                           (ev.flags & F_ADJOIN) != 0 && (ev.flags & (F_COLLECT | F_REFLECT | F_DISCON)) == 0;
        // worth it for kMoveAgainMin lanes of a full wave -- or for half of the lanes that moved, in
        // a wave that is running thin (the drain of a launch, the flush of a chain)
        const unsigned n_again = (unsigned)__popcll(__ballot(again)), n_went = (unsigned)__popcll(__ballot(go));
        if (n_again < kMoveAgainMin && 2u * n_again < n_went) break;
        tally(st.iterations != 0, kEv + R3D_EV_ITERATIONS);   // the move just made ...
        st.iterations = 0;
        tally(again, kEv + R3D_EV_TRANSFER);                  // ... and the hand-over taken here
        if (again) p.cell = cell_neighbor(T.cells[p.cell], ev.face);
        report(again, 4, p);   // CEL
        go = again;
      }
    }
    const bool moved = run && fate == FATE_ALIVE;
    R3D_STAMP(1);  // move
    report(moved && (ev.flags & F_COLLECT) != 0, 3, p);   // COL: the incident state

    // ---- seismometers.  Each arriving lane looks up its own hash cell; the wave then deals
    //      all (arrival, candidate receiver) pairs to its lanes, 64 per pass (collect_pairs) ----
    uint32_t k0 = 0, k1 = 0;
    if (moved && (ev.flags & F_COLLECT)) {
      st.collect++;
      const SeisGrid& g = a.grid;
      const double fx = (p.loc.x - g.origin[0]) * g.inv_h;
      const double fy = (p.loc.y - g.origin[1]) * g.inv_h;
      const double fz = (p.loc.z - g.origin[2]) * g.inv_h;
      if (fx >= 0 && fy >= 0 && fz >= 0 && fx < g.dim_f[0] && fy < g.dim_f[1] && fz < g.dim_f[2]) {
        const int cellid = ((int)fz * g.dim[1] + (int)fy) * g.dim[0] + (int)fx;
        if (grid_in_lds) k0 = lds_gstart[cellid], k1 = lds_gstart[cellid + 1];
        else k0 = g.start[cellid], k1 = g.start[cellid + 1];
      }
    }
#ifdef R3D_ABLATE_COLLECT  // timing-only developer build
    k1 = k0;
#endif
    if (__any(k1 > k0))
      collect_pairs<KIND, TRACE>(a, T, p, ev.vel, k0, k1, grid_in_lds ? lds_gitems : nullptr, lane,
                                 st.n_catch, bc, queue, q_count);

    R3D_STAMP(2);  // collect

    // ---- second half.  Scatter, bend and hand-over are served at once.  The
    //      reflection/transmission solve is the one long divergent branch (in the tetra
    //      models about a fifth of the lanes per iteration): lanes that need it park until
    //      rt_batch of them have gathered -- or nothing else can run -- and then take it
    //      together.  Draws are per-history counters, so the order in which lanes are
    //      served does not change any history.  (Parking the scattering draw the same way
    //      was tried and measured: no gain on any model.) ----
    constexpr bool kPark = (KIND == CELL_TET);   // (see r3d_engine_create: measured per cell kind)
    if (moved) {
      const bool heavy = kPark && ev.face >= 0 && (ev.flags & (F_REFLECT | F_DISCON)) != 0;
      if (heavy) parked = true;
      else fate = step_event<KIND, kPark ? EV_LIGHT : EV_ALL>(a, T, p, rng, st, ev);
    }
    R3D_STAMP(3);  // light events
    if (kPark) {
      const unsigned n_parked = (unsigned)__popcll(__ballot(parked));
      // (in a wave that is running thin -- the drain of a launch, the flush of a chain -- a third
      //  of the lanes still alive is company enough: waiting for rt_batch there would stretch
      //  the longest histories, which are what the drain waits for)
      const unsigned n_alive = (unsigned)__popcll(__ballot(alive && fate == FATE_ALIVE));
      if (n_parked >= a.rt_batch || (n_parked > 0 && 3u * n_parked >= n_alive)) {
        if (parked) {
          fate = step_event<KIND, EV_RT>(a, T, p, rng, st, ev);
          parked = false;
        }
      }
    }

    R3D_STAMP(4);  // parked R/T
    // ---- book-keeping: this iteration's events, and lanes whose history ended ----
    const bool died = alive && fate != FATE_ALIVE;
    if (TRACE && a.evlog) {
      report(st.scatter != 0, 1, p);    // SCT
      report(st.reflect != 0, 2, p);    // REF
      report(st.transfer != 0, 4, p);   // CEL
      report(died && fate == FATE_LOST, 5, p);
      report(died && fate == FATE_TIMEOUT, 6, p);
      report(died && fate == FATE_INVALID, 7, p);
    }
    tally(st.iterations != 0, kEv + R3D_EV_ITERATIONS);
    tally(st.scatter != 0, kEv + R3D_EV_SCATTER);
    tally(st.collect != 0, kEv + R3D_EV_COLLECT);
    tally(st.reflect != 0, kEv + R3D_EV_REFLECT);
    tally(st.transfer != 0, kEv + R3D_EV_TRANSFER);
    tally(st.rtsolve != 0, kEv + R3D_EV_RTSOLVE);
    if (__any(st.n_catch != 0)) {   // a lane can be caught by several receivers at once
      unsigned long long c = st.n_catch;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
      if (lane == 0) atomicAdd(&s_tally[kEv + R3D_EV_CATCH], c);
    }
    if (__any(died)) {
      tally(died && fate == FATE_LOST, 0);
      tally(died && fate == FATE_TIMEOUT, 1);
    }
    if (TRACE) lane_catches += st.n_catch;
    if (__any(died && fate == FATE_INVALID)) {   // rare
      tally(died && fate == FATE_INVALID, 2);
#pragma unroll
      for (int r = 0; r < R3D_INV_NUM; r++) tally(died && fate == FATE_INVALID && reason == r, 3 + r);
    }
    if (died) {
      alive = false;
      if (TRACE && a.finals) {   // (the diagnostic kernel also runs for the report stream alone)
        r3d_final* f = reinterpret_cast<r3d_final*>(a.finals) + (my_id - a.first_id);
        f->time = p.t, f->path = p.path, f->amp = p.amp;
        f->loc[0] = p.loc.x, f->loc[1] = p.loc.y, f->loc[2] = p.loc.z;
        f->dir[0] = p.dir.x, f->dir[1] = p.dir.y, f->dir[2] = p.dir.z;
        f->moves = p.moves;
        f->fate = (uint8_t)fate;
        f->type = (uint8_t)p.type;
        f->n_catch = (uint16_t)(lane_catches > 65535u ? 65535u : lane_catches);
      }
    }
    R3D_STAMP(5);  // tallies + deaths
  }

  // ---- drain the wave's catch queue, then flush the block's tallies to HBM ----
  flush_catches(a, queue, q_count, lane);
  __syncthreads();
  if (bc.on) {
    for (uint32_t i = threadIdx.x; i <= bc.mask; i += kBlock) {
      const uint32_t bin = bc.key[i];
      if (bin == kEmpty) continue;
      double* e = a.energy + (size_t)bin * 5;
#pragma unroll
      for (int c = 0; c < 5; c++) {
        const double v = bc.e[i * 5u + c];
        if (v != 0.0) unsafeAtomicAdd(e + c, v);
      }
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const uint32_t n = bc.cnt[i * 2u + t];
        if (n) atomicAdd(a.counts + (size_t)bin * 2 + t, (unsigned long long)n);
      }
    }
  }
#ifdef R3D_PHASE_TIMING
  if (threadIdx.x < 8) atomicAdd(&g_phase_cycles[threadIdx.x], s_phase[threadIdx.x]);
#endif
  if (threadIdx.x < R3D_N_SCALARS && s_tally[threadIdx.x] != 0ull)
    atomicAdd(a.scalars + threadIdx.x, s_tally[threadIdx.x]);
}

// The traversal kernel, and the same body under a second name for the flush launch of a carry
// chain (no new ids, only the histories carried over), so that profiles list the two apart.
template <int KIND, int RES, bool TRACE>
__global__ __launch_bounds__(kBlock) void propagate_kernel(const KArgs a) {
  propagate_body<KIND, RES, TRACE>(a);
}
template <int KIND, int RES>
__global__ __launch_bounds__(kBlock) void drain_kernel(const KArgs a) {
  propagate_body<KIND, RES, false>(a);
}

}  // namespace r3d
#include "r3d_pool.h"
namespace r3d {

// ------------------------------------------------------------------- engine --
thread_local std::string g_error;

#define R3D_HIP_OK(call)                                                              \
  do {                                                                                \
    hipError_t err__ = (call);                                                        \
    if (err__ != hipSuccess) {                                                        \
      g_error = std::string(#call) + ": " + hipGetErrorString(err__);                 \
      return fail_value;                                                              \
    }                                                                                 \
  } while (0)

// Entry points make the engine's device current for their own calls and restore the caller's
// on return: a host program (or torch) whose current device differs from the engine's keeps it.
struct DeviceGuard {
  int prev = -1;
  hipError_t status;
  explicit DeviceGuard(int device) {
    status = hipGetDevice(&prev);
    if (status != hipSuccess) prev = -1;
    if (prev == device) prev = -1;   // nothing to switch, nothing to restore
    else status = hipSetDevice(device);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define R3D_ON_DEVICE(dev)          \
  DeviceGuard guard__(dev);         \
  R3D_HIP_OK(guard__.status)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t upload(const void* src, size_t n) {
    bytes = n;
    hipError_t e = hipMalloc(&p, n ? n : 8);
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpy(p, src, n, hipMemcpyHostToDevice);
    return e;
  }
  hipError_t alloc_zero(size_t n) {
    bytes = n;
    hipError_t e = hipMalloc(&p, n ? n : 8);
    if (e != hipSuccess) return e;
    return hipMemset(p, 0, n ? n : 8);
  }
};

}  // namespace r3d

using namespace r3d;

struct r3d_engine {
  int device = 0;
  int kind = 0;
  int res = RES_ALL;   // which tables live in LDS (see propagate_kernel)
  bool pool = true;    // the pool kernel (r3d_pool.h); false: the lane-resident kernel (R3D_KERNEL=lanes)
  size_t carry_bytes = 0;
  int n_seis = 0;
  uint32_t n_bins = 0;
  KArgs args{};
  size_t lds_bytes = 0;
  int grid_blocks = 0;
  hipStream_t stream = nullptr;
  std::vector<std::unique_ptr<DevBuf>> bufs;
  DevBuf d_energy, d_counts, d_scalars, d_next;
  // launches on different streams may be in flight together (a caller overlapping one
  // batch's drain with the next batch): each takes its own work counter from a small ring
  static constexpr unsigned kCounters = 64;
  unsigned launch_seq = 0;
  // ... and its own pair of timing events, so that r3d_kernel_ms(e, launch) reads the launch
  // it names and not whichever recorded last
  hipEvent_t ev0[kCounters] = {}, ev1[kCounters] = {};
  uint64_t launches = 0;   // launches enqueued so far; launch id k used slot (k - 1) % kCounters
  std::unique_ptr<DevBuf> d_carry;   // CarrySlot per work-item of the grid (r3d_run_device_carry)
  bool carry_pending = false;
  uint64_t carry_seed = 0;
  std::unique_ptr<DevBuf> d_volume;
  void* volume_ext = nullptr;   // caller-owned counters (r3d_engine_set_volume_buffer)
  size_t volume_len = 0;
  std::unique_ptr<DevBuf> d_evlog, d_evlog_count;
  struct ScatStats {
    double mfp[2], dipole[2], total[4];
  };
  std::vector<ScatStats> scat_stats;
  std::vector<ScatPtrs> scat_ptrs;   // device addresses of every scatterer's tables
  uint64_t n_toa = 0;

  DevBuf* keep(std::unique_ptr<DevBuf> b) {
    bufs.push_back(std::move(b));
    return bufs.back().get();
  }
  ~r3d_engine() {
    for (unsigned i = 0; i < kCounters; i++) {
      if (ev0[i]) (void)hipEventDestroy(ev0[i]);
      if (ev1[i]) (void)hipEventDestroy(ev1[i]);
    }
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

template <class T>
const T* upload_vec(r3d_engine* e, const std::vector<T>& v, hipError_t* err) {
  auto b = std::make_unique<DevBuf>();
  hipError_t r = b->upload(v.data(), v.size() * sizeof(T));
  if (r != hipSuccess) *err = r;
  return reinterpret_cast<const T*>(e->keep(std::move(b))->p);
}
const double* upload_doubles(r3d_engine* e, const double* src, size_t n, hipError_t* err) {
  auto b = std::make_unique<DevBuf>();
  hipError_t r = b->upload(src, n * sizeof(double));
  if (r != hipSuccess) *err = r;
  return reinterpret_cast<const double*>(e->keep(std::move(b))->p);
}

// Call f(kind, res) with the engine's cell kind and table residency as compile-time constants.
template <class F>
hipError_t with_kernel(const r3d_engine* e, F&& f) {
  auto by_res = [&](auto kind) {
    switch (e->res) {
      case RES_ALL: return f(kind, std::integral_constant<int, RES_ALL>{});
      case RES_TABLES: return f(kind, std::integral_constant<int, RES_TABLES>{});
      default: return f(kind, std::integral_constant<int, RES_NONE>{});
    }
  };
  switch (e->kind) {
    case R3D_CELL_CYLINDER: return by_res(std::integral_constant<int, CELL_CYL>{});
    case R3D_CELL_TETRA: return by_res(std::integral_constant<int, CELL_TET>{});
    default: return by_res(std::integral_constant<int, CELL_SPH>{});
  }
}

// pool kernels: res 0 = cell records + scatterer heads in LDS, 1 = scatterer heads only, 2 = neither
template <class F>
hipError_t with_pool_kernel(const r3d_engine* e, F&& f) {
  return with_kernel(e, [&](auto kind, auto res) {
    constexpr int K = decltype(kind)::value, R = decltype(res)::value;
    return f(kind, std::integral_constant<bool, R == RES_ALL>{}, std::integral_constant<bool, R != RES_NONE>{});
  });
}

hipError_t launch_any(r3d_engine* e, const KArgs& a, bool trace, hipStream_t s, bool drain_only = false) {
  if (e->pool)
    return with_pool_kernel(e, [&](auto kind, auto cells, auto scat) {
      constexpr int K = decltype(kind)::value;
      constexpr bool C = decltype(cells)::value, H = decltype(scat)::value;
      if (drain_only && !trace)
        hipLaunchKernelGGL((pool_drain_kernel<K, C, H>), dim3(e->grid_blocks), dim3(kPoolBlock), e->lds_bytes, s, a);
      else if (trace)
        hipLaunchKernelGGL((pool_kernel<K, C, H, true>), dim3(e->grid_blocks), dim3(kPoolBlock), e->lds_bytes, s, a);
      else
        hipLaunchKernelGGL((pool_kernel<K, C, H, false>), dim3(e->grid_blocks), dim3(kPoolBlock), e->lds_bytes, s, a);
      return hipGetLastError();
    });
  return with_kernel(e, [&](auto kind, auto res) {
    constexpr int K = decltype(kind)::value, R = decltype(res)::value;
    if (drain_only && !trace)
      hipLaunchKernelGGL((drain_kernel<K, R>), dim3(e->grid_blocks), dim3(kBlock), e->lds_bytes, s, a);
    else if (trace)
      hipLaunchKernelGGL((propagate_kernel<K, R, true>), dim3(e->grid_blocks), dim3(kBlock), e->lds_bytes, s, a);
    else
      hipLaunchKernelGGL((propagate_kernel<K, R, false>), dim3(e->grid_blocks), dim3(kBlock), e->lds_bytes, s, a);
    return hipGetLastError();
  });
}

hipError_t set_lds_attr(const r3d_engine* e) {
  if (e->pool)
    return with_pool_kernel(e, [&](auto kind, auto cells, auto scat) {
      constexpr int K = decltype(kind)::value;
      constexpr bool C = decltype(cells)::value, H = decltype(scat)::value;
      hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_kernel<K, C, H, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
      if (r != hipSuccess) return r;
      r = hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_drain_kernel<K, C, H>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
      if (r != hipSuccess) return r;
      return hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_kernel<K, C, H, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
    });
  return with_kernel(e, [&](auto kind, auto res) {
    constexpr int K = decltype(kind)::value, R = decltype(res)::value;
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&propagate_kernel<K, R, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
    if (r != hipSuccess) return r;
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(&drain_kernel<K, R>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
    if (r != hipSuccess) return r;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&propagate_kernel<K, R, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes);
  });
}

hipError_t blocks_per_cu(const r3d_engine* e, int* per_cu) {
  return with_kernel(e, [&](auto kind, auto res) {
    constexpr int K = decltype(kind)::value, R = decltype(res)::value;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, propagate_kernel<K, R, false>, kBlock, e->lds_bytes);
  });
}

bool check_model(const r3d_model_desc* m) {
  if (!m) return g_error = "null model", false;
  if (m->cell_kind < 0 || m->cell_kind > 2) return g_error = "unknown cell kind", false;
  if (m->n_cells <= 0 || !m->cells) return g_error = "model has no cells", false;
  if (m->n_scatterers <= 0 || !m->scatterers) return g_error = "model has no scatterers", false;
  if (m->n_toa == 0 || !m->toa) return g_error = "model has no take-off angles", false;
  if (m->params.n_bins == 0) return g_error = "zero time bins", false;
  // index widths of the device layout: guides and samplers hold take-off indices in 32 bits,
  // a catch's (seismometer, bin, type) travels as one 32-bit word
  if (m->n_toa >= (uint64_t(1) << 32)) return g_error = "more than 2^32 take-off angles", false;
  if ((uint64_t)std::max(0, m->n_seismometers) * m->params.n_bins >= (uint64_t(1) << 31))
    return g_error = "seismometers x time bins must stay below 2^31", false;
  if (m->source.cell < 0 || m->source.cell >= m->n_cells) return g_error = "source cell out of range", false;
  const int want_faces = m->cell_kind == R3D_CELL_CYLINDER ? 3 : m->cell_kind == R3D_CELL_TETRA ? 4 : 2;
  for (int i = 0; i < m->n_cells; i++) {
    const r3d_cell& c = m->cells[i];
    if (c.n_faces != want_faces) return g_error = "cell face count does not match cell kind", false;
    if (c.scatterer < 0 || c.scatterer >= m->n_scatterers) return g_error = "cell scatterer out of range", false;
    for (int f = 0; f < c.n_faces; f++) {
      const r3d_face& F = c.faces[f];
      if ((F.flags & R3D_FACE_ADJOIN) && (F.neighbor < 0 || F.neighbor >= m->n_cells))
        return g_error = "face neighbour out of range", false;
    }
  }
  for (int s = 0; s < m->n_scatterers; s++)
    for (int k = 0; k < 4; k++)
      if (m->scatterers[s].cdf[0] && (!m->scatterers[s].cdf[k] || !m->scatterers[s].spol))
        return g_error = "scatterer table missing", false;
  for (int k = 0; k < 3; k++)
    if (!m->source.cdf[k]) return g_error = "source table missing", false;
  return true;
}

}  // namespace

extern "C" {

const char* r3d_last_error(void) { return g_error.c_str(); }
const char* r3d_version(void) { return "radiative3d_amd engine r1 (gfx950, fp64)"; }

r3d_engine* r3d_engine_create(const r3d_model_desc* m, int device) {
  r3d_engine* const fail_value = nullptr;
  if (!check_model(m)) return nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_error = "no HIP device available: the engine has no CPU path";
    return nullptr;
  }
  if (device < 0 || device >= ndev) {
    g_error = "device index out of range";
    return nullptr;
  }
  R3D_ON_DEVICE(device);
  auto e = std::make_unique<r3d_engine>();
  e->device = device;
  e->kind = m->cell_kind;
  e->n_seis = m->n_seismometers;
  e->n_bins = m->params.n_bins;
  KArgs& a = e->args;
  PackedModel pm;
  pack_model(*m, pm);
  a = pm.args;
  const size_t cell_bytes = pm.cell_bytes();
  // Scheduling knobs.  Parking pays where the R/T solve is a minority branch (tetra models:
  // ~1/5 of the lanes per iteration, measured -10 %); in the layered and spherical models it
  // is taken by most lanes anyway (measured +5 % with parking), so they do not park.
  // (compiled in: the tetra kernel parks, the others do not; rt_batch is its trigger level)
  a.rt_batch = (m->cell_kind == R3D_CELL_TETRA) ? kRtBatch : 1u;
  a.refill_min = kRefillMin;
  if (const char* s = getenv("R3D_RT_BATCH")) a.rt_batch = (uint32_t)std::max(1, atoi(s));   // developer tuning
  if (const char* s = getenv("R3D_REFILL_MIN")) a.refill_min = (uint32_t)std::max(1, atoi(s));
  hipError_t err = hipSuccess;
  // ---- move every table into HBM and point the launch arguments at it ----
  switch (m->cell_kind) {
    case R3D_CELL_CYLINDER: a.cells = upload_vec(e.get(), pm.cyl, &err); break;
    case R3D_CELL_TETRA:
      a.cells = upload_vec(e.get(), pm.tet, &err);
      a.rho = upload_vec(e.get(), pm.rho, &err);
      break;
    default: a.cells = upload_vec(e.get(), pm.sph, &err);
  }
  e->n_toa = m->n_toa;
  e->scat_stats.resize(m->n_scatterers);
  const double* d_toa = nullptr;   // (theta, phi) pairs, only while tables are built here
  std::unique_ptr<DevBuf> toa_buf;
  for (int s = 0; s < m->n_scatterers; s++) {
    const r3d_scatterer& S = m->scatterers[s];
    r3d_engine::ScatStats& st = e->scat_stats[s];
    const double nan = std::nan("");
    if (S.cdf[0]) {   // host-built tables: copy them in
      for (int k = 0; k < 4; k++) {
        pm.scat_ptrs[s].cdf[k] = upload_doubles(e.get(), S.cdf[k], m->n_toa, &err);
        pm.scat_ptrs[s].guide[k] = upload_vec(e.get(), pm.scat_guide[s * 4 + k], &err);
        st.total[k] = S.cdf[k][m->n_toa - 1];
      }
      pm.scat_ptrs[s].spol = upload_doubles(e.get(), S.spol, m->n_toa, &err);
      st.mfp[0] = S.mfp[0], st.mfp[1] = S.mfp[1], st.dipole[0] = st.dipole[1] = nan;
      continue;
    }
    // build-on-device form (r3d_tables_build.hip)
    if (!d_toa) {
      toa_buf = std::make_unique<DevBuf>();
      if (hipError_t r = toa_buf->upload(m->toa, m->n_toa * 2 * sizeof(double)); r != hipSuccess) err = r;
      d_toa = reinterpret_cast<const double*>(toa_buf->p);
    }
    double* d_cdf[4];
    for (int k = 0; k < 4; k++) {
      auto b = std::make_unique<DevBuf>();
      if (hipError_t r = b->alloc_zero(m->n_toa * sizeof(double)); r != hipSuccess) err = r;
      d_cdf[k] = reinterpret_cast<double*>(e->keep(std::move(b))->p);
    }
    auto sp = std::make_unique<DevBuf>();
    if (hipError_t r = sp->alloc_zero(m->n_toa * sizeof(double)); r != hipSuccess) err = r;
    double* d_spol = reinterpret_cast<double*>(e->keep(std::move(sp))->p);
    if (err != hipSuccess) break;
    double cos_sums[4];
    if (hipError_t r = build_scatterer_tables(S.het, S.psdf_numer, d_toa, m->n_toa, d_cdf, d_spol, st.total,
                                              cos_sums, nullptr);
        r != hipSuccess) {
      err = r;
      break;
    }
    for (int k = 0; k < 4; k++) {
      auto g = std::make_unique<DevBuf>();
      if (hipError_t r = g->alloc_zero(((size_t(1) << a.guide_bits) + 1) * sizeof(uint32_t)); r != hipSuccess) err = r;
      uint32_t* d_guide = reinterpret_cast<uint32_t*>(e->keep(std::move(g))->p);
      if (err == hipSuccess)
        if (hipError_t r = build_guide_on_device(d_cdf[k], m->n_toa, a.guide_bits, d_guide, nullptr); r != hipSuccess)
          err = r;
      pm.scat_ptrs[s].cdf[k] = d_cdf[k], pm.scat_ptrs[s].guide[k] = d_guide;
      pm.scat_head[s].total[k] = st.total[k];
    }
    pm.scat_ptrs[s].spol = d_spol;
    // Mean free paths, conversion table and dipole moments from the totals, as the host
    // builder derives them (scatterers.cpp:172-220, :244-259)
    const double* tot = st.total;
    const double n = (double)m->n_toa;
    st.mfp[0] = S.mfp_fixed ? S.mfp[0] : 1.0 / ((tot[0] + tot[1]) / n);
    st.mfp[1] = S.mfp_fixed ? S.mfp[1] : 1.0 / ((tot[2] + tot[3]) / n);
    auto frac = [](double part, double whole) { return whole == 0 ? 0.0 : part / whole; };
    st.dipole[0] = frac(cos_sums[0], tot[0]) * frac(tot[0], tot[0] + tot[1]) +
                   frac(cos_sums[1], tot[1]) * frac(tot[1], tot[0] + tot[1]);
    st.dipole[1] = frac(cos_sums[2], tot[2]) * frac(tot[2], tot[2] + tot[3]) +
                   frac(cos_sums[3], tot[3]) * frac(tot[3], tot[2] + tot[3]);
    const double wp[2][4] = {{tot[0], tot[1], 0, 0}, {0, 0, tot[2], tot[3]}};
    for (int t = 0; t < 2; t++) {
      pm.scat_head[s].mfp[t] = st.mfp[t];
      double acc = 0;
      for (int k = 0; k < 4; k++) pm.scat_head[s].whole[t][k] = (acc += wp[t][k]);
    }
  }
  if (err == hipSuccess && d_toa) err = hipDeviceSynchronize();   // guides done before toa_buf goes
  toa_buf.reset();
  e->scat_ptrs = pm.scat_ptrs;
  a.scat_head = upload_vec(e.get(), pm.scat_head, &err);
  a.scat_ptrs = upload_vec(e.get(), pm.scat_ptrs, &err);
  a.toa_xyz = upload_vec(e.get(), pm.toa_xyz, &err);
  for (int k = 0; k < 3; k++) {
    a.src_cdf[k] = upload_doubles(e.get(), m->source.cdf[k], m->n_toa, &err);
    a.src_guide[k] = upload_vec(e.get(), pm.src_guide[k], &err);
  }
  a.seis_scan = upload_vec(e.get(), pm.seis_scan, &err);
  a.seis_hit = upload_vec(e.get(), pm.seis_hit, &err);
  a.grid.start = upload_vec(e.get(), pm.grid_start, &err);
  a.grid.items = upload_vec(e.get(), pm.grid_items, &err);
  if (err != hipSuccess) {
    g_error = std::string("uploading model tables: ") + hipGetErrorString(err);
    return nullptr;
  }

  // ---- LDS carve-up and launch geometry ----
  if (const char* k = getenv("R3D_KERNEL")) e->pool = std::string(k) != "lanes";   // developer A/B
  a.grid_n_items = (uint32_t)pm.grid_items.size();
  if (e->pool) {
    // Pool kernel (r3d_pool.h): the cell records (when there are few of them) and the scatterer
    // heads are staged in LDS, then a minimum of bin accumulators, and everything else goes to
    // the pool: S slots of 124 B (field-major) plus the rings of 16-bit slot numbers.
    auto align16 = [](size_t x) { return (x + 15) & ~size_t(15); };
    const size_t kLds = 160 * 1024, kStatic = 1024;   // static: queue control words, tallies
    const size_t scat_bytes = (size_t)m->n_scatterers * sizeof(ScatHead);
    const bool scat_fit = scat_bytes <= 24 * 1024;
    const bool cells_fit = scat_fit && cell_bytes + scat_bytes <= 24 * 1024;
    e->res = cells_fit ? RES_ALL : scat_fit ? RES_TABLES : RES_NONE;
    size_t off = 0;
    a.lds_cells_off = a.lds_scat_off = a.lds_seis_off = a.lds_hit_off = a.lds_grid_off = 0xFFFFFFFFu;
    if (cells_fit) a.lds_cells_off = (uint32_t)off, off = align16(off + cell_bytes);
    if (scat_fit) a.lds_scat_off = (uint32_t)off, off = align16(off + scat_bytes);
    uint32_t acc_bits = m->n_seismometers > 0 ? 8u : 0u;   // 256 accumulators = 13 KB ...
    if (const char* s = getenv("R3D_ACC_BITS")) acc_bits = (uint32_t)atoi(s);   // developer tuning
    if (acc_bits && acc_bits < 5) acc_bits = 0;
    a.lds_acc_off = (uint32_t)off, a.acc_bits = acc_bits;
    if (acc_bits) off = align16(off + (kAccEntryBytes << acc_bits));
    const size_t left = kLds - kStatic - off;
    // S slots need 124 S bytes + one ring of (power of two >= S) u16 per queue: try the largest first
    uint32_t slots = 0, cap = 0;
    for (uint32_t s_try = 2048; s_try >= 128; s_try -= 64) {
      uint32_t c = 64;
      while (c < s_try) c <<= 1;
      if ((size_t)s_try * kSlotBytes + (size_t)Q_NUM * c * sizeof(uint16_t) <= left) {
        slots = s_try, cap = c;
        break;
      }
    }
    if (const char* s = getenv("R3D_POOL_SLOTS")) {   // developer tuning (never more than fits)
      const uint32_t want = (uint32_t)atoi(s) / 64 * 64;
      if (want >= 64 && want < slots) {
        slots = want, cap = 64;
        while (cap < slots) cap <<= 1;
      }
    }
    if (slots < (uint32_t)kPoolBlock) {
      g_error = "internal error: the model's LDS tables leave no room for the phonon pool";
      return nullptr;
    }
    a.pool_slots = slots, a.pool_ring_mask = cap - 1;
    a.lds_pool_off = (uint32_t)off, off = align16(off + (size_t)slots * kSlotBytes);
    a.lds_ring_off = (uint32_t)off, off = align16(off + (size_t)Q_NUM * cap * sizeof(uint16_t));
    e->lds_bytes = off;
    if (e->lds_bytes + kStatic > kLds) {
      g_error = "internal error: LDS carve-up exceeds the 160 KB of a CU";
      return nullptr;
    }
  } else {
    auto align16 = [](size_t x) { return (x + 15) & ~size_t(15); };
    size_t off = 0;
    // static LDS of the kernel: the waves' catch queues and the block's tallies
    const size_t kStaticLds = sizeof(CatchQueue) * kWaves + 1024;
    const size_t scat_bytes = (size_t)m->n_scatterers * sizeof(ScatHead);
    const size_t scan_bytes = (size_t)std::max(1, m->n_seismometers) * sizeof(SeisScan);
    // The scatterer heads and the receiver scan records go to LDS unless they would leave
    // less than 16 KB for everything else (thousands of scatterers or receivers); the cell
    // records go with them when they are small (layered and spherical models).
    const bool tables_fit = scat_bytes + scan_bytes + kStaticLds + 16 * 1024 <= 160 * 1024;
    const bool cells_fit = tables_fit && cell_bytes <= 48 * 1024 &&
                           cell_bytes + scat_bytes + scan_bytes + kStaticLds + 16 * 1024 <= 160 * 1024;
    e->res = cells_fit ? RES_ALL : tables_fit ? RES_TABLES : RES_NONE;
    a.lds_cells_off = a.lds_scat_off = a.lds_seis_off = 0xFFFFFFFFu;
    if (cells_fit) a.lds_cells_off = (uint32_t)off, off = align16(off + cell_bytes);
    if (tables_fit) {
      a.lds_scat_off = (uint32_t)off, off = align16(off + scat_bytes);
      a.lds_seis_off = (uint32_t)off, off = align16(off + scan_bytes);
    }
    // the receiver "hit" records go to LDS only if everything still fits in the CU's 160 KB
    const size_t hit_bytes = (size_t)std::max(1, m->n_seismometers) * sizeof(SeisHit);
    a.lds_hit_off = 0xFFFFFFFFu;
    if (tables_fit && off + hit_bytes + kStaticLds + 8192 <= 160 * 1024)
      a.lds_hit_off = (uint32_t)off, off = align16(off + hit_bytes);
    // the seismometer hash, when it is small (it is for the reference's survey lines and arrays:
    // a few thousand cells): both levels of the lookup become LDS reads instead of two
    // dependent global loads in every collecting iteration
    a.lds_grid_off = 0xFFFFFFFFu, a.grid_n_items = (uint32_t)pm.grid_items.size();
    {
      const size_t grid_bytes = ((size_t)a.grid.n_cells + 1) * sizeof(uint32_t) + pm.grid_items.size() * sizeof(uint16_t);
      if (m->n_seismometers > 0 && m->n_seismometers <= 65535 && grid_bytes <= 40 * 1024 &&
          off + grid_bytes + kStaticLds + 8192 <= 160 * 1024)
        a.lds_grid_off = (uint32_t)off, off = align16(off + grid_bytes);
    }
    // what is left (minus a little slack) goes to the bin accumulators
    a.lds_acc_off = (uint32_t)off, a.acc_bits = 0;
    if (m->n_seismometers > 0) {
      const size_t left = 160 * 1024 - std::min<size_t>(160 * 1024, off + kStaticLds + 2048);
      uint32_t bits = 0;
      while (bits < 11 && (kAccEntryBytes << (bits + 1)) <= left) bits++;
      if (const char* s = getenv("R3D_ACC_BITS")) bits = std::min<uint32_t>(bits, (uint32_t)atoi(s));   // developer tuning
      if (bits >= 5) a.acc_bits = bits, off = align16(off + (kAccEntryBytes << bits));
    }
    e->lds_bytes = off;
    if (e->lds_bytes + kStaticLds > 160 * 1024) {
      g_error = "internal error: LDS carve-up exceeds the 160 KB of a CU";
      return nullptr;
    }
  }
  hipDeviceProp_t prop;
  R3D_HIP_OK(hipGetDeviceProperties(&prop, device));
  if (e->lds_bytes > 64 * 1024) R3D_HIP_OK(set_lds_attr(e.get()));
  // persistent grid: exactly the workgroups that can be resident at once (register- and
  // LDS-limited), so every block is running while there is work and none queues behind
  int per_cu = 1;
  if (!e->pool) R3D_HIP_OK(blocks_per_cu(e.get(), &per_cu));   // (the pool takes the CU's whole LDS: one workgroup)
  per_cu = std::max(1, std::min(per_cu, 8));
  e->grid_blocks = prop.multiProcessorCount * per_cu;
  e->carry_bytes = e->pool ? (size_t)e->grid_blocks * a.pool_slots * kSlotBytes
                           : (size_t)e->grid_blocks * kBlock * sizeof(CarrySlot);

  // ---- result scratch, work counter, stream, events ----
  R3D_HIP_OK(e->d_energy.alloc_zero((size_t)std::max(1, e->n_seis) * e->n_bins * R3D_N_ENERGY * sizeof(double)));
  R3D_HIP_OK(e->d_counts.alloc_zero((size_t)std::max(1, e->n_seis) * e->n_bins * R3D_N_COUNT * sizeof(uint64_t)));
  R3D_HIP_OK(e->d_scalars.alloc_zero(R3D_N_SCALARS * sizeof(uint64_t)));
  R3D_HIP_OK(e->d_next.alloc_zero(r3d_engine::kCounters * sizeof(unsigned long long)));
  R3D_HIP_OK(hipStreamCreate(&e->stream));
  for (unsigned i = 0; i < r3d_engine::kCounters; i++) {
    R3D_HIP_OK(hipEventCreate(&e->ev0[i]));
    R3D_HIP_OK(hipEventCreate(&e->ev1[i]));
  }
  return e.release();
}

int r3d_engine_carry_pending(const r3d_engine* e) { return (e && e->carry_pending) ? 1 : 0; }

int r3d_engine_close(r3d_engine* e) {
  const int fail_value = 1;
  if (!e) return 0;
  if (e->carry_pending)
    return g_error = "histories carried over from the last r3d_run_device_carry launch are still in the engine: "
                     "flush them (final != 0) before closing it, or their tallies and bins are lost", 1;
  {
    R3D_ON_DEVICE(e->device);
    R3D_HIP_OK(hipDeviceSynchronize());
    delete e;
  }
  return 0;
}

void r3d_engine_destroy(r3d_engine* e) {
  if (!e) return;
  if (e->carry_pending)   // (a void call cannot refuse: leave the fact where the caller can find it)
    g_error = "r3d_engine_destroy: carried histories were dropped unflushed (see r3d_engine_close)";
  DeviceGuard guard(e->device);
  delete e;
}

size_t r3d_energy_len(const r3d_engine* e) { return e ? (size_t)e->n_seis * e->n_bins * R3D_N_ENERGY : 0; }
size_t r3d_counts_len(const r3d_engine* e) { return e ? (size_t)e->n_seis * e->n_bins * R3D_N_COUNT : 0; }

// carry: 0 none, 1 resume carried histories and park the unfinished ones, 2 resume and finish all
static int enqueue(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, double* d_energy,
                   uint64_t* d_counts, uint64_t* d_scalars, r3d_final* d_finals, hipStream_t s,
                   int carry = 0) {
  const int fail_value = 1;
  if (!e) return g_error = "null engine", 1;
  if (!d_energy || !d_counts || !d_scalars) return g_error = "null result buffer", 1;
  R3D_ON_DEVICE(e->device);
  KArgs a = e->args;
  a.n = n, a.first_id = first_id, a.seed = seed;
  const unsigned slot = e->launch_seq++ % r3d_engine::kCounters;
  a.next = reinterpret_cast<unsigned long long*>(e->d_next.p) + slot;
  a.energy = d_energy;
  a.counts = reinterpret_cast<unsigned long long*>(d_counts);
  a.scalars = reinterpret_cast<unsigned long long*>(d_scalars);
  a.finals = d_finals;
  a.carry_in = a.carry_out = nullptr;
  bool must_launch = n > 0;
  if (carry) {
    if (d_finals) return g_error = "final records and carry-over cannot be combined", 1;
    if (e->carry_pending && seed != e->carry_seed)
      return g_error = "carried histories were started under another seed", 1;
    if (!e->d_carry) {
      auto buf = std::make_unique<DevBuf>();
      R3D_HIP_OK(buf->alloc_zero(e->carry_bytes));
      e->d_carry = std::move(buf);
    }
    if (e->carry_pending) a.carry_in = e->d_carry->p, must_launch = true;
    if (carry == 1) a.carry_out = e->d_carry->p;
    e->carry_pending = (carry == 1) && (n > 0 || e->carry_pending);
    e->carry_seed = seed;
  }
  R3D_HIP_OK(hipMemsetAsync(a.next, 0, sizeof(unsigned long long), s));
  R3D_HIP_OK(hipEventRecord(e->ev0[slot], s));
  if (must_launch)
    R3D_HIP_OK(launch_any(e, a, d_finals != nullptr || a.evlog != nullptr, s, /*drain_only*/ carry == 2 && n == 0));
  R3D_HIP_OK(hipEventRecord(e->ev1[slot], s));
  e->launches++;
  return 0;
}

int r3d_run_device(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, double* d_energy,
                   uint64_t* d_counts, uint64_t* d_scalars, r3d_final* d_finals, void* stream) {
  // NULL means HIP's default (null) stream, which is also what torch's
  // default stream is, so later work queued there is ordered after the kernel.
  return enqueue(e, n, first_id, seed, d_energy, d_counts, d_scalars, d_finals,
                 reinterpret_cast<hipStream_t>(stream));
}

int r3d_run_device_carry(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, double* d_energy,
                         uint64_t* d_counts, uint64_t* d_scalars, void* stream, int final) {
  return enqueue(e, n, first_id, seed, d_energy, d_counts, d_scalars, nullptr,
                 reinterpret_cast<hipStream_t>(stream), final ? 2 : 1);
}

static int run_host(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, r3d_result* out,
                    r3d_final* finals) {
  const int fail_value = 1;
  if (!e) return g_error = "null engine", 1;
  if (!out || !out->energy || !out->counts) return g_error = "null result", 1;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipMemsetAsync(e->d_energy.p, 0, e->d_energy.bytes, e->stream));
  R3D_HIP_OK(hipMemsetAsync(e->d_counts.p, 0, e->d_counts.bytes, e->stream));
  R3D_HIP_OK(hipMemsetAsync(e->d_scalars.p, 0, e->d_scalars.bytes, e->stream));
  DevBuf d_finals;
  if (finals) R3D_HIP_OK(d_finals.alloc_zero(n * sizeof(r3d_final)));
  if (enqueue(e, n, first_id, seed, reinterpret_cast<double*>(e->d_energy.p),
              reinterpret_cast<uint64_t*>(e->d_counts.p), reinterpret_cast<uint64_t*>(e->d_scalars.p),
              finals ? reinterpret_cast<r3d_final*>(d_finals.p) : nullptr, e->stream))
    return 1;
  R3D_HIP_OK(hipStreamSynchronize(e->stream));
  const size_t ne = r3d_energy_len(e), nc = r3d_counts_len(e);
  std::vector<double> he(ne);
  std::vector<uint64_t> hc(nc);
  uint64_t hs[R3D_N_SCALARS];
  if (ne) R3D_HIP_OK(hipMemcpy(he.data(), e->d_energy.p, ne * sizeof(double), hipMemcpyDeviceToHost));
  if (nc) R3D_HIP_OK(hipMemcpy(hc.data(), e->d_counts.p, nc * sizeof(uint64_t), hipMemcpyDeviceToHost));
  R3D_HIP_OK(hipMemcpy(hs, e->d_scalars.p, sizeof hs, hipMemcpyDeviceToHost));
  if (finals) R3D_HIP_OK(hipMemcpy(finals, d_finals.p, n * sizeof(r3d_final), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < ne; i++) out->energy[i] += he[i];
  for (size_t i = 0; i < nc; i++) out->counts[i] += hc[i];
  out->n_lost += hs[0], out->n_timeout += hs[1], out->n_invalid += hs[2];
  for (int r = 0; r < R3D_INV_NUM; r++) out->invalid_reasons[r] += hs[3 + r];
  for (int k = 0; k < R3D_EV_NUM; k++) out->events[k] += hs[3 + R3D_INV_NUM + k];
  return 0;
}

int r3d_run(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, r3d_result* out) {
  return run_host(e, n, first_id, seed, out, nullptr);
}

int r3d_run_model(const r3d_model_desc* model, uint64_t n, uint64_t first_id, uint64_t seed, int n_gpus,
                  r3d_result* out) {
  if (!model || !out || !out->energy || !out->counts) return g_error = "null argument", 1;
  if (n_gpus < 1) return g_error = "n_gpus must be at least 1 (the engine has no CPU path)", 1;
  const size_t ne = (size_t)model->n_seismometers * model->params.n_bins * R3D_N_ENERGY;
  const size_t nc = (size_t)model->n_seismometers * model->params.n_bins * R3D_N_COUNT;
  struct Shard {
    std::vector<double> energy;
    std::vector<uint64_t> counts;
    r3d_result res{};
    std::string error;
  };
  std::vector<Shard> shards(n_gpus);
  std::vector<std::thread> pool;
  for (int g = 0; g < n_gpus; g++) {
    pool.emplace_back([&, g] {
      Shard& sh = shards[g];
      sh.energy.assign(std::max<size_t>(ne, 1), 0.0), sh.counts.assign(std::max<size_t>(nc, 1), 0);
      sh.res.energy = sh.energy.data(), sh.res.counts = sh.counts.data();
      const uint64_t lo = n / n_gpus * g + std::min<uint64_t>(g, n % n_gpus);
      const uint64_t cnt = n / n_gpus + ((uint64_t)g < n % n_gpus ? 1 : 0);
      r3d_engine* e = r3d_engine_create(model, g);
      if (!e) {
        sh.error = g_error;   // (thread-local: this thread's message)
        return;
      }
      if (r3d_run(e, cnt, first_id + lo, seed, &sh.res)) sh.error = g_error;
      r3d_engine_destroy(e);
    });
  }
  for (auto& t : pool) t.join();
  for (const Shard& sh : shards)
    if (!sh.error.empty()) return g_error = sh.error, 1;
  for (const Shard& sh : shards) {
    for (size_t i = 0; i < ne; i++) out->energy[i] += sh.energy[i];
    for (size_t i = 0; i < nc; i++) out->counts[i] += sh.counts[i];
    out->n_lost += sh.res.n_lost, out->n_timeout += sh.res.n_timeout, out->n_invalid += sh.res.n_invalid;
    for (int r = 0; r < R3D_INV_NUM; r++) out->invalid_reasons[r] += sh.res.invalid_reasons[r];
    for (int k = 0; k < R3D_EV_NUM; k++) out->events[k] += sh.res.events[k];
  }
  return 0;
}

int r3d_run_traced(r3d_engine* e, uint64_t n, uint64_t first_id, uint64_t seed, r3d_result* out,
                   r3d_final* finals) {
  if (!finals) return g_error = "null finals", 1;
  return run_host(e, n, first_id, seed, out, finals);
}

#ifdef R3D_PHASE_TIMING
// diagnostic builds only: per queue of the pool kernel, batches served / lanes filled / wave cycles
// since the last call (slot 6 of the first row: idle polls)
int r3d_debug_pool_stats(unsigned long long out[24]) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pool_stats), 24 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long zero[24] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pool_stats), zero, sizeof zero) != hipSuccess;
}
// diagnostic builds only: cumulative per-phase wave cycles since the last call
int r3d_debug_phase_cycles(unsigned long long out[8]) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_cycles), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zero, sizeof zero) != hipSuccess;
}
#endif

int r3d_engine_scatterer_stats(const r3d_engine* e, int s, double out[8]) {
  if (!e || !out || s < 0 || s >= (int)e->scat_stats.size()) return g_error = "scatterer index out of range", 1;
  const r3d_engine::ScatStats& st = e->scat_stats[s];
  out[0] = st.mfp[0], out[1] = st.mfp[1], out[2] = st.dipole[0], out[3] = st.dipole[1];
  for (int k = 0; k < 4; k++) out[4 + k] = st.total[k];
  return 0;
}

int r3d_engine_download_scatterer(r3d_engine* e, int s, double* cdf[4], double* spol) {
  const int fail_value = 1;
  if (!e || s < 0 || s >= (int)e->scat_ptrs.size()) return g_error = "scatterer index out of range", 1;
  R3D_ON_DEVICE(e->device);
  const size_t bytes = e->n_toa * sizeof(double);
  for (int k = 0; k < 4; k++)
    if (cdf && cdf[k]) R3D_HIP_OK(hipMemcpy(cdf[k], e->scat_ptrs[s].cdf[k], bytes, hipMemcpyDeviceToHost));
  if (spol) R3D_HIP_OK(hipMemcpy(spol, e->scat_ptrs[s].spol, bytes, hipMemcpyDeviceToHost));
  return 0;
}

int r3d_engine_set_event_log(r3d_engine* e, uint32_t mask, uint64_t capacity) {
  const int fail_value = 1;
  if (!e) return g_error = "null engine", 1;
  R3D_ON_DEVICE(e->device);
  KArgs& a = e->args;
  a.evlog = nullptr, a.evlog_count = nullptr, a.evlog_cap = 0, a.evlog_mask = 0;
  e->d_evlog.reset(), e->d_evlog_count.reset();
  if (mask == 0 || capacity == 0) return 0;
  auto buf = std::make_unique<DevBuf>(), cnt = std::make_unique<DevBuf>();
  R3D_HIP_OK(buf->alloc_zero(capacity * sizeof(r3d_event)));
  R3D_HIP_OK(cnt->alloc_zero(sizeof(unsigned long long)));
  a.evlog = buf->p, a.evlog_count = reinterpret_cast<unsigned long long*>(cnt->p);
  a.evlog_cap = capacity, a.evlog_mask = mask & R3D_RPT_ALL;
  e->d_evlog = std::move(buf), e->d_evlog_count = std::move(cnt);
  return 0;
}

uint64_t r3d_event_log_count(r3d_engine* e) {
  const uint64_t fail_value = ~uint64_t(0);
  if (!e || !e->d_evlog_count) return 0;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipDeviceSynchronize());
  unsigned long long n = 0;
  R3D_HIP_OK(hipMemcpy(&n, e->d_evlog_count->p, sizeof n, hipMemcpyDeviceToHost));
  return n;
}

uint64_t r3d_event_log_read(r3d_engine* e, r3d_event* out, uint64_t max, int reset) {
  const uint64_t fail_value = ~uint64_t(0);
  if (!e || !e->d_evlog) return g_error = "no event log attached", fail_value;
  const uint64_t total = r3d_event_log_count(e);
  if (total == fail_value) return fail_value;
  const uint64_t n = std::min<uint64_t>(std::min<uint64_t>(total, e->args.evlog_cap), max);
  if (n && !out) return g_error = "null output", fail_value;
  if (n) R3D_HIP_OK(hipMemcpy(out, e->d_evlog->p, n * sizeof(r3d_event), hipMemcpyDeviceToHost));
  if (reset) R3D_HIP_OK(hipMemset(e->d_evlog_count->p, 0, sizeof(unsigned long long)));
  return n;
}

static int attach_volume(r3d_engine* e, const r3d_volume_desc* v, void* d_counters) {
  const int fail_value = 1;
  if (!e) return g_error = "null engine", 1;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipStreamSynchronize(e->stream));
  e->d_volume.reset();
  e->volume_len = 0;
  e->volume_ext = nullptr;
  KArgs& a = e->args;
  a.vol = nullptr;
  if (!v) return 0;
  if (!(v->frame_dt > 0) || v->n_frames == 0 || v->dims[0] == 0 || v->dims[1] == 0 || v->dims[2] == 0 ||
      !(v->cell_size[0] > 0) || !(v->cell_size[1] > 0) || !(v->cell_size[2] > 0))
    return g_error = "volume grid: dimensions, cell sizes and frame length must be positive", 1;
  const size_t len = (size_t)2 * v->n_frames * v->dims[2] * v->dims[1] * v->dims[0];
  if (d_counters) {
    e->volume_ext = d_counters;
  } else {
    auto buf = std::make_unique<DevBuf>();
    R3D_HIP_OK(buf->alloc_zero(len * sizeof(unsigned int)));
    e->d_volume = std::move(buf);
  }
  for (int k = 0; k < 3; k++) {
    a.vol_origin[k] = v->origin[k], a.vol_inv_cell[k] = 1.0 / v->cell_size[k], a.vol_dim[k] = v->dims[k];
    a.vol_dim_f[k] = (double)v->dims[k];
  }
  a.vol_frames = v->n_frames, a.vol_frames_f = (double)v->n_frames;
  a.vol_inv_dt = 1.0 / v->frame_dt;
  a.vol = reinterpret_cast<unsigned int*>(d_counters ? d_counters : e->d_volume->p);
  e->volume_len = len;
  return 0;
}

int r3d_engine_set_volume(r3d_engine* e, const r3d_volume_desc* v) { return attach_volume(e, v, nullptr); }

int r3d_engine_set_volume_buffer(r3d_engine* e, const r3d_volume_desc* v, uint32_t* d_counters) {
  if (v && !d_counters) return g_error = "null volume buffer", 1;
  return attach_volume(e, v, d_counters);
}

size_t r3d_volume_len(const r3d_engine* e) { return e ? e->volume_len : 0; }

void* r3d_volume_device_ptr(r3d_engine* e) {
  if (!e) return nullptr;
  return e->volume_ext ? e->volume_ext : (e->d_volume ? e->d_volume->p : nullptr);
}

int r3d_volume_read(r3d_engine* e, uint32_t* out, int reset) {
  const int fail_value = 1;
  void* const vol = r3d_volume_device_ptr(e);
  if (!vol) return g_error = "no volume grid attached", 1;
  if (!out) return g_error = "null output", 1;
  R3D_ON_DEVICE(e->device);
  R3D_HIP_OK(hipDeviceSynchronize());   // (runs may have been enqueued on caller streams)
  R3D_HIP_OK(hipMemcpy(out, vol, e->volume_len * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (reset) R3D_HIP_OK(hipMemset(vol, 0, e->volume_len * sizeof(uint32_t)));
  return 0;
}

uint64_t r3d_launch_count(const r3d_engine* e) { return e ? e->launches : 0; }

double r3d_kernel_ms(r3d_engine* e, uint64_t launch) {
  if (!e || launch == 0 || launch > e->launches || e->launches - launch >= r3d_engine::kCounters) return -1.0;
  DeviceGuard guard(e->device);
  const unsigned slot = (unsigned)((launch - 1) % r3d_engine::kCounters);
  if (hipEventSynchronize(e->ev1[slot]) != hipSuccess) return -1.0;
  float ms = 0;
  if (hipEventElapsedTime(&ms, e->ev0[slot], e->ev1[slot]) != hipSuccess) return -1.0;
  return (double)ms;
}

double r3d_last_kernel_ms(r3d_engine* e) { return e ? r3d_kernel_ms(e, e->launches) : -1.0; }

}  // extern "C"
