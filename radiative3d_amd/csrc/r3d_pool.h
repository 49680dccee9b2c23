// r3d_pool.h -- the traversal kernel as a PHONON POOL: included by r3d_engine.hip (device code).
//
// One-phonon-per-lane-for-life leaves a wave's lanes scattered over the phases of the loop:
// at any moment some lanes need the boundary search, a few a reflection / transmission solve, a
// few a scattering-table draw, a few a fresh history -- and every phase is executed under a
// partial mask (measured on the lane-resident kernel: 36 % of the lanes active in an average
// vector instruction for the layered models, 59 % for the spherical Earth, 65 % for the tetra
// model, with the vector issue slots 82-92 % full).  Here the histories in flight live in LDS
// instead -- a pool of S 128-byte slots per workgroup -- and lanes are only workers:
//
//   * every slot is, between phases, in exactly one of six queues (rings of slot numbers in
//     LDS): MOVE, COLLECT, RT, BEND, SCATTER, FREE;
//   * a wave takes up to 64 slots OF ONE QUEUE, loads their state into registers, runs that
//     phase's code for all of them at once, stores what changed and hands each slot to the queue
//     of its next phase.  With S about twice the workgroup's lanes the queues are deep enough
//     that nearly every batch is a full one, so each phase runs at (close to) 64 lanes;
//   * phases:  FREE -> (claim ids from the global counter, source spray) -> MOVE
//              MOVE -> (termination checks, boundary search, free-path draw, advance) ->
//                      SCATTER | COLLECT | RT | BEND | MOVE (plain hand-over) | FREE (history ended)
//              COLLECT -> (receiver hash, (arrival, receiver) pairs dealt over the lanes, bins) ->
//                      RT | BEND | MOVE | FREE
//              RT, BEND, SCATTER -> MOVE
//   * histories are keyed by id and draw from per-history counters (r3d_rng.h), and every phase
//     is the same per-history code as before (r3d_step.h), so which wave runs which phase of a
//     history, and in what order histories are served, changes no result.
//
// Queue protocol (multi-producer, multi-consumer among the waves of one workgroup, LDS only):
// a ring of 16-bit slot numbers with free-running head / tail tickets and a count of published
// entries.  A consumer takes min(count, 64) with a compare-and-swap on the count, then a range
// of head tickets, and spins on each of its entries until it is no longer EMPTY (the entry may
// belong to a producer that has its ticket but has not written yet), reads it and marks it
// EMPTY.  A producer takes tail tickets, waits until its entries are EMPTY (a consumer that
// holds the ticket of the previous lap may not have read yet), writes them, then adds to the
// count.  A slot's state is written before its number is published and read after it is taken
// (LDS operations of a wave complete in order; release / acquire fences at workgroup scope
// keep the compiler from moving them).
#ifndef R3D_POOL_H_
#define R3D_POOL_H_

namespace r3d {

#ifndef R3D_POOL_BLOCK
#define R3D_POOL_BLOCK 512
#endif
constexpr int kPoolBlock = R3D_POOL_BLOCK;   // 8 waves = 2 per SIMD: 256 registers per lane, no spills
constexpr int kPoolWaves = kPoolBlock / 64;

enum { Q_MOVE = 0, Q_COLLECT = 1, Q_RT = 2, Q_BEND = 3, Q_SCATTER = 4, Q_FREE = 5, Q_NUM = 6 };
constexpr uint16_t kRingEmpty = 0xFFFFu;

// One history in flight.  meta: bit 0 ray type | bits 1-3 pending face + 1 (0: scattered inside
// the cell) | bits 8-15 that face's flags | bits 16-18 the queue the slot is in (for carry-over).
struct alignas(16) Slot {
  double t, path, recent, amp;
  double loc[3], dir[3];
  double pc, ps;
  uint32_t cell, moves, k, meta;
  uint32_t id_lo, id_hi, catches, spare;
};
static_assert(sizeof(Slot) == 128, "a pool slot is 128 bytes");

struct PoolCtl {
  uint32_t head[Q_NUM], tail[Q_NUM], count[Q_NUM];
  uint32_t drained;   // the global id counter ran out
};

__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t meta_pack(int type, int face, uint32_t flags, int queue) {
  return (uint32_t)type | ((uint32_t)(face + 1) << 1) | ((flags & 0xFFu) << 8) | ((uint32_t)queue << 16);
}

// Take up to `want` (<= 64) slot numbers from queue q; returns how many (wave-uniform); lane l < k
// gets its slot in `id`.
__device__ __forceinline__ unsigned q_pop(PoolCtl& ctl, uint16_t* ring, uint32_t mask, int q, unsigned lane,
                                          unsigned want, unsigned& id) {
  unsigned k = 0, pos = 0;
  if (lane == 0) {
    uint32_t c = lds_ld(&ctl.count[q]);
    while (c) {
      const uint32_t t = c < want ? c : want;
      const uint32_t seen = atomicCAS(&ctl.count[q], c, c - t);
      if (seen == c) {
        k = t;
        break;
      }
      c = seen;
    }
    if (k) pos = atomicAdd(&ctl.head[q], k);
  }
  k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
  pos = (unsigned)__builtin_amdgcn_readfirstlane((int)pos);
  id = 0;
  if (lane < k) {
    volatile uint16_t* e = ring + ((pos + lane) & mask);
    uint16_t v;
    do {
      v = *e;
    } while (v == kRingEmpty);
    *e = kRingEmpty;
    id = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return k;
}

// Hand the slots of the lanes with `cond` to queue q.
__device__ __forceinline__ void q_push(PoolCtl& ctl, uint16_t* ring, uint32_t mask, int q, unsigned lane, bool cond,
                                       unsigned id) {
  const unsigned long long m = __ballot(cond);
  if (!m) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the slots' state before their numbers
  const unsigned k = (unsigned)__popcll(m);
  const int first = __ffsll((long long)m) - 1;
  unsigned pos = 0;
  if ((int)lane == first) pos = atomicAdd(&ctl.tail[q], k);
  pos = (unsigned)__shfl((int)pos, first);
  if (cond) {
    volatile uint16_t* e = ring + ((pos + rank_in(m)) & mask);
    while (*e != kRingEmpty) {
    }
    *e = (uint16_t)id;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((int)lane == first) atomicAdd(&ctl.count[q], k);
}

// Seismometer collection for a batch of arrivals (one per lane with k1 > k0): the same tests and
// bin updates as collect() in r3d_step.h (reference dataout.cpp:103-216, :545-568).  The
// (arrival, candidate receiver) pairs of the whole batch are numbered through a prefix sum of
// the candidate counts and dealt to the 64 lanes, 64 pairs per pass; a pair's lane finds its
// arrival by bisection over the prefix sums and fetches the arrival's state from that lane.
// Catches go to the workgroup's bin accumulators when those admit them, else straight to HBM as
// native fp64 / u64 atomics.  Returns the number of catches of the batch (wave-uniform);
// lane_catches counts per arriving lane when TRACE.
template <int KIND, bool TRACE>
__device__ __forceinline__ uint32_t pool_collect_pairs(const KArgs& a, const Tables<KIND>& T, const Phonon& p,
                                                       double vel_lane, uint32_t k0, uint32_t k1,
                                                       const uint16_t* lds_items /* or null: a.grid.items */,
                                                       unsigned lane, uint32_t& lane_catches, const BinCache& bc) {
  const uint32_t cnt = k1 - k0;
  uint32_t incl = cnt;   // inclusive prefix sum over the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(incl, off);
    if (lane >= (unsigned)off) incl += y;
  }
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  const uint32_t excl = incl - cnt;
  V3 dopm = p.dir;
  if (cnt && p.type != RAY_P) {
    V3 th, ph;
    sph_basis(p.dir, th, ph);
    dopm = p.pc * th + p.ps * ph;
  }
  const double amp2 = p.amp * p.amp;
  const double inv_vel = 1.0 / vel_lane;
  uint32_t n_hits = 0;
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
    uint32_t hit_bin = 0;
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
          const SeisHit& H = T.seis_hit[s];
          const double xf = dot(dm, v3(H.axes[0])), yf = dot(dm, v3(H.axes[1])), zf = dot(dm, v3(H.axes[2]));
          et = a2 * H.inv_norm[type];
          ex = et * (xf * xf), ey = et * (yf * yf), ez = et * (zf * zf);
          hit_bin = s * a.n_bins + (uint32_t)fl;
          hit = true;
        }
      }
    }
    const unsigned long long hm = __ballot(hit);
    if (!hm) continue;
    n_hits += (uint32_t)__popcll(hm);
    if (TRACE) {   // per-history catch counts for the final records
      for (unsigned long long r = hm; r; r &= r - 1ull) {
        const int b = __ffsll((long long)r) - 1;
        if ((int)lane == __builtin_amdgcn_readlane((int)src, b)) lane_catches++;
      }
    }
#ifndef R3D_ABLATE_CATCH
    if (hit && !(bc.on && bin_cache_add(bc, hit_bin, (uint32_t)type, ex, ey, ez, et))) {
      double* e = a.energy + (size_t)hit_bin * 5;
      unsafeAtomicAdd(e + 0, ex);
      unsafeAtomicAdd(e + 1, ey);
      unsafeAtomicAdd(e + 2, ez);
      unsafeAtomicAdd(e + 3 + type, et);
      atomicAdd(a.counts + (size_t)hit_bin * 2 + type, 1ull);
    }
#endif
  }
  return n_hits;
}

#ifdef R3D_PHASE_TIMING
// diagnostic build: per queue, batches served, lanes filled, wave cycles spent; slot 6: idle polls
__device__ unsigned long long g_pool_stats[3][8];
#endif

// LDS_CELLS / LDS_SCAT: the cell records / the scatterer heads are staged in LDS (models with a few
// dozen cells; all but models with thousands of scatterers).  The receiver tables are read through
// L1 / L2 (a collection phase serves 64 arrivals at once, so their latency is paid per batch).
template <int KIND, bool LDS_CELLS, bool LDS_SCAT, bool TRACE>
__device__ __forceinline__ void pool_body(const KArgs& a) {
  using Cell = typename CellOf<KIND>::type;
  constexpr bool LDS_SEIS = false;
  extern __shared__ __align__(16) unsigned char smem[];
  const unsigned tid = threadIdx.x, lane = tid & 63u;

  // ---- stage the small tables in LDS ----
  {
    auto copy_words = [&](void* dst, const void* src, size_t bytes) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
      const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
      for (size_t i = tid; i < bytes / 8; i += kPoolBlock) d[i] = s[i];
    };
    if (LDS_CELLS) copy_words(smem + a.lds_cells_off, a.cells, (size_t)a.n_cells * sizeof(Cell));
    if (LDS_SCAT) copy_words(smem + a.lds_scat_off, a.scat_head, (size_t)a.n_scat * sizeof(ScatHead));
    if (LDS_SEIS) {
      copy_words(smem + a.lds_seis_off, a.seis_scan, (size_t)a.n_seis * sizeof(SeisScan));
      uint32_t* gs = reinterpret_cast<uint32_t*>(smem + a.lds_grid_off);
      for (uint32_t i = tid; i <= (uint32_t)a.grid.n_cells; i += kPoolBlock) gs[i] = a.grid.start[i];
      uint16_t* gi = reinterpret_cast<uint16_t*>(gs + a.grid.n_cells + 1);
      for (uint32_t i = tid; i < a.grid_n_items; i += kPoolBlock) gi[i] = (uint16_t)a.grid.items[i];
    }
    if (a.acc_bits) {   // energies and counts zero, keys empty
      const size_t n = (size_t)1 << a.acc_bits;
      unsigned long long* z = reinterpret_cast<unsigned long long*>(smem + a.lds_acc_off);
      for (size_t i = tid; i < n * 5; i += kPoolBlock) z[i] = 0ull;
      uint32_t* k = reinterpret_cast<uint32_t*>(smem + a.lds_acc_off + n * 5 * sizeof(double));
      for (size_t i = tid; i < n * 3; i += kPoolBlock) k[i] = (i < n) ? kEmpty : 0u;
    }
  }
  const uint32_t* lds_gstart = reinterpret_cast<const uint32_t*>(smem + (LDS_SEIS ? a.lds_grid_off : 0u));
  const uint16_t* lds_gitems = reinterpret_cast<const uint16_t*>(lds_gstart + a.grid.n_cells + 1);
  BinCache bc;
  bc.on = a.acc_bits != 0;
  bc.e = reinterpret_cast<double*>(smem + a.lds_acc_off);
  bc.key = reinterpret_cast<uint32_t*>(smem + a.lds_acc_off + ((size_t)5 * sizeof(double) << a.acc_bits));
  bc.cnt = bc.key + ((size_t)1 << a.acc_bits);
  bc.mask = (1u << a.acc_bits) - 1u, bc.shift = 32u - a.acc_bits;
  Tables<KIND> T;
  T.cells = LDS_CELLS ? reinterpret_cast<const Cell*>(smem + a.lds_cells_off) : reinterpret_cast<const Cell*>(a.cells);
  T.scat_head = LDS_SCAT ? reinterpret_cast<const ScatHead*>(smem + a.lds_scat_off) : a.scat_head;
  T.seis_scan = LDS_SEIS ? reinterpret_cast<const SeisScan*>(smem + a.lds_seis_off) : a.seis_scan;
  T.seis_hit = a.seis_hit;   // fetched on a hit only: stays in HBM / L2

  // ---- the pool, its queues, the block's tallies ----
  Slot* const pool = reinterpret_cast<Slot*>(smem + a.lds_pool_off);
  uint16_t* const rings = reinterpret_cast<uint16_t*>(smem + a.lds_ring_off);
  const uint32_t S = a.pool_slots, rmask = a.pool_ring_mask, rcap = a.pool_ring_mask + 1u;
  __shared__ PoolCtl ctl;
  __shared__ unsigned long long s_tally[R3D_N_SCALARS];
  for (uint32_t i = tid; i < Q_NUM * rcap; i += kPoolBlock) rings[i] = kRingEmpty;
  if (tid < R3D_N_SCALARS) s_tally[tid] = 0ull;
  if (tid == 0) {
    for (int q = 0; q < Q_NUM; q++) ctl.head[q] = ctl.tail[q] = ctl.count[q] = 0u;
    ctl.drained = 0u;
  }
  __syncthreads();
  auto ring = [&](int q) { return rings + (size_t)q * rcap; };
  if (a.carry_in) {
    // resume: the pool image this workgroup parked at the end of the engine's previous launch;
    // every slot goes back to the queue named in its meta word
    const Slot* img = reinterpret_cast<const Slot*>(a.carry_in) + (size_t)blockIdx.x * S;
    for (uint32_t base = 0; base < S; base += kPoolBlock) {
      const uint32_t s = base + tid;
      int tag = -1;
      if (s < S) {
        pool[s] = img[s];
        tag = (int)((pool[s].meta >> 16) & 7u);
      }
#pragma unroll
      for (int q = 0; q < Q_NUM; q++) q_push(ctl, ring(q), rmask, q, lane, tag == q, s);
    }
  } else {
    for (uint32_t s = tid; s < S; s += kPoolBlock) {
      ring(Q_FREE)[s] = (uint16_t)s;
      pool[s].meta = meta_pack(0, -1, 0u, Q_FREE);
    }
    if (tid == 0) ctl.tail[Q_FREE] = S, ctl.count[Q_FREE] = S;
  }
  __syncthreads();

  constexpr int kEv = 3 + R3D_INV_NUM;
  auto tally_n = [&](int slot, unsigned long long n) {
    if (lane == 0 && n) atomicAdd(&s_tally[slot], n);
  };
  auto tally = [&](bool cond, int slot) { tally_n(slot, (unsigned long long)__popcll(__ballot(cond))); };
  // Report stream (diagnostic kernel only; include/r3d.h r3d_event): the lanes for which `cond`
  // holds append one record each; the wave claims the slots with one atomic.
  auto report = [&](bool cond, int tag, const Phonon& q, uint64_t hid) {
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
      r->id = hid;
      r->time = q.t, r->path = q.path, r->amp = q.amp;
      r->loc[0] = q.loc.x, r->loc[1] = q.loc.y, r->loc[2] = q.loc.z;
      r->dir[0] = q.dir.x, r->dir[1] = q.dir.y, r->dir[2] = q.dir.z;
      r->cell = (uint32_t)q.cell, r->moves = q.moves;
      r->tag = (uint8_t)tag, r->type = (uint8_t)q.type;
    }
  };
  // A history ended (lanes with `died`): loss counters, the final record, the slot back to FREE.
  auto finish = [&](bool died, int fate, int reason, const Phonon& p, uint64_t hid, uint32_t catches, unsigned id) {
    if (!__any(died)) return;
    tally(died && fate == FATE_LOST, 0);
    tally(died && fate == FATE_TIMEOUT, 1);
    if (__any(died && fate == FATE_INVALID)) {   // rare
      tally(died && fate == FATE_INVALID, 2);
#pragma unroll
      for (int r = 0; r < R3D_INV_NUM; r++) tally(died && fate == FATE_INVALID && reason == r, 3 + r);
    }
    if (TRACE) {
      report(died && fate == FATE_LOST, 5, p, hid);
      report(died && fate == FATE_TIMEOUT, 6, p, hid);
      report(died && fate == FATE_INVALID, 7, p, hid);
      if (died && a.finals) {   // (the diagnostic kernel also runs for the report stream alone)
        r3d_final* f = reinterpret_cast<r3d_final*>(a.finals) + (hid - a.first_id);
        f->time = p.t, f->path = p.path, f->amp = p.amp;
        f->loc[0] = p.loc.x, f->loc[1] = p.loc.y, f->loc[2] = p.loc.z;
        f->dir[0] = p.dir.x, f->dir[1] = p.dir.y, f->dir[2] = p.dir.z;
        f->moves = p.moves;
        f->fate = (uint8_t)fate;
        f->type = (uint8_t)p.type;
        f->n_catch = (uint16_t)(catches > 65535u ? 65535u : catches);
      }
    }
    if (died) pool[id].meta = meta_pack(0, -1, 0u, Q_FREE);
    q_push(ctl, ring(Q_FREE), rmask, Q_FREE, lane, died, id);
  };
  // Where a phonon that sits on face `face` with flags `fl` goes next (phonons.cpp:640-676):
  // lost | full reflection / transmission solve | plain hand-over (taken at once) | Snell bend or
  // run-time test.  Returns the destination queue, or -1 if the history ends here (lost).
  auto face_dest = [&](uint32_t fl) -> int {
    if (!(fl & (F_REFLECT | F_ADJOIN))) return -1;
    if (fl & (F_REFLECT | F_DISCON)) return Q_RT;
    if (fl & F_SMOOTH) return Q_MOVE;
    return Q_BEND;
  };
  auto load_phonon = [&](unsigned id, Phonon& p, Rng& rng, uint32_t& meta) {
    const Slot& s = pool[id];
    p.t = s.t, p.path = s.path, p.recent = s.recent, p.amp = s.amp;
    p.loc = v3(s.loc), p.dir = v3(s.dir);
    p.pc = s.pc, p.ps = s.ps;
    p.cell = (int32_t)s.cell, p.moves = s.moves;
    meta = s.meta;
    p.type = (int32_t)(meta & 1u);
    rng.k = s.k, rng.id_lo = s.id_lo, rng.id_hi = s.id_hi;
  };
#ifdef R3D_PHASE_TIMING
  __shared__ unsigned long long s_stats[3][8];
  if (tid < 24) s_stats[tid / 8][tid % 8] = 0ull;
  __syncthreads();
#endif

  for (;;) {
    // ---- choose a queue: a full batch of a minor phase first (they all feed MOVE), then a
    //      refill, then MOVE; with no full batch anywhere, the fullest queue ----
    const uint32_t cnt = lane < Q_NUM ? lds_ld(&ctl.count[lane]) : 0u;
    const uint32_t drained = lds_ld(&ctl.drained);
    if (drained && a.carry_out) break;   // no ids left: the pool is parked as it is for the next launch
    uint32_t c[Q_NUM];
#pragma unroll
    for (int q = 0; q < Q_NUM; q++) c[q] = (uint32_t)__builtin_amdgcn_readlane((int)cnt, q);
    if (drained) {
      if (c[Q_FREE] == S) break;   // every slot is free and nothing is left to hand out
      c[Q_FREE] = 0u;              // (free slots are of no use any more)
    }
    int q = -1;
    if (c[Q_RT] >= 64u) q = Q_RT;
    else if (c[Q_COLLECT] >= 64u) q = Q_COLLECT;
    else if (c[Q_SCATTER] >= 64u) q = Q_SCATTER;
    else if (c[Q_BEND] >= 64u) q = Q_BEND;
    else if (c[Q_FREE] >= 64u) q = Q_FREE;
    else if (c[Q_MOVE] >= 64u) q = Q_MOVE;
    else {
      uint32_t best = 0u;
#pragma unroll
      for (int j = 0; j < Q_NUM; j++)
        if (c[j] > best) best = c[j], q = j;
    }
    if (q < 0) {   // everything in flight is in other waves' hands
#ifdef R3D_PHASE_TIMING
      if (lane == 0) atomicAdd(&s_stats[0][6], 1ull);
#endif
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    unsigned id;
    const unsigned k = q_pop(ctl, ring(q), rmask, q, lane, 64u, id);
    if (k == 0) continue;   // another wave was quicker
    const bool act = lane < k;
#ifdef R3D_PHASE_TIMING
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif

    if (q == Q_FREE) {
      // ---- fresh histories: ids from the global counter, source spray (events.cpp:111-124) ----
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(a.next, (unsigned long long)k);
      base = __shfl(base, 0);
      const unsigned take = base >= a.n ? 0u : (a.n - base < k ? (unsigned)(a.n - base) : k);
      if (take < k && lane == 0) __hip_atomic_store(&ctl.drained, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const bool fresh = lane < take;
      Phonon p;
      uint64_t hid = 0;
      if (fresh) {
        Rng rng;
        hid = a.first_id + base + lane;
        rng_init(rng, hid);
        spray(a, p, rng);
        Slot& s = pool[id];
        s.t = p.t, s.path = p.path, s.recent = p.recent, s.amp = p.amp;
        s.loc[0] = p.loc.x, s.loc[1] = p.loc.y, s.loc[2] = p.loc.z;
        s.dir[0] = p.dir.x, s.dir[1] = p.dir.y, s.dir[2] = p.dir.z;
        s.pc = p.pc, s.ps = p.ps;
        s.cell = (uint32_t)p.cell, s.moves = p.moves, s.k = rng.k;
        s.meta = meta_pack(p.type, -1, 0u, Q_MOVE);
        s.id_lo = rng.id_lo, s.id_hi = rng.id_hi, s.catches = 0u, s.spare = 0u;
      }
      report(fresh, 0, p, hid);   // GEN
      tally_n(kEv + R3D_EV_GENERATED, take);
      q_push(ctl, ring(Q_MOVE), rmask, Q_MOVE, lane, fresh, id);
      q_push(ctl, ring(Q_FREE), rmask, Q_FREE, lane, act && !fresh, id);
    } else if (q == Q_MOVE) {
      // ---- termination checks, boundary search, free-path draw, advance (phonons.cpp:549-623) ----
      Phonon p;
      Rng rng;
      uint32_t meta = 0;
      int fate = FATE_ALIVE, reason = 0;
      LaneStats st = {0, 0, 0, 0, 0, 0, 0};
      Pending ev;
      ev.vel = 0.0, ev.face = -1, ev.flags = 0u;
      uint64_t hid = 0;
      if (act) {
        load_phonon(id, p, rng, meta);
        hid = ((uint64_t)rng.id_hi << 32) | rng.id_lo;
        fate = step_move<KIND>(a, T, p, rng, st, &reason, ev);
      }
      tally(st.iterations != 0, kEv + R3D_EV_ITERATIONS);
      const bool moved = act && fate == FATE_ALIVE;
      int dest = -1;   // where the slot goes; -1: the history ended
      if (moved) {
        if (ev.face < 0) dest = Q_SCATTER;
        else if (ev.flags & F_COLLECT) dest = Q_COLLECT;
        else dest = face_dest(ev.flags);
        if (dest < 0) fate = FATE_LOST;   // phonons.cpp:675
      }
      const bool handover = moved && dest == Q_MOVE;   // nothing to do on this face but change cells
      if (handover) p.cell = cell_neighbor(T.cells[p.cell], ev.face);
      tally(handover, kEv + R3D_EV_TRANSFER);
      report(handover, 4, p, hid);   // CEL
      const bool died = act && dest < 0;
      if (act && !died) {
        Slot& s = pool[id];
        s.t = p.t, s.path = p.path, s.recent = p.recent, s.amp = p.amp;
        s.loc[0] = p.loc.x, s.loc[1] = p.loc.y, s.loc[2] = p.loc.z;
        s.dir[0] = p.dir.x, s.dir[1] = p.dir.y, s.dir[2] = p.dir.z;
        s.cell = (uint32_t)p.cell, s.moves = p.moves, s.k = rng.k;
        s.meta = meta_pack(p.type, handover ? -1 : ev.face, handover ? 0u : ev.flags, dest);
      }
      q_push(ctl, ring(Q_SCATTER), rmask, Q_SCATTER, lane, dest == Q_SCATTER, id);
      q_push(ctl, ring(Q_COLLECT), rmask, Q_COLLECT, lane, dest == Q_COLLECT, id);
      q_push(ctl, ring(Q_RT), rmask, Q_RT, lane, dest == Q_RT, id);
      q_push(ctl, ring(Q_BEND), rmask, Q_BEND, lane, dest == Q_BEND, id);
      q_push(ctl, ring(Q_MOVE), rmask, Q_MOVE, lane, dest == Q_MOVE, id);
      finish(died, fate, reason, p, hid, TRACE && died ? pool[id].catches : 0u, id);
    } else if (q == Q_COLLECT) {
      // ---- arrival at a collection face: the receivers, with the incident state
      //      (phonons.cpp:629-631), then on to what the face itself asks for ----
      Phonon p;
      Rng rng;
      uint32_t meta = 0;
      uint64_t hid = 0;
      uint32_t k0 = 0, k1 = 0, catches = 0;
      double vel = 1.0;
      int face = 0;
      if (act) {
        load_phonon(id, p, rng, meta);
        hid = ((uint64_t)rng.id_hi << 32) | rng.id_lo;
        face = (int)((meta >> 1) & 7u) - 1;
        vel = cell_velocity(T.cells[p.cell], p.loc, p.type);
        const SeisGrid& g = a.grid;
        const double fx = (p.loc.x - g.origin[0]) * g.inv_h;
        const double fy = (p.loc.y - g.origin[1]) * g.inv_h;
        const double fz = (p.loc.z - g.origin[2]) * g.inv_h;
        if (fx >= 0 && fy >= 0 && fz >= 0 && fx < g.dim_f[0] && fy < g.dim_f[1] && fz < g.dim_f[2]) {
          const int cellid = ((int)fz * g.dim[1] + (int)fy) * g.dim[0] + (int)fx;
          if (LDS_SEIS) k0 = lds_gstart[cellid], k1 = lds_gstart[cellid + 1];
          else k0 = g.start[cellid], k1 = g.start[cellid + 1];
        }
      }
#ifdef R3D_ABLATE_COLLECT  // timing-only developer build
      k1 = k0;
#endif
      report(act, 3, p, hid);   // COL: the incident state
      tally_n(kEv + R3D_EV_COLLECT, k);
      if (__any(k1 > k0)) {
        const uint32_t hits = pool_collect_pairs<KIND, TRACE>(a, T, p, vel, k0, k1, LDS_SEIS ? lds_gitems : nullptr,
                                                              lane, catches, bc);
        tally_n(kEv + R3D_EV_CATCH, hits);
      }
      const uint32_t fl = (meta >> 8) & 0xFFu;
      const int dest = act ? face_dest(fl) : Q_NUM;
      const bool handover = act && dest == Q_MOVE;
      if (handover) p.cell = cell_neighbor(T.cells[p.cell], face);
      tally(handover, kEv + R3D_EV_TRANSFER);
      report(handover, 4, p, hid);   // CEL
      const bool died = act && dest < 0;
      if (act && !died) {
        Slot& s = pool[id];
        if (handover) s.cell = (uint32_t)p.cell;
        s.meta = meta_pack(p.type, handover ? -1 : face, handover ? 0u : fl, dest);
        if (TRACE) s.catches += catches;
      }
      q_push(ctl, ring(Q_RT), rmask, Q_RT, lane, dest == Q_RT, id);
      q_push(ctl, ring(Q_BEND), rmask, Q_BEND, lane, dest == Q_BEND, id);
      q_push(ctl, ring(Q_MOVE), rmask, Q_MOVE, lane, dest == Q_MOVE, id);
      finish(died, FATE_LOST, 0, p, hid, TRACE && died ? pool[id].catches + catches : 0u, id);
    } else {
      // ---- RT: reflection / transmission solve; BEND: Snell bend or hand-over after the
      //      run-time velocity-step test; SCATTER: deflection drawn from the scatterer's tables
      //      (phonons.cpp:611-618, :640-661) ----
      Phonon p;
      Rng rng;
      uint32_t meta = 0;
      uint64_t hid = 0;
      LaneStats st = {0, 0, 0, 0, 0, 0, 0};
      if (act) {
        load_phonon(id, p, rng, meta);
        hid = ((uint64_t)rng.id_hi << 32) | rng.id_lo;
        Pending ev;
        ev.vel = 0.0, ev.face = (int)((meta >> 1) & 7u) - 1, ev.flags = (meta >> 8) & 0xFFu;
        if (q == Q_RT) step_event<KIND, EV_RT>(a, T, p, rng, st, ev);
        else if (q == Q_BEND) step_event<KIND, EV_BEND>(a, T, p, rng, st, ev);
        else step_event<KIND, EV_SCATTER>(a, T, p, rng, st, ev);
        Slot& s = pool[id];
        s.dir[0] = p.dir.x, s.dir[1] = p.dir.y, s.dir[2] = p.dir.z;
        s.pc = p.pc, s.ps = p.ps;
        s.cell = (uint32_t)p.cell, s.k = rng.k;
        s.meta = meta_pack(p.type, -1, 0u, Q_MOVE);
      }
      if (q == Q_SCATTER) {
        tally_n(kEv + R3D_EV_SCATTER, k);
        report(act, 1, p, hid);   // SCT
      } else {
        if (q == Q_RT) tally_n(kEv + R3D_EV_RTSOLVE, k);
        tally(st.reflect != 0, kEv + R3D_EV_REFLECT);
        tally(st.transfer != 0, kEv + R3D_EV_TRANSFER);
        report(st.reflect != 0, 2, p, hid);    // REF
        report(st.transfer != 0, 4, p, hid);   // CEL
      }
      q_push(ctl, ring(Q_MOVE), rmask, Q_MOVE, lane, act, id);
    }
#ifdef R3D_PHASE_TIMING
    if (lane == 0) {
      atomicAdd(&s_stats[0][q], 1ull);
      atomicAdd(&s_stats[1][q], (unsigned long long)k);
      atomicAdd(&s_stats[2][q], __builtin_readcyclecounter() - t_begin);
    }
#endif
  }

  __syncthreads();
  if (a.carry_out) {   // park the pool for the engine's next launch
    Slot* img = reinterpret_cast<Slot*>(a.carry_out) + (size_t)blockIdx.x * S;
    for (uint32_t s = tid; s < S; s += kPoolBlock) img[s] = pool[s];
  }
  // ---- the block's bin accumulators and tallies to HBM ----
  if (bc.on) {
    for (uint32_t i = tid; i <= bc.mask; i += kPoolBlock) {
      const uint32_t bin = bc.key[i];
      if (bin == kEmpty) continue;
      double* e = a.energy + (size_t)bin * 5;
#pragma unroll
      for (int cc = 0; cc < 5; cc++) {
        const double v = bc.e[i * 5u + cc];
        if (v != 0.0) unsafeAtomicAdd(e + cc, v);
      }
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const uint32_t n = bc.cnt[i * 2u + t];
        if (n) atomicAdd(a.counts + (size_t)bin * 2 + t, (unsigned long long)n);
      }
    }
  }
#ifdef R3D_PHASE_TIMING
  if (tid < 24) atomicAdd(&g_pool_stats[tid / 8][tid % 8], s_stats[tid / 8][tid % 8]);
#endif
  if (tid < R3D_N_SCALARS && s_tally[tid] != 0ull) atomicAdd(a.scalars + tid, s_tally[tid]);
}

// The traversal kernel, and the same body under a second name for the flush launch of a carry
// chain (no new ids, only the histories carried over), so that profiles list the two apart.
template <int KIND, bool LDS_CELLS, bool LDS_SCAT, bool TRACE>
__global__ __launch_bounds__(kPoolBlock) void pool_kernel(const KArgs a) {
  pool_body<KIND, LDS_CELLS, LDS_SCAT, TRACE>(a);
}
template <int KIND, bool LDS_CELLS, bool LDS_SCAT>
__global__ __launch_bounds__(kPoolBlock) void pool_drain_kernel(const KArgs a) {
  pool_body<KIND, LDS_CELLS, LDS_SCAT, false>(a);
}

}  // namespace r3d
#endif
