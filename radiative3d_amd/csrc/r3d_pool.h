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
//   * every slot is, between phases, in exactly one of five queues (rings of slot numbers in
//     LDS): MOVE, COLLECT, RT, SCATTER, FREE;
//   * a wave takes up to 64 slots OF ONE QUEUE, loads their state into registers, runs that
//     phase's code for all of them at once, stores what changed and hands each slot to the queue
//     of its next phase (one merged hand-off for all destinations).  With S above the workgroup's
//     lane count the queues are deep enough that nearly every batch is a full one, so each phase
//     runs at (close to) 64 lanes;
//   * phases:  FREE -> (claim ids from the global counter, source spray) -> MOVE
//              MOVE -> (termination checks, boundary search, free-path draw, advance; a plain
//                      hand-over or a Snell bend is served inline, and lanes that can simply move
//                      again do so in registers: up to kPoolMoves moves while >= kMoveAgainLanes
//                      lanes go on) -> SCATTER | COLLECT | RT | MOVE | FREE (history ended)
//              COLLECT -> (receiver hash, (arrival, receiver) pairs dealt over the lanes, bins) ->
//                      RT | MOVE | FREE
//              RT, SCATTER -> MOVE
//   * histories are keyed by id and draw from per-history counters (r3d_rng.h), and every phase
//     is the same per-history code as the host emulation runs (r3d_step.h), so which wave runs
//     which phase of a history, and in what order histories are served, changes no result.
//
// Queue protocol (multi-producer, multi-consumer among the waves of one workgroup, LDS only):
// a ring of 16-bit slot numbers; the head ticket and the count of published entries share one
// control word, the tail ticket has its own.  A consumer takes min(count, 64) entries with ONE
// compare-and-swap on the control word it has just read (head advanced, count reduced), then
// spins on each of its entries until it is no longer EMPTY (the entry may belong to a producer
// that has its ticket but has not written yet) and carries its own ticket's lap (below), reads it
// and marks it EMPTY.  A producer takes
// tail tickets -- for all destination queues with one LDS atomic instruction, lane q serving
// queue q -- waits until its entries are EMPTY (a consumer holding the ticket of the previous lap
// may not have read yet), writes them, then adds to the count.  A slot's state is written before
// its number is published and read after it is taken (LDS operations of a wave complete in
// order; release / acquire fences at workgroup scope keep the compiler from moving them).
#ifndef R3D_POOL_H_
#define R3D_POOL_H_

#include "r3d_kernels.h"
#include "r3d_wave.h"

namespace r3d {

// Wave priority: raised while a wave is between batches (hand-off, scheduling, take) -- serial LDS
// round trips during which it holds slots other waves may be polling for -- and lowered for the
// phase's arithmetic: from the hand-off until the next batch's state is in registers.
#define R3D_PRIO_HIGH() __builtin_amdgcn_s_setprio(2)
#define R3D_PRIO_LOW() __builtin_amdgcn_s_setprio(0)
#define R3D_PRIO_MOVE() __builtin_amdgcn_s_setprio(1)
constexpr int kPoolWaves = kPoolBlock / 64;

constexpr uint16_t kRingEmpty = 0xFFFFu;
// The rings are polled with volatile accesses.  Through a plain (generic) pointer those compile to
// FLAT instructions with system-scope cache bits and a wait for every outstanding memory operation
// of the wave -- a thousand cycles per take and per hand-off, measured; through a pointer in the
// LDS address space they are the ds_read_u16 / ds_write_b16 they should be.
typedef __attribute__((address_space(3))) uint16_t lds_u16;


// Queue control: word[q] = head ticket << 16 | published entries (a consumer moves both with one
// compare-and-swap), tail[q] = next ticket for producers; word[kDrainedWord] = the global id
// counter ran out.
constexpr int kDrainedWord = 6;
struct PoolCtl {
  uint32_t word[8];
  uint32_t tail[8];
};

__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t meta_pack(int type, int face, uint32_t flags, int queue) {
  return (uint32_t)type | ((uint32_t)(face + 1) << 1) | ((flags & 0xFFu) << 8) | ((uint32_t)queue << 16);
}

// A ring entry: the slot number in its low 11 bits (S <= 1984), above it the low 5 bits of the LAP of
// the ticket it was written for.  A consumer accepts an entry only with its own ticket's lap on it:
// should a wave ever stall between taking its tickets and reading its entries for as long as the
// other waves need to push a whole ring's worth (tens of thousands of cycles -- a pre-empted queue),
// the consumer a lap later waits instead of reading the stalled wave's entry.
constexpr uint32_t kSlotBits = 11;
__device__ __forceinline__ uint32_t lap_of(uint32_t ticket, uint32_t log2cap) { return (ticket >> log2cap) & 31u; }

// Take up to 64 slot numbers from queue q, whose control word was last seen as `seen`; returns how
// many (wave-uniform); lane l < k gets its slot in `id`.
__device__ __forceinline__ unsigned q_pop(PoolCtl& ctl, lds_u16* ring, uint32_t mask, uint32_t log2cap, int q,
                                          unsigned lane, uint32_t seen, unsigned& id) {
  // (everything but the compare-and-swap itself is wave-uniform and written so: scalar instructions and
  //  scalar branches, where the same loop under "lane 0 only" ran on the vector unit behind exec masks)
  unsigned k = 0, pos = 0;
  {
    uint32_t w = seen;
    // (never fewer than the scheduler counted on: when another wave was quicker and left a
    //  remainder, taking that handful would make a batch of a few lanes -- look again instead)
    const uint32_t need = (seen & 0xFFFFu) < 64u ? (seen & 0xFFFFu) : 64u;
    while ((w & 0xFFFFu) >= need && (w & 0xFFFFu)) {
      const uint32_t c = w & 0xFFFFu, t = c < 64u ? c : 64u;
      const uint32_t next = ((w + (t << 16)) & 0xFFFF0000u) | (c - t);
      uint32_t was = 0;
      if (lane == 0) was = atomicCAS(&ctl.word[q], w, next);
      was = (uint32_t)__builtin_amdgcn_readfirstlane((int)was);
      if (was == w) {
        k = t, pos = w >> 16;
        break;
      }
      w = was;
    }
  }
  id = 0;
  if (lane < k) {
    volatile lds_u16* e = ring + ((pos + lane) & mask);
    const uint32_t want = lap_of(pos + lane, log2cap);
    uint16_t v;
    do {
      v = *e;
    } while (v == kRingEmpty || (uint32_t)(v >> kSlotBits) != want);
    id = v & ((1u << kSlotBits) - 1u);
  }
  R3D_LDS_ACQUIRE();
  // (the entry is marked empty after the fence: nothing waits for this store, and the slot's state is
  //  requested right behind it)
  if (lane < k) ring[(pos + lane) & mask] = kRingEmpty;
  return k;
}

// Hand every active lane's slot to the queue named in its `dest` (all queues in one go: lane q
// takes the tail tickets of queue q and publishes its count, so the whole distribution costs one
// round of LDS atomics each way).
__device__ __forceinline__ void q_push_all(PoolCtl& ctl, lds_u16* rings, uint32_t rcap, uint32_t log2cap,
                                           unsigned lane, bool act, int dest_, unsigned id) {
  const int dest = act ? dest_ : -1;   // (no queue: the compares below then ARE the lane masks)
  unsigned long long m[Q_NUM];
  uint32_t kq = 0, rank = 0;
#pragma unroll
  for (int q = 0; q < Q_NUM; q++) {
    m[q] = ballot(dest == q);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(kq) : "s"((uint32_t)__popcll(m[q])), "n"(q));   // lane q: entries for queue q
    rank = (dest == q) ? rank_in(m[q]) : rank;
  }
  // the tail tickets are asked for first; their round trip also sees the slots' state written (this
  // wave's LDS accesses return in order), and the release fence sits where it is needed: between
  // the state and the slot numbers
  uint32_t pos = 0;
  if (lane < Q_NUM && kq) pos = atomicAdd(&ctl.tail[lane], kq);
  R3D_LDS_RELEASE();
  // (each lane's queue's ticket through scalar registers: a cross-lane fetch would be one more LDS
  //  round trip in a chain of them)
  uint32_t t = rank;
#pragma unroll
  for (int q = 0; q < Q_NUM; q++) {
    const uint32_t pq = (uint32_t)__builtin_amdgcn_readlane((int)pos, q);
    t = (dest == q) ? rank + pq : t;
  }
  if (act) {
    volatile lds_u16* e = rings + (((uint32_t)dest << log2cap) | (t & (rcap - 1u)));   // (rcap = 1 << log2cap)
    while (*e != kRingEmpty) {
    }
    *e = (uint16_t)((lap_of(t, log2cap) << kSlotBits) | id);
  }
  R3D_LDS_RELEASE();
  if (lane < Q_NUM && kq) atomicAdd(&ctl.word[lane], kq);
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
  const uint32_t incl = wave_scan_add(cnt);   // inclusive prefix sum over the wave (r3d_wave.h: no LDS round trips)
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  const uint32_t excl = incl - cnt;
  V3 dopm = p.dir;
  if (cnt && p.type != RAY_P) {
    V3 th, ph;
    sph_basis(p.dir, th, ph);
    dopm = p.pc * th + p.ps * ph;
  }
  const double amp2 = amplitude2(p);
  const double inv_vel = 1.0 / vel_lane;
  uint32_t n_hits = 0;
  for (uint32_t base = 0; base < total; base += 64u) {
    const uint32_t j = base + lane;
    uint32_t src = 0;   // smallest lane whose inclusive sum exceeds j
    {
      // Who owns pair j?  Every arrival whose range STARTS inside this pass's window of 64 pairs sends its lane number
      // to the lane of the pair it starts at (one forward permute; every such lane has its own target: arrivals without
      // candidates do not take part); all other lanes send the owner of the window's first pair, src0, to lane 0 -- the
      // same value from all of them --; a running maximum over the lanes then gives every pair its owner (owners ascend
      // with the pairs).  One crossbar round trip and six data-parallel instructions where the bisection over the
      // prefix sums took six dependent round trips.
      const uint32_t src0 = (uint32_t)__popcll(ballot(incl <= base));   // first lane whose inclusive sum exceeds `base`
      const int pos = (int)excl - (int)base;
      const bool starts = cnt != 0u && pos > 0 && pos < 64;
      const uint32_t mark = (uint32_t)__builtin_amdgcn_ds_permute((starts ? pos : 0) << 2, (int)((starts ? lane : src0) + 1u));
      src = wave_scan_max(mark) - 1u;
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
    const unsigned long long hm = ballot(hit);
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
// diagnostic build: per queue, batches served, lanes filled, wave cycles spent; slot 6: idle polls;
// slot 7 of rows 0 / 1: move sub-iterations run / lanes live in them
static __device__ unsigned long long g_pool_stats[5][8];   // rows 3 / 4: cycles in the take / in the hand-off
#endif

// Moves a batch may make before its slots go back to the queues: lanes whose move ends with nothing
// to do but change cells (or bend) stay in registers and move again while at least kMoveAgainLanes
// of them do, so the pool round trip is paid once per several moves.
constexpr int kPoolMoves = 4;
constexpr int kPoolMovesThin = 32;   // (bounded, so that a drained launch with carry-over still parks promptly)
constexpr unsigned kMoveAgainLanes = 44;
// The drain of a launch that finishes its own stragglers (the ids are out, ever fewer histories are
// alive): a wave that has served a THIN batch -- no more than kTailBatch slots -- KEEPS those slots
// for as long as they all want the same next phase, and serves that phase itself: no hand-off, no
// take, nobody to wait for.  What a drain waits for are the longest histories' serial chains, and
// alone on the chip a history's queue round trips were 15 % (tetra) to 33 % (layered: every move of a
// reverberating phonon ends in a reflection) of its chain (tools/lone_history_stats.py).  Kept lanes
// that want DIFFERENT phases would be served one phase after the other, where the queues hand them to
// different waves at once (whole batches kept whatever they want: LopNor's flush 11.1 -> 14.1 ms): a
// batch that disagrees goes back to the queues, which also regroup the stragglers.  No more than
// kPoolWaves - kTailServers waves keep lanes at a time, so that what waits in a queue is always served.
constexpr uint32_t kTailBatch = 16, kTailServers = 2;
constexpr int kKeepersWord = 7;   // PoolCtl::word[kKeepersWord]: waves that keep lanes at the moment
// Developer builds (make variant DEFS=-DR3D_STEP_FINALS=1): the final record's code in the chain-step kernel too, which
// the shipped kernel is compiled without (3.5 % of its launch).  tests/test_gpu_parity.py holds that sibling compilation
// per history against the oracle.
#ifndef R3D_STEP_FINALS
#define R3D_STEP_FINALS 0
#endif
// Layered models: a full MOVE batch of which at least this many lanes end on an interface that wants the reflection /
// transmission solve serves it itself, on the state it holds (the MOVE phase below); a thin batch: when at least half
// of its histories do.  A lone LopNor history 4.27 -> 3.57 us per move, flush 9.3 -> 8.0 ms, step launch 6.60 -> 6.38.
// The same for the thin batches of a tetra model's drain (kernels with a tail: a lone NSCP history 2.91 -> 2.72 us per
// move, flush -4 %); not the shell kernel (a spherical model's flush is a twentieth of its step and gains nothing, and
// the solve's registers beside the move's cost its self-contained launch 3 %).  profiles/r06/inline_rt_ab.log.
constexpr unsigned kInlineRtFullBatch = 40;
// A refill starts kRefillBatches batches of histories at a time, their table fetches set going together (the
// FREE phase below); the FREE queue counts as full for the scheduler at kRefillFull entries.
constexpr int kRefillBatches = 2;
constexpr uint32_t kRefillFull = 64 * kRefillBatches;
// Entries a minor phase's queue must hold before a wave goes for it.
constexpr uint32_t kMinorFull = 64;

// LDS_CELLS / LDS_SCAT: the cell records / the scatterer heads are staged in LDS (models with a few
// dozen cells; all but models with thousands of scatterers).  The receiver tables are read through
// L1 / L2 (a collection phase serves 64 arrivals at once, so their latency is paid per batch).
// TAIL: the launch drains its own stragglers (no carry-over to a next launch), and its last stretch keeps
// lanes (kTailBatch above); the step launches of a carry chain are compiled without that code.
template <int KIND, bool LDS_CELLS, bool LDS_SCAT, bool TRACE, bool TAIL>
__device__ __forceinline__ void pool_body(const KArgs& a) {
  using Cell = typename CellOf<KIND>::type;
  constexpr bool LDS_SEIS = false;
  // refills in pairs of batches (the FREE phase): not in the tetra kernel, whose boundary search leaves the
  // register allocator no slack at all -- the pair's second queue take alone had it spill two registers in the
  // move, and 64 slots waiting for their partners are a smaller pool (NSCP +3 % with it, half-space -10 %)
  constexpr bool kPairs = KIND != CELL_TET;
  extern __shared__ __align__(16) unsigned char smem[];
  const unsigned tid = threadIdx.x, lane = tid & 63u;

  // ---- stage the small tables in LDS ----
  {
    auto copy_words = [&](void* dst, const void* src, size_t bytes) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(dst);
      const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
      for (size_t i = tid; i < bytes / 8; i += kPoolBlock) d[i] = s[i];
    };
    if (LDS_CELLS) copy_words(smem + a.lds_cells_off, a.cells, (size_t)2 * a.n_cells * sizeof(Cell));   // (a record per cell and ray type)
    if (LDS_SCAT) {
      copy_words(smem + a.lds_scat_off, a.scat_head, (size_t)a.n_scat * sizeof(ScatHead));
      copy_words(smem + a.lds_scatptr_off, a.scat_ptrs, (size_t)a.n_scat * sizeof(ScatPtrs));
    }
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
  // (the table addresses too: a scatter then starts its guide fetch after an LDS read instead of a
  //  round trip to L2 for the pointer)
  T.scat_ptrs = LDS_SCAT ? reinterpret_cast<const ScatPtrs*>(smem + a.lds_scatptr_off) : a.scat_ptrs;
  T.seis_scan = LDS_SEIS ? reinterpret_cast<const SeisScan*>(smem + a.lds_seis_off) : a.seis_scan;
  T.seis_hit = a.seis_hit;   // fetched on a hit only: stays in HBM / L2

  // ---- the pool, its queues, the block's tallies ----
  const uint32_t S = a.pool_slots, rcap = a.pool_ring_mask + 1u, rmask = a.pool_ring_mask;
  const uint32_t rlog = 31u - (uint32_t)__builtin_clz(rcap);   // (a power of two)
  // A slot's state is eight 16-byte records -- six pairs of doubles, two quadruples of words --, each
  // kind of record an array of kSlotStride entries whatever S is (record-major: a batch's 64 random
  // slots then spread over the banks; slot-major they would all start on the same ones).  One
  // ds_read_b128 fetches a record: half the LDS instructions of 8-byte fields and, at 16 lanes a
  // bank group, fewer array cycles under the conflicts random slot numbers bring (two-address LDS
  // instructions take twice the array cycles of two plain ones: MI355X_MICROARCH, LDS table).
  constexpr uint32_t F = kSlotStride;
  struct alignas(16) D2 { double a, b; };
  struct alignas(16) U4 { uint32_t a, b, c, d; };
  enum { P_T_PATH, P_RECENT_LAMP, P_LX_LY, P_LZ_DX, P_DY_DZ, P_PC_PS, P_NUM };   // pairs of doubles
  enum { U_STATE /* cell, moves, draws, meta */, U_ID /* id lo, id hi, catches, pending neighbour */, U_NUM };
  static_assert((P_NUM + U_NUM) * 16 == kSlotBytes, "slot layout");
  D2* const pd = reinterpret_cast<D2*>(smem + a.lds_pool_off);   // [P_NUM][F]
  U4* const pu = reinterpret_cast<U4*>(pd + (size_t)P_NUM * F);   // [U_NUM][F]
  lds_u16* const rings = (lds_u16*)(smem + a.lds_ring_off);                      // [Q_NUM][rcap]
  __shared__ PoolCtl ctl;
  __shared__ unsigned long long s_tally[R3D_N_SCALARS];
  for (uint32_t i = tid; i < Q_NUM * rcap; i += kPoolBlock) rings[i] = kRingEmpty;
  if (tid < R3D_N_SCALARS) s_tally[tid] = 0ull;
  if (tid < 8) ctl.word[tid] = 0u, ctl.tail[tid] = 0u;
  __syncthreads();
  auto ring = [&](int q) { return rings + (uint32_t)q * rcap; };
  const size_t image_words = (size_t)F * (kSlotBytes / 4);   // the pool's state as 32-bit words
  if (a.carry_in) {
    // resume: the pool image this workgroup parked at the end of the engine's previous launch;
    // every slot goes back to the queue named in its meta word
    const uint32_t* img = reinterpret_cast<const uint32_t*>(a.carry_in) + (size_t)blockIdx.x * image_words;
    uint32_t* dst = reinterpret_cast<uint32_t*>(pd);
    for (size_t i = tid; i < image_words; i += kPoolBlock) dst[i] = img[i];
    __syncthreads();
    for (uint32_t base = 0; base < S; base += kPoolBlock) {
      const uint32_t s = base + tid;
      const bool have = s < S;
      const int tag = have ? (int)((pu[U_STATE * F + s].d >> 16) & 7u) : 0;
      q_push_all(ctl, rings, rcap, rlog, lane, have, tag, s);
    }
  } else {
    for (uint32_t s = tid; s < S; s += kPoolBlock) {
      ring(Q_FREE)[s] = (uint16_t)s;
      pu[U_STATE * F + s].d = meta_pack(0, -1, 0u, Q_FREE);
    }
    if (tid == 0) ctl.tail[Q_FREE] = S, ctl.word[Q_FREE] = S;
  }
  __syncthreads();

  constexpr int kEv = 3 + R3D_INV_NUM;
  // Tallies: the per-lane event counters of r3d_step.h (LaneStats) start from zero for every move or
  // face event, so "this lane's counter moved" is a lane mask the compiler already holds (the
  // condition the increment sat under); the masks are counted with scalar instructions and the
  // batch's sums go to the block's LDS tallies once, from lane 0.  (Counters that ran on in registers
  // for the whole launch were ten registers that every phase had to carry -- or spill -- around its
  // peak; per-batch counters summed by a wave butterfly were ~40 vector instructions a batch.)
  // (rare ones -- invalid histories -- go one by one; the rest as ONE LDS instruction per batch, below)
  auto tally_n = [&](int slot, unsigned long long n) {
    if (lane == 0 && n) atomicAdd(&s_tally[slot], n);
  };
  auto tally = [&](bool cond, int slot) { tally_n(slot, (unsigned long long)__popcll(ballot(cond))); };
  auto count = [&](bool cond) { return (uint32_t)__popcll(ballot(cond)); };
  // Report stream (diagnostic kernel only; include/r3d.h r3d_event): the lanes for which `cond`
  // holds append one record each; the wave claims the slots with one atomic.
  auto report = [&](bool cond, int tag, const Phonon& q, uint64_t hid) {
    if (!TRACE || !a.evlog || !((a.evlog_mask >> tag) & 1u)) return;
    const unsigned long long m = ballot(cond);
    if (!m) return;
    const int first = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if ((int)lane == first) base = atomicAdd(a.evlog_count, (unsigned long long)__popcll(m));
    base = __shfl(base, first);
    const unsigned long long at = base + (unsigned long long)rank_in(m);
    if (cond && at < a.evlog_cap) {
      r3d_event* r = reinterpret_cast<r3d_event*>(a.evlog) + at;
      r->id = hid;
      r->time = q.t, r->path = q.path, r->amp = amplitude(q);
      r->loc[0] = q.loc.x, r->loc[1] = q.loc.y, r->loc[2] = q.loc.z;
      r->dir[0] = q.dir.x, r->dir[1] = q.dir.y, r->dir[2] = q.dir.z;
      r->cell = (uint32_t)q.cell, r->moves = q.moves;
      r->tag = (uint8_t)tag, r->type = (uint8_t)q.type;
    }
  };
  // A history ended (lanes with `died`): loss counters, report line, final record.
  // A batch's counts go to the block's tallies as ONE LDS instruction where the batch ends (lane j adds
  // to tally j), in the layered and the spherical kernel; the tetra kernel, with no vector register to
  // spare for that, adds them one by one where they arise (the vector form there: +0.8 %).
  constexpr bool kVectorTally = KIND != CELL_TET;
  uint32_t n_lost = 0, n_timeout = 0;   // (per batch, like the event counts: reset where a batch starts)
  auto finish = [&](bool died, int fate, int reason, const Phonon& p, uint64_t hid, uint32_t catches, unsigned slot) {
    if (!any_lane(died)) return;
    if constexpr (kVectorTally) {
      n_lost += count(died && fate == FATE_LOST), n_timeout += count(died && fate == FATE_TIMEOUT);
    } else {
      tally(died && fate == FATE_LOST, 0);
      tally(died && fate == FATE_TIMEOUT, 1);
    }
    if (any_lane(died && fate == FATE_INVALID)) {   // rare
      tally(died && fate == FATE_INVALID, 2);
#pragma unroll
      for (int r = 0; r < R3D_INV_NUM; r++) tally(died && fate == FATE_INVALID && reason == r, 3 + r);
    }
    if (!TRACE && (TAIL || R3D_STEP_FINALS)) {
      // (the production kernels' own witness, when a buffer is attached: everything but the catch count, which
      //  only the diagnostic kernel keeps per history.  In the self-contained and the drain kernel; compiled into
      //  the chain-step kernel too it cost the NSCP launch 3.5 % -- four to six registers spilled -- with no
      //  buffer attached, so that kernel's histories are held through the bins and counters they leave)
      if (a.pfinals) {
        if (died) {
          // (the id from the slot, not from `hid`: nothing but this would keep that pair of registers alive
          //  through the move)
          const U4 ids = pu[U_ID * F + slot];
          r3d_final* f = reinterpret_cast<r3d_final*>(a.pfinals) + (((uint64_t)ids.b << 32) | ids.a);
          // (the amplitude as its logarithm, which is what the kernel carries: r3d_production_finals_read
          //  exponentiates on the host -- an exponential here is thirty instructions under full register load)
          f->time = p.t, f->path = p.path, f->amp = p.lamp;
          f->loc[0] = p.loc.x, f->loc[1] = p.loc.y, f->loc[2] = p.loc.z;
          f->dir[0] = p.dir.x, f->dir[1] = p.dir.y, f->dir[2] = p.dir.z;
          f->moves = p.moves;
          f->fate = (uint8_t)fate;
          f->type = (uint8_t)p.type;
          f->n_catch = (uint16_t)0xFFFFu;
        }
      }
    }
    if (TRACE) {
      report(died && fate == FATE_LOST, 5, p, hid);
      report(died && fate == FATE_TIMEOUT, 6, p, hid);
      report(died && fate == FATE_INVALID, 7, p, hid);
      if (died && a.finals) {   // (the diagnostic kernel also runs for the report stream alone)
        r3d_final* f = reinterpret_cast<r3d_final*>(a.finals) + (hid - a.first_id);
        f->time = p.t, f->path = p.path, f->amp = amplitude(p);
        f->loc[0] = p.loc.x, f->loc[1] = p.loc.y, f->loc[2] = p.loc.z;
        f->dir[0] = p.dir.x, f->dir[1] = p.dir.y, f->dir[2] = p.dir.z;
        f->moves = p.moves;
        f->fate = (uint8_t)fate;
        f->type = (uint8_t)p.type;
        f->n_catch = (uint16_t)(catches > 65535u ? 65535u : catches);
      }
    }
  };
  // (nbr: the cell behind the pending face, left by the move for the phase that serves the face)
  auto load_state = [&](unsigned id, Phonon& p, Rng& rng, uint32_t& meta, uint32_t& nbr) {
    const D2* d = pd + id;
    const D2 q0 = d[P_T_PATH * F], q1 = d[P_RECENT_LAMP * F], q2 = d[P_LX_LY * F], q3 = d[P_LZ_DX * F],
             q4 = d[P_DY_DZ * F], q5 = d[P_PC_PS * F];
    const U4 u0 = pu[U_STATE * F + id], u1 = pu[U_ID * F + id];
    p.t = q0.a, p.path = q0.b, p.recent = q1.a, p.lamp = q1.b;
    p.loc = v3(q2.a, q2.b, q3.a);
    p.dir = v3(q3.b, q4.a, q4.b);
    p.pc = q5.a, p.ps = q5.b;
    p.cell = (int32_t)u0.a, p.moves = u0.b, rng.k = u0.c, meta = u0.d;
    p.type = (int32_t)(meta & 1u);
    rng.id_lo = u1.a, rng.id_hi = u1.b, nbr = u1.d;
  };
  // what a face event or a scattering changes: direction, polarisation, cell, draws, meta
  auto store_event = [&](unsigned id, const Phonon& p, const Rng& rng, uint32_t meta) {
    D2* d = pd + id;
    d[P_LZ_DX * F].b = p.dir.x;
    d[P_DY_DZ * F] = D2{p.dir.y, p.dir.z}, d[P_PC_PS * F] = D2{p.pc, p.ps};
    pu[U_STATE * F + id] = U4{(uint32_t)p.cell, p.moves, rng.k, meta};
  };
  // (the fields a move or a face event can change: everything but the history id)
  auto store_state = [&](unsigned id, const Phonon& p, const Rng& rng, uint32_t meta) {
    D2* d = pd + id;
    d[P_T_PATH * F] = D2{p.t, p.path}, d[P_RECENT_LAMP * F] = D2{p.recent, p.lamp};
    d[P_LX_LY * F] = D2{p.loc.x, p.loc.y}, d[P_LZ_DX * F] = D2{p.loc.z, p.dir.x};
    d[P_DY_DZ * F] = D2{p.dir.y, p.dir.z}, d[P_PC_PS * F] = D2{p.pc, p.ps};
    pu[U_STATE * F + id] = U4{(uint32_t)p.cell, p.moves, rng.k, meta};
  };
#ifdef R3D_PHASE_TIMING
  __shared__ unsigned long long s_stats[5][8];
  if (tid < 40) s_stats[tid / 8][tid % 8] = 0ull;
  __syncthreads();
#endif

  // The launch arguments are re-read for every batch: `args` below is the kernel-argument segment
  // behind a pointer the compiler cannot see through, so what a phase needs of the ~90 words is
  // loaded (scalar loads, scalar cache) where the phase starts and dropped where it ends.  Left to
  // itself the compiler fetches every argument once, ahead of the loop, and then holds -- or
  // spills to vector-register lanes and scratch -- all of them across it: 84 scalar and 10
  // vector registers spilled, against 15-43 and 0-4 this way.
  typedef const __attribute__((address_space(4))) KArgs* KernArgs;
  KernArgs args = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();   // (KArgs is the kernels' only parameter)
  // the drain (kTailBatch above): the lanes that keep their slot, and the phase they all want next
  constexpr bool kChainMove = true;   // a batch that goes on to its move as a whole is served by the wave that has it
  constexpr int kChainFlag = 8;   // (above the queue numbers 0 .. Q_NUM - 1)
  // the drain's kept lanes four to a history (tetra cells; not in the diagnostic kernel, whose report stream is per lane)
  constexpr bool kQuad = TAIL && !TRACE && KIND == CELL_TET;
  constexpr bool kChainCollect = kChainMove;
  // the interface solve inside the MOVE phase (layered models: the MOVE phase below)
  constexpr bool kInlineRt = KIND == CELL_CYL || (TAIL && !TRACE && KIND == CELL_TET);
  constexpr unsigned kInlineRtLanes = KIND == CELL_CYL ? kInlineRtFullBatch : 65u;   // (65: thin batches only)
  bool held = false;
  unsigned k_chain = 0;   // (wave-uniform) slots of the batch just served that all want MOVE next and stay with this wave
  bool ids_out = false;   // (wave-uniform) the id counter was seen exhausted
  // The drain's thin batches: what a launch that finishes its own stragglers waits for is the history with most of its
  // time to live still ahead (the long ones are the ones that run into the limit), and it shares its SIMD with waves whose
  // histories have less to go -- so a thin batch's arithmetic runs at a priority that falls with the age of its youngest
  // history: longest remaining chain first.  LopNor's flush 7.9 -> 7.1 ms, its self-contained 1e7 launch 14.75 -> 14.15
  // (profiles/r06/drain_priority_ab.log); a tetra model's is unchanged; not in the shell kernel (every history of a
  // whole-Earth run lives to the limit, its drain is a twentieth of its step: +1 %).
  constexpr bool kDrainPrio = TAIL && KIND != CELL_SPH;
  int drain_level = 0;    // (wave-uniform) 0, or the priority 2 / 3 the drain gave this wave's batch at its last move
  auto drain_prio = [&](double t_alive, bool thin_batch, double ttl) {   // (called by the active lanes of a MOVE batch)
    if constexpr (kDrainPrio) {
      drain_level = 0;
      if (thin_batch && ids_out) {
        const double third = ttl * (1.0 / 3.0);
        if (any_lane(t_alive < third)) drain_level = 3, __builtin_amdgcn_s_setprio(3);
        else if (any_lane(t_alive < 2.0 * third)) drain_level = 2, __builtin_amdgcn_s_setprio(2);
      }
    }
  };
  auto drain_prio_again = [&]() {   // (the phases a kept batch goes through between its moves: at its move's priority)
    if (kDrainPrio && drain_level == 3) __builtin_amdgcn_s_setprio(3);
    else if (kDrainPrio && drain_level == 2) __builtin_amdgcn_s_setprio(2);
    else R3D_PRIO_LOW();
  };
  int dest = Q_FREE;
  unsigned id = 0;
  for (;;) {
    asm volatile("" : "+s"(args));
    const KArgs& a = *(const KArgs*)args;   // (shadows the parameter: the same values, fetched afresh)
    int q;
    unsigned k;
    bool act;
#ifdef R3D_PHASE_TIMING
    unsigned long long t_pop = __builtin_readcyclecounter();
#endif
    const unsigned long long held_m = TAIL ? ballot(held) : 0ull;
    const bool was_held = held_m != 0ull;
    // (kernels with a tail: a batch chained on to its move travels as held lanes whose `dest` carries kChainFlag;
    //  it is not one of the tail's keepers)
    bool was_kept = was_held;
    if (was_held) {
      // ---- kept lanes: they all want the same phase ----
      q = __builtin_amdgcn_readlane(dest, __ffsll((long long)held_m) - 1);
      if (kChainMove) was_kept = (q & kChainFlag) == 0, q &= kChainFlag - 1;
      act = held;
      k = (unsigned)__popcll(held_m);
    } else if (kChainMove && !TAIL && k_chain) {
      // ---- the batch this wave has just served (R/T or scattering: on to its move; a collection all of whose
      //      arrivals are reflected: on to that solve), in the same lanes; k_chain = slots | queue << 8 ----
      const unsigned kc = (unsigned)__builtin_amdgcn_readfirstlane((int)k_chain);
      q = (int)(kc >> 8), k = kc & 0xFFu, act = lane < k;
      k_chain = 0;
    } else {
      // ---- choose a queue: a full batch of a minor phase first (they all feed MOVE), then a
      //      refill, then MOVE; with no full batch anywhere, the fullest queue ----
      // (lane j looks at queue j: one compare for all of them, then scalar tests of the lane mask)
      const uint32_t snap = lane < 8 ? lds_ld(&ctl.word[lane]) : 0u;
      const uint32_t drained = (uint32_t)__builtin_amdgcn_readlane((int)snap, kDrainedWord);
      if (drained && a.carry_out) break;   // no ids left: the pool is parked as it is for the next launch
      const uint32_t cnt = snap & 0xFFFFu;
      const bool minor = lane == (unsigned)Q_RT || lane == (unsigned)Q_COLLECT || lane == (unsigned)Q_SCATTER;
      // (a refill waits for two batches of free slots: it starts them together, see the FREE phase)
      uint32_t full = (uint32_t)ballot(cnt >= (minor ? kMinorFull : (kPairs && lane == (unsigned)Q_FREE) ? kRefillFull : 64u)) & ((1u << Q_NUM) - 1u);
      if (drained) {
        if ((uint32_t)__builtin_amdgcn_readlane((int)cnt, Q_FREE) == S) break;   // every slot is free and nothing is left to hand out
        full &= ~(1u << Q_FREE);                                                  // (free slots are of no use any more)
      }
      q = (full & (1u << Q_RT)) ? Q_RT : (full & (1u << Q_COLLECT)) ? Q_COLLECT : (full & (1u << Q_SCATTER)) ? Q_SCATTER
          : (full & (1u << Q_FREE)) ? Q_FREE : (full & (1u << Q_MOVE)) ? Q_MOVE : -1;
      if (q < 0) {   // no full batch anywhere: the fullest queue
        uint32_t best = 0u;
#pragma unroll
        for (int j = 0; j < Q_NUM; j++) {
          const uint32_t cj = (drained && j == Q_FREE) ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)cnt, j);
          if (cj > best) best = cj, q = j;
        }
      }
      if (q < 0) {   // everything in flight is in other waves' hands
#ifdef R3D_PHASE_TIMING
        if (lane == 0) atomicAdd(&s_stats[0][6], 1ull);
#endif
        R3D_PRIO_LOW();   // (an idle wave must not outrank the ones that work)
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      const uint32_t wq = (uint32_t)__builtin_amdgcn_readlane((int)snap, q);
      ids_out = drained != 0u;
      R3D_PRIO_HIGH();
      k = q_pop(ctl, ring(q), rmask, rlog, q, lane, wq, id);
      if (k == 0) {   // another wave was quicker
        R3D_PRIO_LOW();
        continue;
      }
      act = lane < k;
    }
    const bool thin = k < 64u;
#ifdef R3D_PHASE_TIMING
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
    // where each active lane's slot goes after this phase (kept lanes outside this batch: as it stands)
    if constexpr (TAIL) dest = act ? Q_FREE : dest;
    else dest = Q_FREE;
    // the batch's event counts (wave-uniform; scalar registers), added to the block's tallies together below
    uint32_t n_iter = 0, n_transfer = 0, n_reflect = 0, n_generated = 0, n_collect = 0, n_catch = 0, n_rtsolve = 0;
    uint32_t n_volout = 0;   // (video runs: events outside the attached grid)
    n_lost = 0, n_timeout = 0;

    if (q == Q_FREE) R3D_PRIO_LOW();
    if (q == Q_FREE) {
      // ---- fresh histories: ids from the global counter, source spray (events.cpp:111-124) ----
      // A refill starts TWO batches of histories at a time (kRefillBatches; the scheduler lets the FREE queue
      // fill that far before a wave goes for it).  A spray is two dependent
      // fetches from tables of hundreds of MB (the guide cell of the take-off draw, then the direction
      // record) with a few dozen instructions between them: a refill spends its time waiting for them, and
      // the single-receiver half-space run is all refill (67 % of its wave-cycles were such waits).  So the
      // fetches of BOTH batches are set going first -- one word of every guide cell, then, as those come in,
      // one word of every direction record (spray_touch) -- and the sprays proper find their tables near:
      // two memory round trips per pair of batches where there were four (single-receiver half-space run: step
      // launch 2.00 -> 1.81 ms; three or four batches at a time: 1.77 / 1.79).  Nothing is held in registers
      // between the passes but the touched words (both cells in registers at once made the register
      // allocator spill forty of the kernel's long-lived values, in every phase); each pass evaluates the
      // generator again.
      unsigned ids[kRefillBatches];   // (only ever indexed by constants: registers)
      ids[0] = id;
      unsigned n_b = 1u;                // (wave-uniform) batches in hand; batch h takes the ids behind batch h - 1's
      if (kPairs && k == 64u) {
#pragma unroll
        for (int h = 1; h < kRefillBatches; h++) {
          ids[h] = 0u;
          if (n_b == (unsigned)h) {
            const uint32_t w2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_ld(&ctl.word[Q_FREE]));
            if ((w2 & 0xFFFFu) >= 64u && q_pop(ctl, ring(Q_FREE), rmask, rlog, Q_FREE, lane, w2, ids[h]) == 64u) n_b++;
          }
        }
      }
      const unsigned want = k + (n_b - 1u) * 64u;
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(a.next, (unsigned long long)want);
      base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(base >> 32), 0) << 32) |
             (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
      const unsigned take = base >= a.n ? 0u : (a.n - base < want ? (unsigned)(a.n - base) : want);
      if (take < want && lane == 0)
        __hip_atomic_store(&ctl.word[kDrainedWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (kPairs && n_b > 1u) {
        uint32_t near_w = 0;
#pragma nounroll
        for (unsigned h = 0; h < n_b; h++)
          if (lane + h * 64u < take) near_w ^= spray_touch<0>(a, a.first_id + base + h * 64u + lane);
        R3D_SCHED_FENCE();
#pragma nounroll
        for (unsigned h = 0; h < n_b; h++)
          if (lane + h * 64u < take) near_w ^= spray_touch<1>(a, a.first_id + base + h * 64u + lane);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(near_w));   // (the touches are not to be dropped; their words have arrived by the time they are asked for)
#endif
        (void)near_w;
      }
      Phonon p;
      uint64_t hid = 0;
      bool fresh = false;
#pragma nounroll
      for (unsigned h = 0; h < n_b; h++) {
        unsigned slot = ids[0];
#pragma unroll
        for (int j = 1; j < kRefillBatches; j++) slot = h == (unsigned)j ? ids[j] : slot;
        fresh = lane + h * 64u < take;
        hid = a.first_id + base + h * 64u + lane;
        if (fresh) {
          Rng rng;
          rng_init(rng, hid);
          spray(a, p, rng);
          store_state(slot, p, rng, meta_pack(p.type, -1, 0u, Q_MOVE));
          pu[U_ID * F + slot] = U4{rng.id_lo, rng.id_hi, 0u, 0u};
        }
        report(fresh, 0, p, hid);   // GEN
        if (h == 0) dest = fresh ? Q_MOVE : dest;
        else {   // the later batches' slots are handed on here; the first's with everyone else's below
          R3D_PRIO_HIGH();
          q_push_all(ctl, rings, rcap, rlog, lane, true, fresh ? Q_MOVE : Q_FREE, slot);
          R3D_PRIO_LOW();
        }
      }
      if constexpr (kVectorTally) n_generated = take;
      else tally_n(kEv + R3D_EV_GENERATED, take);
    } else if (q == Q_MOVE) {
      // ---- termination checks, boundary search, free-path draw, advance (phonons.cpp:549-623),
      //      and what the face reached asks for when that is little: a plain hand-over, a Snell
      //      bend (phonons.cpp:640-661).  Lanes that can simply move again do so here. ----
      Phonon p;
      Rng rng;
      uint32_t meta = 0, nbr0 = 0;
      int fate = FATE_ALIVE, reason = 0;
      uint64_t hid = 0;
      // The drain's kept lanes, four lanes to a history (kQuad): the k <= 16 kept slots are spread over the wave,
      // history j onto lanes 4j .. 4j+3; all four lanes load the slot's state and run the move -- the same values in
      // every lane of a quad -- except that the boundary search's four faces are one lane's each
      // (r3d_physics.h tet_fast_exit_quad).  Lane 4j leads: it alone counts, stores and hands the slot on.
      bool quad = false;    // (wave-uniform)
      bool lead = true;
      unsigned fl = 0u;
      if constexpr (kQuad) {
        quad = was_kept && k <= 16u;
        if (quad) {
          unsigned long long hm = held_m;
          unsigned qid = 0u;
          for (unsigned j = 0; j < k; j++) {   // (wave-uniform: at most sixteen rounds of scalar work)
            const int src = __ffsll((long long)hm) - 1;
            hm &= hm - 1ull;
            const unsigned sid = (unsigned)__builtin_amdgcn_readlane((int)id, src);
            qid = (lane >> 2) == j ? sid : qid;
          }
          id = qid, act = (lane >> 2) < k, fl = lane & 3u, lead = fl == 0u;
          dest = act ? Q_FREE : dest;
        }
      }
      bool live = act;   // still moving in registers
      if (act) {
        load_state(id, p, rng, meta, nbr0);
        R3D_PRIO_MOVE();   // (the phase every other one waits for: above them, below a wave between batches)
        hid = ((uint64_t)rng.id_hi << 32) | rng.id_lo;
        drain_prio(p.t, thin, a.ttl);
      }
#pragma nounroll
      for (int rep = 0;; rep++) {
        Pending ev;
        ev.vel = 0.0, ev.face = -1, ev.flags = 0u, ev.nbr = -1;
        bool leaving = false;
        LaneStats st = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (kQuad) st.quiet = (quad && !lead) ? 1u : 0u;
        if (live) {
          fate = step_move<KIND, kQuad>(a, T, p, rng, st, &reason, ev, quad, fl);
          leaving = true;
          if (fate != FATE_ALIVE) {
            dest = Q_FREE;
          } else if (ev.face < 0) {
            dest = Q_SCATTER;
          } else if (ev.flags & F_COLLECT) {
            dest = Q_COLLECT;
          } else if (!(ev.flags & (F_REFLECT | F_ADJOIN))) {
            dest = Q_FREE, fate = FATE_LOST;   // phonons.cpp:675
          } else if (ev.flags & (F_REFLECT | F_DISCON)) {
            dest = Q_RT;
          } else {
            leaving = false;   // hand-over or bend: served right here, and on to the next move
          }
        }
        bool light = live && !leaving;
        if (light) step_event<KIND, EV_BEND>(a, T, p, rng, st, ev, ev.nbr);
        if constexpr (kInlineRt) {
          // Layered models: nearly every move ends on an interface that wants the solve, so when most of the batch
          // does, the wave serves it here, on the state it holds -- no hand-off, no take, no store and reload of the
          // slots -- and those lanes go on to their next move with the others (a thin batch: whenever any does).
          const bool wants = live && leaving && dest == Q_RT;
          const unsigned n_wants = count(wants && lead);
          // (a thin batch: when at least half of its histories want it -- the solve for a few lanes of many keeps the
          //  others from their next move for two thousand cycles, and the queue would batch those few with others')
          if (n_wants != 0u && n_wants >= (thin ? (quad ? 1u : (k + 1u) / 2u) : kInlineRtLanes)) {
            if (wants) {
              step_event<KIND, EV_RT>(a, T, p, rng, st, ev, ev.nbr);
              leaving = false, light = true, dest = Q_MOVE;
            }
            n_rtsolve += n_wants;
          }
        }
        if (TRACE) {
          report(st.reflect != 0u, 2, p, hid);    // REF
          report(st.transfer != 0u, 4, p, hid);   // CEL
        }
        n_iter += count(lead && st.iterations != 0u), n_transfer += count(lead && st.transfer != 0u), n_reflect += count(lead && st.reflect != 0u);
        if (a.vol) n_volout += count(lead && st.vol_out != 0u);
        live = light;
        const unsigned n_live = (unsigned)__popcll(ballot(live));
        // a full batch goes on while most of its lanes can (the others' slots are wanted by the
        // queues); a thin one -- the tail of a launch, where nobody waits for these lanes -- goes on
        // while any can: the longest histories are what a drain waits for, and a move in registers
        // costs a fraction of a round trip through the pool
        const bool last = !thin ? (rep + 1 >= kPoolMoves) || (n_live < kMoveAgainLanes)
                                : (rep + 1 >= kPoolMovesThin) || (n_live == 0u);
#ifdef R3D_PHASE_TIMING
        if (lane == 0) atomicAdd(&s_stats[0][7], 1ull), atomicAdd(&s_stats[1][7], (unsigned long long)(n_live));
#endif
        if (last && live) dest = Q_MOVE;
        if ((leaving || (last && live)) && lead) {
          const bool keep = dest != Q_FREE;   // (a history that ended leaves nothing to keep)
          if (keep) {
            store_state(id, p, rng, meta_pack(p.type, dest == Q_MOVE ? -1 : ev.face, dest == Q_MOVE ? 0u : ev.flags, dest));
            if (dest == Q_RT || dest == Q_COLLECT)   // (the cell's record is at hand here: the later phase need not wait for it)
              pu[U_ID * F + id].d = (uint32_t)ev.nbr;
          } else
            pu[U_STATE * F + id].d = meta_pack(0, -1, 0u, Q_FREE);
        }
        if (last) break;
      }
      if (kQuad) act = act && lead;   // (the helpers' work is done: the leaders hand the slots on)
      const bool died = act && dest == Q_FREE;
      finish(died, fate, reason, p, hid, (TRACE && died) ? pu[U_ID * F + id].c : 0u, id);
    } else if (q == Q_COLLECT) {
      // ---- arrival at a collection face: the receivers, with the incident state
      //      (phonons.cpp:629-631), then on to what the face itself asks for ----
      Phonon p;
      Rng rng;
      uint32_t meta = 0, nbr = 0;
      uint64_t hid = 0;
      uint32_t k0 = 0, k1 = 0, catches = 0;
      double vel = 1.0;
      Pending ev;
      ev.vel = 0.0, ev.face = 0, ev.flags = 0u, ev.nbr = -1;
      if (act) {
        load_state(id, p, rng, meta, nbr);
        drain_prio_again();
        hid = ((uint64_t)rng.id_hi << 32) | rng.id_lo;
        ev.face = (int)((meta >> 1) & 7u) - 1, ev.flags = (meta >> 8) & 0xFFu;
        vel = velocity_in<KIND>(T, p.cell, p.loc, p.type);
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
      if constexpr (kVectorTally) n_collect = k;
      else tally_n(kEv + R3D_EV_COLLECT, k);
      if (any_lane(k1 > k0)) {
        const uint32_t hits = pool_collect_pairs<KIND, TRACE>(a, T, p, vel, k0, k1, LDS_SEIS ? lds_gitems : nullptr,
                                                              lane, catches, bc);
        if constexpr (kVectorTally) n_catch = hits;
        else tally_n(kEv + R3D_EV_CATCH, hits);
      }
      bool died = false;
      if (act) {
        if (!(ev.flags & (F_REFLECT | F_ADJOIN))) {
          died = true;   // lost through this face (phonons.cpp:675)
        } else if (ev.flags & (F_REFLECT | F_DISCON)) {
          dest = Q_RT;
        } else {
          dest = Q_MOVE;
        }
      }
      const bool light = act && dest == Q_MOVE;
      LaneStats st = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (light) step_event<KIND, EV_BEND>(a, T, p, rng, st, ev, (int)nbr);
      if (TRACE) {
        report(st.reflect != 0u, 2, p, hid);    // REF
        report(st.transfer != 0u, 4, p, hid);   // CEL
      }
      n_transfer += count(st.transfer != 0u), n_reflect += count(st.reflect != 0u);
      if (a.vol) n_volout += count(st.vol_out != 0u);
      if (act) {
        const uint32_t m2 = died ? meta_pack(0, -1, 0u, Q_FREE)
                                 : meta_pack(p.type, light ? -1 : ev.face, light ? 0u : ev.flags, dest);
        if (light) store_event(id, p, rng, m2);
        else pu[U_STATE * F + id].d = m2;
        if (TRACE) pu[U_ID * F + id].c += catches;
      }
      finish(died, FATE_LOST, 0, p, hid, (TRACE && died) ? pu[U_ID * F + id].c : 0u, id);
    } else {
      // ---- RT: reflection / transmission solve; SCATTER: deflection drawn from the scatterer's
      //      tables (phonons.cpp:611-618, :640-661) ----
      Phonon p;
      Rng rng;
      uint32_t meta = 0, nbr = 0;
      uint64_t hid = 0;
      LaneStats st = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (act) {
        load_state(id, p, rng, meta, nbr);
        drain_prio_again();
        hid = ((uint64_t)rng.id_hi << 32) | rng.id_lo;
        Pending ev;
        ev.vel = 0.0, ev.face = (int)((meta >> 1) & 7u) - 1, ev.flags = (meta >> 8) & 0xFFu, ev.nbr = -1;
        if (q == Q_RT) {
#ifdef R3D_ABLATE_RT
          step_event<KIND, EV_RT>(a, T, p, rng, st, ev, (int)nbr);
#else
          // the solve in two halves with nothing but the choice (four words), the draw counter and the slot
          // number carried across (everything else is read again from the slot and the tables):
          // what is live while the weights are formed decides whether three waves fit a SIMD
          const RtChoice ch = rt_event_choose<KIND>(a, T, p, rng, st, ev, (int)nbr);
          const uint32_t draws = rng.k;
          asm volatile("" ::: "memory");   // (the second half must not reuse the first half's loads)
          load_state(id, p, rng, meta, nbr);
          rng.k = draws;
          Pending ev2;
          ev2.vel = 0.0, ev2.face = (int)((meta >> 1) & 7u) - 1, ev2.flags = (meta >> 8) & 0xFFu, ev2.nbr = -1;
          rt_event_apply<KIND>(a, T, p, st, ev2, (int)nbr, ch);
#endif
        } else {
          step_event<KIND, EV_SCATTER>(a, T, p, rng, st, ev);
        }
        store_event(id, p, rng, meta_pack(p.type, -1, 0u, Q_MOVE));
        dest = Q_MOVE;
      }
      if (TRACE) {
        report(act && q == Q_SCATTER, 1, p, hid);   // SCT
        report(st.reflect != 0u, 2, p, hid);        // REF
        report(st.transfer != 0u, 4, p, hid);       // CEL
      }
      if (q == Q_RT) n_transfer += count(st.transfer != 0u), n_reflect += count(st.reflect != 0u);
      if (a.vol) n_volout += count(st.vol_out != 0u);
    }
    if constexpr (!kVectorTally) {
      if (q == Q_SCATTER) tally_n(kEv + R3D_EV_SCATTER, k);
      tally_n(kEv + R3D_EV_ITERATIONS, n_iter), tally_n(kEv + R3D_EV_TRANSFER, n_transfer);
      tally_n(kEv + R3D_EV_REFLECT, n_reflect);
      tally_n(kEv + R3D_EV_VOLUME_OUT, n_volout);
      tally_n(kEv + R3D_EV_RTSOLVE, q == Q_RT ? k : n_rtsolve);
    } else {
      // lane j holds what the batch adds to tally j (include/r3d.h: lost, timeout, invalid + reasons, the
      // eight event counters): one LDS add for all of them
      static_assert(kEv + R3D_EV_NUM == R3D_N_SCALARS && R3D_N_SCALARS <= 64, "one lane per tally");
      uint32_t tv = 0;
#define R3D_PUT(slot, n)                                                         \
  do {                                                                           \
    uint32_t put_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n));          \
    asm("" : "+s"(put_)); /* (a scalar REGISTER: a count known to be zero would fold into an operand the instruction does not take) */ \
    asm("v_writelane_b32 %0, %1, %2" : "+v"(tv) : "s"(put_), "n"(slot));         \
  } while (0)
      R3D_PUT(0, n_lost);
      R3D_PUT(1, n_timeout);
      R3D_PUT(kEv + R3D_EV_GENERATED, n_generated);
      R3D_PUT(kEv + R3D_EV_ITERATIONS, n_iter);
      R3D_PUT(kEv + R3D_EV_SCATTER, q == Q_SCATTER ? k : 0u);
      R3D_PUT(kEv + R3D_EV_COLLECT, n_collect);
      R3D_PUT(kEv + R3D_EV_CATCH, n_catch);
      R3D_PUT(kEv + R3D_EV_REFLECT, n_reflect);
      R3D_PUT(kEv + R3D_EV_VOLUME_OUT, n_volout);
      R3D_PUT(kEv + R3D_EV_TRANSFER, n_transfer);
      R3D_PUT(kEv + R3D_EV_RTSOLVE, q == Q_RT ? k : n_rtsolve);
#undef R3D_PUT
      if (lane < (unsigned)R3D_N_SCALARS && tv) atomicAdd(&s_tally[lane], (unsigned long long)tv);
    }
#ifdef R3D_PHASE_TIMING
    const unsigned long long t_push = __builtin_readcyclecounter();
#endif
    R3D_PRIO_HIGH();
    bool keep_lanes = false;   // (wave-uniform) this batch's slots stay with the wave
    bool chain_set = false;    // (wave-uniform; kernels with a tail) ... as a chained batch: `held` was set for it below
    if constexpr (TAIL) {
      const bool alive = act && dest != Q_FREE;
      const unsigned long long alive_m = ballot(alive);
      if (kTailBatch != 0u && ids_out && !a.carry_out && k <= kTailBatch && alive_m) {
        const int d0 = __builtin_amdgcn_readlane(dest, __ffsll((long long)alive_m) - 1);
        keep_lanes = !any_lane(alive && dest != d0);   // they agree on what comes next
      }
      if (keep_lanes && !was_kept) {   // one more wave that keeps lanes: only while enough others serve the queues
        uint32_t n = 0;
        if (lane == 0) n = atomicAdd(&ctl.word[kKeepersWord], 1u);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)n) >= (uint32_t)kPoolWaves - kTailServers) {
          if (lane == 0) atomicSub(&ctl.word[kKeepersWord], 1u);
          keep_lanes = false;
        }
      } else if (was_kept && !keep_lanes) {
        if (lane == 0) atomicSub(&ctl.word[kKeepersWord], 1u);
      }
    }
    if (TAIL && keep_lanes) {
      // the slots' state is in LDS as for a hand-off and their next phase reads it back; only the slots of
      // histories that ended go back (the workgroup's count of free slots is how everyone learns that the
      // launch is over)
      const bool ended = act && dest == Q_FREE;
      if (any_lane(ended)) q_push_all(ctl, rings, rcap, rlog, lane, ended, Q_FREE, id);
      held = act && dest != Q_FREE;
      // (kept lanes sit in the wave's low lanes or anywhere: a phase only looks at `act`)
    } else if (kChainCollect && !was_kept && q == Q_COLLECT && k == 64u && !any_lane(act && dest != Q_RT)) {
      // (a full collection batch whose arrivals are all reflected -- a free surface: every one of them --: its solve next)
      if (!TAIL) k_chain = (unsigned)__builtin_amdgcn_readfirstlane((int)(k | ((unsigned)Q_RT << 8)));
      else held = act, dest = act ? (Q_RT | kChainFlag) : dest, chain_set = true;
    } else if (!was_kept && (q == Q_RT || q == Q_SCATTER)) {
      // (all on to MOVE -- every R/T and scattering batch; a batch fresh from a queue, or chained, sits in lanes 0 .. k-1)
      if (!TAIL) k_chain = (unsigned)__builtin_amdgcn_readfirstlane((int)(k | ((unsigned)Q_MOVE << 8)));   // served by this wave next: see the top of the loop
      else held = act, dest = act ? (Q_MOVE | kChainFlag) : dest, chain_set = true;   // ... as held lanes, in the kernels with a tail
    }
    else q_push_all(ctl, rings, rcap, rlog, lane, act, dest, id);
    if constexpr (TAIL) {
      if (!keep_lanes && !chain_set) held = false;
    }
#ifdef R3D_PHASE_TIMING
    if (lane == 0) {
      const unsigned long long t_end = __builtin_readcyclecounter();
      atomicAdd(&s_stats[0][q], 1ull);
      atomicAdd(&s_stats[1][q], (unsigned long long)k);
      atomicAdd(&s_stats[2][q], t_end - t_pop);
      atomicAdd(&s_stats[3][q], t_begin - t_pop);
      atomicAdd(&s_stats[4][q], t_end - t_push);
    }
#endif
  }

  __syncthreads();
  asm volatile("" : "+s"(args));
  const KArgs& a_end = *(const KArgs*)args;   // (for what follows the loop, fetched after it)
  if (a_end.carry_out) {   // park the pool for the engine's next launch
    uint32_t* img = reinterpret_cast<uint32_t*>(a_end.carry_out) + (size_t)blockIdx.x * image_words;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(pd);
    for (size_t i = tid; i < image_words; i += kPoolBlock) img[i] = src[i];
  }
  // ---- the block's bin accumulators and tallies to HBM ----
  if (bc.on) {
    for (uint32_t i = tid; i <= bc.mask; i += kPoolBlock) {
      const uint32_t bin = bc.key[i];
      if (bin == kEmpty) continue;
      double* e = a_end.energy + (size_t)bin * 5;
#pragma unroll
      for (int cc = 0; cc < 5; cc++) {
        const double v = bc.e[i * 5u + cc];
        if (v != 0.0) unsafeAtomicAdd(e + cc, v);
      }
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const uint32_t n = bc.cnt[i * 2u + t];
        if (n) atomicAdd(a_end.counts + (size_t)bin * 2 + t, (unsigned long long)n);
      }
    }
  }
#ifdef R3D_PHASE_TIMING
  if (tid < 40) atomicAdd(&g_pool_stats[tid / 8][tid % 8], s_stats[tid / 8][tid % 8]);
#endif
  if (tid < R3D_N_SCALARS && s_tally[tid] != 0ull) atomicAdd(a_end.scalars + tid, s_tally[tid]);
}

// The traversal kernel under four names, so that profiles list the roles apart:
//   pool_kernel<..., false>  a step launch of a carry chain (parks what is unfinished: no tail)
//   pool_job_kernel          a self-contained production launch (r3d_run, r3d_run_device, the last launch
//                            of a chain when it brings ids of its own): drains its own stragglers
//   pool_drain_kernel        the flush launch of a carry chain (no new ids, only the histories carried over)
//   pool_kernel<..., true>   the diagnostic kernel (final records, report stream)
template <int KIND, bool LDS_CELLS, bool LDS_SCAT, bool TRACE>
__global__ __launch_bounds__(kPoolBlock) void pool_kernel(const KArgs a) {
  pool_body<KIND, LDS_CELLS, LDS_SCAT, TRACE, TRACE>(a);
}
template <int KIND, bool LDS_CELLS, bool LDS_SCAT>
__global__ __launch_bounds__(kPoolBlock) void pool_job_kernel(const KArgs a) {
  pool_body<KIND, LDS_CELLS, LDS_SCAT, false, true>(a);
}
template <int KIND, bool LDS_CELLS, bool LDS_SCAT>
__global__ __launch_bounds__(kPoolBlock) void pool_drain_kernel(const KArgs a) {
  pool_body<KIND, LDS_CELLS, LDS_SCAT, false, true>(a);
}

}  // namespace r3d
#endif
