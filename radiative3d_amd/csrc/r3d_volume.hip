// r3d_volume.hip -- the scatter-event grid between ranks: compaction of a range of the grid's
// counters into (index, count) pairs and the add of such pairs into a grid (include/r3d.h
// r3d_volume_compact / r3d_volume_scatter_add).
//
// Why.  The grid of BASELINE config 5 is 2 x 300 x 64 x 256 x 256 uint32 = 10 GB per rank, and the
// reference's semantics for replicas is "add them" (vis/seisplot/combine.m:26-33; the grid is the
// per-frame histogram vis/scattervid/scattervid_above.m:111 builds from the event stream).  A
// rank's 1.25e7 histories leave ~10 events each: at most 1.25e8 of the 2.5e9 cells are touched
// (< 5 %; measured: profiles/).  Moving the grid as stored costs every rank (N-1)/N x 10 GB on the
// wire even as a reduce-scatter; moving the touched cells as 8-byte pairs costs a tenth of that.
// So each rank compacts, per owner of a frame range, the non-zero counters of that range; the
// pairs travel point to point; the owner adds what it receives into its own range.
//
// Both kernels are HBM-bound streaming work: 16-byte loads, one atomic per 32 KB tile to claim
// output space, pairs written as 8-byte stores; the add is one 32-bit atomic per pair.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/r3d.h"

namespace r3d {
extern thread_local std::string g_error;

namespace {

constexpr int kCompactBlock = 256;
constexpr int kQuadsPerThread = 8;                       // 8 x 16 B per thread in flight
constexpr uint64_t kTileCounters = (uint64_t)kCompactBlock * kQuadsPerThread * 4;   // 8192 counters = 32 KB per tile

// Non-zero counters of [begin, end) as (global index, count) pairs, in no particular order.
// One workgroup per 32 KB tile, grid-stride: every thread holds eight 16-byte quads of the tile in
// registers (quad q of thread t = quad q * 256 + t of the tile: coalesced), counts its non-zeros,
// the workgroup's exclusive scan gives each thread its place, ONE atomic claims the tile's output
// range.  Pairs beyond `capacity` are counted but not written (the caller sees *n > capacity and
// takes the dense path).
__global__ __launch_bounds__(kCompactBlock) void volume_compact_kernel(const uint32_t* __restrict__ counters,
                                                                        uint64_t begin, uint64_t end,
                                                                        uint2* __restrict__ pairs, uint64_t capacity,
                                                                        unsigned long long* __restrict__ n_out) {
  __shared__ uint32_t s_waves[2][kCompactBlock / 64];   // (two sets, used in turn: a tile's sums are still being read while the next tile's are written)
  __shared__ unsigned long long s_base;
  unsigned turn = 0;
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t n_tiles = (end - begin + kTileCounters - 1) / kTileCounters;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t t0 = begin + tile * kTileCounters;
    uint4 q[kQuadsPerThread];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kQuadsPerThread; k++) {
      const uint64_t at = t0 + ((uint64_t)k * kCompactBlock + tid) * 4;
      if (at + 4 <= end && (at & 3u) == 0) {
        q[k] = *reinterpret_cast<const uint4*>(counters + at);   // (hipMalloc'ed base, index a multiple of 4: 16-byte aligned)
      } else {   // the ragged end of the range, or a range that does not start on a quad
        q[k].x = at + 0 < end ? counters[at + 0] : 0u;
        q[k].y = at + 1 < end ? counters[at + 1] : 0u;
        q[k].z = at + 2 < end ? counters[at + 2] : 0u;
        q[k].w = at + 3 < end ? counters[at + 3] : 0u;
      }
      mine += (q[k].x != 0u) + (q[k].y != 0u) + (q[k].z != 0u) + (q[k].w != 0u);
    }
    // exclusive scan over the workgroup: within the wave by shuffles, across the four waves through LDS
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(incl, off);
      if (lane >= (unsigned)off) incl += y;
    }
    uint32_t* const s_wave = s_waves[turn];
    turn ^= 1u;
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kCompactBlock / 64; w++) {
      const uint32_t c = s_wave[w];
      before += (unsigned)w < wave ? c : 0u;
      total += c;
    }
    if (total == 0) continue;   // (a tile without a single event -- most of a 0.5 % full grid -- costs one barrier)
    if (tid == 0) s_base = atomicAdd(n_out, (unsigned long long)total);
    __syncthreads();
    {
      unsigned long long at_out = s_base + before + (incl - mine);
#pragma unroll
      for (int k = 0; k < kQuadsPerThread; k++) {
        const uint64_t at = t0 + ((uint64_t)k * kCompactBlock + tid) * 4;
        const uint32_t v[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (v[j]) {
            if (at_out < capacity) pairs[at_out] = make_uint2((uint32_t)(at + j), v[j]);
            at_out++;
          }
      }
    }
    // (no barrier here: the next tile writes the OTHER set of sums, and s_base only behind its own first barrier,
    //  which every thread reaches after it has read this tile's)
  }
}

// counters[index] += count for every pair, saturating at 2^32 - 1: an add whose sum wraps pins the cell
// to the ceiling (whatever the order of the adds that follow, the last one to see a wrap pins it
// again); *saturated counts the cells that reached the ceiling (an add that finds the cell already
// there is not counted again; two wrapping adds to one cell that race may both count).
__global__ void volume_scatter_add_kernel(uint32_t* __restrict__ counters, uint64_t len, const uint2* __restrict__ pairs,
                                          uint64_t n, unsigned long long* __restrict__ saturated,
                                          unsigned long long* __restrict__ out_of_range) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint2 p = pairs[i];
    if (p.x >= len) {   // (a pair for another owner's range: a protocol error, reported, never written)
      atomicAdd(out_of_range, 1ull);
      continue;
    }
    const uint32_t old = atomicAdd(counters + p.x, p.y);
    if (old + p.y < old) {
      atomicExch(counters + p.x, 0xFFFFFFFFu);
      if (old != 0xFFFFFFFFu) atomicAdd(saturated, 1ull);
    }
  }
}

}  // namespace
}  // namespace r3d

using namespace r3d;

extern "C" {

int r3d_volume_compact(int device, const uint32_t* d_counters, uint64_t begin, uint64_t end, uint32_t* d_pairs,
                       uint64_t capacity, uint64_t* d_n, void* stream) {
  if (!d_counters || !d_n || (capacity && !d_pairs)) return g_error = "r3d_volume_compact: null argument", 1;
  if (end < begin) return g_error = "r3d_volume_compact: end before begin", 1;
  if (end > (uint64_t(1) << 32)) return g_error = "r3d_volume_compact: indices beyond 2^32 do not fit a pair", 1;
  if (end == begin) return 0;
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) return g_error = "r3d_volume_compact: no HIP device", 1;
  if (prev != device && hipSetDevice(device) != hipSuccess) return g_error = "r3d_volume_compact: bad device", 1;
  const uint64_t tiles = (end - begin + kTileCounters - 1) / kTileCounters;
  // (enough workgroups to keep every CU's memory pipeline full -- 8 x 256 threads per CU --, grid-stride beyond)
  const unsigned grid = (unsigned)(tiles < 2048 * 4 ? tiles : 2048 * 4);
  volume_compact_kernel<<<dim3(grid), dim3(kCompactBlock), 0, reinterpret_cast<hipStream_t>(stream)>>>(
      d_counters, begin, end, reinterpret_cast<uint2*>(d_pairs), capacity, reinterpret_cast<unsigned long long*>(d_n));
  const hipError_t err = hipGetLastError();
  if (prev != device) (void)hipSetDevice(prev);
  if (err != hipSuccess) return g_error = std::string("r3d_volume_compact: ") + hipGetErrorString(err), 1;
  return 0;
}

int r3d_volume_scatter_add(int device, uint32_t* d_counters, uint64_t len, const uint32_t* d_pairs, uint64_t n,
                           uint64_t* d_flags, void* stream) {
  if (!d_counters || !d_flags || (n && !d_pairs)) return g_error = "r3d_volume_scatter_add: null argument", 1;
  if (n == 0) return 0;
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) return g_error = "r3d_volume_scatter_add: no HIP device", 1;
  if (prev != device && hipSetDevice(device) != hipSuccess) return g_error = "r3d_volume_scatter_add: bad device", 1;
  const uint64_t blocks = (n + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 2048 * 8 ? blocks : 2048 * 8);
  unsigned long long* flags = reinterpret_cast<unsigned long long*>(d_flags);
  volume_scatter_add_kernel<<<dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
      d_counters, len, reinterpret_cast<const uint2*>(d_pairs), n, flags, flags + 1);
  const hipError_t err = hipGetLastError();
  if (prev != device) (void)hipSetDevice(prev);
  if (err != hipSuccess) return g_error = std::string("r3d_volume_scatter_add: ") + hipGetErrorString(err), 1;
  return 0;
}

}  // extern "C"
