// hbm_gather.hip -- how many random 64-byte sectors per second does the memory system deliver?
// The single-receiver half-space run (BASELINE config 1) is a chain of table look-ups: a take-off spray and a
// scattering per history, each two dependent fetches of one 64-byte sector from tables of 0.3-2 GB (guide cell,
// then direction record): 3.7 GB per 1e7 histories in 64-byte pieces.  This measures the rate such gathers reach
// on the chip, by occupancy (waves per CU) and by independent fetches in flight per lane, each round's addresses
// depending on the data of the round before (as a direction record's address depends on its guide cell).
// Output: sectors / s and GB / s (sectors x 64 B).   ./hbm_gather [table GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NL>
__global__ void gather(const uint4* __restrict__ table, uint32_t sector_mask, int iters, uint32_t* sink) {
  uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint4 v[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
      uint32_t x = h + (uint32_t)k * 0x9E3779B9u;
      x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
      v[k] = table[(size_t)(x & sector_mask) * 4u];   // the first 16 bytes of a random 64-byte sector
    }
#pragma unroll
    for (int k = 0; k < NL; k++) acc += v[k].x;
    h = h * 1664525u + 1013904223u + acc;   // (the next round's addresses wait for this round's data)
  }
  if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 4.0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  const int cus = prop.multiProcessorCount;
  size_t sectors = 1;
  while (sectors * 2 * 64 <= (size_t)(gib * 1073741824.0)) sectors *= 2;
  uint4* table;
  uint32_t* sink;
  if (hipMalloc(&table, sectors * 64) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  if (hipMemset(table, 1, sectors * 64) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  printf("%d CUs, table %.1f GiB = %zu sectors of 64 B\n", cus, sectors * 64 / 1073741824.0, sectors);
  auto run = [&](auto kern, int nl, int threads, int blocks_per_cu) {
    const int iters = 4000 / nl;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(cus * blocks_per_cu), dim3(threads), 0, 0, table, (uint32_t)(sectors - 1), iters, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double n = (double)cus * blocks_per_cu * threads * iters * nl;
    printf("%2d waves/CU, %d in flight per lane: %8.3f ms  %7.2f G sectors/s = %6.2f TB/s of 64-B sectors; round trip %6.0f ns\n",
           threads * blocks_per_cu / 64, nl, ms, n / ms * 1e-6, n * 64 / ms * 1e-9, ms * 1e6 / iters);
  };
  for (int occ = 0; occ < 3; occ++) {
    const int threads = occ == 0 ? 768 : 1024, bpc = occ == 2 ? 2 : 1;   // 12, 16, 32 waves per CU
    run(gather<1>, 1, threads, bpc);
    run(gather<2>, 2, threads, bpc);
    run(gather<4>, 4, threads, bpc);
    run(gather<8>, 8, threads, bpc);
  }
  return 0;
}
