// gather.hip -- what does a 64-lane gather from L2-resident records cost the texture-addresser / L1 path?
// Each wave repeatedly loads `NL` x (WIDTH bytes) per lane from a record chosen per lane; variants:
//   mode 0: every lane its own random record (the traversal's cell-record fetch)
//   mode 1: the four lanes of a quad read the four 16-byte quarters of ONE 64-byte line (16 lines / instruction)
//   mode 2: all lanes the same record (broadcast)
//   mode 3: twelve consecutive lanes read the twelve 16-byte pieces of ONE 192-byte record (five records / instruction)
//   mode 4: sixteen consecutive lanes read sixteen consecutive 16-byte pieces (256 bytes) of one record (four / instruction)
// Records live in a table of `ncell` x `rec` bytes (L2-resident for the sizes of interest).
// Output: ns per load instruction per CU at 12 waves / CU (the traversal's occupancy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NL, int W>   // NL loads of W dwords per lane and iteration
__global__ __launch_bounds__(768) void gather(const uint32_t* __restrict__ table, const uint32_t* __restrict__ idx,
                                              int iters, int rec_dwords, int mode, double* sink) {
  const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t h = wave * 2654435761u + lane * 40503u + 12345u;
  double acc = 0;
  for (int it = 0; it < iters; it++) {
    h = h * 1664525u + 1013904223u;
    uint32_t r = idx[(h >> 8) & 0xFFFFu];
    if (mode == 2) r = __builtin_amdgcn_readfirstlane(r);
    if (mode == 1) r = __shfl(r, lane & ~3u);
    if (mode == 3) r = __shfl(r, (lane / 12u) * 12u);
    if (mode == 4) r = __shfl(r, lane & ~15u);
    const uint32_t* p = table + (size_t)r * rec_dwords + (mode == 1 ? (lane & 3u) * 4u : mode == 3 ? (lane % 12u) * 4u : mode == 4 ? (lane & 15u) * 4u : 0u);
    typedef uint32_t vec __attribute__((ext_vector_type(W)));
    vec v[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) v[k] = *reinterpret_cast<const vec*>(p + k * (mode == 1 ? 16 : mode >= 3 ? 0 : W));
#pragma unroll
    for (int k = 0; k < NL; k++) acc += (double)v[k][0];
    h ^= (uint32_t)acc;
  }
  if (acc == 12345.678) *sink = acc;
}

int main(int argc, char** argv) {
  const int ncell = argc > 1 ? atoi(argv[1]) : 2275;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::vector<uint32_t> idx(65536);
  for (auto& x : idx) x = rand() % ncell;
  uint32_t *d_table, *d_idx;
  double* d_sink;
  hipMalloc(&d_table, (size_t)ncell * 512 + 4096);
  hipMemset(d_table, 1, (size_t)ncell * 512 + 4096);
  hipMalloc(&d_idx, idx.size() * 4);
  hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&d_sink, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 2000;
  auto run = [&](const char* name, auto kern, int nl, int rec_bytes, int mode) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(cus), dim3(768), 0, 0, d_table, d_idx, iters, rec_bytes / 4, mode, d_sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = 12.0 * iters * nl;   // wave-level load instructions per CU
    printf("%-44s rec %3d B mode %d: %7.3f ms  -> %6.1f ns = %5.0f cycles@2.4GHz per load instruction per CU, %6.1f ns per record\n",
           name, rec_bytes, mode, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4, ms * 1e6 / (12.0 * iters));
  };
  printf("%d CUs, table of %d records\n", cus, ncell);
  for (int mode = 0; mode < 5; mode++) {
    run("16 x dwordx4 (256-B record)", gather<16, 4>, 16, 256, mode);
    run("12 x dwordx4 (192-B record)", gather<12, 4>, 12, 192, mode);
    run(" 8 x dwordx4 (128-B record)", gather<8, 4>, 8, 128, mode);
    run("24 x dwordx2 (192-B record)", gather<24, 2>, 24, 192, mode);
    run(" 4 x dwordx4 ( 64-B record)", gather<4, 4>, 4, 64, mode);
    run(" 1 x dwordx4 ( 64-B record)", gather<1, 4>, 1, 64, mode);
  }
  return 0;
}
