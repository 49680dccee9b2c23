// horner.hip -- what does a polynomial coefficient cost the vector unit?  A 13-term Horner chain in fp64,
// CHAINS (1 or 4) independent chains per lane, 12 waves per CU (the traversal's occupancy), three ways:
//   lit : as the compiler writes it from literals: two v_mov_b32 per coefficient + v_fmac_f64 (VOP2)
//   sreg: v_fma_f64 (VOP3) with the coefficient as a SCALAR operand, materialised by two s_mov_b32
//   stbl: the same with the coefficients s_load'ed from constant memory, sixteen scalar registers at a time
#include <hip/hip_runtime.h>
#include <cstdio>

#define COEFS(X) X(1.0/27) X(1.0/25) X(1.0/23) X(1.0/21) X(1.0/19) X(1.0/17) X(1.0/15) X(1.0/13) X(1.0/11) X(1.0/9) X(1.0/7) X(1.0/5) X(1.0/3)
__constant__ double kC[16] = {1.0/27,1.0/25,1.0/23,1.0/21,1.0/19,1.0/17,1.0/15,1.0/13,1.0/11,1.0/9,1.0/7,1.0/5,1.0/3, 1.0, 0, 0};

__device__ __forceinline__ double fma_s(double p, double t, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(t), "s"(c));
  return r;
}

template <int MODE, int CHAINS>
__global__ __launch_bounds__(768) void k(const double* x, double* out, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double t0 = x[i], t1 = t0 * 0.5, t2 = t0 * 0.25, t3 = t0 * 0.125;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int it = 0; it < iters; it++) {
    double p0, p1, p2, p3;
    if (MODE == 0) {
      p0 = p1 = p2 = p3 = 1.0 / 29;
#define X(c) p0 = __builtin_fma(p0, t0, c); if (CHAINS == 4) { p1 = __builtin_fma(p1, t1, c); p2 = __builtin_fma(p2, t2, c); p3 = __builtin_fma(p3, t3, c); }
      COEFS(X)
#undef X
    } else if (MODE == 1) {
      p0 = p1 = p2 = p3 = 1.0 / 29;
#define X(c) p0 = fma_s(p0, t0, c); if (CHAINS == 4) { p1 = fma_s(p1, t1, c); p2 = fma_s(p2, t2, c); p3 = fma_s(p3, t3, c); }
      COEFS(X)
#undef X
    } else {
      p0 = p1 = p2 = p3 = 1.0 / 29;
#pragma unroll
      for (int q = 0; q < 13; q++) {
        const double c = kC[q];
        p0 = fma_s(p0, t0, c);
        if (CHAINS == 4) p1 = fma_s(p1, t1, c), p2 = fma_s(p2, t2, c), p3 = fma_s(p3, t3, c);
      }
    }
    a0 += p0, a1 += p1, a2 += p2, a3 += p3;
    t0 += 1e-9, t1 += 1e-9, t2 += 1e-9, t3 += 1e-9;
  }
  out[i] = a0 + a1 + a2 + a3;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, n = cus * 768, iters = 20000;
  double *x, *out;
  hipMalloc(&x, n * 8), hipMalloc(&out, n * 8);
  hipMemset(x, 0, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int chains) {
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(cus), dim3(768), 0, 0, x, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    // per SIMD: 3 waves x iters x 4 chains x 13 coefficient steps
    const double steps = 3.0 * iters * chains * 13;
    printf("%-6s x%d %8.3f ms -> %.2f cycles@2.4GHz per coefficient step per SIMD (fma alone: 4)\n", name, chains, ms, ms * 1e-3 * 2.4e9 / steps);
  };
  run("lit", k<0, 4>, 4), run("sreg", k<1, 4>, 4), run("stbl", k<2, 4>, 4);
  run("lit", k<0, 1>, 1), run("sreg", k<1, 1>, 1), run("stbl", k<2, 1>, 1);
  return 0;
}
