// fp64 VALU issue rate on MI355X as the row kernels see it: ONE workgroup per CU, 1 or 2 waves per SIMD, K independent
// dependency chains per thread of v_fma_f64 / v_add_f64 / v_mul_f64.  Prints nanoseconds and (at 2.4 GHz nominal) cycles per
// wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int K, int OP>
__global__ void k_valu(double* out, int iters, double seed) {
  double a[K];
#pragma unroll
  for (int i = 0; i < K; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
  const double m = 1.0 + seed * 1e-9, c = seed * 1e-7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int i = 0; i < K; ++i) {
        if (OP == 0) a[i] = __builtin_fma(a[i], m, c);
        else if (OP == 1) a[i] = a[i] + c;
        else if (OP == 2) a[i] = a[i] * m;
        else { a[i] = a[i] + a[(i + 1) % K]; }      // butterfly-like: adds between different chains
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int OP>
void run(const char* name, int threads, double* out) {
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_valu<K, OP>), dim3(256), dim3(threads), 0, 0, out, 10, 1.0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k_valu<K, OP>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double ninstr_per_simd = (double)iters * 8 * K * (threads / 256.0);      // wave-instructions per SIMD
  const double ns = ms * 1e6 / ninstr_per_simd;
  printf("%-10s K=%2d threads=%4d (%d wave/SIMD): %.3f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz\n", name, K, threads,
         threads / 256, ns, ns * 2.4);
}

int main() {
  double* out;
  CK(hipMalloc(&out, 256 * 1024 * 8));
  run<16, 0>("fma", 256, out);
  run<16, 0>("fma", 512, out);
  run<16, 0>("fma", 1024, out);
  run<16, 1>("add", 256, out);
  run<16, 1>("add", 512, out);
  run<16, 2>("mul", 256, out);
  run<16, 2>("mul", 512, out);
  run<16, 3>("add-mix", 256, out);
  run<16, 3>("add-mix", 512, out);
  run<4, 0>("fma", 256, out);
  run<4, 0>("fma", 512, out);
  run<2, 0>("fma", 256, out);
  run<1, 0>("fma", 256, out);
  run<1, 1>("add", 256, out);
  return 0;
}
