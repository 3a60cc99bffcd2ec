// Do fp64 VALU work and LDS stores of the waves of ONE workgroup overlap on MI355X?  Each thread alternates a block of KV
// independent v_fma_f64 with a block of KS ds_write_b128 (no barriers, no dependence between the two).  Prints the time of
// the VALU blocks alone, of the store blocks alone and of both, for 1 / 2 waves per SIMD (256 / 512 threads, one workgroup
// per CU).  overlap = (T_valu + T_lds - T_both) / min(T_valu, T_lds): 1 = perfect, 0 = the two pipelines serialise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int KV, int KS, int MODE, bool READS>     // MODE 1: VALU only, 2: LDS only, 3: both
__global__ void k_mix(double* out, int iters, double seed) {
  double2* lds = reinterpret_cast<double2*>(smem);
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
  const double m = 1.0 + seed * 1e-9, c = seed * 1e-7;
  double2 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = make_double2(seed + i, seed - i);
  double2* mine = lds + threadIdx.x;                 // conflict-free: lanes contiguous
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
#pragma unroll
      for (int rep = 0; rep < KV / 16; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_fma(a[i], m, c);
    }
    if (MODE & 2) {
#pragma unroll
      for (int s = 0; s < KS; ++s) mine[(s & 7) * 512] = v[s & 7];
      if (READS) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = mine[s * 512];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x ^ 1].x;
}

template <int KV, int KS, int MODE, bool READS>
float run1(int threads, double* out) {
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  auto k = k_mix<KV, KS, MODE, READS>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  hipLaunchKernelGGL(k, dim3(256), dim3(threads), 81920, 0, out, 10, 1.0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(256), dim3(threads), 81920, 0, out, iters, 1.0);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e6f / iters;       // ns per iteration
}

template <int KV, int KS, bool READS>
void report(int threads, double* out) {
  const float tv = run1<KV, KS, 1, READS>(threads, out), tl = run1<KV, KS, 2, READS>(threads, out), tb = run1<KV, KS, 3, READS>(threads, out);
  printf("threads=%4d  %3d fma + %2d ds_write_b128%s per iteration:  VALU %7.1f ns  LDS %7.1f ns  both %7.1f ns  overlap %.2f\n", threads, KV, KS,
         READS ? " + 8 ds_read_b128" : "", tv, tl, tb, (tv + tl - tb) / (tv < tl ? tv : tl));
}

int main() {
  double* out;
  CK(hipMalloc(&out, 256 * 1024 * 8));
  for (int threads : {256, 512}) {
    report<96, 8, false>(threads, out);
    report<96, 8, true>(threads, out);
    report<48, 8, true>(threads, out);
    report<192, 8, true>(threads, out);
    report<96, 16, true>(threads, out);
  }
  return 0;
}
