// Does data WRITTEN by one kernel stay in the Infinity Cache (256 MiB) for the next kernel's reads?
// kernel A writes S bytes, kernel B reads them back (sum into a sink); compare B's bandwidth with B after a 1 GiB flush.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_write(double2* p, size_t n, double v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = make_double2(v, (double)i);
}
__global__ void k_read(const double2* p, size_t n, double* sink) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  double a = 0;
  for (; i < n; i += st) { double2 v = p[i]; a += v.x + v.y; }
  if (a == 12345.678) sink[0] = a;
}
int main() {
  const size_t GiB = 1ull << 30;
  double2 *buf, *flush; double* sink;
  CK(hipMalloc(&buf, GiB)); CK(hipMalloc(&flush, GiB)); CK(hipMalloc(&sink, 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int G = 256 * 16, T = 256;
  for (size_t mb : {16, 32, 64, 128, 192, 256, 512}) {
    const size_t n = mb * (1ull << 20) / 16;
    float ms_hot = 0, ms_cold = 0, ms_rr = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_write, dim3(G), dim3(T), 0, 0, buf, n, 1.0);          // write, then read at once
      CK(hipEventRecord(a)); hipLaunchKernelGGL(k_read, dim3(G), dim3(T), 0, 0, buf, n, sink); CK(hipEventRecord(b));
      CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_hot, a, b));
      CK(hipEventRecord(a)); hipLaunchKernelGGL(k_read, dim3(G), dim3(T), 0, 0, buf, n, sink); CK(hipEventRecord(b));   // read after read
      CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_rr, a, b));
      hipLaunchKernelGGL(k_write, dim3(G), dim3(T), 0, 0, flush, GiB / 16, 2.0);  // flush caches with 1 GiB of other stores
      CK(hipEventRecord(a)); hipLaunchKernelGGL(k_read, dim3(G), dim3(T), 0, 0, buf, n, sink); CK(hipEventRecord(b));
      CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_cold, a, b));
    }
    const double gb = n * 16 / 1e9;
    printf("%4zu MiB: read right after write %6.2f TB/s | read after read %6.2f TB/s | read after 1 GiB flush %6.2f TB/s\n", mb,
           gb / ms_hot, gb / ms_rr, gb / ms_cold);
  }
  return 0;
}
