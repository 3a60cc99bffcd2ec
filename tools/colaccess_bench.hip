// Column-tile access microbenchmark: what HBM rate does a workgroup get when it owns NC whole columns of a (4096, pitch)
// complex fp64 plane, i.e. reads and writes NC*16-byte segments at a stride of one row?  (Design question: a single-pass
// register-resident column transform needs NC = 4; the two-pass kernels use 32 columns = 512-byte segments.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NC, int PER>
__global__ __launch_bounds__(512) void k_cols(const cd* __restrict__ in, cd* __restrict__ out, int pitch, int ntiles) {
  const int tid = threadIdx.x, col = tid % NC, r0 = tid / NC;
  constexpr int RS = 512 / NC;                 // rows covered by one load instruction of the workgroup
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const size_t base = (size_t)t * NC + col;
    cd v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] = in[(size_t)(r0 + RS * j) * pitch + base];
#pragma unroll
    for (int j = 0; j < PER; ++j) { v[j].x = v[j].x * 1.0000001 + v[(j + 1) % PER].y; }
#pragma unroll
    for (int j = 0; j < PER; ++j) out[(size_t)(r0 + RS * j) * pitch + base] = v[j];
  }
}

template <int NC, int PER>
static void run(const char* name, const cd* in, cd* out, int N, int pitch, int width, int grid) {
  // PER loads per thread cover PER * 512 / NC rows; loop the kernel over row blocks by offsetting the pointers
  const int rows_per = PER * 512 / NC, nblk = N / rows_per, ntiles = width / NC;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(a));
    for (int rb = 0; rb < nblk; ++rb)
      hipLaunchKernelGGL((k_cols<NC, PER>), dim3(grid ? grid : ntiles), dim3(512), 0, 0, in + (size_t)rb * rows_per * pitch, out + (size_t)rb * rows_per * pitch, pitch, ntiles);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double bytes = 2.0 * N * (double)width * 16.0;
  printf("%-44s pitch %5d width %5d grid %5d launches %2d: %7.3f ms  %7.1f GB/s\n", name, pitch, width, grid ? grid : ntiles, nblk, best, bytes / best * 1e-6);
}

int main() {
  const int N = 4096;
  cd *in, *out;
  const size_t elems = (size_t)N * 4224;
  CK(hipMalloc(&in, elems * sizeof(cd))); CK(hipMalloc(&out, elems * sizeof(cd)));
  CK(hipMemset(in, 0, elems * sizeof(cd))); CK(hipMemset(out, 0, elems * sizeof(cd)));
  for (int pitch : {4096, 4104, 2056}) {
    const int width = pitch >= 4096 ? 4096 : 2048;
    run<32, 32>("32 columns (512 B), 32 per thread, 512 rows", in, out, N, pitch, width, 0);
    run<8, 32>("8 columns (128 B), 32 per thread, 2048 rows", in, out, N, pitch, width, 0);
    run<4, 32>("4 columns (64 B), whole column per WG", in, out, N, pitch, width, 0);
    run<4, 32>("4 columns (64 B), whole column, 256 WGs", in, out, N, pitch, width, 256);
    run<4, 32>("4 columns (64 B), whole column, 512 WGs", in, out, N, pitch, width, 512);
    run<2, 16>("2 columns (32 B), whole column per WG", in, out, N, pitch, width, 0);
    run<4, 16>("4 columns (64 B), half column per launch", in, out, N, pitch, width, 0);
  }
  return 0;
}
