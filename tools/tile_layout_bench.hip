// Two measurements the round-2 review asked for before deciding about a single-pass column transform at 4096 rows
// ("the third y sub-pass"):
//  (i)  column tiles of 4 columns (64-byte segments, what a register-resident 4096-point column transform can own) with the two
//       tiles that share a 128-byte line placed on the SAME XCD (blocks b and b + 8) and running at the same time;
//  (ii) a tile-blocked layout [k_tile][y][NC] (NC = 4 or 8 columns): the column kernel then streams a contiguous 256 / 512 KB
//       tile, and the ROW kernels pay instead -- one 64 / 128-byte piece per tile, pieces 256 / 512 KB apart.  Measured: the
//       column-side stream, and the row-side copy with the row kernels' thread mapping (one workgroup = one row, lanes along kx,
//       persistent grid of 256), with and without pairing the rows 2m, 2m + 1 (which share every line) on one XCD.
// Every kernel copies `in` to `out` (1 read + 1 write per element); four 256 MiB planes per side are cycled so that nothing is
// served by the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int N = 4096, NPL = 4;

__device__ __forceinline__ int paired(int b) { return 2 * ((b / 16) * 8 + b % 8) + (b / 8) % 2; }     // blocks b, b + 8 -> 2p, 2p + 1

// row-major plane [y][k]: a workgroup owns NC whole columns (k_cols of colaccess_bench), optionally XCD-paired
template <int NC, int PER, bool PAIR>
__global__ __launch_bounds__(512) void k_cols(const cd* __restrict__ in, cd* __restrict__ out, int ntiles) {
  const int tid = threadIdx.x, col = tid % NC, r0 = tid / NC;
  constexpr int RS = 512 / NC;
  for (int t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
    const int t = PAIR ? paired(t0) : t0;
    const size_t base = (size_t)t * NC + col;
    cd v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] = in[(size_t)(r0 + RS * j) * N + base];
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j].x = v[j].x * 1.0000001 + v[(j + 1) % PER].y;
#pragma unroll
    for (int j = 0; j < PER; ++j) out[(size_t)(r0 + RS * j) * N + base] = v[j];
  }
}
// tile-blocked plane [k_tile][y][NC]: the column side streams one contiguous tile (NC * N elements) per workgroup
template <int NC, int THREADS>
__global__ __launch_bounds__(THREADS) void k_tile_stream(const cd* __restrict__ in, cd* __restrict__ out, int ntiles) {
  constexpr int PER = NC * N / THREADS;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const size_t base = (size_t)t * NC * N + threadIdx.x;
    cd v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] = in[base + (size_t)j * THREADS];
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j].x = v[j].x * 1.0000001 + v[(j + 1) % PER].y;
#pragma unroll
    for (int j = 0; j < PER; ++j) out[base + (size_t)j * THREADS] = v[j];
  }
}
// the row side: one workgroup of 512 threads = one row of N elements, 8 per thread, lanes along kx; NC = 0: row-major plane
template <int NC, bool PAIR>
__global__ __launch_bounds__(512) void k_rows(const cd* __restrict__ in, cd* __restrict__ out, int nrows) {
  for (int r0 = blockIdx.x; r0 < nrows; r0 += gridDim.x) {
    const int y = PAIR ? paired(r0) : r0;
    cd v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int kx = threadIdx.x + 512 * t;
      const size_t at = NC ? (size_t)(kx / (NC ? NC : 1)) * (NC * N) + (size_t)y * NC + kx % (NC ? NC : 1) : (size_t)y * N + kx;
      v[t] = in[at];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t].x = v[t].x * 1.0000001 + v[(t + 1) % 8].y;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int kx = threadIdx.x + 512 * t;
      const size_t at = NC ? (size_t)(kx / (NC ? NC : 1)) * (NC * N) + (size_t)y * NC + kx % (NC ? NC : 1) : (size_t)y * N + kx;
      out[at] = v[t];
    }
  }
}

template <typename L>
static void timeit(const char* name, cd** in, cd** out, L launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    for (int p = 0; p < NPL; ++p) launch(in[p], out[p]);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double bytes = 2.0 * NPL * (double)N * N * 16.0;
  printf("%-78s %7.3f ms per plane  %7.1f GB/s\n", name, best / NPL, bytes / best * 1e-6);
}

int main() {
  cd *in[NPL], *out[NPL];
  for (int p = 0; p < NPL; ++p) {
    CK(hipMalloc(&in[p], (size_t)N * N * sizeof(cd))); CK(hipMalloc(&out[p], (size_t)N * N * sizeof(cd)));
    CK(hipMemset(in[p], 0, (size_t)N * N * sizeof(cd))); CK(hipMemset(out[p], 0, (size_t)N * N * sizeof(cd)));
  }
  printf("column side, row-major plane [y][k] (what the kernels have today)\n");
  timeit("  32 columns (512 B segments), 512 rows per launch x 8 [the two-pass tiles]", in, out, [](cd* i, cd* o) {
    for (int rb = 0; rb < 8; ++rb) hipLaunchKernelGGL((k_cols<32, 32, false>), dim3(N / 32), dim3(512), 0, 0, i + (size_t)rb * 512 * N, o + (size_t)rb * 512 * N, N / 32); });
  timeit("  4 columns (64 B), whole column per workgroup, 1024 tiles in launch order", in, out, [](cd* i, cd* o) {
    hipLaunchKernelGGL((k_cols<4, 32, false>), dim3(1024), dim3(512), 0, 0, i, o, 1024); });
  timeit("  (i) 4 columns, the two tiles of a 128-B line on blocks b, b+8 (same XCD)", in, out, [](cd* i, cd* o) {
    hipLaunchKernelGGL((k_cols<4, 32, true>), dim3(1024), dim3(512), 0, 0, i, o, 1024); });
  timeit("  (i) the same, persistent grid of 256", in, out, [](cd* i, cd* o) {
    hipLaunchKernelGGL((k_cols<4, 32, true>), dim3(256), dim3(512), 0, 0, i, o, 1024); });
  timeit("  8 columns (128 B), half a column per launch x 2", in, out, [](cd* i, cd* o) {
    for (int rb = 0; rb < 2; ++rb) hipLaunchKernelGGL((k_cols<8, 32, false>), dim3(512), dim3(512), 0, 0, i + (size_t)rb * 2048 * N, o + (size_t)rb * 2048 * N, 512); });
  printf("(ii) column side, tile-blocked plane [k_tile][y][NC]: contiguous tiles\n");
  timeit("  NC = 4: 256 KB per tile, 1024 threads x 16 elements, one tile per workgroup", in, out, [](cd* i, cd* o) {
    hipLaunchKernelGGL((k_tile_stream<4, 1024>), dim3(1024), dim3(1024), 0, 0, i, o, 1024); });
  timeit("  NC = 4, persistent grid of 256 (one workgroup per CU, as the registers allow)", in, out, [](cd* i, cd* o) {
    hipLaunchKernelGGL((k_tile_stream<4, 1024>), dim3(256), dim3(1024), 0, 0, i, o, 1024); });
  printf("(ii) row side (512 threads = one row, lanes along kx, persistent grid of 256)\n");
  timeit("  row-major plane (today)", in, out, [](cd* i, cd* o) { hipLaunchKernelGGL((k_rows<0, false>), dim3(256), dim3(512), 0, 0, i, o, N); });
  timeit("  tile-blocked NC = 4 (64-B pieces 256 KB apart), rows in launch order", in, out, [](cd* i, cd* o) { hipLaunchKernelGGL((k_rows<4, false>), dim3(256), dim3(512), 0, 0, i, o, N); });
  timeit("  tile-blocked NC = 4, rows 2m, 2m+1 on blocks b, b+8 (same XCD)", in, out, [](cd* i, cd* o) { hipLaunchKernelGGL((k_rows<4, true>), dim3(256), dim3(512), 0, 0, i, o, N); });
  timeit("  tile-blocked NC = 8 (128-B pieces 512 KB apart), rows in launch order", in, out, [](cd* i, cd* o) { hipLaunchKernelGGL((k_rows<8, false>), dim3(256), dim3(512), 0, 0, i, o, N); });
  timeit("  tile-blocked NC = 8, rows 2m, 2m+1 paired on one XCD", in, out, [](cd* i, cd* o) { hipLaunchKernelGGL((k_rows<8, true>), dim3(256), dim3(512), 0, 0, i, o, N); });
  return 0;
}
