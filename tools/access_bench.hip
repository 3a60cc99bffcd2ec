// Access-pattern microbenchmark: how fast can one MI355X move a 4096x4096 complex128 plane
// (256 MiB) when it is read/written as column tiles of C columns (C*16-byte segments at a 64 KiB
// pitch), compared with a plain streaming copy?  Decides the FFT pass structure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// streaming copy, 16 B per lane
__global__ void copy_stream(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

// column-tile copy: each workgroup owns C adjacent columns, all N rows. Threads: tid%C = column,
// tid/C = row within a group of blockDim/C rows; P loads in flight per thread (rows r + t*T).
template <int C, int P>
__global__ void copy_coltile(const double2* __restrict__ in, double2* __restrict__ out, int N, int pitch, int swz) {
  int b = blockIdx.x;
  if (swz) {  // G=swz neighbouring tiles -> same XCD (b%8 equal) and adjacent dispatch slots
    int x = b & 7, r = b >> 3;          // r: slot within XCD
    b = ((r / swz) * 8 + x) * swz + (r % swz);
  }
  int c = threadIdx.x % C, j = threadIdx.x / C, T = blockDim.x / C;
  size_t col = (size_t)b * C + c;
  for (int base = 0; base < N; base += T * P) {
    double2 v[P];
#pragma unroll
    for (int t = 0; t < P; ++t) v[t] = in[(size_t)(base + j + t * T) * pitch + col];
#pragma unroll
    for (int t = 0; t < P; ++t) { v[t].x += 1.0; out[(size_t)(base + j + t * T) * pitch + col] = v[t]; }
  }
}

// 3-pass style tile: 64 rows at stride 64 rows, W contiguous columns (W*16 B segments)
template <int P>
__global__ void copy_strided_rows(const double2* __restrict__ in, double2* __restrict__ out, int N, int pitch, int W) {
  // tile id -> (y1 in [0,64), column block)
  int tiles_x = pitch / W;
  int y1 = blockIdx.x / tiles_x, cb = blockIdx.x % tiles_x;
  int c = threadIdx.x % W, j = threadIdx.x / W, T = blockDim.x / W;   // T rows per sweep
  for (int base = 0; base < 64; base += T * P) {
    double2 v[P];
#pragma unroll
    for (int t = 0; t < P; ++t) v[t] = in[(size_t)(y1 + 64 * (base + j + t * T)) * pitch + cb * W + c];
#pragma unroll
    for (int t = 0; t < P; ++t) { v[t].x += 1.0; out[(size_t)(y1 + 64 * (base + j + t * T)) * pitch + cb * W + c] = v[t]; }
  }
}

template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
  const int N = 4096; const size_t n = (size_t)N * N; const double bytes = 2.0 * n * 16;
  double2 *a, *b, *c2, *d2;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c2, n * 16)); CK(hipMalloc(&d2, n * 16));
  CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16)); CK(hipMemset(c2, 0, n * 16)); CK(hipMemset(d2, 0, n * 16));
  // alternate between two buffer pairs so the 256 MiB Infinity Cache cannot serve the reads
  int flip = 0;
  auto pr = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms * 1e-6); fflush(stdout); };
  pr("stream copy 2048 blocks x 256", timeit([&] { flip ^= 1; copy_stream<<<2048, 256>>>(flip ? a : c2, flip ? b : d2, n); }, 20));
#define COL(C, P, TH, SW) pr("coltile C=" #C " P=" #P " threads=" #TH " swz=" #SW, timeit([&] { flip ^= 1; copy_coltile<C, P><<<N / C, TH>>>(flip ? a : c2, flip ? b : d2, N, N, SW); }, 10))
  COL(1, 16, 256, 0); COL(2, 16, 512, 0); COL(2, 16, 512, 2); COL(2, 16, 512, 4); COL(2, 16, 512, 8); COL(1, 16, 256, 16); COL(1, 16, 256, 8);
  COL(4, 16, 1024, 0); COL(4, 16, 1024, 2); COL(4, 16, 1024, 4); COL(4, 16, 1024, 8); COL(4, 16, 1024, 16); COL(4, 16, 1024, 32); COL(4, 16, 512, 4); COL(4, 16, 512, 8); COL(4, 8, 1024, 0); COL(4, 16, 512, 0); COL(4, 16, 256, 0);
  COL(8, 16, 1024, 0); COL(8, 8, 1024, 0); COL(16, 16, 1024, 0); COL(16, 4, 1024, 0); COL(64, 4, 1024, 0);
#define STR(P, TH, W) pr("strided-rows 64xW  W=" #W " P=" #P " threads=" #TH, timeit([&] { flip ^= 1; copy_strided_rows<P><<<64 * (N / W), TH>>>(flip ? a : c2, flip ? b : d2, N, N, W); }, 10))
  STR(4, 256, 16); STR(4, 256, 64); STR(1, 1024, 64); STR(4, 1024, 256);
  return 0;
}
