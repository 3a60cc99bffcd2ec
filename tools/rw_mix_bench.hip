// Read/write mix microbenchmark: what does one MI355X stream when a kernel reads R arrays and writes W arrays (each 512 MiB,
// 16 B per lane, grid-stride)?  Question behind it: k_s_q moves 79 % reads and reaches 4.6 TB/s, k_y_A moves 50 % / 50 %
// and reaches 5.7 -- is the read side the limit?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Ptrs { cd* p[8]; };

template <int R, int W>
__global__ __launch_bounds__(256) void k_mix(Ptrs in, Ptrs out, size_t n, double* sink) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (; i < n; i += stride) {
    cd v = make_double2(1.0, 2.0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const cd x = in.p[r][i];
      v.x += x.x;
      v.y += x.y;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) out.p[w][i] = make_double2(v.x + w, v.y);
    if (W == 0) acc += v.x + v.y;
  }
  if (W == 0 && acc == 12345.678) *sink = acc;      // keeps the loads alive
}

template <int R, int W>
static void run(Ptrs in, Ptrs out, size_t n, double* sink) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_mix<R, W>), dim3(256 * 8), dim3(256), 0, 0, in, out, n, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double gb = (double)(R + W) * n * 16.0 * 1e-9;
  printf("%d read + %d write streams: %7.3f ms  total %6.0f GB/s  (read %5.0f, write %5.0f)\n", R, W, best, gb / best * 1e3,
         R * n * 16e-9 / best * 1e3, W * n * 16e-9 / best * 1e3);
}

int main(int argc, char** argv) {
  // argv[1]: stagger in bytes -- array i starts i * stagger bytes into its allocation (0: all arrays at offsets that are
  // multiples of 512 MiB apart, i.e. the same index of every array maps to the same DRAM channel and bank)
  const size_t stagger = argc > 1 ? (size_t)atol(argv[1]) : 0;
  const size_t n = (size_t)512 * 1024 * 1024 / 16;
  printf("stagger %zu bytes\n", stagger);
  Ptrs in, out;
  for (int i = 0; i < 8; ++i) {
    char* b; CK(hipMalloc(&b, n * 16 + 16 * stagger)); CK(hipMemset(b, 0, n * 16 + 16 * stagger));
    in.p[i] = reinterpret_cast<cd*>(b + i * stagger);
  }
  for (int i = 0; i < 4; ++i) {
    char* b; CK(hipMalloc(&b, n * 16 + 16 * stagger)); CK(hipMemset(b, 0, n * 16 + 16 * stagger));
    out.p[i] = reinterpret_cast<cd*>(b + (8 + i) * stagger);
  }
  for (int i = 4; i < 8; ++i) out.p[i] = nullptr;
  double* sink; CK(hipMalloc(&sink, 8));
  run<1, 0>(in, out, n, sink); run<2, 0>(in, out, n, sink); run<4, 0>(in, out, n, sink); run<8, 0>(in, out, n, sink);
  run<0, 1>(in, out, n, sink); run<0, 2>(in, out, n, sink); run<0, 4>(in, out, n, sink);
  run<1, 1>(in, out, n, sink); run<2, 2>(in, out, n, sink); run<4, 4>(in, out, n, sink);
  run<2, 1>(in, out, n, sink); run<4, 1>(in, out, n, sink); run<5, 2>(in, out, n, sink); run<7, 2>(in, out, n, sink); run<4, 2>(in, out, n, sink);
  run<1, 2>(in, out, n, sink); run<2, 4>(in, out, n, sink);
  return 0;
}
