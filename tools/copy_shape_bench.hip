// What access SHAPE does a 1-read + 1-write stream want on MI355X?  The plain grid-stride copy of tools/rw_mix_bench.hip gets
// 5.0 TB/s; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy.  Every HBM-bound kernel of the step (k_y_A, k_s_phi, k_s_q,
// k_s_invert: 70 % of the 4096^2 step) runs at the grid-stride rate, so if some shape streams faster it is worth knowing which.
// Sweep: grid-stride vs one contiguous chunk per workgroup, loads-then-stores unrolled U deep, non-temporal loads / stores,
// workgroups per CU.  512 MiB in, 512 MiB out (nothing served by the 256 MB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool CHUNK, int NT>     // NT: 0 plain, 1 nt stores, 2 nt loads + nt stores
__global__ __launch_bounds__(256) void k_copy(const cd* __restrict__ in, cd* __restrict__ out, size_t n) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  size_t i, step, end;
  if (CHUNK) {                       // workgroup b streams the contiguous range [b * per, (b + 1) * per)
    const size_t per = n / gridDim.x;
    i = (size_t)blockIdx.x * per + threadIdx.x;
    step = blockDim.x;
    end = (size_t)(blockIdx.x + 1) * per;
  } else {
    i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    step = nthreads;
    end = n;
  }
  for (; i + (U - 1) * step < end; i += U * step) {
    cd v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT == 2) {
        v[u].x = __builtin_nontemporal_load(&in[i + u * step].x);
        v[u].y = __builtin_nontemporal_load(&in[i + u * step].y);
      } else {
        v[u] = in[i + u * step];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT >= 1) {
        __builtin_nontemporal_store(v[u].x, &out[i + u * step].x);
        __builtin_nontemporal_store(v[u].y, &out[i + u * step].y);
      } else {
        out[i + u * step] = v[u];
      }
    }
  }
}

template <int U, bool CHUNK, int NT>
static void run(const char* name, const cd* in, cd* out, size_t n, int wg_per_cu) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  const int grid = 256 * wg_per_cu;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_copy<U, CHUNK, NT>), dim3(grid), dim3(256), 0, 0, in, out, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("%-58s %2d WG/CU  %7.3f ms  %6.0f GB/s\n", name, wg_per_cu, best, 2.0 * n * 16.0 / best * 1e-6);
}

int main() {
  const size_t n = (size_t)512 * 1024 * 1024 / 16;
  cd *in, *out;
  CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16));
  CK(hipMemset(in, 1, n * 16)); CK(hipMemset(out, 0, n * 16));
  for (int w : {2, 4, 8, 16, 32}) {
    run<1, false, 0>("grid-stride, 1 element per iteration", in, out, n, w);
    run<4, false, 0>("grid-stride, 4 loads then 4 stores", in, out, n, w);
    run<8, false, 0>("grid-stride, 8 loads then 8 stores", in, out, n, w);
    run<4, true, 0>("contiguous chunk per workgroup, 4 deep", in, out, n, w);
    run<8, true, 0>("contiguous chunk per workgroup, 8 deep", in, out, n, w);
    run<4, false, 1>("grid-stride, 4 deep, non-temporal stores", in, out, n, w);
    run<4, false, 2>("grid-stride, 4 deep, non-temporal loads and stores", in, out, n, w);
    run<8, true, 2>("contiguous chunk, 8 deep, non-temporal loads and stores", in, out, n, w);
  }
  return 0;
}
