// What does a phase boundary cost on MI355X when the phases of a small-grid step (QGModel 256^2: 8 phases per step, 5-14 us
// each) are (a) separate dependent kernels, (b) the phases of ONE persistent kernel separated by a grid
// barrier, with the workgroups spread over the 8 XCDs, (c) the same with every active workgroup on ONE XCD (workgroups are dealt
// round-robin to the XCDs: of 8 G launched blocks only those with blockIdx % 8 == 0 stay), where the data that crosses a phase
// stays in that XCD's L2?  Every phase reads what its neighbour workgroup wrote in the previous phase (one complex per thread) and
// writes its own, so a barrier that does not make the data visible shows up as a wrong checksum.
// Safety: the spin has an iteration bound; a barrier that times out sets a flag and every workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int THREADS = 192;

__device__ __forceinline__ double phase_work(const double* __restrict__ in, int src_wg, int iters) {
  double v = in[(size_t)src_wg * THREADS + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9;      // a dependent chain standing for the transforms of a phase
  return v;
}

// (a) one phase per launch
__global__ void __launch_bounds__(THREADS) k_phase(const double* __restrict__ in, double* __restrict__ out, int nwg, int iters) {
  const int w = blockIdx.x;
  out[(size_t)w * THREADS + threadIdx.x] = phase_work(in, (w + 1) % nwg, iters);
}

// (b), (c) all phases in one launch; stride = 1: every block is active, stride = 8: blocks with blockIdx % 8 == 0
template <bool AGENT_FENCE>
__global__ void __launch_bounds__(THREADS) k_persistent(double* __restrict__ a, double* __restrict__ b, int nwg, int stride,
                                                        int phases, int iters, unsigned* __restrict__ bar, int* __restrict__ fail,
                                                        unsigned* __restrict__ xcc_seen) {
  if (blockIdx.x % stride) return;
  const int w = blockIdx.x / stride;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    atomicOr(xcc_seen, 1u << (id & 7));
  }
  double* in = a;
  double* out = b;
  for (int p = 0; p < phases; ++p) {
    out[(size_t)w * THREADS + threadIdx.x] = phase_work(in, (w + 1) % nwg, iters);
    // ---- grid barrier
    // AGENT_FENCE: release at agent scope = write the XCD's L2 back (buffer_wbl2 sc1), what other XCDs need to see the data.
    // Without: only wait until this workgroup's stores have reached its own L2 -- enough when every reader shares that L2.
    if (AGENT_FENCE) __threadfence();
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1u, AGENT_FENCE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)nwg * (unsigned)(p + 1);
      long spins = 0;
      while (__hip_atomic_load(bar, AGENT_FENCE ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 2000000L || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (AGENT_FENCE) __threadfence();                     // acquire
    else asm volatile("buffer_inv sc1" ::: "memory");     // drop this CU's L1 lines (cheap); no L2 write-back anywhere
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    double* t = in; in = out; out = t;
  }
}

static double checksum(const double* d, size_t n) {
  std::vector<double> h(n);
  CK(hipMemcpy(h.data(), d, n * sizeof(double), hipMemcpyDeviceToHost));
  double s = 0;
  for (double x : h) s += x;
  return s;
}

int main() {
  const int phases = 64, reps = 20;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nwg : {32, 64, 128}) {
    const size_t n = (size_t)nwg * THREADS;
    double *a, *b;
    unsigned *bar, *xcc;
    int* fail;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&bar, 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&xcc, 4));
    std::vector<double> init(n, 1.0);
    for (int iters : {0, 400}) {           // 0: the bare boundary; 400 dependent fp64 ops: ~1.5 us of phase
      // (a)
      CK(hipMemcpy(a, init.data(), n * 8, hipMemcpyHostToDevice));
      float best = 1e9f;
      for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        double *in = a, *out = b;
        for (int p = 0; p < phases; ++p) {
          hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(THREADS), 0, 0, in, out, nwg, iters);
          double* t = in; in = out; out = t;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      CK(hipMemcpy(a, init.data(), n * 8, hipMemcpyHostToDevice));
      { double *in = a, *out = b; for (int p = 0; p < phases; ++p) { hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(THREADS), 0, 0, in, out, nwg, iters); double* t = in; in = out; out = t; } }
      CK(hipDeviceSynchronize());
      const double want = checksum(a, n);      // phases is even: the result is back in a
      printf("%3d workgroups, %3d-op phases: (a) dependent kernels            %6.2f us per phase\n", nwg, iters, best * 1e3 / phases);
      // (b), (c)
      for (int stride : {1, 8}) {
        for (int fence = 1; fence >= 0; --fence) {
          float bestp = 1e9f;
          int failed = 0;
          unsigned seen = 0;
          double got = 0;
          for (int r = 0; r < reps && !failed; ++r) {
            CK(hipMemcpy(a, init.data(), n * 8, hipMemcpyHostToDevice));
            CK(hipMemset(bar, 0, 4)); CK(hipMemset(fail, 0, 4)); CK(hipMemset(xcc, 0, 4));
            CK(hipEventRecord(e0));
            if (fence) hipLaunchKernelGGL((k_persistent<true>), dim3(nwg * stride), dim3(THREADS), 0, 0, a, b, nwg, stride, phases, iters, bar, fail, xcc);
            else hipLaunchKernelGGL((k_persistent<false>), dim3(nwg * stride), dim3(THREADS), 0, 0, a, b, nwg, stride, phases, iters, bar, fail, xcc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < bestp) bestp = ms;
            CK(hipMemcpy(&failed, fail, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&seen, xcc, 4, hipMemcpyDeviceToHost));
            got = checksum(a, n);
          }
          int nx = 0;
          for (int i = 0; i < 8; ++i) nx += (seen >> i) & 1;
          printf("%3d workgroups, %3d-op phases: (%s) grid barrier, %s, %s  %6.2f us per phase   XCDs used %d   %s%s\n", nwg, iters,
                 stride == 1 ? "b" : "c", stride == 1 ? "blocks on all XCDs  " : "blocks with b%8 == 0",
                 fence ? "agent-scope fences  " : "L1 invalidate only  ", bestp * 1e3 / phases, nx,
                 failed ? "BARRIER TIMED OUT" : (got == want ? "checksum ok" : "CHECKSUM WRONG (stale data)"), "");
        }
      }
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(bar)); CK(hipFree(fail)); CK(hipFree(xcc));
  }
  return 0;
}
