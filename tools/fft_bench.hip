// Row-FFT engine microbenchmark: how long does ONE CU take per 4096-point fp64 complex FFT when the data never
// leaves registers/LDS?  Separates the transform engine (VALU + LDS exchange + barriers) from HBM traffic in the
// fused row kernels.  Each workgroup loads one row, runs ITERS forward transforms on it
// (output fed back as input), stores the row.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../niwqg_amd/csrc/nq_fft.hpp"
#include "../niwqg_amd/csrc/nq_generic.hpp"
using namespace nq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int N, int PP, int WG, bool LDS_TW>
__global__ void __launch_bounds__((XPlanT<N, PP, WG, PP>::THREADS), (XPlanT<N, PP, WG, PP>::MIN_WAVES))
k_fft_loop(const cd* __restrict__ in, cd* __restrict__ out, const cd* __restrict__ tw, const cd* __restrict__ twx, int iters) {
  typedef XPlanT<N, PP, WG, PP> X;
  typedef typename X::F F;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  cd* twl = lds + F::LDS_ELEMS;
  cd r[P];
#pragma unroll
  for (int t = 0; t < P; ++t) r[t] = in[(size_t)blockIdx.x * N + j + t * T];
  if constexpr (LDS_TW) {
    for (int i = threadIdx.x; i < F::TW_LDS_ELEMS; i += X::THREADS) twl[i] = twx[i];
    wg_barrier_all();
    typename F::TwLds src{twl};
    for (int it = 0; it < iters; ++it) {
      F::template run<false>(r, j, c, lds, src);
#pragma unroll
      for (int t = 0; t < P; ++t) r[t] = cscale(r[t], 1.0 / 64.0);
    }
  } else {
    typename F::Tw twr;
    F::load_tw(twr, j, tw, 1);
    for (int it = 0; it < iters; ++it) {
      F::template run<false>(r, j, c, lds, twr);
#pragma unroll
      for (int t = 0; t < P; ++t) r[t] = cscale(r[t], 1.0 / 64.0);
    }
  }
#pragma unroll
  for (int t = 0; t < P; ++t) out[(size_t)blockIdx.x * N + j + t * T] = r[t];
}

// two transforms in flight per workgroup (WgFft::run2), two exchange areas
template <int N, int PP>
__global__ void __launch_bounds__((XPlanT<N, PP, 1, PP>::THREADS), (XPlanT<N, PP, 1, PP>::MIN_WAVES))
k_fft_loop2(const cd* __restrict__ in, cd* __restrict__ out, const cd* __restrict__ twx, int iters) {
  typedef XPlanT<N, PP, 1, PP> X;
  typedef typename X::F F;
  constexpr int P = X::P, T = X::T;
  const int j = threadIdx.x % T, c = threadIdx.x / T;
  cd* lds = reinterpret_cast<cd*>(nq_smem);
  cd* lds_b = lds + F::LDS_ELEMS;
  cd* twl = lds_b + F::LDS_ELEMS;
  cd a[P], b[P];
#pragma unroll
  for (int t = 0; t < P; ++t) {
    a[t] = in[(size_t)(2 * blockIdx.x) * N + j + t * T];
    b[t] = in[(size_t)(2 * blockIdx.x + 1) * N + j + t * T];
  }
  for (int i = threadIdx.x; i < F::TW_LDS_ELEMS; i += X::THREADS) twl[i] = twx[i];
  wg_barrier_all();
  typename F::TwLds src{twl};
  for (int it = 0; it < iters; ++it) {
    F::template run2<false>(a, b, j, c, lds, lds_b, src);
#pragma unroll
    for (int t = 0; t < P; ++t) {
      a[t] = cscale(a[t], 1.0 / 64.0);
      b[t] = cscale(b[t], 1.0 / 64.0);
    }
  }
#pragma unroll
  for (int t = 0; t < P; ++t) {
    out[(size_t)(2 * blockIdx.x) * N + j + t * T] = a[t];
    out[(size_t)(2 * blockIdx.x + 1) * N + j + t * T] = b[t];
  }
}

static std::vector<double> stage_table(int N, int PP, const std::vector<double>& twh) {
  std::vector<double> st;
  for (int sidx = 1; sidx < plan_stages(N, PP); ++sidx) {
    const int R = plan_radix(N, PP, sidx), NS = plan_ns(N, PP, sidx);
    for (int pw = 1; pw <= 8; pw *= (pw == 1 ? 4 : 2)) {
      if ((pw == 4 && plan_tw_rows(N, PP, R) < 2) || (pw == 8 && plan_tw_rows(N, PP, R) < 3)) continue;
      for (int jr = 0; jr < NS; ++jr) {
        const long long m = ((long long)pw * jr * (N / (NS * R))) % N;
        st.push_back(twh[2 * m]);
        st.push_back(twh[2 * m + 1]);
      }
    }
  }
  return st;
}

template <int N, int PP>
void run_dual(const cd* in, cd* out, const std::vector<double>& twh, int iters, int nwg) {
  typedef XPlanT<N, PP, 1, PP> X;
  std::vector<double> st = stage_table(N, PP, twh);
  cd* twx;
  CK(hipMalloc(&twx, st.size() * 8 + 16));
  CK(hipMemcpy(twx, st.data(), st.size() * 8, hipMemcpyHostToDevice));
  const size_t ldsb = X::LDS_BYTES + X::F::LDS_ELEMS * sizeof(cd);
  auto k = k_fft_loop2<N, PP>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(X::THREADS), ldsb, 0, in, out, twx, 2);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(X::THREADS), ldsb, 0, in, out, twx, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / ((double)nwg / 256.0 * iters * 2);
  printf("N=%d P=%2d threads=%4d DUAL (two transforms in flight) lds=%6zu B: %8.3f ms, %.3f us per FFT per CU\n", N, PP, X::THREADS, ldsb, ms, us);
  CK(hipFree(twx));
}

template <int N, int PP, int WG, bool LDS_TW>
void run(const cd* in, cd* out, const cd* tw, const std::vector<double>& twh, int iters, int nwg) {
  typedef XPlanT<N, PP, WG, PP> X;
  std::vector<double> st;
  for (int sidx = 1; sidx < plan_stages(N, PP); ++sidx) {
    const int R = plan_radix(N, PP, sidx), NS = plan_ns(N, PP, sidx);
    for (int pw = 1; pw <= 8; pw *= (pw == 1 ? 4 : 2)) {
      if ((pw == 4 && plan_tw_rows(N, PP, R) < 2) || (pw == 8 && plan_tw_rows(N, PP, R) < 3)) continue;
      for (int jr = 0; jr < NS; ++jr) {
        const long long m = ((long long)pw * jr * (N / (NS * R))) % N;
        st.push_back(twh[2 * m]);
        st.push_back(twh[2 * m + 1]);
      }
    }
  }
  cd* twx;
  CK(hipMalloc(&twx, st.size() * 8 + 16));
  CK(hipMemcpy(twx, st.data(), st.size() * 8, hipMemcpyHostToDevice));
  auto k = k_fft_loop<N, PP, WG, LDS_TW>;
  // register twiddles need no table behind the exchange area: 64.5 KB per workgroup, so that TWO 512-thread workgroups fit a CU
  // (with the table the 8-point plan asks for 84.7 KB and "wg/cu=2" silently runs one per CU: round 4)
  const size_t lds_bytes = LDS_TW ? X::LDS_BYTES : (size_t)X::F::LDS_ELEMS * sizeof(cd) + 512;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(X::THREADS), lds_bytes, 0, in, out, tw, twx, 2);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(nwg), dim3(X::THREADS), lds_bytes, 0, in, out, tw, twx, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  // nwg workgroups over 256 CUs: FFTs per CU = nwg/256*iters
  const double us_per_fft_cu = ms * 1e3 / ((double)nwg / 256.0 * iters);
  printf("N=%d P=%2d threads=%4d wg/cu=%d tw=%s lds=%6zu B: %8.3f ms, %.3f us per FFT per CU (%.1f GFLOP/s/CU nominal 5NlogN)\n", N, PP,
         X::THREADS, WG, LDS_TW ? "lds" : "reg", lds_bytes, ms, us_per_fft_cu, 5.0 * N * log2((double)N) / us_per_fft_cu * 1e-3);
  CK(hipFree(twx));
}

int main() {
  constexpr int N = 4096;
  const int rows = 2048;
  std::vector<double> h((size_t)rows * N * 2), twh(2 * N);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
  for (int m = 0; m < N; ++m) { twh[2 * m] = cos(-2.0 * M_PI * m / N); twh[2 * m + 1] = sin(-2.0 * M_PI * m / N); }
  cd *in, *out, *tw;
  CK(hipMalloc(&in, h.size() * 8));
  CK(hipMalloc(&out, h.size() * 8));
  CK(hipMalloc(&tw, twh.size() * 8));
  CK(hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(tw, twh.data(), twh.size() * 8, hipMemcpyHostToDevice));
  const int iters = 200;
  run<N, 8, 1, true>(in, out, tw, twh, iters, 256);
  run<N, 8, 1, false>(in, out, tw, twh, iters, 256);
  run<N, 8, 2, true>(in, out, tw, twh, iters, 512);
  run<N, 8, 2, false>(in, out, tw, twh, iters, 512);
  run<N, 16, 1, true>(in, out, tw, twh, iters, 256);
  run<N, 16, 2, true>(in, out, tw, twh, iters, 512);
  run<N, 16, 2, false>(in, out, tw, twh, iters, 512);
  run<N, 16, 4, true>(in, out, tw, twh, iters, 1024);
  run<N, 4, 1, true>(in, out, tw, twh, iters, 256);
  run_dual<N, 8>(in, out, twh, iters, 256);
  run_dual<N, 16>(in, out, twh, iters, 256);
  return 0;
}
